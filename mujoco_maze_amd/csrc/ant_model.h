// ant_model.h — fp32 device constants of the Ant maze stepper and their
// derivation from the compiled `mz_model` (include/mazestep.h).
//
// The HIP kernel is specialised for the reference's Ant topology
// (mujoco_maze/assets/ant.xml:21-66): free torso + four legs,
// each leg = welded capsule -> hip hinge (body z) -> ankle hinge.  `ant_dev_from_model`
// checks that the compiled model really has that shape and refuses anything
// else (MZ_ERR_UNSUPPORTED) instead of silently computing the wrong robot.
//
// Plain C++ (no HIP): shared by csrc/mazestep.hip and the CPU emulation of the
// kernel logic in tests/emu/.
#pragma once
// ------------------------------------------------------------------ experiment switches
// The kernel sources carry compile-time switches of the form MZ_EXP_<NAME>: instrumentation (PROF, SWPROF, STAMPS, SUBTICK, SUBTICK2,
// TRACE, RK4TICK, GENPROF) and TIMING EXPERIMENTS that switch parts of the physics off or reroute them (NOSECOND, NOSEARCH, NODETECT, NOFACEPATH,
// SPHEREONLY, NOCOLLISION, NOARROW, NONEWTON, NOWALL, ONECAND, NOREFINE, NOLAUNDER, BRANCHY, NOBLOCKCACHE, LCRELOAD, NOXCD — several of
// them WRONG PHYSICS by design; tools/README.md).  None of them may reach the product library by accident: a translation unit
// that sees any MZ_EXP_* macro without -DMZ_EXPERIMENTS does not compile, csrc/Makefile's default target refuses flags that carry
// either, and the experiment builders (make dev / dev1 DEVFLAGS=..., tools/exp_build.sh) pass -DMZ_EXPERIMENTS themselves and write
// to libmazestep_dev.so / exp_*.so, never to libmazestep.so.
#if !defined(MZ_EXPERIMENTS) && (defined(MZ_EXP_NOSECOND) || defined(MZ_EXP_NOSEARCH) || defined(MZ_EXP_NODETECT) || defined(MZ_EXP_NOFACEPATH) || \
    defined(MZ_EXP_SPHEREONLY) || defined(MZ_EXP_NOCOLLISION) || defined(MZ_EXP_NOARROW) || defined(MZ_EXP_NONEWTON) || defined(MZ_EXP_NOWALL) || \
    defined(MZ_EXP_ONECAND) || defined(MZ_EXP_NOREFINE) || defined(MZ_EXP_NOLAUNDER) || defined(MZ_EXP_BRANCHY) || defined(MZ_EXP_NOBLOCKCACHE) || \
    defined(MZ_EXP_LCRELOAD) || defined(MZ_EXP_NOXCD) || defined(MZ_EXP_PROF) || defined(MZ_EXP_SWPROF) || defined(MZ_EXP_STAMPS) || \
    defined(MZ_EXP_SUBTICK) || defined(MZ_EXP_SUBTICK2) || defined(MZ_EXP_TRACE) || defined(MZ_EXP_RK4TICK) || defined(MZ_EXP_GENPROF))
#error "MZ_EXP_* switches build instrumented / timing-experiment kernels (several with wrong physics): pass -DMZ_EXPERIMENTS with them, and never into libmazestep.so"
#endif
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/mazestep.h"

#define ANT_NBODY 13  // moving bodies: torso + 4 x (leg, aux, ankle)
#define ANT_NV 14
#define ANT_NQ 15
#define ANT_NU 8
#define ANT_OBS 30

struct PairDev {  // mixed contact parameters of one geom-pair class
  float margin, mu, K, B, solimp[7];  // d0 dmax width midpoint power | 1 - d0, 1 - dmax rounded from float64 (impedance_pair, ant_dyn.h)
};

// Task constants.  Everything a flag depends on is float64, exactly the reference's values (maze_task.py:26-47): the
// goal predicate `np.linalg.norm(obs[:dim] - pos) <= threshold` is evaluated in fp64 on the returned observation.
// thr_sq[g] = the largest double s with sqrt(s) <= threshold (sqrt correctly rounded, as numpy's): `s <= thr_sq` is
// then the same predicate without a square root, so no floating-point build flag can change a flag.
struct TaskDev {
  int ngoal, reward_kind, reward_slot, reward_binary, term_slot, max_steps;
  int goal_dim[MZ_MAX_GOAL];
  double goal_pos[MZ_MAX_GOAL][3], thr[MZ_MAX_GOAL], thr_sq[MZ_MAX_GOAL], rscale[MZ_MAX_GOAL];
  double penalty, task_scale, inner_scale, fwd_w, ctrl_w;
  // per-env goal POSITIONS (mz_bind_env_goals; device pointer, [N][MZ_MAX_GOAL][3] float64, or NULL: the batch shares goal_pos).  The
  // reference resamples a task's goals at EVERY episode reset (maze_env.py:374-376: one task object per env); thresholds, reward
  // scales and dims stay the task class's
  const double* env_goals;
};

struct MazeDev {
  int rows, cols;
  uint32_t rowmask[MZ_MAX_GRID];  // bit j set <=> cell (i, j) is a BLOCK
  float scale, tx, ty, half_xy, half_z, center_z;
  // elevated mazes (Fall / MultiFall): a platform box (same footprint, z from 0 to 2 half_z, centre half_z) under every cell
  // of the grid that is not a CHASM; the walls stand on top (center_z = half_z + height offset)
  int elevated;
  uint32_t platmask[MZ_MAX_GRID];  // bit j set <=> cell (i, j) carries a platform
};

struct AntDev {
  float h, gz;
  int frame_skip;
  // body classes: 0 torso (sphere), 1 welded leg capsule, 2 aux capsule, 3 ankle capsule
  float mass[4], ilat[4], iax[4], half_len[4], radius[4], bw_tran[4];
  float legoff;  // |x| = |y| offset of aux / ankle body origins in the parent frame
  float sx[4], sy[4];
  float ank_axis[4][3];
  float hip_lo, hip_hi, ank_lo[4], ank_hi[4];
  float armature, damping, dofw_hip, dofw_ank;
  float lim_K, lim_B, lim_solimp[7];
  float ctrl_lo, ctrl_hi, gear;
  int act_dof[ANT_NU];
  PairDev floor, wall;
  MazeDev maze;
  TaskDev task;
  float qpos0[ANT_NQ + 7];  // robot, then the pose of a free-joint object ball
  int reset_kind;
  // object ball on a free joint (AntSmallBilliard; AntEnv.OBJBALL_TYPE, maze_env.py:539-560): a sphere of radius ball_r whose
  // centre sits ball_h above the body origin (the body frame starts on the floor), isotropic inertia about the centre
  int nball, observe_balls;
  float ball_mass, ball_inertia, ball_r, ball_h, ball_bw_tran;
  PairDev ball_floor, ball_wall, ball_robot;
  // movable XY blocks (all the same size: one maze cell); see maze_env.py:563-660
  int nblock, observe_blocks;
  float block_mass, block_bw_tran, block_half[3], block_pos0[4][3];
  // the two slide axes of a block: (x, y) in the Push family; (y, z) — LIMITED, gravity on the z slide — in the Fall mazes
  // (x, y, z) for MultiFall's three-slide block.  block_axis[a] = coordinate axis of slide a (increasing), block_nax = 2 or 3
  int block_axis[3], block_nax, block_limited;
  float block_lo[3], block_hi[3], blim_margin, blim_K, blim_B, blim_solimp[7], blim_w;  // joint-limit rows of the slides
  // float64 copies of the geometry behind ONE decision that sits on an exact tie in a registered maze: a falling block is
  // shrunk to 99 % (maze_env.py:579-582), so at maze scale 2 (AntMultiFall) the faces of the neighbouring platforms /
  // walls are 0.01 away — exactly the contact margin.  `gap < margin` is false in the reference's float64 arithmetic
  // (2 - 1.99 = 0.010000000000000009); in fp32 it would be true.  The block-vs-cell gaps are therefore evaluated in fp64.
  double d_scale, d_tx, d_ty, d_half_xy, d_half_z, d_center_z, d_wall_margin, d_block_half[3], d_block_pos0[4][3];
  // solver
  int max_iter, ls_iter, trust_exact;
  int ls_fast_iters, ls_fast;  // the first `ls_fast_iters` Newton iterations of an evaluation search the line with at most `ls_fast` evaluations (ant_newton_rows.h)
  float tol, rtol, inv_scale;  // inv_scale = 1 / (meaninertia * nv)
};

static inline int ant_fail(char* err, int n, const char* msg) {
  if (err && n > 0) { strncpy(err, msg, (size_t)n - 1); err[n - 1] = 0; }
  return MZ_ERR_UNSUPPORTED;
}

static inline void pair_from(PairDev* p, const mz_model* m, const double* f1, const double* sr1, const double* si1, double mg1,
                             const double* f2, const double* sr2, const double* si2, double mg2) {
  double sr[2], si[5];
  for (int k = 0; k < 2; k++) sr[k] = 0.5 * (sr1[k] + sr2[k]);
  for (int k = 0; k < 5; k++) si[k] = 0.5 * (si1[k] + si2[k]);
  p->margin = (float)fmax(mg1, mg2);
  p->mu = (float)fmax(f1[0], f2[0]);
  double tc = fmax(sr[0], 2.0 * m->timestep), dr = sr[1], dmax = si[1];
  p->K = (float)(1.0 / (dmax * dmax * tc * tc * dr * dr));
  p->B = (float)(2.0 / (dmax * tc));
  for (int k = 0; k < 5; k++) p->solimp[k] = (float)si[k];
  p->solimp[5] = (float)(1.0 - si[0]); p->solimp[6] = (float)(1.0 - si[1]);
}

// largest double s with sqrt(s) <= thr (host libm sqrt is correctly rounded); -1 for a negative threshold (never matches)
static inline double mz_sqrt_le_bound(double thr) {
  if (!(thr >= 0.0)) return -1.0;
  if (isinf(thr)) return thr;
  double s = thr * thr;
  while (sqrt(s) > thr) s = nextafter(s, 0.0);
  while (sqrt(nextafter(s, INFINITY)) <= thr) s = nextafter(s, INFINITY);
  return s;
}

static inline void task_dev_from_model(TaskDev* t, const mz_model* m) {
  memset(t, 0, sizeof(*t));
  t->ngoal = m->ngoal; t->reward_kind = m->reward_kind; t->reward_slot = m->reward_slot;
  t->reward_binary = m->reward_binary; t->term_slot = m->term_slot; t->max_steps = m->max_episode_steps;
  for (int g = 0; g < m->ngoal; g++) {
    t->goal_dim[g] = m->goal_dim[g];
    for (int k = 0; k < 3; k++) t->goal_pos[g][k] = m->goal_pos[g][k];
    t->thr[g] = m->goal_threshold[g];
    t->thr_sq[g] = mz_sqrt_le_bound(m->goal_threshold[g]);
    t->rscale[g] = m->goal_reward_scale[g];
  }
  t->penalty = m->penalty; t->task_scale = m->task_scale; t->inner_scale = m->inner_reward_scaling;
  t->fwd_w = m->forward_reward_weight; t->ctrl_w = m->ctrl_cost_weight;
}

static inline void maze_dev_from_model(MazeDev* z, const mz_model* m) {
  memset(z, 0, sizeof(*z));
  z->rows = m->grid_rows; z->cols = m->grid_cols;
  for (int i = 0; i < m->grid_rows; i++)
    for (int j = 0; j < m->grid_cols; j++)
      if (m->grid[i][j] == MZ_CELL_BLOCK) z->rowmask[i] |= (1u << j);
  z->elevated = m->elevated;
  for (int i = 0; i < m->grid_rows && m->elevated; i++)
    for (int j = 0; j < m->grid_cols; j++)
      if (m->grid[i][j] != MZ_CELL_CHASM) z->platmask[i] |= (1u << j);
  z->scale = (float)m->maze_scale; z->tx = (float)m->torso_x; z->ty = (float)m->torso_y;
  z->half_xy = (float)m->wall_half_xy; z->half_z = (float)m->wall_half_z; z->center_z = (float)m->wall_center_z;
}

static inline int ant_dev_from_model(AntDev* a, const mz_model* m, char* err, int errlen) {
  memset(a, 0, sizeof(*a));
  const int nb = m->nblock, nball = m->nball;
  if (nb < 0 || nb > 4) return ant_fail(err, errlen, "ant kernel: at most 4 movable blocks");
  if (nball < 0 || nball > 1 || (nball && nb)) return ant_fail(err, errlen, "ant kernel: at most one object ball, and not together with movable blocks");
  int nbdof = 0;
  for (int k = 0; k < nb; k++) nbdof += m->body_jntnum[m->block_bodyid[k]];
  if (m->robot != MZ_ROBOT_ANT || m->nbody != 14 + nb + nball || m->nv != ANT_NV + nbdof + 6 * nball || m->nq != ANT_NQ + nbdof + 7 * nball ||
      m->nu != ANT_NU || m->ngeom != 14 + nb + nball)
    return ant_fail(err, errlen, "ant kernel: model is not the 14-body / 14-dof ant (+ movable blocks or one free-joint ball)");
  a->nblock = nb; a->observe_blocks = m->observe_blocks;
  a->nball = nball; a->observe_balls = m->observe_balls;
  if (nball) {
    const int b = m->ball_bodyid[0], g = m->ball_geomid[0], j = m->body_jntadr[b];
    const double* I = m->body_inertia[b];
    if (b != 14 || g != 14 || m->geom_type[g] != MZ_GEOM_SPHERE || m->body_jntnum[b] != 1 || m->jnt_type[j] != MZ_JNT_FREE ||
        m->jnt_qposadr[j] != ANT_NQ || m->jnt_dofadr[j] != ANT_NV || fabs(I[0] - I[1]) > 1e-12 * I[0] || fabs(I[0] - I[2]) > 1e-12 * I[0] ||
        fabs(I[3]) + fabs(I[4]) + fabs(I[5]) > 1e-12 * I[0] || fabs(m->body_ipos[b][0]) + fabs(m->body_ipos[b][1]) > 1e-12 ||
        fabs(m->geom_pos[g][2] - m->body_ipos[b][2]) > 1e-12)
      return ant_fail(err, errlen, "ant kernel: the object ball is one sphere on a free joint, centred above its body origin");
    a->ball_mass = (float)m->body_mass[b]; a->ball_inertia = (float)I[0]; a->ball_r = (float)m->geom_size[g][0];
    a->ball_h = (float)m->body_ipos[b][2]; a->ball_bw_tran = (float)m->body_invweight0[b][0];
    pair_from(&a->ball_floor, m, m->geom_friction[0], m->geom_solref[0], m->geom_solimp[0], m->geom_margin[0], m->geom_friction[g],
              m->geom_solref[g], m->geom_solimp[g], m->geom_margin[g]);
    pair_from(&a->ball_wall, m, m->geom_friction[g], m->geom_solref[g], m->geom_solimp[g], m->geom_margin[g], m->wall_friction,
              m->wall_solref, m->wall_solimp, m->wall_margin);
    pair_from(&a->ball_robot, m, m->geom_friction[1], m->geom_solref[1], m->geom_solimp[1], m->geom_margin[1], m->geom_friction[g],
              m->geom_solref[g], m->geom_solimp[g], m->geom_margin[g]);
  }
  if (m->elevated && nb == 0) return ant_fail(err, errlen, "ant kernel: an elevated maze needs a movable block (the platform code lives in the block instantiations)");
  a->block_axis[0] = 0; a->block_axis[1] = 1; a->block_axis[2] = 2; a->block_nax = 2;
  for (int k = 0; k < nb; k++) {
    int b = m->block_bodyid[k], g = m->block_geomid[k], j0 = m->body_jntadr[b];
    const int nax = m->body_jntnum[b];
    int ax[3] = {-1, -1, -1};
    bool ok = (nax == 2 || (nax == 3 && nb == 1)) && b == 14 + k && m->geom_type[g] == MZ_GEOM_BOX && m->jnt_dofadr[j0] == ANT_NV + nax * k;
    for (int q = 0; ok && q < nax; q++) {
      for (int c = 0; c < 3; c++) if (fabs(m->jnt_axis[j0 + q][c] - 1.0) < 1e-12) ax[q] = c;
      ok = m->jnt_type[j0 + q] == MZ_JNT_SLIDE && ax[q] >= 0 && (q == 0 || ax[q] > ax[q - 1]) && m->jnt_limited[j0 + q] == m->jnt_limited[j0];
    }
    if (ok && k > 0) ok = nax == a->block_nax && ax[0] == a->block_axis[0] && ax[1] == a->block_axis[1] && m->jnt_limited[j0] == a->block_limited;
    if (!ok)
      return ant_fail(err, errlen, "ant kernel: a movable block is a box body with two slides along increasing coordinate axes (x y, y z or x z), "
                                   "or — a single block — with three (x y z)");
    a->block_nax = nax; a->block_limited = m->jnt_limited[j0];
    for (int q = 0; q < nax; q++) { a->block_axis[q] = ax[q]; a->block_lo[q] = (float)m->jnt_range[j0 + q][0]; a->block_hi[q] = (float)m->jnt_range[j0 + q][1]; }
    {
      double tc = fmax(m->jnt_solref[j0][0], 2.0 * m->timestep), dr = m->jnt_solref[j0][1], dmax = m->jnt_solimp[j0][1];
      a->blim_K = (float)(1.0 / (dmax * dmax * tc * tc * dr * dr)); a->blim_B = (float)(2.0 / (dmax * tc));
      a->blim_margin = (float)m->jnt_margin[j0]; a->blim_w = (float)m->dof_invweight0[m->jnt_dofadr[j0]];
      for (int q = 0; q < 5; q++) a->blim_solimp[q] = (float)m->jnt_solimp[j0][q];
      a->blim_solimp[5] = (float)(1.0 - m->jnt_solimp[j0][0]); a->blim_solimp[6] = (float)(1.0 - m->jnt_solimp[j0][1]);
    }
    for (int q = 0; q < 3; q++) {
      a->block_pos0[k][q] = (float)m->body_pos[b][q]; a->block_half[q] = (float)m->geom_size[g][q];
      a->d_block_pos0[k][q] = m->body_pos[b][q]; a->d_block_half[q] = m->geom_size[g][q];
    }
    a->block_mass = (float)m->body_mass[b];
    a->block_bw_tran = (float)m->body_invweight0[b][0];
    {  // the enumeration gives a block 3 x 3 grid cells (geom_contacts): its bounding sphere + margin must stay inside that
      const double* hb = m->geom_size[g];
      if (sqrt(hb[0] * hb[0] + hb[1] * hb[1] + hb[2] * hb[2]) + fmax(m->geom_margin[g], m->wall_margin) >= m->maze_scale)
        return ant_fail(err, errlen, "ant kernel: a movable block must be smaller than a maze cell's reach (its bounding sphere < one cell size)");
    }
  }
  if (m->jnt_type[0] != MZ_JNT_FREE || m->geom_type[0] != MZ_GEOM_PLANE || m->geom_type[1] != MZ_GEOM_SPHERE)
    return ant_fail(err, errlen, "ant kernel: expected free root joint, floor plane, torso sphere");
  a->h = (float)m->timestep; a->gz = (float)m->gravity[2]; a->frame_skip = m->frame_skip;
  // classes from leg 0 (bodies 2,3,4 / geoms 2,3,4), torso = body 1 / geom 1
  const int cb[4] = {1, 2, 3, 4};
  for (int c = 0; c < 4; c++) {
    int b = cb[c], g = b;  // one geom per body, same order (model.py emits floor first)
    a->mass[c] = (float)m->body_mass[b];
    a->ilat[c] = (float)m->body_inertia[b][2];  // zz: capsules lie in the body xy plane
    a->iax[c] = (float)(c == 0 ? m->body_inertia[b][2] : m->body_inertia[b][2] + 2.0 * (m->body_inertia[b][0] - m->body_inertia[b][2]));
    a->radius[c] = (float)m->geom_size[g][0];
    a->half_len[c] = (float)(c == 0 ? 0.0 : m->geom_size[g][1]);
    a->bw_tran[c] = (float)m->body_invweight0[b][0];
  }
  a->legoff = (float)fabs(m->body_pos[3][0]);
  for (int l = 0; l < 4; l++) {
    int b_leg = 2 + 3 * l, b_aux = 3 + 3 * l, b_ank = 4 + 3 * l, j_hip = 1 + 2 * l, j_ank = 2 + 2 * l;
    if (m->body_parent[b_leg] != 1 || m->body_parent[b_aux] != b_leg || m->body_parent[b_ank] != b_aux ||
        m->body_jntnum[b_leg] != 0 || m->jnt_type[j_hip] != MZ_JNT_HINGE || m->jnt_type[j_ank] != MZ_JNT_HINGE ||
        m->jnt_dofadr[j_hip] != 6 + 2 * l || m->geom_type[b_leg] != MZ_GEOM_CAPSULE || fabs(m->jnt_axis[j_hip][2] - 1.0) > 1e-12)
      return ant_fail(err, errlen, "ant kernel: leg topology differs from ant.xml");
    a->sx[l] = m->body_pos[b_aux][0] > 0 ? 1.f : -1.f;
    a->sy[l] = m->body_pos[b_aux][1] > 0 ? 1.f : -1.f;
    if (fabs(fabs(m->body_pos[b_aux][0]) - a->legoff) > 1e-6 || fabs(fabs(m->body_pos[b_ank][1]) - a->legoff) > 1e-6 ||
        fabs(m->body_mass[b_leg] - m->body_mass[2]) > 1e-12 || fabs(m->body_mass[b_ank] - m->body_mass[4]) > 1e-12)
      return ant_fail(err, errlen, "ant kernel: legs are not mirror images");
    for (int k = 0; k < 3; k++) a->ank_axis[l][k] = (float)m->jnt_axis[j_ank][k];
    a->ank_lo[l] = (float)m->jnt_range[j_ank][0]; a->ank_hi[l] = (float)m->jnt_range[j_ank][1];
    if (!m->jnt_limited[j_hip] || !m->jnt_limited[j_ank]) return ant_fail(err, errlen, "ant kernel: hinges must be limited");
  }
  a->hip_lo = (float)m->jnt_range[1][0]; a->hip_hi = (float)m->jnt_range[1][1];
  a->armature = (float)m->dof_armature[6]; a->damping = (float)m->dof_damping[6];
  a->dofw_hip = (float)m->dof_invweight0[6]; a->dofw_ank = (float)m->dof_invweight0[7];
  {
    double tc = fmax(m->jnt_solref[1][0], 2.0 * m->timestep), dr = m->jnt_solref[1][1], dmax = m->jnt_solimp[1][1];
    a->lim_K = (float)(1.0 / (dmax * dmax * tc * tc * dr * dr));
    a->lim_B = (float)(2.0 / (dmax * tc));
    for (int k = 0; k < 5; k++) a->lim_solimp[k] = (float)m->jnt_solimp[1][k];
    a->lim_solimp[5] = (float)(1.0 - m->jnt_solimp[1][0]); a->lim_solimp[6] = (float)(1.0 - m->jnt_solimp[1][1]);
  }
  a->ctrl_lo = (float)m->act_ctrlrange[0][0]; a->ctrl_hi = (float)m->act_ctrlrange[0][1]; a->gear = (float)m->act_gear[0];
  for (int u = 0; u < ANT_NU; u++) a->act_dof[u] = m->act_dofid[u];
  pair_from(&a->floor, m, m->geom_friction[0], m->geom_solref[0], m->geom_solimp[0], m->geom_margin[0], m->geom_friction[1],
            m->geom_solref[1], m->geom_solimp[1], m->geom_margin[1]);
  pair_from(&a->wall, m, m->geom_friction[1], m->geom_solref[1], m->geom_solimp[1], m->geom_margin[1], m->wall_friction,
            m->wall_solref, m->wall_solimp, m->wall_margin);
  maze_dev_from_model(&a->maze, m);
  a->d_scale = m->maze_scale; a->d_tx = m->torso_x; a->d_ty = m->torso_y; a->d_half_xy = m->wall_half_xy; a->d_half_z = m->wall_half_z;
  a->d_center_z = m->wall_center_z; a->d_wall_margin = fmax(m->geom_margin[1], m->wall_margin);
  task_dev_from_model(&a->task, m);
  for (int k = 0; k < ANT_NQ + 7 * nball; k++) a->qpos0[k] = (float)m->qpos0[k];
  a->reset_kind = m->reset_qvel_kind;
  a->max_iter = 50;  // (the plain ant ran with 10 until round 3: one env in 2.5e8 env-steps of the 60 000-step soak needed an eleventh iteration)
  a->ls_iter = 12; a->ls_fast_iters = 5; a->ls_fast = 0; a->trust_exact = 1; a->tol = 1e-6f; a->rtol = 1e-6f;
  a->inv_scale = (float)(1.0 / (m->meaninertia * m->nv));
  return MZ_OK;
}
