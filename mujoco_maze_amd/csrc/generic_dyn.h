// generic_dyn.h — the GENERAL ENGINE: MazeEnv.step for any compiled `mz_model`, as lane-group SPMD code in float64.
//
// The specialised kernels (ant_dyn.h, planar_dyn.h, swimmer_dyn.h) hard-wire the reference's four robots in the mazes its
// registry builds.  Everything else the reference's plugin surface admits steps here instead of being refused:
//   * a user's AgentModel of any tree topology (mujoco_maze/agent_model.py:12-41, README.md:127 "you can define your own robot"):
//     free / ball / slide / hinge joints, sphere / capsule / box geoms, motors — in any maze, movable bodies included;
//   * SPIN plates (maze_env.py:119-120,575,649-660, maze_env_utils.py:33,74-75, maze_task.py:67 PUT_SPIN_NEAR_AGENT): a thin box on two
//     slides and a BALL joint next to the robot — a box of general orientation against floor, walls, blocks and the robot;
//   * any built-in robot in any maze when the caller asks for it (mz_model.engine = 1): a second, independent device
//     implementation the specialised kernels are compared with (tests/test_gpu_general_engine.py).
// The kernel walks the kinematic tree of the compiled model itself (bodies, joints, dofs, geoms, motors; the maze's wall /
// platform boxes come from the cell grid).  Slower than the specialised paths by design — tree walks are serial, the mass matrix
// and the Newton system are dense, one wavefront steps one env — but the same physics: what the reference gets from
// `do_simulation(action, frame_skip)` (ant.py:61-63, swimmer.py:37-39) or from the Point's teleport step (point.py:44-61),
// i.e. frame_skip x mj_step with RK4:
//   kinematics -> spatial inertias, composite rigid bodies, dense M -> collision: every geom pair the contype / conaffinity and
//   parent-child filters admit, and every moving geom against the maze boxes under its bounding square (plane-sphere / -capsule /
//   -box, sphere-sphere / -capsule / -box, capsule-box = mjc_CapsuleBox, box-box = mjc_BoxBox with its edge-edge case, all as restated
//   in oracle/mzo_physics.c and DESIGN.md section 5) -> joint-limit and pyramidal contact rows between TWO moving bodies ->
//   recursive Newton-Euler bias, joint damping, MuJoCo's inertia-box fluid forces, clamped motors -> qacc_smooth (dense
//   Cholesky) -> primal Newton with exact line search -> RK4 with manifold quaternion update (free and ball joints).
// Then MazeEnv.step's own part (maze_env.py:448-481): the Point's manual wall bounce (the bit-exact detector of point_dyn.h),
// the observation with the object balls' / movable blocks' body positions (maze_env.py:351-369), reward and termination.
// One env per lane group; its working set (GenScratch) lives in LDS for the whole step.  Serial tree walks run on the group's
// first lane, everything indexed by body / dof / geom / collision item / constraint row is an MZ_FOR over the lanes.
#pragma once
#include <stddef.h>
#include "ant_dyn.h"    // MZ_FOR, MZ_HD, HostCtx, TaskDev / MazeDev, task_eval_dev
#include "point_dyn.h"  // point_bounce: CollisionDetector.detect + the bounce rule, operation by operation (Point step shape)

#define GN_NB MZ_MAX_BODY  // bodies, world included
#define GN_NJ MZ_MAX_JNT
#define GN_NV MZ_MAX_DOF
#define GN_NQ MZ_MAX_Q
#define GN_NG MZ_MAX_GEOM
#define GN_NI 192   // collision items: geom pairs + one "maze boxes" item per moving geom
#define GN_POOL 96  // contacts found by the narrow phase in one evaluation (active or within the margin)
#define GN_NC 64    // simultaneous ACTIVE contacts (rows of the solver)
#define GN_NL 24    // joint-limit rows
#define GN_KEY 64   // pool key = item * GN_KEY + emission index inside the item

// collision item kinds (geom1 -> geom2 in MuJoCo's order: by type, then by id; the maze boxes are world geoms with ids below
// every robot geom's)
enum { GK_PLANE_SPHERE = 0, GK_PLANE_CAPSULE, GK_PLANE_BOX, GK_SPHERE_SPHERE, GK_SPHERE_CAPSULE, GK_SPHERE_BOX, GK_CAPSULE_BOX, GK_BOX_BOX,
       GK_SPHERE_WALL, GK_CAPSULE_WALL, GK_WALL_BOX, GK_CAPSULE_CAPSULE };
// step shapes (mz_model.step_kind; 0 = the robot family's)
#define GN_STEP_MOTORS 1  // ant.py:61-73 / swimmer.py:37-48: clamped motors, frame_skip x mj_step, forward reward, control cost
#define GN_STEP_POINT 2   // point.py:44-61: heading += a[1] (wrapped), teleport a[0] along it, velocity clip, no control, no inner reward

struct GenPair { double margin, gap, mu, K, B, solimp[5], tran; int condim, pad; };
struct GenItem { int kind, g1, g2, b1, b2, pad; GenPair P; };  // g = -1: the maze boxes (world body)

struct GenDev {
  mz_model m;   // the compiled model as it is (float64): read with uniform (scalar) loads
  TaskDev task;
  MazeDev maze;
  int nitem, step_kind;
  int servos, pad_servos;  // some actuator has a position / velocity term (mz_model.act_biasprm)
  GenItem item[GN_NI];
  double lim_K[GN_NJ], lim_B[GN_NJ];
  int max_iter, ls_iter;
  double tol, inv_scale;
  int dof_parent[GN_NV];
  int limj[GN_NJ], nlimj;            // the limited hinge / slide joints (candidates of the limit rows)
  int body_depth[GN_NB], max_depth;
  unsigned subtree[GN_NB];  // bit b: body b is body p itself or one of its descendants  // tree level of every body (world 0): the kinematic / RNE passes walk level by level, the bodies of a level side by side
  // the Point's manual wall bounce (maze_env.py:451-464): point_bounce reads these three names
  int nseg, obs_extra;
  double seg[MZ_MAX_SEG][4], restitution;
};

static inline int gen_fail(char* err, int n, const char* msg) {
  if (err && n > 0) { strncpy(err, msg, (size_t)n - 1); err[n - 1] = 0; }
  return MZ_ERR_UNSUPPORTED;
}

static inline void gen_pair(GenPair* p, const mz_model* m, const double* f1, const double* sr1, const double* si1, double mg1, double gp1, int cd1,
                            const double* f2, const double* sr2, const double* si2, double mg2, double gp2, int cd2, double tran) {
  double sr[2], si[5];
  for (int k = 0; k < 2; k++) sr[k] = 0.5 * (sr1[k] + sr2[k]);
  for (int k = 0; k < 5; k++) si[k] = 0.5 * (si1[k] + si2[k]);
  p->margin = fmax(mg1, mg2); p->gap = fmax(gp1, gp2); p->mu = fmax(f1[0], f2[0]);
  const double tc = fmax(sr[0], 2.0 * m->timestep), dr = sr[1], dmax = si[1];
  p->K = 1.0 / fmax(1e-15, dmax * dmax * tc * tc * dr * dr);
  p->B = 2.0 / fmax(1e-15, dmax * tc);
  for (int k = 0; k < 5; k++) p->solimp[k] = si[k];
  p->condim = cd1 > cd2 ? cd1 : cd2; p->pad = 0; p->tran = tran;
}

// does the model need this engine (no specialised kernel steps it)?  mazestep.hip asks before it dispatches.
static inline int gen_model_needs_general_engine(const mz_model* m) {
  if (m->robot == MZ_ROBOT_GENERIC || m->engine == 1 || !m->integrator_rk4) return 1;
  for (int j = 0; j < m->njnt; j++) if (m->jnt_type[j] == MZ_JNT_BALL || m->jnt_stiffness[j] != 0.0) return 1;  // SPIN plates; joint springs
  for (int a = 0; a < m->nu; a++) if (m->act_gainprm[a] != 1.0 || m->act_biasprm[a][0] != 0.0 || m->act_biasprm[a][1] != 0.0 || m->act_biasprm[a][2] != 0.0) return 1;  // servos
  if (m->nblock > 3) return 1;
  for (int k = 0; k < m->nblock; k++)  // a three-slide block (MultiFall's XYZ block) has a specialised kernel for the one-block ant only
    if (m->body_jntnum[m->block_bodyid[k]] != 2 && !(m->robot == MZ_ROBOT_ANT && m->nblock == 1)) return 1;
  for (int a = 1; a < m->nblock; a++)  // blocks of several sizes in one maze (half blocks next to full ones)
    for (int k = 0; k < 3; k++) if (m->geom_size[m->block_geomid[a]][k] != m->geom_size[m->block_geomid[0]][k]) return 1;
  return 0;
}

static inline int gen_dev_from_model(GenDev* g, const mz_model* m, char* err, int errlen) {
  memset(g, 0, sizeof(*g));
  if (m->nbody > GN_NB || m->njnt > GN_NJ || m->nv > GN_NV || m->nq > GN_NQ || m->ngeom > GN_NG || m->nbody < 2)
    return gen_fail(err, errlen, "general engine: at most 23 bodies, 24 joints / dofs, 28 coordinates, 24 geoms");
  if (m->geom_type[0] != MZ_GEOM_PLANE || m->geom_bodyid[0] != 0) return gen_fail(err, errlen, "general engine: geom 0 must be the floor plane on the world body");
  for (int b = 1; b < m->nbody; b++)
    if (m->body_parent[b] >= b) return gen_fail(err, errlen, "general engine: bodies must be listed parents first");
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_type[j] == MZ_JNT_BALL && m->jnt_limited[j]) return gen_fail(err, errlen, "general engine: limited ball joints are not implemented");
    if (m->jnt_type[j] == MZ_JNT_FREE && m->body_jntnum[m->jnt_bodyid[j]] != 1) return gen_fail(err, errlen, "general engine: a free joint must be its body's only joint");
  }
  g->step_kind = m->step_kind ? m->step_kind : (m->robot == MZ_ROBOT_POINT ? GN_STEP_POINT : GN_STEP_MOTORS);
  if (g->step_kind == GN_STEP_POINT && (m->nq < 3 || m->jnt_type[0] != MZ_JNT_SLIDE || m->jnt_type[1] != MZ_JNT_SLIDE || m->jnt_type[2] != MZ_JNT_HINGE || m->nu < 2))
    return gen_fail(err, errlen, "general engine: the Point's step shape needs a slide-x, slide-y, hinge-z root and two action entries (point.py:44-61)");
  if (m->manual_collision && m->nseg <= 0) return gen_fail(err, errlen, "general engine: MANUAL_COLLISION without wall segments");
  g->m = *m;
  task_dev_from_model(&g->task, m);
  maze_dev_from_model(&g->maze, m);
  // ---- collision items, in the order the oracle walks them (mzo_physics.c collision()): the explicit geom pairs g1 < g2, then
  // every moving geom against the maze boxes
  int n = 0;
  for (int a = 0; a < m->ngeom && !m->collision_predefined; a++)  // (collision="predefined": no dynamic pairs at all, swimmer.xml:3)
    for (int b = a + 1; b < m->ngeom; b++) {
      const int ba = m->geom_bodyid[a], bb = m->geom_bodyid[b];
      if (ba == bb) continue;
      if (!((m->geom_contype[a] & m->geom_conaffinity[b]) || (m->geom_contype[b] & m->geom_conaffinity[a]))) continue;
      if (ba != 0 && bb != 0 && (m->body_parent[ba] == bb || m->body_parent[bb] == ba)) continue;  // parent-child filter (not against the world)
      int g1 = a, g2 = b;
      if (m->geom_type[g1] > m->geom_type[g2]) { g1 = b; g2 = a; }  // MuJoCo orders a pair by geom type
      const int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
      int kind = -1;
      if (t1 == MZ_GEOM_PLANE) kind = t2 == MZ_GEOM_SPHERE ? GK_PLANE_SPHERE : t2 == MZ_GEOM_CAPSULE ? GK_PLANE_CAPSULE : t2 == MZ_GEOM_BOX ? GK_PLANE_BOX : -1;
      else if (t1 == MZ_GEOM_SPHERE) kind = t2 == MZ_GEOM_SPHERE ? GK_SPHERE_SPHERE : t2 == MZ_GEOM_CAPSULE ? GK_SPHERE_CAPSULE : t2 == MZ_GEOM_BOX ? GK_SPHERE_BOX : -1;
      else if (t1 == MZ_GEOM_CAPSULE) kind = t2 == MZ_GEOM_BOX ? GK_CAPSULE_BOX : t2 == MZ_GEOM_CAPSULE ? GK_CAPSULE_CAPSULE : -1;
      else if (t1 == MZ_GEOM_BOX) kind = t2 == MZ_GEOM_BOX ? GK_BOX_BOX : -1;
      if (kind < 0) return gen_fail(err, errlen, "general engine: a geom pair that can collide has no narrow phase here (geom types: plane, sphere, capsule, box)");
      if (n >= GN_NI) return gen_fail(err, errlen, "general engine: too many geom pairs");
      GenItem& it = g->item[n++];
      it.kind = kind; it.g1 = g1; it.g2 = g2; it.b1 = m->geom_bodyid[g1]; it.b2 = m->geom_bodyid[g2]; it.pad = 0;
      gen_pair(&it.P, m, m->geom_friction[g1], m->geom_solref[g1], m->geom_solimp[g1], m->geom_margin[g1], m->geom_gap[g1], m->geom_condim[g1],
               m->geom_friction[g2], m->geom_solref[g2], m->geom_solimp[g2], m->geom_margin[g2], m->geom_gap[g2], m->geom_condim[g2],
               m->body_invweight0[it.b1][0] + m->body_invweight0[it.b2][0]);
      if (it.P.condim != 1 && it.P.condim != 3) return gen_fail(err, errlen, "general engine: condim must be 1 or 3");
    }
  for (int a = 1; a < m->ngeom && !m->collision_predefined; a++) {
    if (m->geom_bodyid[a] == 0) continue;
    if (!((m->geom_contype[a] & m->wall_conaffinity) || (m->wall_contype & m->geom_conaffinity[a]))) continue;
    const int t = m->geom_type[a];
    if (t != MZ_GEOM_SPHERE && t != MZ_GEOM_CAPSULE && t != MZ_GEOM_BOX) return gen_fail(err, errlen, "general engine: geom types are plane, sphere, capsule, box");
    if (n >= GN_NI) return gen_fail(err, errlen, "general engine: too many geom pairs");
    GenItem& it = g->item[n++];
    const bool box = t == MZ_GEOM_BOX;  // a maze box is geom1 against another box (same type, lower id), geom2 against spheres / capsules
    it.kind = t == MZ_GEOM_SPHERE ? GK_SPHERE_WALL : t == MZ_GEOM_CAPSULE ? GK_CAPSULE_WALL : GK_WALL_BOX;
    it.g1 = box ? -1 : a; it.g2 = box ? a : -1; it.b1 = box ? 0 : m->geom_bodyid[a]; it.b2 = box ? m->geom_bodyid[a] : 0; it.pad = 0;
    gen_pair(&it.P, m, m->geom_friction[a], m->geom_solref[a], m->geom_solimp[a], m->geom_margin[a], m->geom_gap[a], m->geom_condim[a],
             m->wall_friction, m->wall_solref, m->wall_solimp, m->wall_margin, m->wall_gap, m->wall_condim,
             m->body_invweight0[m->geom_bodyid[a]][0] + m->body_invweight0[0][0]);
    if (it.P.condim != 1 && it.P.condim != 3) return gen_fail(err, errlen, "general engine: condim must be 1 or 3");
  }
  g->nitem = n;
  for (int j = 0; j < m->njnt; j++) {
    const double tc = fmax(m->jnt_solref[j][0], 2.0 * m->timestep), dr = m->jnt_solref[j][1], dmax = m->jnt_solimp[j][1];
    g->lim_K[j] = 1.0 / fmax(1e-15, dmax * dmax * tc * tc * dr * dr);
    g->lim_B[j] = 2.0 / fmax(1e-15, dmax * tc);
  }
  for (int i = 0; i < m->nv; i++) {  // previous dof up the tree, -1 at a root
    const int b = m->dof_bodyid[i];
    int p = -1;
    if (i > m->body_dofadr[b]) p = i - 1;
    else
      for (int a = m->body_parent[b]; a > 0; a = m->body_parent[a])
        if (m->body_dofnum[a] > 0) { p = m->body_dofadr[a] + m->body_dofnum[a] - 1; break; }
    g->dof_parent[i] = p;
  }
  g->nlimj = 0;
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_limited[j] && (m->jnt_type[j] == MZ_JNT_HINGE || m->jnt_type[j] == MZ_JNT_SLIDE)) g->limj[g->nlimj++] = j;
  g->servos = 0;
  for (int a = 0; a < m->nu; a++) if (m->act_biasprm[a][1] != 0.0 || m->act_biasprm[a][2] != 0.0) g->servos = 1;
  g->max_depth = 0;
  g->body_depth[0] = 0;
  for (int b = 0; b < GN_NB; b++) g->subtree[b] = 0u;
  for (int b = 1; b < m->nbody; b++) for (int a = b; a > 0; a = m->body_parent[a]) g->subtree[a] |= 1u << b;
  for (int b = 1; b < m->nbody; b++) {  // (a parent's index is below its children's: mjcf.py / MuJoCo body order)
    if (m->body_parent[b] >= b) return gen_fail(err, errlen, "general engine: bodies must be ordered parents first");
    g->body_depth[b] = g->body_depth[m->body_parent[b]] + 1;
    if (g->body_depth[b] > g->max_depth) g->max_depth = g->body_depth[b];
  }
  g->max_iter = 100; g->ls_iter = 50; g->tol = 1e-10;
  g->inv_scale = 1.0 / (m->meaninertia * (m->nv > 1 ? m->nv : 1));
  g->nseg = m->manual_collision ? m->nseg : 0;
  for (int k = 0; k < m->nseg && k < MZ_MAX_SEG; k++) for (int c = 0; c < 4; c++) g->seg[k][c] = m->seg[k][c];
  g->restitution = m->restitution;
  g->obs_extra = 3 * ((m->observe_balls ? m->nball : 0) + (m->observe_blocks ? m->nblock : 0));
  if (m->nq_robot < 3 || m->obs_dim != m->nq_robot + m->nv_robot + 1 + g->obs_extra + (m->top_down_view ? MZ_VIEW_DIM : 0))
    return gen_fail(err, errlen, "general engine: obs_dim must be nq_robot + nv_robot + 1 + 3 per observed ball / block (+ the top-down view)");
  if (m->nq_robot + m->nv_robot + 1 + g->obs_extra > MZ_MAX_OBS) return gen_fail(err, errlen, "general engine: observation wider than MZ_MAX_OBS");
  return MZ_OK;
}

// The model's tree topology as the kernel walks it, staged in LDS once per step (gen_load_topology): every one of these small integers is
// the head of a dependent chain (body -> parent -> joint -> dof ...), and from the constant block in global memory each link cost a
// full L2 round trip on a lone wavefront.  int8: all counts are <= MZ_MAX_* = 24 (28 for qpos addresses); -1 = none.
struct GenTopo {
  int8_t body_parent[GN_NB], body_jntadr[GN_NB], body_jntnum[GN_NB], body_dofadr[GN_NB], body_dofnum[GN_NB], body_depth[GN_NB];
  int8_t jnt_type[GN_NJ], jnt_dofadr[GN_NJ], jnt_qposadr[GN_NJ], jnt_bodyid[GN_NJ];
  int8_t dof_bodyid[GN_NV], dof_parent[GN_NV];
  int8_t geom_bodyid[GN_NG];
  uint32_t subtree[GN_NB];  // bit b: body b belongs to the subtree of this body (itself included): composite inertias and subtree forces are
                            // summed straight from the bodies' own values by whoever needs them — no up-pass over the tree
};
struct alignas(16) GenScratch {
  GenTopo tp;
  double qpos[GN_NQ], qvel[GN_NV], warm[GN_NV], fact[GN_NV], x0q[GN_NQ], x0v[GN_NV], accv[GN_NV], accf[GN_NV], dxv[GN_NV];
  double qacc[GN_NV], qas[GN_NV], qfs[GN_NV], bias[GN_NV], passive[GN_NV];
  double xpos[GN_NB][3], xquat[GN_NB][4], xmat[GN_NB][9], xipos[GN_NB][3];
  double cinert[GN_NB][10], cvel[GN_NB][6], cacc[GN_NB][6], cfrc[GN_NB][6], ffl[GN_NB][6];  // (cfrc: each body's OWN inertial + velocity-product force)
  double xanchor[GN_NJ][3], xaxis[GN_NJ][3];
  double gpos[GN_NG][3], gmat[GN_NG][9];
  double S[GN_NV][6], refpoint[3];
  double M[GN_NV][GN_NV], H[GN_NV][GN_NV];
  // contacts: found by the items into a pool (arrival order), then the active ones compacted in the oracle's order.  The pool is dead
  // once compacted: the solver's per-contact iterates share its bytes
  int npool, ncon, nlim, status, iters, pad;
  union {
    struct { int pkey[GN_POOL]; double pdist[GN_POOL], ppos[GN_POOL][3], pnrm[GN_POOL][3], phint[GN_POOL][3]; };
    struct {
      double cu[GN_NC][3], cjv[GN_NC][3];  // J qacc - aref and J search of the Newton iterate
      double cg[GN_NC][3], cW[GN_NC][5];   // gradient and Hessian weights of the contact's pyramid there (gen_contact_eval)
    };
  };
  int citem[GN_NC];
  double cdist[GN_NC], cpos[GN_NC][3], cnrm[GN_NC][3], chint[GN_NC][3];
  double caref[GN_NC][3], cD[GN_NC];
  int ldof[GN_NL], lflag[2 * GN_NJ];
  double lsign[GN_NL], lD[GN_NL], laref[GN_NL], ljar[GN_NL], ljv[GN_NL];
  double grad[GN_NV], search[GN_NV], Mx[GN_NV], Ms[GN_NV], red[8];
  double ctmp[GN_NV], cdinv[GN_NV];  // gen_chol_solve: the column in flight, reciprocals of the factor's diagonal
#ifdef MZ_EXP_GENPROF
  unsigned long long prof[20], prof_t0;  // phase timers of the instrumented build (tools/exp_general_prof.sh)
#endif
  // LAST: the contacts' Jacobian rows [contact][normal, mu t1, mu t2] at a row stride of the MODEL's nv (gen_cj).  The step kernel
  // allocates the block only as far as 3 nv GN_NC doubles of it (gen_scratch_bytes): 30 KB for a 20-dof model instead of 37 — what
  // keeps two envs per CU up to 22 dofs (generic_kernels.hip)
  double cJf[GN_NC * 3 * GN_NV];
};
MZ_HD size_t gen_scratch_bytes(int nv) { return (offsetof(GenScratch, cJf) + sizeof(double) * 3 * (size_t)nv * GN_NC + 15) / 16 * 16; }
#if defined(MZ_EXP_GENPROF) && defined(__HIP_DEVICE_COMPILE__)
#define GEN_TICK(id) do { if (cx.lane0() == 0) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); s.prof[id] += now_ - s.prof_t0; s.prof_t0 = now_; } } while (0)
#else
#define GEN_TICK(id) do {} while (0)
#endif

MZ_HD double* gen_cj(GenScratch& s, int nv, int k, int a) { return s.cJf + (size_t)(3 * k + a) * nv; }
MZ_HD const double* gen_cj(const GenScratch& s, int nv, int k, int a) { return s.cJf + (size_t)(3 * k + a) * nv; }

template <class C>
MZ_HD void gen_load_topology(const C& cx, const GenDev& K, GenScratch& s) {
  const mz_model& m = K.m;
  MZ_FOR(b, m.nbody) {
    s.tp.body_parent[b] = (int8_t)m.body_parent[b]; s.tp.body_jntadr[b] = (int8_t)m.body_jntadr[b]; s.tp.body_jntnum[b] = (int8_t)m.body_jntnum[b];
    s.tp.body_dofadr[b] = (int8_t)m.body_dofadr[b]; s.tp.body_dofnum[b] = (int8_t)m.body_dofnum[b]; s.tp.body_depth[b] = (int8_t)K.body_depth[b];
    s.tp.subtree[b] = K.subtree[b];
  }
  MZ_FOR(j, m.njnt) {
    s.tp.jnt_type[j] = (int8_t)m.jnt_type[j]; s.tp.jnt_dofadr[j] = (int8_t)m.jnt_dofadr[j]; s.tp.jnt_qposadr[j] = (int8_t)m.jnt_qposadr[j];
    s.tp.jnt_bodyid[j] = (int8_t)m.jnt_bodyid[j];
  }
  MZ_FOR(i, m.nv) { s.tp.dof_bodyid[i] = (int8_t)m.dof_bodyid[i]; s.tp.dof_parent[i] = (int8_t)K.dof_parent[i]; }
  MZ_FOR(g, m.ngeom) s.tp.geom_bodyid[g] = (int8_t)m.geom_bodyid[g];
  cx.sync();
}

// slot of the contact pool (LDS counter; a wavefront's arrival order is whatever it is — the compaction sorts by key)
MZ_HD int gen_take(int* counter) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(counter, 1);
#else
  return (*counter)++;
#endif
}

// ------------------------------------------------------------------ small float64 helpers
MZ_HD double gd_dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
MZ_HD void gd_cross(double* r, const double* a, const double* b) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
MZ_HD void gd_mulmat(double* r, const double* m, const double* v) {
  const double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
MZ_HD void gd_mulmatT(double* r, const double* m, const double* v) {
  const double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2], z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
MZ_HD void gd_quat_mul(double* r, const double* a, const double* b) {
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
MZ_HD void gd_quat_norm(double* q) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-15) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
MZ_HD void gd_quat2mat(double* m, const double* q) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
MZ_HD void gd_inertia_mul(double* r, const double* I, const double* v) {  // compact spatial inertia [m, h(3), Ibar(6)] times motion vector
  const double* h = I + 1;
  const double* J = I + 4;
  double a[3], b[3];
  gd_cross(a, h, v + 3);
  gd_cross(b, v, h);
  r[0] = J[0] * v[0] + J[3] * v[1] + J[4] * v[2] + a[0];
  r[1] = J[3] * v[0] + J[1] * v[1] + J[5] * v[2] + a[1];
  r[2] = J[4] * v[0] + J[5] * v[1] + J[2] * v[2] + a[2];
  r[3] = I[0] * v[3] + b[0]; r[4] = I[0] * v[4] + b[1]; r[5] = I[0] * v[5] + b[2];
}
MZ_HD void gd_motion_cross(double* r, const double* v, const double* s) {
  double a[3], b[3], c[3];
  gd_cross(a, v, s); gd_cross(b, v, s + 3); gd_cross(c, v + 3, s);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
MZ_HD void gd_force_cross(double* r, const double* v, const double* f) {
  double a[3], b[3], c[3];
  gd_cross(a, v, f); gd_cross(b, v + 3, f + 3); gd_cross(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
// dot products of run-time length over LDS operands: four partial sums, so that the loads of four terms are in flight together (a
// rolled loop with one accumulator pays one LDS round trip per term on a lone wavefront)
MZ_HD double gd_dotn(const double* a, const double* b, int n) {
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  int k = 0;
  for (; k + 4 <= n; k += 4) { t0 += a[k] * b[k]; t1 += a[k + 1] * b[k + 1]; t2 += a[k + 2] * b[k + 2]; t3 += a[k + 3] * b[k + 3]; }
  for (; k < n; k++) t0 += a[k] * b[k];
  return (t0 + t1) + (t2 + t3);
}
// the three rows of a contact (normal, mu t1, mu t2: consecutive in the Jacobian store at a stride of n) against one vector — its loads shared
MZ_HD void gd_dot3n(const double* J, const double* x, int n, double* out) {
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, c0 = 0.0, c1 = 0.0;
  const double* J1 = J + n;
  const double* J2 = J + 2 * n;
  int k = 0;
  for (; k + 2 <= n; k += 2) {
    const double x0 = x[k], x1 = x[k + 1];
    a0 += J[k] * x0; a1 += J[k + 1] * x1; b0 += J1[k] * x0; b1 += J1[k + 1] * x1; c0 += J2[k] * x0; c1 += J2[k + 1] * x1;
  }
  if (k < n) { const double x0 = x[k]; a0 += J[k] * x0; b0 += J1[k] * x0; c0 += J2[k] * x0; }
  out[0] = a0 + a1; out[1] = b0 + b1; out[2] = c0 + c1;
}
MZ_HD double gd_dotn_diff(const double* a, const double* x, const double* y, int n) {  // a . (x - y)
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  int k = 0;
  for (; k + 4 <= n; k += 4) { t0 += a[k] * (x[k] - y[k]); t1 += a[k + 1] * (x[k + 1] - y[k + 1]); t2 += a[k + 2] * (x[k + 2] - y[k + 2]); t3 += a[k + 3] * (x[k + 3] - y[k + 3]); }
  for (; k < n; k++) t0 += a[k] * (x[k] - y[k]);
  return (t0 + t1) + (t2 + t3);
}
MZ_HD double gd_dot6(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }
MZ_HD double gd_impedance(const double* si, double x) {
  const double d0 = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
  if (d0 == dmax || width <= 1e-15) return 0.5 * (d0 + dmax);
  const double xn = x / width;
  if (xn >= 1.0) return dmax;
  if (xn <= 0.0) return d0;
  double y;
  if (power <= 1.0 + 1e-12) y = xn;
  else if (power == 2.0) y = xn <= mid ? xn * xn / mid : 1.0 - (1.0 - xn) * (1.0 - xn) / (1.0 - mid);  // MuJoCo's default power: no pow()
  else if (xn <= mid) y = pow(xn, power) / pow(mid, power - 1.0);
  else y = 1.0 - pow(1.0 - xn, power) / pow(1.0 - mid, power - 1.0);
  return d0 + y * (dmax - d0);
}

// ------------------------------------------------------------------ kinematics (parents before children, level by level)
// pose of body b from its parent's (mj_kinematics); the bodies of one tree level are independent of each other
MZ_HD void gen_kin_body(const GenDev& K, GenScratch& s, int b) {
  const mz_model& m = K.m;
  const int p = s.tp.body_parent[b], j0 = s.tp.body_jntadr[b], jn = s.tp.body_jntnum[b];
  double pos[3], quat[4];
  if (jn == 1 && s.tp.jnt_type[j0] == MZ_JNT_FREE) {
    const int qa = s.tp.jnt_qposadr[j0];
    for (int k = 0; k < 3; k++) pos[k] = s.qpos[qa + k];
    for (int k = 0; k < 4; k++) quat[k] = s.qpos[qa + 3 + k];
    gd_quat_norm(quat);
    for (int k = 0; k < 3; k++) { s.xanchor[j0][k] = pos[k]; s.xaxis[j0][k] = m.jnt_axis[j0][k]; }
  } else {
    double t[3];
    gd_mulmat(t, s.xmat[p], m.body_pos[b]);
    for (int k = 0; k < 3; k++) pos[k] = s.xpos[p][k] + t[k];
    gd_quat_mul(quat, s.xquat[p], m.body_quat[b]);
    for (int j = j0; j < j0 + jn; j++) {
      double mat[9], axis[3], anchor[3];
      gd_quat2mat(mat, quat);
      gd_mulmat(axis, mat, m.jnt_axis[j]);
      gd_mulmat(anchor, mat, m.jnt_pos[j]);
      for (int k = 0; k < 3; k++) anchor[k] += pos[k];
      if (s.tp.jnt_type[j] == MZ_JNT_SLIDE) {
        const double q = s.qpos[s.tp.jnt_qposadr[j]] - m.qpos0[s.tp.jnt_qposadr[j]];
        for (int k = 0; k < 3; k++) pos[k] += axis[k] * q;
      } else {  // hinge: rotate about the joint axis through the anchor; ball: the coordinates ARE the relative quaternion
        double ql[4], qn[4], v[3];
        if (s.tp.jnt_type[j] == MZ_JNT_BALL) {
          for (int k = 0; k < 4; k++) ql[k] = s.qpos[s.tp.jnt_qposadr[j] + k];
          gd_quat_norm(ql);
        } else {
          const double q = s.qpos[s.tp.jnt_qposadr[j]] - m.qpos0[s.tp.jnt_qposadr[j]];
          double sh, ch;
          sincos(0.5 * q, &sh, &ch);  // (one argument reduction for the two)
          ql[0] = ch; ql[1] = m.jnt_axis[j][0] * sh; ql[2] = m.jnt_axis[j][1] * sh; ql[3] = m.jnt_axis[j][2] * sh;
        }
        gd_quat_mul(qn, quat, ql);
        for (int k = 0; k < 4; k++) quat[k] = qn[k];
        gd_quat2mat(mat, quat);
        gd_mulmat(v, mat, m.jnt_pos[j]);
        for (int k = 0; k < 3; k++) pos[k] = anchor[k] - v[k];
      }
      for (int k = 0; k < 3; k++) { s.xaxis[j][k] = axis[k]; s.xanchor[j][k] = anchor[k]; }
    }
  }
  gd_quat_norm(quat);
  for (int k = 0; k < 3; k++) s.xpos[b][k] = pos[k];
  for (int k = 0; k < 4; k++) s.xquat[b][k] = quat[k];
  gd_quat2mat(s.xmat[b], quat);
  double t[3];
  gd_mulmat(t, s.xmat[b], m.body_ipos[b]);
  for (int k = 0; k < 3; k++) s.xipos[b][k] = s.xpos[b][k] + t[k];
}
// kinematics of the whole tree, level by level (GenDev::body_depth): ends behind a fence
template <class C>
MZ_HD void gen_kinematics(const C& cx, const GenDev& K, GenScratch& s) {
  const mz_model& m = K.m;
  MZ_FOR(one, 1) {
    for (int k = 0; k < 3; k++) s.xpos[0][k] = 0.0;
    s.xquat[0][0] = 1.0; s.xquat[0][1] = s.xquat[0][2] = s.xquat[0][3] = 0.0;
    gd_quat2mat(s.xmat[0], s.xquat[0]);
  }
  for (int d = 1; d <= K.max_depth; d++) {
    cx.sync();
    MZ_FOR(b, m.nbody) if (s.tp.body_depth[b] == d) gen_kin_body(K, s, b);
  }
  cx.sync();
  MZ_FOR(k, 3) s.refpoint[k] = s.xpos[1][k];
  cx.sync();
}

// geom poses, motion axes, body spatial inertias about the reference point (one item per geom / joint / body)
MZ_HD void gen_geom_item(const GenDev& K, GenScratch& s, int g) {
  const mz_model& m = K.m;
  const int b = s.tp.geom_bodyid[g];
  double t[3], q[4];
  gd_mulmat(t, s.xmat[b], m.geom_pos[g]);
  for (int k = 0; k < 3; k++) s.gpos[g][k] = s.xpos[b][k] + t[k];
  gd_quat_mul(q, s.xquat[b], m.geom_quat[g]);
  gd_quat2mat(s.gmat[g], q);
}
MZ_HD void gen_axis_item(const GenDev& K, GenScratch& s, int j) {
  const int b = s.tp.jnt_bodyid[j], d0 = s.tp.jnt_dofadr[j];
  const double* c = s.refpoint;
  double off[3];
  if (s.tp.jnt_type[j] == MZ_JNT_FREE || s.tp.jnt_type[j] == MZ_JNT_BALL) {
    // the rotational dofs are the angular velocity in the child frame: rotations about the body's own axes through the anchor
    const int r0 = s.tp.jnt_type[j] == MZ_JNT_FREE ? d0 + 3 : d0;
    const double* anchor = s.tp.jnt_type[j] == MZ_JNT_FREE ? s.xpos[b] : s.xanchor[j];
    for (int k = 0; k < 3; k++) {
      if (s.tp.jnt_type[j] == MZ_JNT_FREE) {
        for (int e = 0; e < 6; e++) s.S[d0 + k][e] = 0.0;
        s.S[d0 + k][3 + k] = 1.0;
      }
      const double ax[3] = {s.xmat[b][k], s.xmat[b][3 + k], s.xmat[b][6 + k]};
      for (int e = 0; e < 3; e++) { off[e] = c[e] - anchor[e]; s.S[r0 + k][e] = ax[e]; }
      gd_cross(s.S[r0 + k] + 3, ax, off);
    }
  } else if (s.tp.jnt_type[j] == MZ_JNT_SLIDE) {
    for (int e = 0; e < 3; e++) { s.S[d0][e] = 0.0; s.S[d0][3 + e] = s.xaxis[j][e]; }
  } else {
    for (int e = 0; e < 3; e++) { off[e] = c[e] - s.xanchor[j][e]; s.S[d0][e] = s.xaxis[j][e]; }
    gd_cross(s.S[d0] + 3, s.xaxis[j], off);
  }
}
MZ_HD void gen_inertia_item(const GenDev& K, GenScratch& s, int b) {
  const mz_model& m = K.m;
  const double* I6 = m.body_inertia[b];
  const double Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
  const double* R = s.xmat[b];
  double tmp[9], Iw[9], r[3];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) tmp[3 * i + k] = R[3 * i] * Ib[k] + R[3 * i + 1] * Ib[3 + k] + R[3 * i + 2] * Ib[6 + k];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) Iw[3 * i + k] = tmp[3 * i] * R[3 * k] + tmp[3 * i + 1] * R[3 * k + 1] + tmp[3 * i + 2] * R[3 * k + 2];
  for (int k = 0; k < 3; k++) r[k] = s.xipos[b][k] - s.refpoint[k];
  const double ms = m.body_mass[b], rr = gd_dot3(r, r);
  double* I = s.cinert[b];
  I[0] = ms; I[1] = ms * r[0]; I[2] = ms * r[1]; I[3] = ms * r[2];
  I[4] = Iw[0] + ms * (rr - r[0] * r[0]); I[5] = Iw[4] + ms * (rr - r[1] * r[1]); I[6] = Iw[8] + ms * (rr - r[2] * r[2]);
  I[7] = Iw[1] - ms * r[0] * r[1]; I[8] = Iw[2] - ms * r[0] * r[2]; I[9] = Iw[5] - ms * r[1] * r[2];
}
// row i of the joint-space inertia: M[i][j] for the dofs j on the path from i to its root
MZ_HD void gen_mass_item(const GenDev& K, GenScratch& s, int i) {
  const mz_model& m = K.m;
  double F[6];
  for (int j = 0; j < m.nv; j++) s.M[i][j] = 0.0;
  double crb[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // composite inertia of the dof's subtree (mj_crb), summed here from the bodies' own
  for (uint32_t mk = s.tp.subtree[s.tp.dof_bodyid[i]]; mk; mk &= mk - 1) {
    const double* I = s.cinert[__builtin_ctz(mk)];
    for (int k = 0; k < 10; k++) crb[k] += I[k];
  }
  gd_inertia_mul(F, crb, s.S[i]);
  s.M[i][i] = gd_dot6(s.S[i], F) + m.dof_armature[i];
  for (int j = s.tp.dof_parent[i]; j >= 0; j = s.tp.dof_parent[j]) s.M[i][j] = gd_dot6(s.S[j], F);
}

// ------------------------------------------------------------------ narrow phase
// two spheres (mjraw_SphereSphere): normal from the first to the second; returns the number of contacts emitted
template <class E>
MZ_HD int gen_sphere_pair(const double* c1, double r1, const double* c2, double r2, double margin, E& emit) {
  const double dv[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]}, cd = sqrt(gd_dot3(dv, dv)), dist = cd - r1 - r2;
  if (dist > margin) return 0;
  double nrm[3] = {1.0, 0.0, 0.0}, pos[3];
  if (!(cd < 1e-15)) for (int k = 0; k < 3; k++) nrm[k] = dv[k] / cd;
  for (int k = 0; k < 3; k++) pos[k] = c1[k] + nrm[k] * (r1 + 0.5 * dist);
  emit(dist, pos, nrm, nullptr);
  return 1;
}
// capsule vs capsule (mjraw_CapsuleCapsule as restated in oracle/mzo_physics.c capsule_capsule): nearest points of the axis segments, clamped
// one coordinate after the other, then sphere-sphere; parallel axes: the segment ends, up to two contacts
template <class E>
MZ_HD void gen_capsule_vs_capsule(const double* pos1, const double* mat1, double r1, double hl1, const double* pos2, const double* mat2, double r2,
                                  double hl2, double margin, E& emit) {
  const double a1[3] = {mat1[2] * hl1, mat1[5] * hl1, mat1[8] * hl1}, a2[3] = {mat2[2] * hl2, mat2[5] * hl2, mat2[8] * hl2};
  const double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  const double ma = gd_dot3(a1, a1), mb = -gd_dot3(a1, a2), mc = gd_dot3(a2, a2), u = -gd_dot3(a1, dif), v = gd_dot3(a2, dif), det = ma * mc - mb * mb;
  auto clamp1 = [](double x) { return x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x); };
  auto pair_at = [&](double x1, double x2) {
    const double v1[3] = {pos1[0] + a1[0] * x1, pos1[1] + a1[1] * x1, pos1[2] + a1[2] * x1}, v2[3] = {pos2[0] + a2[0] * x2, pos2[1] + a2[1] * x2, pos2[2] + a2[2] * x2};
    return gen_sphere_pair(v1, r1, v2, r2, margin, emit);
  };
  if (fabs(det) >= 1e-15) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > 1.0) { x1 = 1.0; x2 = (v - mb) / mc; }
    else if (x1 < -1.0) { x1 = -1.0; x2 = (v + mb) / mc; }
    if (x2 > 1.0) { x2 = 1.0; x1 = clamp1((u - mb) / ma); }
    else if (x2 < -1.0) { x2 = -1.0; x1 = clamp1((u + mb) / ma); }
    pair_at(x1, x2);
    return;
  }
  int n = pair_at(1.0, clamp1((v - mb) / mc));
  n += pair_at(-1.0, clamp1((v + mb) / mc));
  if (n >= 2) return;
  n += pair_at(clamp1((u - mb) / ma), 1.0);
  if (n >= 2) return;
  pair_at(clamp1((u + mb) / ma), -1.0);
}
// sphere (centre c in box coordinates) vs axis-aligned box; normal from the sphere to the box (mjraw_SphereBox)
MZ_HD bool gen_sphere_box(const double* c, double r, const double* bs, double margin, double* dist, double* pos, double* nrm) {
  double q[3], dd;
  bool inside = true;
  for (int k = 0; k < 3; k++) { q[k] = fmin(fmax(c[k], -bs[k]), bs[k]); if (q[k] != c[k]) inside = false; }
  if (!inside) {
    const double v[3] = {q[0] - c[0], q[1] - c[1], q[2] - c[2]};
    dd = sqrt(gd_dot3(v, v));
    if (dd - r > margin) return false;
    for (int k = 0; k < 3; k++) nrm[k] = v[k] / dd;
    dd -= r;
  } else {
    int kb = 0; double best = 1e30;
    for (int k = 0; k < 3; k++) { const double e = bs[k] - fabs(c[k]); if (e < best) { best = e; kb = k; } }
    nrm[0] = nrm[1] = nrm[2] = 0.0;
    nrm[kb] = c[kb] >= 0.0 ? -1.0 : 1.0;
    dd = -best - r;
  }
  *dist = dd;
  for (int k = 0; k < 3; k++) pos[k] = c[k] + nrm[k] * (r + 0.5 * dd);
  return dd <= margin;
}

// capsule (centre cl in box coordinates, half axis h = geom z * half length) vs axis-aligned box: mjc_CapsuleBox as restated in
// oracle/mzo_physics.c capsule_box — the literal feature search (two segment ends against the faces, then the twelve edges).
// Returns the closest segment parameter t in [-1, 1] and the offset of the second sphere (0: none).
MZ_HD void gen_capsule_box_features(const double* pos, const double* halfaxis, double hl, const double* bsize, double margin, double r,
                                    bool* found, double* t_out, double* second) {
  const double axis[3] = {halfaxis[0] / hl, halfaxis[1] / hl, halfaxis[2] / hl};
  const int axisdir = (halfaxis[0] > 0 ? 1 : 0) + (halfaxis[1] > 0 ? 2 : 0) + (halfaxis[2] > 0 ? 4 : 0);
  const double bestdistmax = margin + 2.0 * (r + hl + bsize[0] + bsize[1] + bsize[2]);
  double bestdist = bestdistmax * bestdistmax, bestsegmentpos = 0.0, bestboxpos = 0.0, secondpos = -4.0;
  int cltype = -4, clface = -1, clcorner = 0, cledge = -1;
  for (int i = -1; i <= 1; i += 2) {
    int nout = 0, face = -1;
    double dist = 0.0;
    for (int k = 0; k < 3; k++) {
      const double e = pos[k] + halfaxis[k] * i;
      if (e < -bsize[k]) { nout++; face = k; dist += (e + bsize[k]) * (e + bsize[k]); }
      else if (e > bsize[k]) { nout++; face = k; dist += (e - bsize[k]) * (e - bsize[k]); }
    }
    if (nout > 1) continue;
    if (dist < bestdist) { bestdist = dist; bestsegmentpos = i; cltype = -2 + i; clface = face; }
  }
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 3; j++) {
      if (i & (1 << j)) continue;
      double mid[3] = {(i & 1 ? 1 : -1) * bsize[0], (i & 2 ? 1 : -1) * bsize[1], (i & 4 ? 1 : -1) * bsize[2]}, dif[3];
      mid[j] = 0.0;
      for (int k = 0; k < 3; k++) dif[k] = mid[k] - pos[k];
      const double u = -bsize[j] * dif[j], v = gd_dot3(halfaxis, dif);
      const double ma = bsize[j] * bsize[j], mb = -bsize[j] * halfaxis[j], mc = hl * hl, det = ma * mc - mb * mb;
      if (fabs(det) < 1e-15) continue;
      double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
      int s1 = 1, s2 = 1;
      if (x1 > 1) { x1 = 1; s1 = 2; x2 = (v - mb) / mc; }
      else if (x1 < -1) { x1 = -1; s1 = 0; x2 = (v + mb) / mc; }
      if (x2 > 1) { x2 = 1; s2 = 2; x1 = (u - mb) / ma; if (x1 > 1) { x1 = 1; s1 = 2; } else if (x1 < -1) { x1 = -1; s1 = 0; } else s1 = 1; }
      else if (x2 < -1) { x2 = -1; s2 = 0; x1 = (u + mb) / ma; if (x1 > 1) { x1 = 1; s1 = 2; } else if (x1 < -1) { x1 = -1; s1 = 0; } else s1 = 1; }
      for (int k = 0; k < 3; k++) dif[k] = mid[k] - (pos[k] + halfaxis[k] * x2);
      dif[j] += bsize[j] * x1;
      const double dist = gd_dot3(dif, dif);
      if (dist < bestdist - 1e-15) {
        bestdist = dist; bestsegmentpos = x2; bestboxpos = x1;
        cltype = 3 * s1 + s2; clcorner = i + (s1 == 2 ? (1 << j) : 0); cledge = j;
      }
    }
  *found = cltype != -4;
  if (cltype == -4) return;
  if (cltype >= 0 && cltype / 3 != 1) {
    int c1 = axisdir ^ clcorner;
    if (c1 != 0 && c1 != 7) {
      double mul = 1.0;
      if (!(c1 == 1 || c1 == 2 || c1 == 4)) { mul = -1.0; c1 = 7 - c1; }
      const int ax = c1 == 1 ? 0 : (c1 == 2 ? 1 : 2), ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (axis[ax] * axis[ax] > 0.5) secondpos = mul * fmin(1.0 - mul * bestsegmentpos, 2.0 * bsize[ax] / fabs(halfaxis[ax]));
      else secondpos = -mul * fmin(1.0 + mul * bestsegmentpos, fmin(2.0 * bsize[ax1] / fabs(halfaxis[ax1]), 2.0 * bsize[ax2] / fabs(halfaxis[ax2])));
    }
  } else if (cltype >= 0) {
    const int c1 = (axisdir ^ clcorner) & (7 - (1 << cledge));
    if (c1 == 1 || c1 == 2 || c1 == 4) {
      const int ax = cledge;
      int ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (fabs(axis[ax1]) > fabs(axis[ax2])) { const int q = ax1; ax1 = ax2; ax2 = q; }
      const double mul = (c1 & (1 << ax2)) ? 1.0 : -1.0;
      double sp = fmin(1.0 - mul * bestsegmentpos, 2.0 * bsize[ax2] / fabs(halfaxis[ax2]));
      const double e2 = (mul * halfaxis[ax] > 0) ? 1.0 - bestboxpos : 1.0 + bestboxpos;
      sp = fmin(sp, bsize[ax] * e2 / fabs(halfaxis[ax]));
      secondpos = mul * sp;
    }
  } else {
    const double travel = -2.0 * bestsegmentpos;
    double frac = 1.0;
    for (int k = 0; k < 3; k++) {
      if (k == clface) continue;
      const double p0 = pos[k] + halfaxis[k] * bestsegmentpos, v = halfaxis[k] * travel;
      if (v > 0 && p0 + v > bsize[k]) frac = fmin(frac, (bsize[k] - p0) / v);
      if (v < 0 && p0 + v < -bsize[k]) frac = fmin(frac, (-bsize[k] - p0) / v);
    }
    secondpos = travel * fmax(frac, 0.0);
  }
  *t_out = bestsegmentpos;
  *second = (secondpos > -3.0 && fabs(secondpos) > 1e-12) ? secondpos : 0.0;
}

// The routines below hand every contact they find to `emit(dist, pos, normal, hint)` — world frame, normal from geom1 to geom2,
// hint = the tangent the contact frame should start from (the capsule's axis for capsule-plane) or nullptr.

// sphere (geom1: centre cs, radius r) vs a box of any orientation (geom2): into the box frame, mjraw_SphereBox, back
template <class E>
MZ_HD void gen_sphere_vs_box(const double* cs, double r, const double* bpos, const double* bmat, const double* bsize, double margin, E& emit) {
  double rel[3] = {cs[0] - bpos[0], cs[1] - bpos[1], cs[2] - bpos[2]}, c[3], dist, pl[3], nl[3], pw[3], nw[3];
  gd_mulmatT(c, bmat, rel);
  if (!gen_sphere_box(c, r, bsize, margin, &dist, pl, nl)) return;
  gd_mulmat(pw, bmat, pl); gd_mulmat(nw, bmat, nl);
  for (int k = 0; k < 3; k++) pw[k] += bpos[k];
  emit(dist, pw, nw, nullptr);
}

// capsule (geom1: centre cpos, unit axis caxis, radius r, half length hl) vs a box of any orientation (geom2): mjc_CapsuleBox —
// the feature search in the box frame, then at most two sphere-box contacts on the axis segment
template <class E>
MZ_HD void gen_capsule_vs_box(const double* cpos, const double* caxis, double r, double hl, const double* bpos, const double* bmat, const double* bsize,
                              double margin, E& emit) {
  double rel[3] = {cpos[0] - bpos[0], cpos[1] - bpos[1], cpos[2] - bpos[2]}, cl[3], al[3];
  gd_mulmatT(cl, bmat, rel); gd_mulmatT(al, bmat, caxis);
  const double h[3] = {al[0] * hl, al[1] * hl, al[2] * hl};
  bool found; double t = 0.0, second = 0.0;
  gen_capsule_box_features(cl, h, hl, bsize, margin, r, &found, &t, &second);
  if (!found) return;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1 && second == 0.0) break;
    const double tt = t + (pass ? second : 0.0), c[3] = {cl[0] + tt * h[0], cl[1] + tt * h[1], cl[2] + tt * h[2]};
    double dist, pl[3], nl[3], pw[3], nw[3];
    if (!gen_sphere_box(c, r, bsize, margin, &dist, pl, nl)) continue;
    gd_mulmat(pw, bmat, pl); gd_mulmat(nw, bmat, nl);
    for (int k = 0; k < 3; k++) pw[k] += bpos[k];
    emit(dist, pw, nw, nullptr);
  }
}

// box (geom1) vs box (geom2), both of any orientation: mjc_BoxBox as restated in oracle/mzo_physics.c box_box — the 15-axis
// search in MuJoCo's order (face axes of box 1 and 2 by index, strict improvement; the nine edge-edge axes win only when clearly
// smaller), the face case (vertices of the incident rectangle clipped by the reference rectangle, each with its own depth, at
// most 8, [ASSUME-12]: an intersection without area makes no contact) and the edge-edge case (one contact midway between the
// closest points of the two edges: a tilted SPIN plate on a wall's edge, a leg-less robot box across a block's edge)
template <class E>
MZ_HD void gen_box_vs_box(const double* pos1, const double* mat1, const double* size1, const double* pos2, const double* mat2, const double* size2,
                          double margin, E& emit) {
  double rot[9], rotabs[9], pos21[3], pos12[3], tmp[3], plen1[3], plen2[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {  // rot[i][j] = axis i of box 1 . axis j of box 2
      rot[3 * i + j] = mat1[i] * mat2[j] + mat1[3 + i] * mat2[3 + j] + mat1[6 + i] * mat2[6 + j];
      rotabs[3 * i + j] = fabs(rot[3 * i + j]);
    }
  for (int k = 0; k < 3; k++) tmp[k] = pos2[k] - pos1[k];
  gd_mulmatT(pos21, mat1, tmp);
  for (int k = 0; k < 3; k++) tmp[k] = pos1[k] - pos2[k];
  gd_mulmatT(pos12, mat2, tmp);
  for (int i = 0; i < 3; i++) {
    plen2[i] = rotabs[3 * i] * size2[0] + rotabs[3 * i + 1] * size2[1] + rotabs[3 * i + 2] * size2[2];
    plen1[i] = rotabs[i] * size1[0] + rotabs[3 + i] * size1[1] + rotabs[6 + i] * size1[2];
  }
  double penetration = margin;
  for (int i = 0; i < 3; i++) penetration += 3.0 * (size1[i] + size2[i]);
  int code = -1;
  for (int i = 0; i < 3; i++) {
    const double c1 = -fabs(pos21[i]) + size1[i] + plen2[i], c2 = -fabs(pos12[i]) + size2[i] + plen1[i];
    if (c1 < -margin || c2 < -margin) return;
    if (c1 < penetration) { penetration = c1; code = i + 3 * (pos21[i] < 0); }
    if (c2 < penetration) { penetration = c2; code = 6 + i + 3 * (pos12[i] < 0); }
  }
  double en[3] = {0, 0, 0};  // edge-edge axis (frame of box 1, pointing from box 1 to box 2)
  int ei = -1, ej = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const double ei3[3] = {i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0}, aj[3] = {rot[j], rot[3 + j], rot[6 + j]};
      double cr[3];
      gd_cross(cr, ei3, aj);
      const double len = sqrt(gd_dot3(cr, cr));
      if (len < 1e-10) continue;
      for (int k = 0; k < 3; k++) cr[k] /= len;
      double r1 = 0, r2 = 0;
      for (int k = 0; k < 3; k++) {
        r1 += size1[k] * fabs(cr[k]);
        r2 += size2[k] * fabs(cr[0] * rot[k] + cr[1] * rot[3 + k] + cr[2] * rot[6 + k]);
      }
      const double sep = gd_dot3(pos21, cr), c3 = r1 + r2 - fabs(sep);
      if (c3 < -margin) return;
      if (c3 < penetration - 1e-10 * (fabs(penetration) + r1 + r2)) {
        penetration = c3; code = 12 + 3 * i + j; ei = i; ej = j;
        for (int k = 0; k < 3; k++) en[k] = sep >= 0 ? cr[k] : -cr[k];
      }
    }
  if (code < 0) return;
  if (code >= 12) {
    double p1[3], p2[3], d1[3] = {ei == 0 ? 1.0 : 0.0, ei == 1 ? 1.0 : 0.0, ei == 2 ? 1.0 : 0.0}, d2[3] = {rot[ej], rot[3 + ej], rot[6 + ej]};
    for (int k = 0; k < 3; k++) p1[k] = k == ei ? 0.0 : (en[k] >= 0 ? size1[k] : -size1[k]);
    for (int k = 0; k < 3; k++) p2[k] = pos21[k];
    for (int k = 0; k < 3; k++) {
      if (k == ej) continue;
      const double ak[3] = {rot[k], rot[3 + k], rot[6 + k]}, sg = gd_dot3(ak, en) >= 0 ? -size2[k] : size2[k];
      for (int e = 0; e < 3; e++) p2[e] += ak[e] * sg;
    }
    const double w[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    const double b = gd_dot3(d1, d2), dd = gd_dot3(d1, w), e = gd_dot3(d2, w), den = 1.0 - b * b;
    const double sc = den > 1e-12 ? (b * e - dd) / den : 0.0, tc = den > 1e-12 ? (e - b * dd) / den : 0.0;
    double mid[3], nw[3], pw[3];
    for (int k = 0; k < 3; k++) mid[k] = 0.5 * ((p1[k] + sc * d1[k]) + (p2[k] + tc * d2[k]));
    gd_mulmat(nw, mat1, en); gd_mulmat(pw, mat1, mid);
    for (int k = 0; k < 3; k++) pw[k] += pos1[k];
    emit(-penetration, pw, nw, nullptr);
    return;
  }
  // face case, worked in the frame of the reference box A
  const int fromB = code >= 6, a = code % 3;
  const double* posA = fromB ? pos2 : pos1; const double* matA = fromB ? mat2 : mat1;
  const double* sA = fromB ? size2 : size1; const double* sB = fromB ? size1 : size2;
  const double* pBA = fromB ? pos12 : pos21;
  double R[9];  // R[k][j] = axis k of A . axis j of B
  for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) R[3 * k + j] = fromB ? rot[3 * j + k] : rot[3 * k + j];
  const double sg = pBA[a] < 0 ? -1.0 : 1.0;  // the reference normal sg * e_a points from A to B
  int b = 0;
  for (int j = 1; j < 3; j++) if (fabs(R[3 * a + j]) > fabs(R[3 * a + b])) b = j;
  const double sb = R[3 * a + b] * sg > 0 ? -1.0 : 1.0;  // incident face: the one of B whose outward normal opposes the reference normal
  const int b1 = (b + 1) % 3, b2 = (b + 2) % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
  double cf[3], u[3], v[3];
  for (int k = 0; k < 3; k++) {
    cf[k] = pBA[k] + sb * sB[b] * R[3 * k + b];
    u[k] = sB[b1] * R[3 * k + b1];
    v[k] = sB[b2] * R[3 * k + b2];
  }
  double cand[24][3];
  int nc = 0;
  const double tol = 1e-12 * (1.0 + sA[a1] + sA[a2]);
  for (int c = 0; c < 4; c++) {  // incident corners inside the reference rectangle
    const double su = (c & 1) ? 1.0 : -1.0, sv = (c & 2) ? 1.0 : -1.0;
    double p[3];
    for (int k = 0; k < 3; k++) p[k] = cf[k] + su * u[k] + sv * v[k];
    if (fabs(p[a1]) <= sA[a1] + tol && fabs(p[a2]) <= sA[a2] + tol) { for (int k = 0; k < 3; k++) cand[nc][k] = p[k]; nc++; }
  }
  for (int e = 0; e < 4; e++) {  // incident edges against the border lines of the reference rectangle
    double p[3], q[3];  // edge = p + t q, t in [-1, 1]
    for (int k = 0; k < 3; k++) {
      if (e < 2) { p[k] = cf[k] + (e ? v[k] : -v[k]); q[k] = u[k]; }
      else { p[k] = cf[k] + (e == 3 ? u[k] : -u[k]); q[k] = v[k]; }
    }
    for (int w = 0; w < 2; w++) {
      const int c = w ? a2 : a1, o = w ? a1 : a2;
      if (fabs(q[c]) < 1e-15) continue;
      for (int side = -1; side <= 1; side += 2) {
        const double t = (side * sA[c] - p[c]) / q[c];
        if (t < -1.0 || t > 1.0) continue;
        if (fabs(p[o] + t * q[o]) > sA[o] + tol) continue;
        for (int k = 0; k < 3; k++) cand[nc][k] = p[k] + t * q[k];
        nc++;
      }
    }
  }
  {  // reference corners inside the incident rectangle, on the incident plane
    const double det = u[a1] * v[a2] - u[a2] * v[a1];
    if (fabs(det) > 1e-15)
      for (int c = 0; c < 4; c++) {
        const double x = ((c & 1) ? sA[a1] : -sA[a1]) - cf[a1], y = ((c & 2) ? sA[a2] : -sA[a2]) - cf[a2];
        const double al = (x * v[a2] - y * v[a1]) / det, be = (u[a1] * y - u[a2] * x) / det;
        if (fabs(al) <= 1.0 + 1e-12 && fabs(be) <= 1.0 + 1e-12) {
          for (int k = 0; k < 3; k++) cand[nc][k] = cf[k] + al * u[k] + be * v[k];
          nc++;
        }
      }
  }
  {
    double lo1 = 1e300, hi1 = -1e300, lo2 = 1e300, hi2 = -1e300;
    for (int c = 0; c < nc; c++) {
      lo1 = fmin(lo1, cand[c][a1]); hi1 = fmax(hi1, cand[c][a1]);
      lo2 = fmin(lo2, cand[c][a2]); hi2 = fmax(hi2, cand[c][a2]);
    }
    if (nc == 0 || hi1 - lo1 <= 1e-6 || hi2 - lo2 <= 1e-6) return;  // MZO_BOX_MINOVERLAP [ASSUME-12]
  }
  int emitted = 0;
  const double dtol = 1e-9 * (1.0 + sA[a1] + sA[a2]);  // candidates closer than this (max norm) are one vertex
  for (int c = 0; c < nc && emitted < 8; c++) {
    bool dup = false;
    for (int e = 0; e < c && !dup; e++)
      if (fabs(cand[c][0] - cand[e][0]) <= dtol && fabs(cand[c][1] - cand[e][1]) <= dtol && fabs(cand[c][2] - cand[e][2]) <= dtol) dup = true;
    if (dup) continue;
    const double dist = sg * cand[c][a] - sA[a];
    if (dist > margin) continue;
    double pl[3] = {cand[c][0], cand[c][1], cand[c][2]}, pw[3], nl[3] = {0, 0, 0}, nw[3];
    pl[a] -= sg * 0.5 * dist;
    nl[a] = fromB ? -sg : sg;  // reported from geom1 to geom2
    gd_mulmat(pw, matA, pl); gd_mulmat(nw, matA, nl);
    for (int k = 0; k < 3; k++) pw[k] += posA[k];
    emit(dist, pw, nw, nullptr);
    emitted++;
  }
}

// ------------------------------------------------------------------ collision: one item per lane
// contacts of collision item `it` into the pool.  The oracle's order (item order, then the order inside a routine) is the pool
// key; the compaction below sorts by it.
MZ_HD void gen_collide_item(const GenDev& K, GenScratch& s, int it) {
  const mz_model& m = K.m;
  const GenItem& I = K.item[it];
  const double margin = I.P.margin;
  int n = 0;
  const double active_below = I.P.margin - I.P.gap;
  auto emit = [&](double dist, const double* pos, const double* nrm, const double* hint) {
    // only ACTIVE contacts enter the pool (dist < margin - gap; the others make no constraint row) — their index n within the item still
    // counts, so the keys order the survivors as the oracle orders them
    if (!(dist < active_below)) { n++; return; }
    const int slot = gen_take(&s.npool);
    if (slot < GN_POOL) {
      s.pkey[slot] = it * GN_KEY + (n < GN_KEY ? n : GN_KEY - 1); s.pdist[slot] = dist;
      for (int k = 0; k < 3; k++) { s.ppos[slot][k] = pos[k]; s.pnrm[slot][k] = nrm[k]; s.phint[slot][k] = hint ? hint[k] : 0.0; }
    }
    n++;
  };
  static const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (I.kind <= GK_PLANE_BOX) {  // the floor plane (geom1) against a sphere / capsule ends / box corners
    const int g = I.g2;
    const double* pm = s.gmat[I.g1];
    const double nz[3] = {pm[2], pm[5], pm[8]};
    const double* gp = s.gpos[g];
    const double* gm = s.gmat[g];
    if (I.kind == GK_PLANE_BOX) {
      // mjc_PlaneBox: the corners in index order (bit 0 x, bit 1 y, bit 2 z); a corner is skipped when it lies beyond the margin or
      // on the far side of the box centre (its offset along the plane normal is positive); at most 4
      const double* sz = m.geom_size[g];
      const double rel[3] = {gp[0] - s.gpos[I.g1][0], gp[1] - s.gpos[I.g1][1], gp[2] - s.gpos[I.g1][2]}, cdist = gd_dot3(rel, nz);
      int cnt = 0;
      for (int ci = 0; ci < 8 && cnt < 4; ci++) {
        const double loc[3] = {(ci & 1 ? 1 : -1) * sz[0], (ci & 2 ? 1 : -1) * sz[1], (ci & 4 ? 1 : -1) * sz[2]};
        double w[3], pos[3];
        gd_mulmat(w, gm, loc);
        const double ldist = gd_dot3(nz, w);
        if (cdist + ldist > margin || ldist > 0) continue;
        const double dist = cdist + ldist;
        for (int k = 0; k < 3; k++) pos[k] = gp[k] + w[k] - nz[k] * (0.5 * dist);
        emit(dist, pos, nz, nullptr);
        cnt++;
      }
      return;
    }
    const double r = m.geom_size[g][0], hl = I.kind == GK_PLANE_CAPSULE ? m.geom_size[g][1] : 0.0;
    const double axis[3] = {gm[2], gm[5], gm[8]};
    const int nend = I.kind == GK_PLANE_CAPSULE ? 2 : 1;
    for (int e = 0; e < nend; e++) {  // [ASSUME-5] end +axis first, then -axis; tangent hint = the capsule's axis
      const double sg = nend == 1 ? 0.0 : (e == 0 ? 1.0 : -1.0);
      double c[3], rel[3], pos[3];
      for (int k = 0; k < 3; k++) { c[k] = gp[k] + sg * axis[k] * hl; rel[k] = c[k] - s.gpos[I.g1][k]; }
      const double dist = gd_dot3(rel, nz) - r;
      if (dist > margin) continue;
      for (int k = 0; k < 3; k++) pos[k] = c[k] - nz[k] * (r + 0.5 * dist);
      emit(dist, pos, nz, nend == 2 ? axis : nullptr);
    }
    return;
  }
  if (I.kind == GK_SPHERE_SPHERE || I.kind == GK_SPHERE_CAPSULE) {
    // sphere-sphere; sphere-capsule = the point of the capsule's axis segment nearest to the sphere centre, then sphere-sphere
    // (mjc_SphereCapsule); normal from geom1 to geom2
    const double* c1 = s.gpos[I.g1];
    double c2[3] = {s.gpos[I.g2][0], s.gpos[I.g2][1], s.gpos[I.g2][2]};
    if (I.kind == GK_SPHERE_CAPSULE) {
      const double* cm = s.gmat[I.g2];
      const double axis[3] = {cm[2], cm[5], cm[8]}, vec[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]}, hl = m.geom_size[I.g2][1];
      const double x = fmin(fmax(gd_dot3(axis, vec), -hl), hl);
      for (int k = 0; k < 3; k++) c2[k] += axis[k] * x;
    }
    const double dv[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]}, cd = sqrt(gd_dot3(dv, dv)), r1 = m.geom_size[I.g1][0], r2 = m.geom_size[I.g2][0];
    const double dist = cd - r1 - r2;
    if (dist > margin) return;
    double nrm[3] = {1.0, 0.0, 0.0}, pos[3];
    if (!(cd < 1e-15)) for (int k = 0; k < 3; k++) nrm[k] = dv[k] / cd;
    for (int k = 0; k < 3; k++) pos[k] = c1[k] + nrm[k] * (r1 + 0.5 * dist);
    emit(dist, pos, nrm, nullptr);
    return;
  }
  if (I.kind == GK_CAPSULE_CAPSULE) {
    gen_capsule_vs_capsule(s.gpos[I.g1], s.gmat[I.g1], m.geom_size[I.g1][0], m.geom_size[I.g1][1], s.gpos[I.g2], s.gmat[I.g2], m.geom_size[I.g2][0],
                           m.geom_size[I.g2][1], margin, emit);
    return;
  }
  if (I.kind == GK_SPHERE_BOX) { gen_sphere_vs_box(s.gpos[I.g1], m.geom_size[I.g1][0], s.gpos[I.g2], s.gmat[I.g2], m.geom_size[I.g2], margin, emit); return; }
  if (I.kind == GK_CAPSULE_BOX) {
    const double* cm = s.gmat[I.g1];
    const double axis[3] = {cm[2], cm[5], cm[8]};
    gen_capsule_vs_box(s.gpos[I.g1], axis, m.geom_size[I.g1][0], m.geom_size[I.g1][1], s.gpos[I.g2], s.gmat[I.g2], m.geom_size[I.g2], margin, emit);
    return;
  }
  if (I.kind == GK_BOX_BOX) { gen_box_vs_box(s.gpos[I.g1], s.gmat[I.g1], m.geom_size[I.g1], s.gpos[I.g2], s.gmat[I.g2], m.geom_size[I.g2], margin, emit); return; }
  // the maze boxes under the geom's bounding square, row-major; per cell, in the order the reference emits the geoms
  // (maze_env.py:124-152): the platform of an elevated maze (every cell but the chasms; z from 0 to height_offset), then the wall
  const int g = I.kind == GK_WALL_BOX ? I.g2 : I.g1;
  const double* gp = s.gpos[g];
  const double sc = m.maze_scale, reach = m.geom_rbound[g] + margin;
  if (gp[2] - reach > m.wall_center_z + m.wall_half_z) return;
  if (!m.elevated && gp[2] + reach < m.wall_center_z - m.wall_half_z) return;
  const int j0 = (int)floor((gp[0] - reach + m.torso_x) / sc + 0.5), j1 = (int)floor((gp[0] + reach + m.torso_x) / sc + 0.5);
  const int i0 = (int)floor((gp[1] - reach + m.torso_y) / sc + 0.5), i1 = (int)floor((gp[1] + reach + m.torso_y) / sc + 0.5);
  const double bs[3] = {m.wall_half_xy, m.wall_half_xy, m.wall_half_z};
  for (int i = i0; i <= i1; i++)
    for (int j = j0; j <= j1; j++) {
      if (i < 0 || j < 0 || i >= m.grid_rows || j >= m.grid_cols) continue;
      for (int layer = 0; layer < 2; layer++) {
        if (layer == 0 && !(m.elevated && m.grid[i][j] != MZ_CELL_CHASM)) continue;
        if (layer == 1 && m.grid[i][j] != MZ_CELL_BLOCK) continue;
        const double bpos[3] = {j * sc - m.torso_x, i * sc - m.torso_y, layer == 0 ? m.wall_half_z : m.wall_center_z};
        if (gp[2] - reach > bpos[2] + m.wall_half_z || gp[2] + reach < bpos[2] - m.wall_half_z) continue;
        if (I.kind == GK_SPHERE_WALL) gen_sphere_vs_box(gp, m.geom_size[g][0], bpos, ident, bs, margin, emit);
        else if (I.kind == GK_CAPSULE_WALL) {
          const double* cm = s.gmat[g];
          const double axis[3] = {cm[2], cm[5], cm[8]};
          gen_capsule_vs_box(gp, axis, m.geom_size[g][0], m.geom_size[g][1], bpos, ident, bs, margin, emit);
        } else gen_box_vs_box(bpos, ident, bs, gp, s.gmat[g], m.geom_size[g], margin, emit);
      }
    }
}

// pool -> the ACTIVE contacts (dist < margin - gap), in key order: entry e's slot = the number of active entries with a smaller key
MZ_HD void gen_compact_item(const GenDev& K, GenScratch& s, int e, int np) {
  const int key = s.pkey[e];
  int r0 = 0, r1 = 0, r2 = 0, r3 = 0, o = 0;  // (four keys per round trip)
  for (; o + 4 <= np; o += 4) {
    const int k0 = s.pkey[o], k1 = s.pkey[o + 1], k2 = s.pkey[o + 2], k3 = s.pkey[o + 3];
    r0 += (k0 < key || (k0 == key && o < e)) ? 1 : 0; r1 += (k1 < key || (k1 == key && o + 1 < e)) ? 1 : 0;
    r2 += (k2 < key || (k2 == key && o + 2 < e)) ? 1 : 0; r3 += (k3 < key || (k3 == key && o + 3 < e)) ? 1 : 0;
  }
  for (; o < np; o++) { const int k0 = s.pkey[o]; r0 += (k0 < key || (k0 == key && o < e)) ? 1 : 0; }
  const int rank = (r0 + r1) + (r2 + r3);
  if (rank >= GN_NC) return;  // (counted by gen_forward)
  s.citem[rank] = key / GN_KEY; s.cdist[rank] = s.pdist[e];
  for (int k = 0; k < 3; k++) { s.cpos[rank][k] = s.ppos[e][k]; s.cnrm[rank][k] = s.pnrm[e][k]; s.chint[rank][k] = s.phint[e][k]; }
}

// joint-limit rows (hinge / slide): candidate e = (limited joint e / 2, lower | upper side).  gen_limit_flag marks the active ones; after a
// fence gen_limit_row fills the row at its rank among them — the serial loop's order (joint by joint, lower side first)
MZ_HD void gen_limit_flag(const GenDev& K, GenScratch& s, int e) {
  const mz_model& m = K.m;
  const int j = K.limj[e >> 1];
  const double q = s.qpos[s.tp.jnt_qposadr[j]], dist = (e & 1) ? m.jnt_range[j][1] - q : q - m.jnt_range[j][0];
  s.lflag[e] = dist < m.jnt_margin[j] ? 1 : 0;
}
MZ_HD void gen_limit_row(const GenDev& K, GenScratch& s, int e) {
  const mz_model& m = K.m;
  if (!s.lflag[e]) return;
  int nl = 0;
  for (int o = 0; o < e; o++) nl += s.lflag[o];
  if (nl >= GN_NL) return;  // (counted by gen_forward: a dropped constraint row is never silent)
  const int j = K.limj[e >> 1], dof = s.tp.jnt_dofadr[j];
  const double q = s.qpos[s.tp.jnt_qposadr[j]], dist = (e & 1) ? m.jnt_range[j][1] - q : q - m.jnt_range[j][0];
  const double imp = gd_impedance(m.jnt_solimp[j], fabs(dist - m.jnt_margin[j]));
  const double R = fmax(1e-15, (1.0 - imp) * m.dof_invweight0[dof] / imp), sg = (e & 1) ? -1.0 : 1.0;
  s.ldof[nl] = dof; s.lsign[nl] = sg; s.lD[nl] = 1.0 / R;
  s.laref[nl] = -K.lim_B[j] * (sg * s.qvel[dof]) - K.lim_K[j] * imp * (dist - m.jnt_margin[j]);
}

// one item per (contact, frame axis): the contact frame, the Jacobian row J(body2) - J(body1), the reference acceleration
MZ_HD void gen_contact_row_item(const GenDev& K, GenScratch& s, int item) {
  const mz_model& m = K.m;
  const int c = item / 3, a = item - 3 * c;
  const GenItem& I = K.item[s.citem[c]];
  const GenPair& P = I.P;
  // frame (mju_makeFrame): the normal, a tangent from the hint (capsule axis for capsule-plane) or the default rule
  double n[3] = {s.cnrm[c][0], s.cnrm[c][1], s.cnrm[c][2]}, y[3] = {s.chint[c][0], s.chint[c][1], s.chint[c][2]}, t1[3], t2[3];
  {
    const double nn = sqrt(gd_dot3(n, n));
    for (int k = 0; k < 3; k++) n[k] /= nn;
    if (sqrt(gd_dot3(y, y)) < 0.5) { y[0] = 0.0; y[1] = (n[1] < 0.5 && n[1] > -0.5) ? 1.0 : 0.0; y[2] = 1.0 - y[1]; }
    double d = gd_dot3(n, y);
    for (int k = 0; k < 3; k++) y[k] -= n[k] * d;
    double yn = sqrt(gd_dot3(y, y));
    if (yn < 1e-10) {
      y[0] = 0.0; y[1] = (n[1] < 0.5 && n[1] > -0.5) ? 1.0 : 0.0; y[2] = 1.0 - y[1];
      d = gd_dot3(n, y);
      for (int k = 0; k < 3; k++) y[k] -= n[k] * d;
      yn = sqrt(gd_dot3(y, y));
    }
    for (int k = 0; k < 3; k++) t1[k] = y[k] / yn;
    gd_cross(t2, n, t1);
  }
  const double* dir = a == 0 ? n : (a == 1 ? t1 : t2);
  const double sc = a == 0 ? 1.0 : P.mu;
  double off[3];
  double* J = gen_cj(s, m.nv, c, a);  // (built in place: a local array indexed by a run-time dof would live in scratch memory)
  for (int k = 0; k < 3; k++) off[k] = s.cpos[c][k] - s.refpoint[k];
  for (int i = 0; i < m.nv; i++) J[i] = 0.0;
  for (int side = 0; side < 2; side++) {  // the contact force acts on geom2's body (+) and reacts on geom1's (-)
    const double sgn = side ? 1.0 : -1.0;
    for (int b = side ? I.b2 : I.b1; b > 0; b = s.tp.body_parent[b])
      for (int i = s.tp.body_dofadr[b]; i >= 0 && i < s.tp.body_dofadr[b] + s.tp.body_dofnum[b]; i++) {
        double wx[3];
        gd_cross(wx, s.S[i], off);
        J[i] += sgn * sc * ((s.S[i][3] + wx[0]) * dir[0] + (s.S[i][4] + wx[1]) * dir[1] + (s.S[i][5] + wx[2]) * dir[2]);
      }
  }
  double vel = 0.0;
  for (int i = 0; i < m.nv; i++) vel += J[i] * s.qvel[i];
  double aref = -P.B * vel;
  if (a == 0) {
    const double imp = gd_impedance(P.solimp, fabs(s.cdist[c] - (P.margin - P.gap)));
    if (P.condim == 1) {
      s.cD[c] = -1.0 / fmax(1e-15, (1.0 - imp) * P.tran / imp);  // (negative: a single frictionless row, see gen_contact_eval)
    } else {
      const double R = fmax(1e-15, (1.0 - imp) * (P.tran + P.mu * P.mu * P.tran) / imp);
      s.cD[c] = 1.0 / (2.0 * P.mu * P.mu * R);
    }
    aref -= P.K * imp * (s.cdist[c] - (P.margin - P.gap));
  }
  s.caref[c][a] = aref;
}

// pyramidal contact rows u0 +- u1, u0 +- u2 (D > 0) or the single frictionless row u0 (stored as D < 0): cost, gradient block
// g[3] and curvature block W[5] (W0 = d2/du0^2, W1 = du0du1, W2 = du0du2, W3 = du1^2, W4 = du2^2)
MZ_HD double gen_contact_eval(double D, const double* u, double* g, double* W) {
  if (D < 0.0) {
    const double Dm = -D, act = u[0] < 0.0 ? 1.0 : 0.0;
    if (g) { g[0] = Dm * act * u[0]; g[1] = 0.0; g[2] = 0.0; }
    if (W) { W[0] = Dm * act; W[1] = W[2] = W[3] = W[4] = 0.0; }
    return 0.5 * Dm * act * u[0] * u[0];
  }
  const double r0 = u[0] + u[1], r1 = u[0] - u[1], r2 = u[0] + u[2], r3 = u[0] - u[2];
  const double a0 = r0 < 0.0 ? 1.0 : 0.0, a1 = r1 < 0.0 ? 1.0 : 0.0, a2 = r2 < 0.0 ? 1.0 : 0.0, a3 = r3 < 0.0 ? 1.0 : 0.0;
  if (g) { g[0] = D * (a0 * r0 + a1 * r1 + a2 * r2 + a3 * r3); g[1] = D * (a0 * r0 - a1 * r1); g[2] = D * (a2 * r2 - a3 * r3); }
  if (W) { W[0] = D * (a0 + a1 + a2 + a3); W[1] = D * (a0 - a1); W[2] = D * (a2 - a3); W[3] = D * (a0 + a1); W[4] = D * (a2 + a3); }
  return 0.5 * D * (a0 * r0 * r0 + a1 * r1 * r1 + a2 * r2 * r2 + a3 * r3 * r3);
}
MZ_HD bool gen_rows_flip(double D, const double* u, const double* w) {  // does any row change its active state between u and w?
  if (D < 0.0) return (u[0] < 0.0) != (w[0] < 0.0);
  return ((u[0] + u[1] < 0.0) != (w[0] + w[1] < 0.0)) || ((u[0] - u[1] < 0.0) != (w[0] - w[1] < 0.0)) || ((u[0] + u[2] < 0.0) != (w[0] + w[2] < 0.0)) ||
         ((u[0] - u[2] < 0.0) != (w[0] - w[2] < 0.0));
}
MZ_HD void gen_rows_slope(double D, const double* u, const double* v, double alpha, double* d1, double* d2) {  // phi'(alpha), phi''(alpha) terms
  const double x0 = u[0] + alpha * v[0], x1 = u[1] + alpha * v[1], x2 = u[2] + alpha * v[2];
  if (D < 0.0) { if (x0 < 0.0) { *d1 += -D * x0 * v[0]; *d2 += -D * v[0] * v[0]; } return; }
  double r, w;
  r = x0 + x1; w = v[0] + v[1]; if (r < 0.0) { *d1 += D * r * w; *d2 += D * w * w; }
  r = x0 - x1; w = v[0] - v[1]; if (r < 0.0) { *d1 += D * r * w; *d2 += D * w * w; }
  r = x0 + x2; w = v[0] + v[2]; if (r < 0.0) { *d1 += D * r * w; *d2 += D * w * w; }
  r = x0 - x2; w = v[0] - v[2]; if (r < 0.0) { *d1 += D * r * w; *d2 += D * w * w; }
}

// ------------------------------------------------------------------ velocities, bias (RNE), passive (damping + fluid), actuation
// velocity, acceleration and inertial + velocity-product force of body b from its parent's (mj_comVel / mj_rne forward pass)
MZ_HD void gen_rne_body(const GenDev& K, GenScratch& s, int b) {
  const int p = s.tp.body_parent[b], j0 = s.tp.body_jntadr[b];
  double v[6], a[6];
  for (int e = 0; e < 6; e++) { v[e] = s.cvel[p][e]; a[e] = s.cacc[p][e]; }
  for (int j = j0; j < j0 + s.tp.body_jntnum[b]; j++) {
    const int d0 = s.tp.jnt_dofadr[j];
    if (s.tp.jnt_type[j] == MZ_JNT_FREE || s.tp.jnt_type[j] == MZ_JNT_BALL) {
      // mj_comVel: the free joint's translations first (world-fixed axes: no derivative); then all three rotation-axis
      // derivatives from the velocity in front of them, then the rotations' own velocity
      int r0 = d0;
      if (s.tp.jnt_type[j] == MZ_JNT_FREE) {
        for (int k = 0; k < 3; k++)
          for (int e = 0; e < 6; e++) v[e] += s.S[d0 + k][e] * s.qvel[d0 + k];
        r0 = d0 + 3;
      }
      double sd[3][6];
      for (int k = 0; k < 3; k++) gd_motion_cross(sd[k], v, s.S[r0 + k]);
      for (int k = 0; k < 3; k++)
        for (int e = 0; e < 6; e++) { a[e] += sd[k][e] * s.qvel[r0 + k]; v[e] += s.S[r0 + k][e] * s.qvel[r0 + k]; }
    } else {
      double sd[6];
      gd_motion_cross(sd, v, s.S[d0]);
      for (int e = 0; e < 6; e++) { a[e] += sd[e] * s.qvel[d0]; v[e] += s.S[d0][e] * s.qvel[d0]; }
    }
  }
  double Ia[6], Iv[6], vf[6];
  gd_inertia_mul(Ia, s.cinert[b], a);
  gd_inertia_mul(Iv, s.cinert[b], v);
  gd_force_cross(vf, v, Iv);
  for (int e = 0; e < 6; e++) { s.cvel[b][e] = v[e]; s.cacc[b][e] = a[e]; s.cfrc[b][e] = Ia[e] + vf[e]; }
}
// The tree pass of an evaluation, level by level (the bodies of a level side by side; round 6 — on one lane the passes were 15 % of an
// Ant env-step): down — velocities / accelerations / each body's own force (RNE forward).  There is no up-pass: composite inertias
// (CRB) and subtree forces (RNE backward) are sums over a subtree, and the dof lanes that need them (gen_mass_item, gen_force_item) add
// them up themselves from the bodies' own values by the subtree bit masks of GenTopo — independent loads instead of level fences.
template <class C>
MZ_HD void gen_tree_passes(const C& cx, const GenDev& K, GenScratch& s) {
  const mz_model& m = K.m;
  MZ_FOR(e, 6) { s.cvel[0][e] = 0.0; s.cacc[0][e] = e >= 3 ? -m.gravity[e - 3] : 0.0; }
  for (int d = 1; d <= K.max_depth; d++) {
    cx.sync();
    MZ_FOR(b, m.nbody) if (s.tp.body_depth[b] == d) gen_rne_body(K, s, b);
  }
  cx.sync();
}
// MuJoCo's inertia-box fluid model (option density / viscosity; swimmer.xml:3): wrench of body b about the reference point
MZ_HD void gen_fluid_item(const GenDev& K, GenScratch& s, int b) {
  const mz_model& m = K.m;
  for (int e = 0; e < 6; e++) s.ffl[b][e] = 0.0;
  const double mass = m.body_mass[b];
  if (b == 0 || mass < 1e-15 || !(m.density > 0.0 || m.viscosity > 0.0)) return;
  // principal moments and the inertial frame ximat = xmat * R(body_iquat) (MuJoCo evaluates the box model there; the
  // diagonal of the body-frame tensor is not the same thing for a tilted link)
  const double* I = m.body_pinertia[b];
  double Ri[9], xim[9];
  gd_quat2mat(Ri, m.body_iquat[b]);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) xim[3 * i + j] = s.xmat[b][3 * i] * Ri[j] + s.xmat[b][3 * i + 1] * Ri[3 + j] + s.xmat[b][3 * i + 2] * Ri[6 + j];
  const double bx[3] = {sqrt(fmax(1e-15, I[1] + I[2] - I[0]) / mass * 6.0), sqrt(fmax(1e-15, I[0] + I[2] - I[1]) / mass * 6.0),
                        sqrt(fmax(1e-15, I[0] + I[1] - I[2]) / mass * 6.0)};
  const double w[3] = {s.cvel[b][0], s.cvel[b][1], s.cvel[b][2]};
  double r[3], wr[3], vc[3], lw[3], lv[3];
  for (int k = 0; k < 3; k++) r[k] = s.xipos[b][k] - s.refpoint[k];
  gd_cross(wr, w, r);
  for (int k = 0; k < 3; k++) vc[k] = s.cvel[b][3 + k] + wr[k];
  gd_mulmatT(lw, xim, w);
  gd_mulmatT(lv, xim, vc);
  double lf[3] = {0, 0, 0}, lt[3] = {0, 0, 0};
  const double pi = 3.14159265358979323846;
  if (m.viscosity > 0.0) {
    const double diam = (bx[0] + bx[1] + bx[2]) / 3.0, sv = -3.0 * m.viscosity * pi * diam, tv = -m.viscosity * pi * diam * diam * diam;
    for (int k = 0; k < 3; k++) { lf[k] += sv * lv[k]; lt[k] += tv * lw[k]; }
  }
  if (m.density > 0.0) {
    lf[0] -= 0.5 * m.density * bx[1] * bx[2] * fabs(lv[0]) * lv[0];
    lf[1] -= 0.5 * m.density * bx[0] * bx[2] * fabs(lv[1]) * lv[1];
    lf[2] -= 0.5 * m.density * bx[0] * bx[1] * fabs(lv[2]) * lv[2];
    auto p4 = [](double x) { return x * x * x * x; };
    lt[0] -= m.density * bx[0] * (p4(bx[1]) + p4(bx[2])) * fabs(lw[0]) * lw[0] / 64.0;
    lt[1] -= m.density * bx[1] * (p4(bx[0]) + p4(bx[2])) * fabs(lw[1]) * lw[1] / 64.0;
    lt[2] -= m.density * bx[2] * (p4(bx[0]) + p4(bx[1])) * fabs(lw[2]) * lw[2] / 64.0;
  }
  double f[3], t[3], rf[3];
  gd_mulmat(f, xim, lf);
  gd_mulmat(t, xim, lt);
  gd_cross(rf, r, f);
  for (int k = 0; k < 3; k++) { s.ffl[b][k] = t[k] + rf[k]; s.ffl[b][3 + k] = f[k]; }
}
MZ_HD void gen_force_item(const GenDev& K, GenScratch& s, int i) {
  const mz_model& m = K.m;
  const uint32_t sub = s.tp.subtree[s.tp.dof_bodyid[i]];
  double f6[6] = {0, 0, 0, 0, 0, 0};  // force on the dof's subtree (mj_rne backward pass), summed here
  for (uint32_t mk = sub; mk; mk &= mk - 1) { const double* f = s.cfrc[__builtin_ctz(mk)]; for (int e = 0; e < 6; e++) f6[e] += f[e]; }
  const double bias = gd_dot6(s.S[i], f6);
  double pas = -m.dof_damping[i] * s.qvel[i];
  {  // joint spring (mj_passive), hinge / slide
    const int j = m.dof_jntid[i];
    const double k = m.jnt_stiffness[j];
    if (k != 0.0 && (s.tp.jnt_type[j] == MZ_JNT_HINGE || s.tp.jnt_type[j] == MZ_JNT_SLIDE))
      pas -= k * (s.qpos[s.tp.jnt_qposadr[j]] - m.qpos0[s.tp.jnt_qposadr[j]] - m.jnt_springref[j]);
  }
  if (m.density > 0.0 || m.viscosity > 0.0)
    for (uint32_t mk = sub; mk; mk &= mk - 1) pas += gd_dot6(s.S[i], s.ffl[__builtin_ctz(mk)]);  // dof i moves body b iff body(i) is b or an ancestor of b
  double act = s.fact[i];
  if (K.servos)  // <position> / <velocity> actuators: bias1 * length + bias2 * velocity of the joint, at every evaluation
    for (int u = 0; u < m.nu; u++)
      if (m.act_dofid[u] == i) {
        const int jq = s.tp.jnt_qposadr[m.dof_jntid[i]];
        const double g = m.act_gear[u];
        act += g * (m.act_biasprm[u][1] * g * (s.qpos[jq] - m.qpos0[jq]) + m.act_biasprm[u][2] * g * s.qvel[i]);
      }
  s.bias[i] = bias; s.passive[i] = pas;
  s.qfs[i] = pas - bias + act;
}

// dense Cholesky + solve on one lane (A = L L^T in the lower triangle of H)
// Cholesky factorisation and solve of the n x n system A x = b (A: M for qacc_smooth, M + J^T W J for the Newton direction), spread
// over the wavefront: lane i owns row i, and the right-hand side rides along as row n of the augmented matrix [A; b^T] — its "factor
// row" IS the forward substitution L y = b, at no extra step.  Column j needs only the finished columns k < j: every lane forms its own
// t_i = A[i][j] - sum_k L[i][k] L[j][k] at once (left-looking), the pivot goes round through LDS, one reciprocal square root per
// column and no division; the back substitution L^T x = y runs column by column the same way.  (Round 6: on ONE lane — n^3 / 6
// dependent LDS round trips — this was 63-71 % of an Ant env-step on the general engine: profiles/r06/general_engine_phases.txt.)
// In: Asrc (lower triangle; may be A itself), x = b.  Out: x = Asrc^-1 b; A holds the factor, s.cdinv the reciprocals of its diagonal.
// False: not positive definite (A and x are garbage then).  Called by every lane of the group; ends behind a fence.
template <class C>
MZ_HD bool gen_chol_solve(const C& cx, GenScratch& s, const double (*Asrc)[GN_NV], double (*A)[GN_NV], int n, double* x) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (C::nlanes == 64) {
    // On the device the rows live in REGISTERS (lane l: row l; lane n: the right-hand side), the loops over columns are unrolled to
    // GN_NV with uniform guards, and a pivot row's entries come by v_readlane — no LDS round trip inside the factorisation (each one
    // is ~150 cycles for a lone wavefront; the LDS form below, kept for the one-lane host build, measured 28 k cycles per solve of
    // a 14 x 14 system).  L^T x = y needs COLUMN l of the factor in lane l: the rows cross LDS once, y with them.
    const int l = cx.l;
    const bool isrow = l < n;
    double r[GN_NV];
    {  // (all GN_NV loads of the lane's row unconditionally — they are in bounds whatever n is — and back to back: ONE LDS round trip;
       //  what the lane must not see is masked afterwards)
      const double* src = isrow ? Asrc[l] : x;
#pragma unroll
      for (int k = 0; k < GN_NV; k++) r[k] = src[k];
#pragma unroll
      for (int k = 0; k < GN_NV; k++) asm volatile("" : "+v"(r[k]));  // (pins the loads where they are: the compiler otherwise sinks each one into a branch of its mask — 24 round trips)
      // (masked with integer ands on ONE per-lane bound: as a select over three conditions this loop became six branches per entry)
      const int kmax = isrow ? (l < n - 1 ? l : n - 1) : (l == n ? n - 1 : -1);
#pragma unroll
      for (int k = 0; k < GN_NV; k++) r[k] = __longlong_as_double(__double_as_longlong(r[k]) & (k <= kmax ? ~0ll : 0ll));
    }
    bool ok = true;
    double dinv = 0.0;
#pragma unroll
    for (int j = 0; j < GN_NV; j++) {
      if (j < n) {
        double t = r[j];
#pragma unroll
        for (int k = 0; k < j; k++) t -= r[k] * C::readlaned(r[k], j);
        const double d = C::readlaned(t, j);
        ok = ok && d >= 1e-15;
        const double inv = rsqrt(d);
        r[j] = l == j ? d * inv : (l > j ? t * inv : 0.0);
        if (l == j) dinv = inv;
      }
    }
    if (!ok) return false;  // (uniform: every lane saw the same pivots)
    cx.sync();              // (every lane has read its row of Asrc)
    if (l <= n) {  // (whole rows, zeros above the diagonal and behind column n included: nobody reads those, and the stores go out back to back)
      double* dst = isrow ? A[l] : x;
#pragma unroll
      for (int k = 0; k < GN_NV; k++) dst[k] = r[k];
    }
    cx.sync();
    const int lc = isrow ? l : 0;
    double y = x[lc];
#pragma unroll
    for (int k = 0; k < GN_NV; k++) r[k] = A[k][lc];  // column l of the factor ...
#pragma unroll
    for (int k = 0; k < GN_NV; k++) asm volatile("" : "+v"(r[k]));
    y = isrow ? y : 0.0;
    const int klo = isrow ? l : GN_NV;
#pragma unroll
    for (int k = 0; k < GN_NV; k++) r[k] = __longlong_as_double(__double_as_longlong(r[k]) & ((k > klo && k < n) ? ~0ll : 0ll));  // ... below the diagonal (rows >= n hold whatever LDS held: masked away, never computed with)
#pragma unroll
    for (int j = GN_NV - 1; j >= 0; j--) {
      if (j < n) {
        const double xj = C::readlaned(y * dinv, j);
        y = l == j ? xj : y - r[j] * xj;
      }
    }
    cx.sync();  // (every lane has read y)
    if (isrow) { x[l] = y; s.cdinv[l] = dinv; }
    cx.sync();
    return true;
  }
#endif
  if (Asrc != A) { MZ_FOR(e, n * n) { const int i = e / n, j = e - n * i; if (j <= i) A[i][j] = Asrc[i][j]; } cx.sync(); }
  for (int j = 0; j < n; j++) {
    MZ_FOR(i, n + 1) if (i >= j) {
      const double* Ai = i < n ? A[i] : x;  // (row n: the right-hand side)
      const double* Aj = A[j];
      double t0 = Ai[j], t1 = 0.0, t2 = 0.0, t3 = 0.0;  // four partial sums: the loads of a group of four are in flight together
      int k = 0;
      for (; k + 4 <= j; k += 4) { t0 -= Ai[k] * Aj[k]; t1 -= Ai[k + 1] * Aj[k + 1]; t2 -= Ai[k + 2] * Aj[k + 2]; t3 -= Ai[k + 3] * Aj[k + 3]; }
      for (; k < j; k++) t0 -= Ai[k] * Aj[k];
      s.ctmp[i] = (t0 + t1) + (t2 + t3);
    }
    cx.sync();
    const double d = s.ctmp[j];
    if (!(d >= 1e-15)) return false;  // (every lane reads the same pivot: a uniform exit)
#if defined(__HIP_DEVICE_COMPILE__)
    const double inv = rsqrt(d);
#else
    const double inv = 1.0 / sqrt(d);
#endif
    MZ_FOR(i, n + 1) if (i >= j) {
      if (i < n) A[i][j] = i == j ? d * inv : s.ctmp[i] * inv; else x[j] = s.ctmp[i] * inv;  // (x[j] <- y_j: entries k > j of b are still to come)
      if (i == j) s.cdinv[j] = inv;
    }
    cx.sync();
  }
  for (int j = n - 1; j >= 0; j--) {  // L^T x = y
    const double xj = x[j] * s.cdinv[j];
    cx.sync();  // (everybody has read x[j] before its owner overwrites it)
    MZ_FOR(i, n) { if (i == j) x[i] = xj; else if (i < j) x[i] -= A[j][i] * xj; }
    cx.sync();
  }
  return true;
}

// ------------------------------------------------------------------ constraint solve: primal Newton, exact line search
template <class C>
MZ_HD void gen_solve(const C& cx, const GenDev& K, GenScratch& s) {
  const int nv = K.m.nv, ncon = s.ncon, nlim = s.nlim;
  if (ncon == 0 && nlim == 0) { MZ_FOR(i, nv) s.qacc[i] = s.qas[i]; MZ_FOR(one, 1) s.iters = 0; cx.sync(); return; }
  auto cost_at = [&](const double* x) {  // partial cost over this lane's rows (+ the smooth part on its dofs)
    double c = 0.0;
    MZ_FOR(i, nv) c += 0.5 * gd_dotn_diff(s.M[i], x, s.qas, nv) * (x[i] - s.qas[i]);
    MZ_FOR(k, ncon) {
      double u[3];
      gd_dot3n(gen_cj(s, nv, k, 0), x, nv, u);
      for (int a = 0; a < 3; a++) u[a] -= s.caref[k][a];
      c += gen_contact_eval(s.cD[k], u, nullptr, nullptr);
    }
    MZ_FOR(l, nlim) { const double jar = s.lsign[l] * x[s.ldof[l]] - s.laref[l]; if (jar < 0.0) c += 0.5 * s.lD[l] * jar * jar; }
    return cx.gsum(c);
  };
  // warm start: the better of qacc_warmstart and qacc_smooth, by cost (mj_fwdConstraint)
  const double cw = cost_at(s.warm), cs = cost_at(s.qas);
  cx.sync();
  MZ_FOR(i, nv) s.qacc[i] = cw < cs ? s.warm[i] : s.qas[i];
  cx.sync();
  bool done = false;
  int it = 0;
  double prev_cost = cw < cs ? cw : cs;
  GEN_TICK(7);
  while (!done && it < K.max_iter) {
    MZ_FOR(i, nv) s.Mx[i] = gd_dotn_diff(s.M[i], s.qacc, s.qas, nv);
    MZ_FOR(k, ncon) {  // a contact's three residuals, and from them its gradient and Hessian weights — once, not once per entry that uses them
      double u[3];
      gd_dot3n(gen_cj(s, nv, k, 0), s.qacc, nv, u);
      for (int a = 0; a < 3; a++) { u[a] -= s.caref[k][a]; s.cu[k][a] = u[a]; }
      gen_contact_eval(s.cD[k], u, s.cg[k], s.cW[k]);
    }
    MZ_FOR(l, nlim) s.ljar[l] = s.lsign[l] * s.qacc[s.ldof[l]] - s.laref[l];
    cx.sync();
    GEN_TICK(8);
    double gpart = 0.0;
    MZ_FOR(i, nv) {
      double g = s.Mx[i];
#pragma unroll 2
      for (int k = 0; k < ncon; k++) { const double* Jk = gen_cj(s, nv, k, 0); g += Jk[i] * s.cg[k][0] + Jk[nv + i] * s.cg[k][1] + Jk[2 * nv + i] * s.cg[k][2]; }
      for (int l = 0; l < nlim; l++) if (s.ldof[l] == i && s.ljar[l] < 0.0) g += s.lsign[l] * s.lD[l] * s.ljar[l];
      s.grad[i] = g; gpart += g * g;
    }
    const double gn = sqrt(cx.gsum(gpart));
    GEN_TICK(9);
    if (K.inv_scale * gn < K.tol) break;
    // H = M + J^T W J: the lower triangle only (all the factorisation reads).  Rows r and nv - 1 - r together hold nv + 1 entries:
    // (nv + 1) * ceil(nv / 2) items cover the triangle (nv odd: the middle row twice, same values)
    MZ_FOR(e, (nv + 1) * ((nv + 1) / 2)) {
      const int r = e / (nv + 1), c = e - (nv + 1) * r;
      const int i = c <= r ? r : nv - 1 - r, j = c <= r ? c : c - r - 1;
      double acc = s.M[i][j];
#pragma unroll 2
      for (int k = 0; k < ncon; k++) {
        const double* W = s.cW[k];
        const double* Jk = gen_cj(s, nv, k, 0);
        const double ni = Jk[i], pi = Jk[nv + i], qi = Jk[2 * nv + i], nj = Jk[j], pj = Jk[nv + j], qj = Jk[2 * nv + j];
        acc += W[0] * ni * nj + W[1] * (ni * pj + pi * nj) + W[2] * (ni * qj + qi * nj) + W[3] * pi * pj + W[4] * qi * qj;
      }
      if (i == j) for (int l = 0; l < nlim; l++) if (s.ldof[l] == i && s.ljar[l] < 0.0) acc += s.lD[l];
      s.H[i][j] = acc;
    }
    cx.sync();
    GEN_TICK(10);
    MZ_FOR(i, nv) s.search[i] = -s.grad[i];
    cx.sync();
    const bool posdef = gen_chol_solve(cx, s, s.H, s.H, nv, s.search);
    GEN_TICK(11);
    if (!posdef) { MZ_FOR(one, 1) s.status |= MZ_STATUS_BAD_STATE; break; }
    // line search on phi(alpha) = cost(qacc + alpha search): unit step when no row changes state, else safeguarded Newton on phi'
    MZ_FOR(i, nv) s.Ms[i] = gd_dotn(s.M[i], s.search, nv);
    MZ_FOR(e, 3 * ncon) s.cjv[e / 3][e % 3] = gd_dotn(s.cJf + (size_t)e * nv, s.search, nv);
    MZ_FOR(l, nlim) s.ljv[l] = s.lsign[l] * s.search[s.ldof[l]];
    cx.sync();
    bool changed = false;
    MZ_FOR(k, ncon) { const double w[3] = {s.cu[k][0] + s.cjv[k][0], s.cu[k][1] + s.cjv[k][1], s.cu[k][2] + s.cjv[k][2]}; changed = changed || gen_rows_flip(s.cD[k], s.cu[k], w); }
    MZ_FOR(l, nlim) changed = changed || ((s.ljar[l] < 0.0) != (s.ljar[l] + s.ljv[l] < 0.0));
    changed = cx.gany(changed);
    double alpha = 1.0;
    if (changed) {
      double p1 = 0.0, p2 = 0.0;
      MZ_FOR(i, nv) { p1 += s.search[i] * s.Mx[i]; p2 += s.search[i] * s.Ms[i]; }
      p1 = cx.gsum(p1); p2 = cx.gsum(p2);
      double lo = 0.0, hi = -1.0, prev_d2 = -1.0;
      for (int ls = 0; ls < K.ls_iter; ls++) {
        double d1 = 0.0, d2 = 0.0;
        MZ_FOR(k, ncon) gen_rows_slope(s.cD[k], s.cu[k], s.cjv[k], alpha, &d1, &d2);
        MZ_FOR(l, nlim) { const double r = s.ljar[l] + alpha * s.ljv[l]; if (r < 0.0) { d1 += s.lD[l] * r * s.ljv[l]; d2 += s.lD[l] * s.ljv[l] * s.ljv[l]; } }
        d1 = cx.gsum(d1) + p1 + alpha * p2;
        d2 = cx.gsum(d2) + p2;
        if (d2 == prev_d2) break;
        prev_d2 = d2;
        if (d1 < 0.0) lo = alpha; else hi = alpha;
        double next = alpha - d1 / d2;
        if (hi >= 0.0 && !(next > lo && next < hi)) next = 0.5 * (lo + hi);
        if (!(next > 0.0)) next = hi >= 0.0 ? 0.5 * (lo + hi) : 0.0;
        if (fabs(next - alpha) <= 1e-15 * fabs(next)) { alpha = next; break; }
        alpha = next;
      }
    }
    cx.sync();
    GEN_TICK(12);
    if (!(alpha > 0.0)) break;
    MZ_FOR(i, nv) s.qacc[i] += alpha * s.search[i];
    cx.sync();
    it++;
    if (!changed) break;  // the unit Newton step stayed inside one active set: exact minimiser
    const double cnow = cost_at(s.qacc);
    GEN_TICK(13);
    if (!(K.inv_scale * (prev_cost - cnow) > 0.0)) break;  // round-off floor
    prev_cost = cnow;
    if (it >= K.max_iter) { MZ_FOR(one, 1) s.status |= MZ_STATUS_SOLVER_MAXITER; }
  }
  MZ_FOR(one, 1) s.iters = it;
  cx.sync();
}

// ------------------------------------------------------------------ one forward-dynamics evaluation
template <class C>
MZ_HD void gen_forward(const C& cx, const GenDev& K, GenScratch& s) {
  const mz_model& m = K.m;
  GEN_TICK(15);
  MZ_FOR(one, 1) s.npool = 0;
  gen_kinematics(cx, K, s);
  GEN_TICK(0);
  MZ_FOR(g, m.ngeom) gen_geom_item(K, s, g);
  MZ_FOR(j, m.njnt) gen_axis_item(K, s, j);
  MZ_FOR(b, m.nbody) if (b > 0) gen_inertia_item(K, s, b);
  cx.sync();
  GEN_TICK(1);
  MZ_FOR(it, K.nitem) gen_collide_item(K, s, it);
  GEN_TICK(2);
  gen_tree_passes(cx, K, s);
  GEN_TICK(3);
  MZ_FOR(i, m.nv) gen_mass_item(K, s, i);
  MZ_FOR(b, m.nbody) gen_fluid_item(K, s, b);
  {
    const int np = s.npool < GN_POOL ? s.npool : GN_POOL;
    double cnt = 0.0, lcnt = 0.0;
    MZ_FOR(e, np) { cnt += 1.0; gen_compact_item(K, s, e, np); }  // (every pool entry is an active contact: gen_collide_item's emit)
    MZ_FOR(e, 2 * K.nlimj) { gen_limit_flag(K, s, e); lcnt += (double)s.lflag[e]; }
    const int c = (int)cx.gsum(cnt), nl = (int)cx.gsum(lcnt);
    cx.sync();  // (the flags are visible)
    MZ_FOR(e, 2 * K.nlimj) gen_limit_row(K, s, e);
    MZ_FOR(one, 1) {
      if (s.npool > GN_POOL || c > GN_NC || nl > GN_NL) s.status |= MZ_STATUS_CONTACT_OVERFLOW;
      s.ncon = c < GN_NC ? c : GN_NC;
      s.nlim = nl < GN_NL ? nl : GN_NL;
    }
  }
  cx.sync();
  GEN_TICK(4);
  MZ_FOR(i, m.nv) gen_force_item(K, s, i);
  MZ_FOR(e, m.nv * m.nv) { const int i = e / m.nv, j = e - m.nv * i; if (i < j) s.M[i][j] = s.M[j][i]; }  // gen_mass_item wrote the lower triangle
  MZ_FOR(item, 3 * s.ncon) gen_contact_row_item(K, s, item);
  cx.sync();
  GEN_TICK(5);
  // qacc_smooth = M^-1 qfrc_smooth (H is free until the solver assembles the Hessian)
  MZ_FOR(i, m.nv) s.qas[i] = s.qfs[i];
  cx.sync();
  if (!gen_chol_solve(cx, s, s.M, s.H, m.nv, s.qas)) { MZ_FOR(one, 1) s.status |= MZ_STATUS_BAD_STATE; }
  cx.sync();
  GEN_TICK(6);
  gen_solve(cx, K, s);
  GEN_TICK(14);
}

// mj_integratePos of joint j (the joints are independent of each other: one lane each)
MZ_HD void gen_integrate_joint(const GenDev& K, GenScratch& s, const double* base, const double* vel, double h, int j) {
  {
    const int qa = s.tp.jnt_qposadr[j], da = s.tp.jnt_dofadr[j];
    if (s.tp.jnt_type[j] == MZ_JNT_FREE || s.tp.jnt_type[j] == MZ_JNT_BALL) {  // mj_integratePos: q <- q * exp(h w / 2), w in the child frame
      int qq = qa, dw = da;
      if (s.tp.jnt_type[j] == MZ_JNT_FREE) {
        for (int k = 0; k < 3; k++) s.qpos[qa + k] = base[qa + k] + h * vel[da + k];
        qq = qa + 3; dw = da + 3;
      }
      const double w[3] = {vel[dw], vel[dw + 1], vel[dw + 2]}, n = sqrt(gd_dot3(w, w));
      double q[4] = {base[qq], base[qq + 1], base[qq + 2], base[qq + 3]};
      if (n > 1e-15) {
        double sh, ch;
        sincos(0.5 * h * n, &sh, &ch);
        sh /= n;
        const double qr[4] = {ch, w[0] * sh, w[1] * sh, w[2] * sh};
        double qn[4];
        gd_quat_mul(qn, q, qr);
        for (int k = 0; k < 4; k++) q[k] = qn[k];
      }
      gd_quat_norm(q);
      for (int k = 0; k < 4; k++) s.qpos[qq + k] = q[k];
    } else {
      s.qpos[qa] = base[qa] + h * vel[da];
    }
  }
}

// one mj_step with RK4; state in s.qpos / s.qvel / s.warm, actuator forces in s.fact
template <class C>
MZ_HD void gen_mj_step(const C& cx, const GenDev& K, GenScratch& s) {
  const mz_model& m = K.m;
  const double h = m.timestep;
  if (!m.integrator_rk4) {
    // MuJoCo's default integrator (mj_EulerSkip as restated in oracle/mzo_physics.c mzo_mj_step): one evaluation per step, semi-implicit,
    // implicit in the joint damping: (M + h diag(damping)) qacc' = M qacc; qvel += h qacc'; qpos integrates the new velocity
    gen_forward(cx, K, s);
    bool damped = false;
    for (int i = 0; i < m.nv; i++) damped = damped || m.dof_damping[i] > 0.0;  // (uniform)
    MZ_FOR(i, m.nv) { s.Mx[i] = damped ? gd_dotn(s.M[i], s.qacc, m.nv) : s.qacc[i]; s.warm[i] = s.qacc[i]; }
    if (damped) {
      MZ_FOR(e, m.nv * m.nv) { const int i = e / m.nv, j = e - m.nv * i; if (j <= i) s.H[i][j] = s.M[i][j] + (i == j ? h * m.dof_damping[i] : 0.0); }
      cx.sync();
      if (!gen_chol_solve(cx, s, s.H, s.H, m.nv, s.Mx)) { MZ_FOR(one, 1) s.status |= MZ_STATUS_BAD_STATE; }
    }
    cx.sync();
    MZ_FOR(i, m.nq) s.x0q[i] = s.qpos[i];
    MZ_FOR(i, m.nv) s.qvel[i] += h * s.Mx[i];
    cx.sync();
    MZ_FOR(j, m.njnt) gen_integrate_joint(K, s, s.x0q, s.qvel, h, j);
    cx.sync();
    return;
  }
  MZ_FOR(i, m.nq) s.x0q[i] = s.qpos[i];
  MZ_FOR(i, m.nv) { s.x0v[i] = s.qvel[i]; s.accv[i] = 0.0; s.accf[i] = 0.0; }
  cx.sync();
  for (int st = 0; st < 4; st++) {
    gen_forward(cx, K, s);
    const double bw = (st == 0 || st == 3) ? 1.0 / 6.0 : 1.0 / 3.0, aw = st == 2 ? 1.0 : 0.5;
    MZ_FOR(i, m.nv) {
      s.accv[i] += bw * s.qvel[i]; s.accf[i] += bw * s.qacc[i];
      s.dxv[i] = aw * s.qvel[i];
      s.Mx[i] = s.x0v[i] + h * aw * s.qacc[i];
    }
    cx.sync();
    if (st < 3) {
      MZ_FOR(j, m.njnt) gen_integrate_joint(K, s, s.x0q, s.dxv, h, j);
      MZ_FOR(i, m.nv) s.qvel[i] = s.Mx[i];
      cx.sync();
    }
  }
  MZ_FOR(j, m.njnt) gen_integrate_joint(K, s, s.x0q, s.accv, h, j);
  MZ_FOR(i, m.nv) { s.qvel[i] = s.x0v[i] + h * s.accf[i]; s.warm[i] = s.qacc[i]; }  // qacc_warmstart: the last stage's qacc (mj_advance)
  cx.sync();
}

// ------------------------------------------------------------------ observation (maze_env.py:351-369)
// get_body_com of a movable body = its frame origin: the spawn position plus its slide coordinates along their axes; a free-joint
// body's origin is its qpos (mzo_env.c body_origin).  Movable bodies hang off the world, so no kinematics pass is needed.
template <class Q>
MZ_HD double gen_body_origin(const mz_model& m, const Q* qpos, int body, int c) {
  const int j0 = m.body_jntadr[body], jn = m.body_jntnum[body];
  if (jn == 1 && m.jnt_type[j0] == MZ_JNT_FREE) return (double)qpos[m.jnt_qposadr[j0] + c];
  double p = m.body_pos[body][c];
  for (int j = j0; j < j0 + jn; j++)
    if (m.jnt_type[j] == MZ_JNT_SLIDE) p += m.jnt_axis[j][c] * ((double)qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]);
  return p;
}
// element i of the observation WITHOUT the view: robot qpos[:3] | object balls' xpos | movable blocks' xpos (3 each, if observed) |
// robot qpos[3:nq_robot] | robot qvel[:nv_robot] | t / 1000
template <class Q>
MZ_HD float gen_obs_elem(const GenDev& K, const Q* qpos, const Q* qvel, int i, int t) {
  const mz_model& m = K.m;
  if (i < 3) return (float)qpos[i];
  if (i < 3 + K.obs_extra) {
    const int k = (i - 3) / 3, c = (i - 3) - 3 * k, nb = m.observe_balls ? m.nball : 0;
    return (float)gen_body_origin(m, qpos, k < nb ? m.ball_bodyid[k] : m.block_bodyid[k - nb], c);
  }
  const int q = i - K.obs_extra;
  if (q < m.nq_robot) return (float)qpos[q];
  if (q < m.nq_robot + m.nv_robot) return (float)qvel[q - m.nq_robot];
  return (float)t * 0.001f;
}
// the row as the API hands it out: `base` = its width without the view; with a top-down view (row width base + MZ_VIEW_DIM) the
// time entry moves behind the view and the movable blocks' x, y are parked in the view's first entries for mzk_view_fill
template <class Q>
MZ_HD void gen_store_obs(const GenDev& K, const Q* qpos, const Q* qvel, int t, float* row, int i) {
  const mz_model& m = K.m;
  const int base = m.nq_robot + m.nv_robot + 1 + K.obs_extra;
  if (i < base - 1) { row[i] = gen_obs_elem(K, qpos, qvel, i, t); return; }
  if (i == base - 1) {
    row[m.top_down_view ? base - 1 + MZ_VIEW_DIM : base - 1] = (float)t * 0.001f;
    if (m.top_down_view)
      for (int b = 0; b < m.nblock && b < 4; b++) {
        row[base - 1 + 2 * b] = (float)gen_body_origin(m, qpos, m.block_bodyid[b], 0);
        row[base + 2 * b] = (float)gen_body_origin(m, qpos, m.block_bodyid[b], 1);
      }
  }
}

// ------------------------------------------------------------------ MazeEnv.step (maze_env.py:448-481)
// The robot's own step in one of two shapes — motors (ant.py:61-73, swimmer.py:37-48: clamped motors, frame_skip x mj_step, forward
// reward |dxy| / dt, control cost on the raw action) or the Point's (point.py:44-61: heading and position moved by the action,
// velocities clipped, frame_skip x mj_step without control, no inner reward) — then the manual wall bounce where the robot asks
// for it (maze_env.py:451-464), the observation, the task's reward and termination.
// obs: a row of obs_dim floats (view entries left open, see gen_store_obs).
template <class C>
MZ_HD void gen_env_step(const C& cx, const GenDev& K, GenScratch& s, const float* action, float* obs, float* reward, uint8_t* done, int* goal_idx,
                        float* info, int* t_io, int env = -1) {
  const mz_model& m = K.m;
  gen_load_topology(cx, K, s);
  MZ_FOR(i, m.nv) s.fact[i] = 0.0;
  MZ_FOR(one, 1) { s.status = 0; s.red[1] = s.qpos[0]; s.red[2] = s.qpos[1]; }
  cx.sync();
  if (K.step_kind == GN_STEP_POINT) {
    MZ_FOR(one, 1) {  // point.py:45-56
      const double pi = 3.14159265358979323846;
      double th = s.qpos[2] + (double)action[1];
      if (th < -pi) th += pi * 2; else if (pi < th) th -= pi * 2;
      s.qpos[2] = th;
      s.qpos[0] += cos(th) * (double)action[0];
      s.qpos[1] += sin(th) * (double)action[0];
    }
    MZ_FOR(i, m.nv) s.qvel[i] = fmin(fmax(s.qvel[i], -m.velocity_limit), m.velocity_limit);
  } else {
    MZ_FOR(one, 1)
      for (int u = 0; u < m.nu; u++) {
        double c = (double)action[u];
        if (m.act_ctrllimited[u]) c = fmin(fmax(c, m.act_ctrlrange[u][0]), m.act_ctrlrange[u][1]);
        s.fact[m.act_dofid[u]] += m.act_gear[u] * (m.act_gainprm[u] * c + m.act_biasprm[u][0]);  // (a servo's state-dependent part: gen_force_item)
      }
  }
  cx.sync();
  for (int f = 0; f < m.frame_skip; f++) gen_mj_step(cx, K, s);
  if (K.nseg > 0) {  // CollisionDetector.detect + bounce / give-up on the robot's xy (maze_env.py:451-464)
    MZ_FOR(one, 1) {
      const double old_xy[2] = {s.red[1], s.red[2]}, new_xy[2] = {s.qpos[0], s.qpos[1]};
      double fin[2];
      const int r = point_bounce(K, old_xy, new_xy, fin, (double*)nullptr);
      if (r < 0) s.status |= MZ_STATUS_COLLINEAR;
      s.qpos[0] = fin[0]; s.qpos[1] = fin[1];
    }
    cx.sync();
  }
  const int t = *t_io + 1, base = m.nq_robot + m.nv_robot + 1 + K.obs_extra;
  MZ_FOR(i, base) gen_store_obs(K, s.qpos, s.qvel, t, obs, i);
  cx.sync();
  MZ_FOR(one, 1) {
    double fwd = 0.0, cc = 0.0, inner = 0.0;
    if (K.step_kind != GN_STEP_POINT) {
      const double dt = m.timestep * m.frame_skip, vx = (s.qpos[0] - s.red[1]) / dt, vy = (s.qpos[1] - s.red[2]) / dt;
      fwd = sqrt(vx * vx + vy * vy);
      for (int u = 0; u < m.nu; u++) cc += (double)action[u] * (double)action[u];
      cc *= K.task.ctrl_w;
      inner = K.task.fwd_w * fwd - cc;
    }
    float o6[6];
    for (int k = 0; k < 6; k++) o6[k] = k < base - 1 ? obs[k] : 0.f;
    float outer; int tm, gi;
    task_eval_dev(K.task, o6, &outer, &tm, &gi, env);
    *reward = (float)(K.task.inner_scale * inner + (double)outer);
    *done = (uint8_t)((tm ? 1 : 0) | (t >= K.task.max_steps ? 2 : 0));
    if (goal_idx) *goal_idx = gi;
    if (info) { info[0] = (float)s.qpos[0]; info[1] = (float)s.qpos[1]; info[2] = (float)fwd; info[3] = (float)-cc; }
    bool badv = false;
    for (int i = 0; i < m.nq; i++) badv = badv || !(fabs(s.qpos[i]) < 1e10);
    for (int i = 0; i < m.nv; i++) badv = badv || !(fabs(s.qvel[i]) < 1e10);
    if (badv) s.status |= MZ_STATUS_BAD_STATE;
  }
  cx.sync();
  MZ_FOR(one, 1) *t_io = t;
  cx.sync();
}
