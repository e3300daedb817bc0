// point_dyn.h — the Point robot: model constants (PointDev), the manual wall detector and the closed-form
// unconstrained dynamics.  The step itself (contacts, Newton solve, RK4, bounce) is the lane-group code in
// planar_dyn.h, which also carries the movable blocks of the Push / BlockMaze family.
//
// Replaces (per env): PointEnv.step (mujoco_maze/point.py:44-61: heading update with
// the single +-2pi wrap, teleport along the heading, qvel clip, mj_step x 1 with ctrl
// never written), the manual wall bounce of MazeEnv.step (maze_env.py:451-464) with
// CollisionDetector.detect / Line.* (maze_env_utils.py:96-123,186-206), the observation
// (point.py:63-69 + maze_env.py:351-369) and MazeTask reward / termination.
//
// Dynamics: the 3-dof model (slide x, slide y, hinge z with the COM offset c on the body
// x axis, point.xml:21-25) has the closed-form forward dynamics
//     qacc = (c w^2 cos th, c w^2 sin th, 0)
// (the mass matrix couples the hinge to the slides through m c; the centripetal bias
// accelerates the joint origin so that the COM moves straight) — integrated with RK4
// exactly as mj_step does.  The arithmetic is fp64: the path is a few hundred flops per env and fp64 keeps
// the wall-hit / give-up decisions identical to the float64 reference logic; the state is stored as fp32.
//
// Near a maze wall MuJoCo's own contacts also fire (the 0.5 sphere and the arrow box are
// collidable while the manual detector uses radius 0.4): sphere-box and (z-rotated) box-box contacts enter as
// pyramidal soft constraints and the dense Newton problem is solved exactly like on the Ant path.
#pragma once
#include <math.h>
#include <stdint.h>

#include "ant_model.h"

#define MZ_STATUS_COLLINEAR 8

struct PtPair {  // mixed contact parameters of one geom-pair class (MuJoCo: max margin / friction, mean solref / solimp)
  double margin, mu, K, B, solimp[5], wsum;  // wsum = body_invweight0 (translation) of the two bodies
};

struct PointDev {
  double h, com_x, vel_limit, restitution;
  int frame_skip, nseg;
  double seg[MZ_MAX_SEG][4];
  double reach;  // arrow tip distance from the torso origin (wall broad phase)
  // MuJoCo contact regime: body, geoms, pair classes 0 robot-wall, 1 robot-block, 2 wall-block, 3 block-block,
  // 4 ball-wall, 5 robot-ball
  double mass, izz, inv_scale;
  double sph_r, sph_z, arr_off, arr_hx, arr_hy, arr_hz, arr_z, arr_rxy;  // arr_rxy: radius of the arrow's circumscribed circle (top view)
  PtPair pair[8];  // + 6 joint-limit rows of a block's slides (margin = joint margin), 7 floor plane vs block
  // movable XY blocks (maze_env.py:563-660), all of one size
  int nblock, observe_blocks;
  double block_mass, block_half[3], block_pos0[3][3];
  // the two slide axes of a block: (x, y) in the Push family, (y, z) — limited, the z slide under gravity — in Fall mazes
  int block_axis[2], block_limited;
  double block_lo[2], block_hi[2], gz;
  // object ball (Billiard, maze_env.py:489-536): slide-x + slide-y + z hinge body, sphere of radius ball_r at height ball_r
  int nball, observe_balls;
  double ball_mass, ball_izz, ball_r, ball_pos0[3];
  MazeDev maze;
  TaskDev task;
  double qpos0[3];
  int reset_kind;
  int unit_steps;  // Newton iterations of a solve that take the unit step before the exact line search takes over (option "ls_fast_iterations"; planar_dyn.h MZ_PL_UNIT_STEPS)
};

static inline void pt_mix_pair(PtPair* p, double h, double m1, double m2, const double* f1, const double* f2, const double* sr1,
                               const double* sr2, const double* si1, const double* si2, double wsum) {
  double sr[2] = {0.5 * (sr1[0] + sr2[0]), 0.5 * (sr1[1] + sr2[1])};
  for (int k = 0; k < 5; k++) p->solimp[k] = 0.5 * (si1[k] + si2[k]);
  p->margin = fmax(m1, m2);
  p->mu = fmax(f1[0], f2[0]);
  double tc = fmax(sr[0], 2.0 * h), dmax = p->solimp[1];
  p->K = 1.0 / (dmax * dmax * tc * tc * sr[1] * sr[1]);
  p->B = 2.0 / (dmax * tc);
  p->wsum = wsum;
}

static inline int point_dev_from_model(PointDev* p, const mz_model* m, char* err, int errlen) {
  memset(p, 0, sizeof(*p));
  const int nb = m->nblock, ns = m->nball;
  if (ns < 0 || ns > 1 || (ns && nb)) return ant_fail(err, errlen, "point kernel: at most one object ball, and not together with movable blocks");
  if (m->robot != MZ_ROBOT_POINT || nb < 0 || nb > 3 || m->nv != 3 + 2 * nb + 3 * ns || m->nq != 3 + 2 * nb + 3 * ns || m->jnt_type[0] != MZ_JNT_SLIDE ||
      m->jnt_type[1] != MZ_JNT_SLIDE || m->jnt_type[2] != MZ_JNT_HINGE || fabs(m->body_ipos[1][1]) > 1e-12)
    return ant_fail(err, errlen, "point kernel: model is not the slide-slide-hinge point robot (+ up to 3 XY blocks)");
  p->h = m->timestep; p->com_x = m->body_ipos[1][0]; p->vel_limit = m->velocity_limit; p->restitution = m->restitution;
  p->frame_skip = m->frame_skip; p->nseg = m->manual_collision ? m->nseg : 0;
  for (int k = 0; k < m->nseg; k++) for (int j = 0; j < 4; j++) p->seg[k][j] = m->seg[k][j];
  p->mass = m->body_mass[1];
  p->izz = m->body_inertia[1][2] + m->body_mass[1] * m->body_ipos[1][0] * m->body_ipos[1][0];
  p->inv_scale = 1.0 / (m->meaninertia * m->nv);
  if (m->ngeom != 3 + nb + ns || m->geom_type[1] != MZ_GEOM_SPHERE || m->geom_type[2] != MZ_GEOM_BOX)
    return ant_fail(err, errlen, "point kernel: expected floor + sphere + arrow box (+ block) geoms");
  p->sph_r = m->geom_size[1][0]; p->sph_z = m->body_pos[1][2] + m->geom_pos[1][2];  // torso body height: 0, or 0.75 + platform height in elevated mazes
  p->arr_off = m->geom_pos[2][0]; p->arr_hx = m->geom_size[2][0]; p->arr_hy = m->geom_size[2][1]; p->arr_hz = m->geom_size[2][2];
  p->arr_z = m->body_pos[1][2] + m->geom_pos[2][2];
  p->arr_rxy = sqrt(p->arr_hx * p->arr_hx + p->arr_hy * p->arr_hy);
  for (int k = 0; k < 5; k++)
    if (m->geom_solimp[1][k] != m->geom_solimp[2][k]) return ant_fail(err, errlen, "point kernel: sphere and arrow must share contact parameters");
  if (m->geom_margin[1] != m->geom_margin[2] || m->geom_margin[0] != 0.0 || m->geom_margin[1] != 0.0)
    return ant_fail(err, errlen, "point kernel: geom margins must be 0 (the floor touches sphere and blocks at dist = 0: no contact)");
  const double bw_robot = m->body_invweight0[1][0];
  pt_mix_pair(&p->pair[0], m->timestep, m->geom_margin[1], m->wall_margin, m->geom_friction[1], m->wall_friction, m->geom_solref[1],
              m->wall_solref, m->geom_solimp[1], m->wall_solimp, bw_robot);
  p->nblock = nb; p->observe_blocks = m->observe_blocks;
  p->gz = m->gravity[2];
  for (int k = 0; k < nb; k++) {
    int b = m->block_bodyid[k], g = m->block_geomid[k], j0 = m->body_jntadr[b];
    int ax[2] = {-1, -1};
    for (int a = 0; a < 2 && m->body_jntnum[b] == 2; a++)
      for (int c = 0; c < 3; c++) if (fabs(m->jnt_axis[j0 + a][c] - 1.0) < 1e-12) ax[a] = c;
    if (m->body_jntnum[b] != 2 || m->jnt_type[j0] != MZ_JNT_SLIDE || m->jnt_type[j0 + 1] != MZ_JNT_SLIDE || m->geom_type[g] != MZ_GEOM_BOX ||
        ax[0] < 0 || ax[1] <= ax[0] || m->body_dofadr[b] != 3 + 2 * k || m->jnt_limited[j0] != m->jnt_limited[j0 + 1] || m->geom_margin[g] != 0.0 ||
        (k > 0 && (ax[0] != p->block_axis[0] || ax[1] != p->block_axis[1] || m->jnt_limited[j0] != p->block_limited)))
      return ant_fail(err, errlen, "point kernel: a movable block is a box body with two slides along increasing coordinate axes (x y, y z or x z), margin 0");
    p->block_axis[0] = ax[0]; p->block_axis[1] = ax[1]; p->block_limited = m->jnt_limited[j0];
    for (int a = 0; a < 2; a++) { p->block_lo[a] = m->jnt_range[j0 + a][0]; p->block_hi[a] = m->jnt_range[j0 + a][1]; }
    if (p->block_limited) {  // limit rows: one-sided single rows with the joint's solref / solimp and margin; R from dof_invweight0
      PtPair* q = &p->pair[6];
      double tc = fmax(m->jnt_solref[j0][0], 2.0 * m->timestep), dmax = m->jnt_solimp[j0][1];
      q->margin = m->jnt_margin[j0]; q->mu = 0.0; q->K = 1.0 / (dmax * dmax * tc * tc * m->jnt_solref[j0][1] * m->jnt_solref[j0][1]);
      q->B = 2.0 / (dmax * tc); q->wsum = m->dof_invweight0[m->jnt_dofadr[j0]];
      for (int c = 0; c < 5; c++) q->solimp[c] = m->jnt_solimp[j0][c];
    }
    for (int q = 0; q < 3; q++) { p->block_pos0[k][q] = m->body_pos[b][q]; p->block_half[q] = m->geom_size[g][q]; }
    p->block_mass = m->body_mass[b];
    if (k > 0 && (m->geom_size[g][0] != m->geom_size[m->block_geomid[0]][0] || m->body_mass[b] != m->body_mass[m->block_bodyid[0]]))
      return ant_fail(err, errlen, "point kernel: movable blocks must share one size and mass");
  }
  if (nb > 0) {
    int g = m->block_geomid[0];
    const double bw_block = m->body_invweight0[m->block_bodyid[0]][0];
    pt_mix_pair(&p->pair[1], m->timestep, m->geom_margin[1], m->geom_margin[g], m->geom_friction[1], m->geom_friction[g], m->geom_solref[1],
                m->geom_solref[g], m->geom_solimp[1], m->geom_solimp[g], bw_robot + bw_block);
    pt_mix_pair(&p->pair[2], m->timestep, m->wall_margin, m->geom_margin[g], m->wall_friction, m->geom_friction[g], m->wall_solref,
                m->geom_solref[g], m->wall_solimp, m->geom_solimp[g], bw_block);
    pt_mix_pair(&p->pair[3], m->timestep, m->geom_margin[g], m->geom_margin[g], m->geom_friction[g], m->geom_friction[g], m->geom_solref[g],
                m->geom_solref[g], m->geom_solimp[g], m->geom_solimp[g], 2.0 * bw_block);
    pt_mix_pair(&p->pair[7], m->timestep, m->geom_margin[0], m->geom_margin[g], m->geom_friction[0], m->geom_friction[g], m->geom_solref[0],
                m->geom_solref[g], m->geom_solimp[0], m->geom_solimp[g], bw_block);
  }
  p->nball = ns; p->observe_balls = m->observe_balls;
  if (ns > 0) {
    int b = m->ball_bodyid[0], g = m->ball_geomid[0], j0 = m->body_jntadr[b];
    if (m->body_jntnum[b] != 3 || m->jnt_type[j0] != MZ_JNT_SLIDE || m->jnt_type[j0 + 1] != MZ_JNT_SLIDE || m->jnt_type[j0 + 2] != MZ_JNT_HINGE ||
        m->geom_type[g] != MZ_GEOM_SPHERE || m->body_dofadr[b] != 3 || fabs(m->jnt_axis[j0][0] - 1.0) > 1e-12 ||
        fabs(m->jnt_axis[j0 + 1][1] - 1.0) > 1e-12 || fabs(m->jnt_axis[j0 + 2][2] - 1.0) > 1e-12 || m->jnt_limited[j0] || m->jnt_limited[j0 + 1] ||
        m->jnt_limited[j0 + 2] || m->geom_margin[g] != 0.0 || fabs(m->geom_pos[g][0]) + fabs(m->geom_pos[g][1]) > 1e-12 ||
        fabs(m->geom_pos[g][2] - m->geom_size[g][0]) > 1e-12 || m->dof_armature[3] != 0.0 || m->dof_damping[3] != 0.0 || m->dof_damping[5] != 0.0)
      return ant_fail(err, errlen, "point kernel: object ball is not the slide-x / slide-y / hinge-z sphere body resting on the floor");
    p->ball_mass = m->body_mass[b]; p->ball_izz = m->body_inertia[b][2]; p->ball_r = m->geom_size[g][0];
    for (int q = 0; q < 3; q++) p->ball_pos0[q] = m->body_pos[b][q];
    const double bw_ball = m->body_invweight0[b][0];
    pt_mix_pair(&p->pair[4], m->timestep, m->geom_margin[g], m->wall_margin, m->geom_friction[g], m->wall_friction, m->geom_solref[g],
                m->wall_solref, m->geom_solimp[g], m->wall_solimp, bw_ball);
    pt_mix_pair(&p->pair[5], m->timestep, m->geom_margin[1], m->geom_margin[g], m->geom_friction[1], m->geom_friction[g], m->geom_solref[1],
                m->geom_solref[g], m->geom_solimp[1], m->geom_solimp[g], bw_robot + bw_ball);
  }
  if (m->wall_margin != 0.0) return ant_fail(err, errlen, "point kernel: wall margin must be 0");
  p->reach = 0.0;
  for (int g = 1; g < 3; g++) {
    double r = hypot(m->geom_pos[g][0], m->geom_pos[g][1]) + m->geom_rbound[g];
    if (r > p->reach) p->reach = r;
  }
  if (p->reach >= m->maze_scale) return ant_fail(err, errlen, "point kernel: maze cells must be wider than the robot's reach");
  if (m->elevated) {
    // the robot has no z dof: lifted with the torso (maze_env.py:102-107) it hovers above the platforms and can touch neither
    // them nor the floor — the kernel relies on that
    const double top = m->height_offset, zoff = m->body_pos[1][2];
    if (zoff + m->geom_pos[1][2] - m->geom_size[1][0] < top || zoff + m->geom_pos[2][2] - m->geom_size[2][2] < top)
      return ant_fail(err, errlen, "point kernel: in an elevated maze the robot's geoms must stay above the platforms");
  }
  maze_dev_from_model(&p->maze, m);
  task_dev_from_model(&p->task, m);
  for (int k = 0; k < 3; k++) p->qpos0[k] = m->qpos0[k];
  p->reset_kind = m->reset_qvel_kind;
  p->unit_steps = 5;  // planar_dyn.h MZ_PL_UNIT_STEPS: the measured optimum (profiles/r05/point_unit_steps_ab.txt); option "ls_fast_iterations"
  return MZ_OK;
}

#if defined(__HIPCC__)
#define MZP_HD __host__ __device__ __forceinline__
#else
#define MZP_HD inline
#endif

// ---- the manual wall detector: float64 arithmetic of the reference, operation for operation.
// The reference computes with Python complex numbers and floats (maze_env_utils.py:84-123): every product, sum and
// quotient is a separately rounded IEEE double operation and abs(complex) is C hypot().  The functions below keep that:
// `#pragma clang fp contract(off) reciprocal(off) reassociate(off)` — no fused multiply-add, IEEE division whatever the
// command line says — a correctly rounded square root (this translation unit is built without -fapprox-func,
// csrc/Makefile), and mz_hypot restates glibc's hypot so that the "nearest collision" comparison sees the same distances.

// glibc 2.35 __hypot (sysdeps/ieee754/dbl-64/e_hypot.c, the non-FMA kernel that x86-64 builds use), valid for the
// magnitudes a maze coordinate can take (no scaling branches: |x|, |y| in [2^-459, 2^511] or zero).  Checked bit for bit
// against libm hypot on the host (tests/test_maze_golden.py) — CPython's abs(complex) calls exactly that function.
MZP_HD double mz_hypot(double x, double y) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  double ax = fabs(x), ay = fabs(y);
  if (ax < ay) { double t = ax; ax = ay; ay = t; }
  if (ax >= ay * 0x1p54) return ax + ay;
  double h = sqrt(ax * ax + ay * ay), t1, t2;
  if (h <= 2.0 * ay) {
    double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  } else {
    double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t1 + t2) / (2.0 * h);
  return h;
}

// (conj(a) * b).imag of maze_env_utils.py:99 with a = ax + i ay, b = bx + i by:  ax * by + (-ay) * bx
MZP_HD double cross2d(double ax, double ay, double bx, double by) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  return ax * by + (-ay) * bx;
}

// CollisionDetector.detect (maze_env_utils.py:186-206).  The loop over the wall segments keeps the nearest collision, the first
// one among equals (`dist < best`, strict).  point_detect_range is that loop body over the segments k0, k0 + kstep, ...: the whole
// table for the serial form (point_detect: the golden-vector kernel mz_debug_detect, the host emulation), one residue class per
// lane for the step kernel's lane groups (planar_dyn.h point_detect_group, which then takes the minimum by (dist, k) — the same
// winner).  The arithmetic of a segment is the same function either way.  PD: anything with nseg / seg / restitution (PointDev;
// the general engine's GenDev, generic_dyn.h).
struct PtCand { int found, degenerate, k; double dist, pt[2], rf[2]; };
template <class PD>
MZP_HD void point_detect_range(const PD& P, const double* o, const double* n, double mvx, double mvy, int k0, int kstep, PtCand& c) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  for (int k = k0; k < P.nseg; k += kstep) {
    const double* s = P.seg[k];
    double wx = s[2] - s[0], wy = s[3] - s[1];
    double c1 = cross2d(wx, wy, o[0] - s[0], o[1] - s[1]), c2 = cross2d(wx, wy, n[0] - s[0], n[1] - s[1]);
    if (!(c1 * c2 <= 0.0)) continue;
    double c3 = cross2d(mvx, mvy, s[0] - o[0], s[1] - o[1]), c4 = cross2d(mvx, mvy, s[2] - o[0], s[3] - o[1]);
    if (!(c3 * c4 <= 0.0)) continue;
    double a = cross2d(wx, wy, mvx, mvy), b = cross2d(wx, wy, s[2] - o[0], s[3] - o[1]);
    if (a == 0.0) { c.degenerate = 1; continue; }
    double r = b / a, px = o[0] + r * mvx, py = o[1] + r * mvy;
    double dist = mz_hypot(px - o[0], py - o[1]);
    if (!c.found || dist < c.dist) {
      c.found = 1; c.dist = dist; c.k = k;
      c.pt[0] = px; c.pt[1] = py;
      double bx = -wx, by = -wy, n2 = mz_hypot(bx, by);
      n2 = n2 * n2;
      double dx = n[0] - s[0], dy = n[1] - s[1];
      double sc = (dx * bx - (-dy) * by) / n2;
      double qx = s[0] + bx * sc, qy = s[1] + by * sc;
      c.rf[0] = n[0] + 2.0 * (qx - n[0]);
      c.rf[1] = n[1] + 2.0 * (qy - n[1]);
    }
  }
}

// 1 hit, 0 none, -1 collinear (the reference raises ZeroDivisionError)
template <class PD>
MZP_HD int point_detect(const PD& P, const double* o, const double* n, double* pt, double* rf) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  double mvx = n[0] - o[0], mvy = n[1] - o[1];
  if (mz_hypot(mvx, mvy) <= 1e-8) return 0;
  PtCand c;
  c.found = 0; c.degenerate = 0; c.k = 0; c.dist = 0.0; c.pt[0] = c.pt[1] = c.rf[0] = c.rf[1] = 0.0;
  point_detect_range(P, o, n, mvx, mvy, 0, 1, c);
  if (c.found) { pt[0] = c.pt[0]; pt[1] = c.pt[1]; rf[0] = c.rf[0]; rf[1] = c.rf[1]; }
  if (c.degenerate && !c.found) return -1;
  return c.found;
}

// Wall bounce of MazeEnv.step (maze_env.py:457-464): 0 no hit, 1 bounced to `fin`, 2 gave up (fin = old position),
// -1 where the reference would have raised (collinear move; fin = new position)
template <class PD>
MZP_HD int point_bounce(const PD& P, const double* old_xy, const double* new_xy, double* fin, double* hit_pt) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  double pt[2] = {0.0, 0.0}, rf[2] = {0.0, 0.0};
  fin[0] = new_xy[0]; fin[1] = new_xy[1];
  int hit = point_detect(P, old_xy, new_xy, pt, rf);
  if (hit_pt) { hit_pt[0] = pt[0]; hit_pt[1] = pt[1]; }
  if (hit <= 0) return hit;
  double pos[2] = {pt[0] + P.restitution * (rf[0] - pt[0]), pt[1] + P.restitution * (rf[1] - pt[1])}, p2[2], r2[2];
  int again = point_detect(P, old_xy, pos, p2, r2);
  if (again < 0) return -1;
  if (again > 0) { fin[0] = old_xy[0]; fin[1] = old_xy[1]; return 2; }
  fin[0] = pos[0]; fin[1] = pos[1];
  return 1;
}

MZP_HD void point_qacc(const PointDev& P, const double* q, const double* v, double* a) {
  double w2 = v[2] * v[2];
  a[0] = P.com_x * w2 * cos(q[2]);
  a[1] = P.com_x * w2 * sin(q[2]);
  a[2] = 0.0;
}

// distance from the torso origin to the nearest BLOCK cell box (xy), used only to flag the
// unmodelled MuJoCo contact regime
MZP_HD bool point_near_wall(const PointDev& P, double x, double y) {
  const MazeDev& z = P.maze;
  double inv = 1.0 / z.scale, reach = P.reach;
  int j0 = (int)floor((x - reach + z.tx) * inv + 0.5), j1 = (int)floor((x + reach + z.tx) * inv + 0.5);
  int i0 = (int)floor((y - reach + z.ty) * inv + 0.5), i1 = (int)floor((y + reach + z.ty) * inv + 0.5);
  for (int i = i0; i <= i1; i++)
    for (int j = j0; j <= j1; j++) {
      if (i < 0 || j < 0 || i >= z.rows || j >= z.cols) continue;
      if (!((z.rowmask[i] >> j) & 1u)) continue;
      double cx = j * (double)z.scale - z.tx, cy = i * (double)z.scale - z.ty;
      double dx = fmax(fabs(x - cx) - z.half_xy, 0.0), dy = fmax(fabs(y - cy) - z.half_xy, 0.0);
      if (dx * dx + dy * dy < reach * reach) return true;
    }
  return false;
}


// The same predicate for a robot whose reach is smaller than a maze cell, straight-line: only the 3 x 3 cells around the cell under
// (x, y) can be that close (point_dev_from_model: reach < scale), and a cell's distance is (dx[dj]^2 + dy[di]^2)^(1/2) with three
// values each — no loops over a lane-dependent cell range.  `reach` may be enlarged by the caller (a conservative pre-test).
MZP_HD bool point_near_wall3(const PointDev& P, double x, double y, double reach) {
  const MazeDev& z = P.maze;
  const double s = z.scale, inv = 1.0 / s;
  const int jc = (int)floor((x + z.tx) * inv + 0.5), ic = (int)floor((y + z.ty) * inv + 0.5);
  double dx2[3], dy2[3];
  unsigned bits[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const int j = jc + d - 1, i = ic + d - 1;
    const double ex = fmax(fabs(x - (j * s - z.tx)) - z.half_xy, 0.0), ey = fmax(fabs(y - (i * s - z.ty)) - z.half_xy, 0.0);
    dx2[d] = ex * ex; dy2[d] = ey * ey;
    const unsigned row = (i >= 0 && i < z.rows && i < MZ_MAX_GRID) ? z.rowmask[i] : 0u;  // (bits of columns >= cols are never set)
    const int sh = jc + 1;  // column c sits at bit c + 2 of row << 2, so the columns jc - 1, jc, jc + 1 are the bits sh, sh + 1, sh + 2
    bits[d] = (sh >= 0 && sh < 16) ? (((row << 2) >> sh) & 7u) : 0u;
  }
  const double r2 = reach * reach;
  bool near = false;
#pragma unroll
  for (int di = 0; di < 3; di++)
#pragma unroll
    for (int dj = 0; dj < 3; dj++) near = near || (((bits[di] >> dj) & 1u) && dx2[dj] + dy2[di] < r2);
  return near;
}

MZP_HD double pt_impedance(const double* si, double x) {
  double d0 = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
  if (d0 == dmax || width <= 1e-15) return 0.5 * (d0 + dmax);
  double xn = x / width;
  if (xn >= 1.0) return dmax;
  if (xn <= 0.0) return d0;
  double y;
  if (power <= 1.0 + 1e-12) y = xn;
  else if (power == 2.0) y = xn <= mid ? xn * xn / mid : 1.0 - (1.0 - xn) * (1.0 - xn) / (1.0 - mid);  // MuJoCo's default power: no pow()
  else if (xn <= mid) y = pow(xn, power) / pow(mid, power - 1.0);
  else y = 1.0 - pow(1.0 - xn, power) / pow(1.0 - mid, power - 1.0);
  return d0 + y * (dmax - d0);
}
