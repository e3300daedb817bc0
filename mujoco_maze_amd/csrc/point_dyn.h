// point_dyn.h — the Point robot's MazeEnv.step, one environment per lane.
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
// exactly as mj_step does.  The arithmetic is fp64 per lane: the path is a few hundred
// flops per env, HBM traffic is the bound, and fp64 keeps the wall-hit / give-up
// decisions identical to the float64 reference logic; the state is stored as fp32.
//
// NOT modelled yet (flagged per env with MZ_STATUS_UNMODELED_CONTACT): the MuJoCo
// sphere-box / box-box contacts that fire when the point's 0.5 sphere or its arrow box
// touch a maze wall.
#pragma once
#include <math.h>
#include <stdint.h>

#include "ant_model.h"

#define MZ_STATUS_UNMODELED_CONTACT 16
#define MZ_STATUS_COLLINEAR 8

struct PointDev {
  double h, com_x, vel_limit, restitution;
  int frame_skip, nseg;
  double seg[MZ_MAX_SEG][4];
  double reach;  // arrow tip distance from the torso origin (contact-regime flag)
  MazeDev maze;
  TaskDev task;
  double qpos0[3];
  int reset_kind;
};

static inline int point_dev_from_model(PointDev* p, const mz_model* m, char* err, int errlen) {
  memset(p, 0, sizeof(*p));
  if (m->robot != MZ_ROBOT_POINT || m->nv != 3 || m->nq != 3 || m->jnt_type[0] != MZ_JNT_SLIDE || m->jnt_type[1] != MZ_JNT_SLIDE ||
      m->jnt_type[2] != MZ_JNT_HINGE || fabs(m->body_ipos[1][1]) > 1e-12)
    return ant_fail(err, errlen, "point kernel: model is not the slide-slide-hinge point robot");
  p->h = m->timestep; p->com_x = m->body_ipos[1][0]; p->vel_limit = m->velocity_limit; p->restitution = m->restitution;
  p->frame_skip = m->frame_skip; p->nseg = m->manual_collision ? m->nseg : 0;
  for (int k = 0; k < m->nseg; k++) for (int j = 0; j < 4; j++) p->seg[k][j] = m->seg[k][j];
  p->reach = 0.0;
  for (int g = 1; g < m->ngeom; g++) {
    double r = hypot(m->geom_pos[g][0], m->geom_pos[g][1]) + m->geom_rbound[g];
    if (r > p->reach) p->reach = r;
  }
  maze_dev_from_model(&p->maze, m);
  task_dev_from_model(&p->task, m);
  for (int k = 0; k < 3; k++) p->qpos0[k] = m->qpos0[k];
  p->reset_kind = m->reset_qvel_kind;
  return MZ_OK;
}

#if defined(__HIPCC__)
#define MZP_HD __host__ __device__ __forceinline__
#else
#define MZP_HD inline
#endif

MZP_HD double cross2d(double ax, double ay, double bx, double by) { return ax * by + (-ay) * bx; }

// CollisionDetector.detect: 1 hit, 0 none, -1 collinear (the reference raises ZeroDivisionError)
MZP_HD int point_detect(const PointDev& P, const double* o, const double* n, double* pt, double* rf) {
  double mvx = n[0] - o[0], mvy = n[1] - o[1];
  if (hypot(mvx, mvy) <= 1e-8) return 0;
  int found = 0, degenerate = 0;
  double best = 0.0;
  for (int k = 0; k < P.nseg; k++) {
    const double* s = P.seg[k];
    double wx = s[2] - s[0], wy = s[3] - s[1];
    double c1 = cross2d(wx, wy, o[0] - s[0], o[1] - s[1]), c2 = cross2d(wx, wy, n[0] - s[0], n[1] - s[1]);
    if (!(c1 * c2 <= 0.0)) continue;
    double c3 = cross2d(mvx, mvy, s[0] - o[0], s[1] - o[1]), c4 = cross2d(mvx, mvy, s[2] - o[0], s[3] - o[1]);
    if (!(c3 * c4 <= 0.0)) continue;
    double a = cross2d(wx, wy, mvx, mvy), b = cross2d(wx, wy, s[2] - o[0], s[3] - o[1]);
    if (a == 0.0) { degenerate = 1; continue; }
    double r = b / a, px = o[0] + r * mvx, py = o[1] + r * mvy;
    double dist = hypot(px - o[0], py - o[1]);
    if (!found || dist < best) {
      found = 1; best = dist;
      pt[0] = px; pt[1] = py;
      double bx = -wx, by = -wy, n2 = hypot(bx, by);
      n2 = n2 * n2;
      double dx = n[0] - s[0], dy = n[1] - s[1];
      double sc = (dx * bx - (-dy) * by) / n2;
      double qx = s[0] + bx * sc, qy = s[1] + by * sc;
      rf[0] = n[0] + 2.0 * (qx - n[0]);
      rf[1] = n[1] + 2.0 * (qy - n[1]);
    }
  }
  if (degenerate && !found) return -1;
  return found;
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

// One MazeEnv.step.  q, v: state in/out (fp64 working copy).  Returns status bits.
MZP_HD int point_env_step(const PointDev& P, double* q, double* v, const double* action, int t_in, double* obs7, double* reward,
                          uint8_t* done, int* goal_idx, double* info4, int* t_out) {
  const double PI = 3.141592653589793;
  int status = 0;
  double old_xy[2] = {q[0], q[1]};
  // point.py:45-56
  double th = q[2] + action[1];
  if (th < -PI) th += PI * 2;
  else if (PI < th) th -= PI * 2;
  q[2] = th;
  q[0] += cos(th) * action[0];
  q[1] += sin(th) * action[0];
  for (int k = 0; k < 3; k++) v[k] = fmin(fmax(v[k], -P.vel_limit), P.vel_limit);
  if (point_near_wall(P, q[0], q[1])) status |= MZ_STATUS_UNMODELED_CONTACT;
  // mj_step x frame_skip, RK4 (point.xml:3)
  for (int f = 0; f < P.frame_skip; f++) {
    const double h = P.h;
    double x0[3] = {q[0], q[1], q[2]}, v0[3] = {v[0], v[1], v[2]}, accv[3] = {0, 0, 0}, accf[3] = {0, 0, 0}, qs[3], vs[3], a[3];
    for (int k = 0; k < 3; k++) { qs[k] = x0[k]; vs[k] = v0[k]; }
    for (int st = 0; st < 4; st++) {
      point_qacc(P, qs, vs, a);
      double bw = (st == 0 || st == 3) ? 1.0 / 6 : 1.0 / 3, aw = st == 2 ? 1.0 : 0.5;
      for (int k = 0; k < 3; k++) {
        accv[k] += bw * vs[k]; accf[k] += bw * a[k];
        double nq = x0[k] + h * (aw * vs[k]), nv = v0[k] + h * (aw * a[k]);
        qs[k] = nq; vs[k] = nv;
      }
    }
    for (int k = 0; k < 3; k++) { q[k] = x0[k] + h * accv[k]; v[k] = v0[k] + h * accf[k]; }
  }
  // maze_env.py:454-464
  if (P.nseg > 0) {
    double new_xy[2] = {q[0], q[1]}, pt[2], rf[2];
    int hit = point_detect(P, old_xy, new_xy, pt, rf);
    if (hit < 0) status |= MZ_STATUS_COLLINEAR;
    if (hit > 0) {
      double pos[2] = {pt[0] + P.restitution * (rf[0] - pt[0]), pt[1] + P.restitution * (rf[1] - pt[1])}, p2[2], r2[2];
      int again = point_detect(P, old_xy, pos, p2, r2);
      if (again < 0) status |= MZ_STATUS_COLLINEAR;
      if (again > 0) { q[0] = old_xy[0]; q[1] = old_xy[1]; }
      else if (again == 0) { q[0] = pos[0]; q[1] = pos[1]; }
    }
  }
  int t = t_in + 1;
  *t_out = t;
  return status;
  (void)obs7; (void)reward; (void)done; (void)goal_idx; (void)info4;
}
