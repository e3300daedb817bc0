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
// Near a maze wall MuJoCo's own contacts also fire (the 0.5 sphere and the arrow box are
// collidable while the manual detector uses radius 0.4): `point_forward` then adds sphere-box and
// (z-rotated) box-box contacts as pyramidal soft constraints and solves the 3-dof Newton problem per lane,
// exactly like the Ant path but dense (3 x 3).  Away from walls it reduces to the closed form above.
#pragma once
#include <math.h>
#include <stdint.h>

#include "ant_model.h"

#define MZ_STATUS_UNMODELED_CONTACT 16 /* kept for ABI stability; no longer raised */
#define PT_NC 8
#define MZ_STATUS_COLLINEAR 8

struct PointDev {
  double h, com_x, vel_limit, restitution;
  int frame_skip, nseg;
  double seg[MZ_MAX_SEG][4];
  double reach;  // arrow tip distance from the torso origin (wall broad phase)
  // MuJoCo contact regime: body, geoms, mixed pair parameters (robot geom x wall)
  double mass, izz, bw_tran, inv_scale;
  double sph_r, sph_z, arr_off, arr_hx, arr_hy, arr_hz, arr_z;
  double margin, mu, K, B, solimp[5];
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
  p->mass = m->body_mass[1];
  p->izz = m->body_inertia[1][2] + m->body_mass[1] * m->body_ipos[1][0] * m->body_ipos[1][0];
  p->bw_tran = m->body_invweight0[1][0];
  p->inv_scale = 1.0 / (m->meaninertia * m->nv);
  if (m->ngeom != 3 || m->geom_type[1] != MZ_GEOM_SPHERE || m->geom_type[2] != MZ_GEOM_BOX)
    return ant_fail(err, errlen, "point kernel: expected floor + sphere + arrow box geoms");
  p->sph_r = m->geom_size[1][0]; p->sph_z = m->geom_pos[1][2];
  p->arr_off = m->geom_pos[2][0]; p->arr_hx = m->geom_size[2][0]; p->arr_hy = m->geom_size[2][1]; p->arr_hz = m->geom_size[2][2];
  p->arr_z = m->geom_pos[2][2];
  {
    double sr[2], si[5];
    for (int k = 0; k < 2; k++) sr[k] = 0.5 * (m->geom_solref[1][k] + m->wall_solref[k]);
    for (int k = 0; k < 5; k++) si[k] = 0.5 * (m->geom_solimp[1][k] + m->wall_solimp[k]);
    p->margin = fmax(m->geom_margin[1], m->wall_margin);
    p->mu = fmax(m->geom_friction[1][0], m->wall_friction[0]);
    double tc = fmax(sr[0], 2.0 * m->timestep), dmax = si[1];
    p->K = 1.0 / (dmax * dmax * tc * tc * sr[1] * sr[1]);
    p->B = 2.0 / (dmax * tc);
    for (int k = 0; k < 5; k++) p->solimp[k] = si[k];
  }
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


// ------------------------------------------------------------------ MuJoCo contact regime of the Point (fp64, one env per lane)
struct PtContact { double J[3][3], aref[3], D; };  // rows: normal, mu*t1, mu*t2; columns: x, y, theta

MZP_HD double pt_impedance(const double* si, double x) {
  double d0 = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
  if (d0 == dmax || width <= 1e-15) return 0.5 * (d0 + dmax);
  double xn = x / width;
  if (xn >= 1.0) return dmax;
  if (xn <= 0.0) return d0;
  double y;
  if (power <= 1.0 + 1e-12) y = xn;
  else if (xn <= mid) y = pow(xn, power) / pow(mid, power - 1.0);
  else y = 1.0 - pow(1.0 - xn, power) / pow(1.0 - mid, power - 1.0);
  return d0 + y * (dmax - d0);
}

// one contact: dist, position p (world xyz), normal n (geom1 -> geom2), sgn = +1 when the robot geom is geom2
MZP_HD void pt_add_contact(const PointDev& P, const double* q, const double* v, PtContact* con, int* ncon, double dist,
                           const double* p, const double* n, double sgn) {
  if (!(dist < P.margin) || *ncon >= PT_NC) return;
  PtContact& c = con[(*ncon)++];
  double y[3] = {0.0, (n[1] < 0.5 && n[1] > -0.5) ? 1.0 : 0.0, 0.0};
  y[2] = 1.0 - y[1];
  double dt = n[0] * y[0] + n[1] * y[1] + n[2] * y[2];
  for (int k = 0; k < 3; k++) y[k] -= n[k] * dt;
  double nn = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
  double t1[3] = {y[0] / nn, y[1] / nn, y[2] / nn};
  double t2[3] = {n[1] * t1[2] - n[2] * t1[1], n[2] * t1[0] - n[0] * t1[2], n[0] * t1[1] - n[1] * t1[0]};
  double rx = p[0] - q[0], ry = p[1] - q[1];
  double imp = pt_impedance(P.solimp, fabs(dist - P.margin));
  double R = fmax(1e-15, (1.0 - imp) / imp * (P.bw_tran + P.mu * P.mu * P.bw_tran));
  c.D = 1.0 / (2.0 * P.mu * P.mu * R);
  for (int a = 0; a < 3; a++) {
    const double* f = a == 0 ? n : (a == 1 ? t1 : t2);
    double sc = sgn * (a == 0 ? 1.0 : P.mu);
    c.J[a][0] = sc * f[0]; c.J[a][1] = sc * f[1]; c.J[a][2] = sc * (-f[0] * ry + f[1] * rx);
    double vel = c.J[a][0] * v[0] + c.J[a][1] * v[1] + c.J[a][2] * v[2];
    c.aref[a] = -P.B * vel - (a == 0 ? P.K * imp * (dist - P.margin) : 0.0);
  }
}

// sphere-box and z-rotated box-box against the wall cells around the robot; returns the number of contacts
MZP_HD int point_collide(const PointDev& P, const double* q, const double* v, PtContact* con) {
  const MazeDev& z = P.maze;
  int ncon = 0;
  double inv = 1.0 / z.scale, reach = P.reach + P.margin;
  int j0 = (int)floor((q[0] - reach + z.tx) * inv + 0.5), j1 = (int)floor((q[0] + reach + z.tx) * inv + 0.5);
  int i0 = (int)floor((q[1] - reach + z.ty) * inv + 0.5), i1 = (int)floor((q[1] + reach + z.ty) * inv + 0.5);
  double co = cos(q[2]), si = sin(q[2]);
  double wh[3] = {z.half_xy, z.half_xy, z.half_z};
  // MuJoCo pair order: by geom type, so the sphere's contacts (sphere < box) come before the arrow's
  for (int pass = 0; pass < 2; pass++)
    for (int i = i0; i <= i1; i++)
      for (int j = j0; j <= j1; j++) {
        if (i < 0 || j < 0 || i >= z.rows || j >= z.cols) continue;
        if (!((z.rowmask[i] >> j) & 1u)) continue;
        double wc[3] = {j * (double)z.scale - z.tx, i * (double)z.scale - z.ty, z.center_z};
        if (pass == 0) {  // sphere (geom1) vs wall (geom2): normal sphere -> wall, J = -J_robot
          double c[3] = {q[0] - wc[0], q[1] - wc[1], P.sph_z - wc[2]}, cl[3], nrm[3], dd;
          bool inside = true;
          for (int k = 0; k < 3; k++) { cl[k] = fmin(fmax(c[k], -wh[k]), wh[k]); if (cl[k] != c[k]) inside = false; }
          if (!inside) {
            double w[3] = {cl[0] - c[0], cl[1] - c[1], cl[2] - c[2]};
            dd = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            if (dd - P.sph_r > P.margin) continue;
            for (int k = 0; k < 3; k++) nrm[k] = w[k] / dd;
            dd -= P.sph_r;
          } else {
            int kb = 0; double best = 1e30;
            for (int k = 0; k < 3; k++) { double e = wh[k] - fabs(c[k]); if (e < best) { best = e; kb = k; } }
            nrm[0] = nrm[1] = nrm[2] = 0.0;
            nrm[kb] = c[kb] >= 0.0 ? -1.0 : 1.0;
            dd = -best - P.sph_r;
          }
          double pos[3] = {q[0] + nrm[0] * (P.sph_r + 0.5 * dd), q[1] + nrm[1] * (P.sph_r + 0.5 * dd), P.sph_z + nrm[2] * (P.sph_r + 0.5 * dd)};
          pt_add_contact(P, q, v, con, &ncon, dd, pos, nrm, -1.0);
        } else {  // wall (geom1) vs arrow box (geom2) [ASSUME-13]: normal wall -> arrow, J = +J_robot
          double bc[2] = {q[0] + P.arr_off * co, q[1] + P.arr_off * si};
          if (fabs(P.arr_z - wc[2]) > P.arr_hz + wh[2] + P.margin) continue;
          double ex[2] = {co, si}, ey[2] = {-si, co}, dx = bc[0] - wc[0], dy = bc[1] - wc[1];
          int best = -1; double bestsep = -1e30, bestsign = 1.0;
          for (int a = 0; a < 4; a++) {
            double nx = a == 0 ? 1.0 : (a == 1 ? 0.0 : (a == 2 ? ex[0] : ey[0])), ny = a == 0 ? 0.0 : (a == 1 ? 1.0 : (a == 2 ? ex[1] : ey[1]));
            double proj = dx * nx + dy * ny;
            double ra = wh[0] * fabs(nx) + wh[1] * fabs(ny);
            double rb = P.arr_hx * fabs(ex[0] * nx + ex[1] * ny) + P.arr_hy * fabs(ey[0] * nx + ey[1] * ny);
            double sep = fabs(proj) - (ra + rb);
            if (sep > bestsep) { bestsep = sep; best = a; bestsign = proj >= 0.0 ? 1.0 : -1.0; }
          }
          if (bestsep > P.margin) continue;
          double nx = best == 0 ? 1.0 : (best == 1 ? 0.0 : (best == 2 ? ex[0] : ey[0])), ny = best == 0 ? 0.0 : (best == 1 ? 1.0 : (best == 2 ? ex[1] : ey[1]));
          double n[3] = {nx * bestsign, ny * bestsign, 0.0};
          double vx[4], vy[4], dep[4], dmin = 1e30;
          for (int k = 0; k < 4; k++) {
            double sx = (k & 1) ? 1.0 : -1.0, sy = (k & 2) ? 1.0 : -1.0;
            if (best < 2) {
              vx[k] = bc[0] + sx * P.arr_hx * ex[0] + sy * P.arr_hy * ey[0];
              vy[k] = bc[1] + sx * P.arr_hx * ex[1] + sy * P.arr_hy * ey[1];
              dep[k] = (vx[k] - wc[0]) * n[0] + (vy[k] - wc[1]) * n[1] - (wh[0] * fabs(n[0]) + wh[1] * fabs(n[1]));
            } else {
              vx[k] = wc[0] + sx * wh[0];
              vy[k] = wc[1] + sy * wh[1];
              dep[k] = (bc[0] - vx[k]) * n[0] + (bc[1] - vy[k]) * n[1] - (best == 2 ? P.arr_hx : P.arr_hy);
            }
            if (dep[k] < dmin) dmin = dep[k];
          }
          for (int k = 0; k < 4; k++)
            if (dep[k] <= dmin + 1e-9) {
              double sg = best < 2 ? -0.5 : 0.5;
              double pos[3] = {vx[k] + sg * n[0] * dep[k], vy[k] + sg * n[1] * dep[k], P.arr_z};
              pt_add_contact(P, q, v, con, &ncon, dep[k], pos, n, 1.0);
            }
        }
      }
  return ncon;
}

MZP_HD double pt_contact_eval(double D, const double* u, double* g, double* W) {
  double r0 = u[0] + u[1], r1 = u[0] - u[1], r2 = u[0] + u[2], r3 = u[0] - u[2];
  double a0 = r0 < 0 ? 1.0 : 0.0, a1 = r1 < 0 ? 1.0 : 0.0, a2 = r2 < 0 ? 1.0 : 0.0, a3 = r3 < 0 ? 1.0 : 0.0;
  if (g) { g[0] = D * (a0 * r0 + a1 * r1 + a2 * r2 + a3 * r3); g[1] = D * (a0 * r0 - a1 * r1); g[2] = D * (a2 * r2 - a3 * r3); }
  if (W) { W[0] = D * (a0 + a1 + a2 + a3); W[1] = D * (a0 - a1); W[2] = D * (a2 - a3); W[3] = D * (a0 + a1); W[4] = D * (a2 + a3); }
  return 0.5 * D * (a0 * r0 * r0 + a1 * r1 * r1 + a2 * r2 * r2 + a3 * r3 * r3);
}

// forward dynamics: qacc from (q, v); returns status bits
MZP_HD int point_forward(const PointDev& P, const double* q, const double* v, double* qacc) {
  double qas[3];
  point_qacc(P, q, v, qas);
  qacc[0] = qas[0]; qacc[1] = qas[1]; qacc[2] = qas[2];
  if (!point_near_wall(P, q[0], q[1])) return 0;
  PtContact con[PT_NC];
  int ncon = point_collide(P, q, v, con);
  if (ncon == 0) return 0;
  double co = cos(q[2]), si = sin(q[2]), mc = P.mass * P.com_x;
  double M[3][3] = {{P.mass, 0.0, -mc * si}, {0.0, P.mass, mc * co}, {-mc * si, mc * co, P.izz}};
  int status = 0;
  for (int it = 0; it < 50; it++) {
    double dq[3] = {qacc[0] - qas[0], qacc[1] - qas[1], qacc[2] - qas[2]}, grad[3], H[3][3], Mx[3];
    for (int i = 0; i < 3; i++) {
      Mx[i] = M[i][0] * dq[0] + M[i][1] * dq[1] + M[i][2] * dq[2];
      grad[i] = Mx[i];
      for (int j = 0; j < 3; j++) H[i][j] = M[i][j];
    }
    double cu[PT_NC][3];
    for (int c = 0; c < ncon; c++) {
      double u[3], g[3], W[5];
      for (int a = 0; a < 3; a++) u[a] = con[c].J[a][0] * qacc[0] + con[c].J[a][1] * qacc[1] + con[c].J[a][2] * qacc[2] - con[c].aref[a];
      for (int a = 0; a < 3; a++) cu[c][a] = u[a];
      pt_contact_eval(con[c].D, u, g, W);
      for (int i = 0; i < 3; i++) {
        double ni = con[c].J[0][i], pi = con[c].J[1][i], qi = con[c].J[2][i];
        grad[i] += ni * g[0] + pi * g[1] + qi * g[2];
        for (int j = 0; j < 3; j++) {
          double nj = con[c].J[0][j], pj = con[c].J[1][j], qj = con[c].J[2][j];
          H[i][j] += W[0] * ni * nj + W[1] * (ni * pj + pi * nj) + W[2] * (ni * qj + qi * nj) + W[3] * pi * pj + W[4] * qi * qj;
        }
      }
    }
    double gn = sqrt(grad[0] * grad[0] + grad[1] * grad[1] + grad[2] * grad[2]);
    if (P.inv_scale * gn < 1e-10) break;
    if (it == 49) status |= MZ_STATUS_SOLVER_MAXITER;
    // 3x3 Cholesky solve H s = -grad
    double l00 = sqrt(H[0][0]), l10 = H[1][0] / l00, l20 = H[2][0] / l00;
    double l11 = sqrt(H[1][1] - l10 * l10), l21 = (H[2][1] - l20 * l10) / l11;
    double l22 = sqrt(H[2][2] - l20 * l20 - l21 * l21);
    double y0 = -grad[0] / l00, y1 = (-grad[1] - l10 * y0) / l11, y2 = (-grad[2] - l20 * y0 - l21 * y1) / l22;
    double s2 = y2 / l22, s1 = (y1 - l21 * s2) / l11, s0 = (y0 - l10 * s1 - l20 * s2) / l00;
    double sr[3] = {s0, s1, s2};
    // exact line search (safeguarded Newton on the piecewise-linear derivative)
    double p1 = sr[0] * Mx[0] + sr[1] * Mx[1] + sr[2] * Mx[2], p2 = 0.0;
    for (int i = 0; i < 3; i++) p2 += sr[i] * (M[i][0] * sr[0] + M[i][1] * sr[1] + M[i][2] * sr[2]);
    double cjv[PT_NC][3];
    for (int c = 0; c < ncon; c++)
      for (int a = 0; a < 3; a++) cjv[c][a] = con[c].J[a][0] * sr[0] + con[c].J[a][1] * sr[1] + con[c].J[a][2] * sr[2];
    double lo = 0.0, hi = -1.0, alpha = 1.0, prev_d2 = -1.0;
    for (int ls = 0; ls < 30; ls++) {
      double d1 = p1 + alpha * p2, d2 = p2;
      for (int c = 0; c < ncon; c++) {
        double D = con[c].D, v0 = cjv[c][0], v1 = cjv[c][1], v2 = cjv[c][2];
        double u0 = cu[c][0] + alpha * v0, u1 = cu[c][1] + alpha * v1, u2 = cu[c][2] + alpha * v2, r, w;
        r = u0 + u1; w = v0 + v1; if (r < 0) { d1 += D * r * w; d2 += D * w * w; }
        r = u0 - u1; w = v0 - v1; if (r < 0) { d1 += D * r * w; d2 += D * w * w; }
        r = u0 + u2; w = v0 + v2; if (r < 0) { d1 += D * r * w; d2 += D * w * w; }
        r = u0 - u2; w = v0 - v2; if (r < 0) { d1 += D * r * w; d2 += D * w * w; }
      }
      if (d2 == prev_d2) break;
      prev_d2 = d2;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double next = alpha - d1 / d2;
      if (hi >= 0 && !(next > lo && next < hi)) next = 0.5 * (lo + hi);
      if (!(next > 0)) next = hi >= 0 ? 0.5 * (lo + hi) : 0.0;
      if (fabs(next - alpha) <= 1e-15 * fabs(next)) { alpha = next; break; }
      alpha = next;
    }
    for (int i = 0; i < 3; i++) qacc[i] += alpha * sr[i];
  }
  return status;
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
  // mj_step x frame_skip, RK4 (point.xml:3)
  for (int f = 0; f < P.frame_skip; f++) {
    const double h = P.h;
    double x0[3] = {q[0], q[1], q[2]}, v0[3] = {v[0], v[1], v[2]}, accv[3] = {0, 0, 0}, accf[3] = {0, 0, 0}, qs[3], vs[3], a[3];
    for (int k = 0; k < 3; k++) { qs[k] = x0[k]; vs[k] = v0[k]; }
    for (int st = 0; st < 4; st++) {
      status |= point_forward(P, qs, vs, a);
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
