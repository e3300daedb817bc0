// planar_dyn.h — the Point robot's MazeEnv.step with NB movable XY blocks, as lane-group SPMD code (fp64).
//
// Same programming model as ant_dyn.h: G lanes advance one environment, its working set (PlanarScratch<NB, NS>) lives
// in LDS for the whole step, a phase is an MZ_FOR over independent items, cx.sync() is the hand-off.  The
// single-lane host context (tests/emu) runs the same source on the CPU.
//
// Replaces, per env: PointEnv.step (mujoco_maze/point.py:44-61), the MuJoCo step behind it on the planar
// system  q = (x, y, theta | bx_0, by_0 | ...)  — robot: slide-slide-hinge body with COM offset (point.xml:21-25);
// blocks: slide-x + slide-y boxes (maze_env.py:563-660) — and the manual wall bounce of MazeEnv.step
// (maze_env.py:451-464).  Dynamics per forward evaluation (4 per env.step: RK4, frame_skip 1):
//   * unconstrained acceleration in closed form (robot: centripetal term of the COM offset; blocks: none);
//   * collision by enumerators spread over the lanes (two passes: count, fill — no atomics):
//       sphere / arrow vs the 3 x 3 wall cells around the robot, sphere / arrow vs each block,
//       each block vs the 3 x 3 wall cells around it, block vs block;
//     every geom margin of the Point model is 0, so a contact exists only under penetration (dist < 0);
//     the floor touches sphere and blocks at dist = 0 exactly (z is not a degree of freedom): never a contact;
//   * contacts as 3 x NV Jacobians [n; mu t1; mu t2] (pyramid edges u0 +- u1, u0 +- u2), dense NV x NV Newton
//     with exact line search, Cholesky by one lane — same cost model and stopping rule as the one-lane code
//     this replaces.
#pragma once
#include "ant_dyn.h"    // MZ_FOR, HostCtx, maze_row
#include "point_dyn.h"  // PointDev, point_detect, pt_impedance

template <int NB, int NS>
struct PlanarDims {
  static_assert(NB == 0 || NS == 0, "no registered maze mixes movable blocks and object balls");
  static_assert(NS <= 1, "one object ball");
  static constexpr int NV = 3 + 2 * NB + 3 * NS;  // robot x, y, theta | block x, y ... | ball x, y, spin
  static constexpr int NC = NB == 0 ? 12 + 4 * NS : (NB == 1 ? 40 : (NB == 2 ? 64 : 96));  // contact slots
  // enumerators: 9 sphere-wall, 9 arrow-wall | per block: sphere-block, arrow-block, 9 block-wall, 9 block-platform (elevated
  //              mazes), block-floor, 2 joint-limit rows (limited slides) | block pairs | per ball: 9 ball-wall, robot
  //              sphere-ball, ball-arrow
  static constexpr int EPB = 23;  // enumerators per block
  static constexpr int NE = 18 + EPB * NB + NB * (NB - 1) / 2 + 11 * NS;
  static constexpr int NOBS = 7 + 3 * NB + 3 * NS;
};

template <int NB, int NS>
struct alignas(16) PlanarScratch {
  using D = PlanarDims<NB, NS>;
  double q[D::NV], v[D::NV];  // state of the current RK4 stage
  double x0[D::NV], v0[D::NV], accv[D::NV], accf[D::NV];
  double qas[D::NV], qacc[D::NV], grad[D::NV], search[D::NV], Mx[D::NV], Ms[D::NV];
  double wd[D::NV];           // constraint-induced acceleration (qacc - qacc_smooth) of the previous RK4 stage: warm start
  double M3[3][3];            // robot block of the mass matrix (blocks: block_mass on the diagonal)
  double H[D::NV][D::NV];
  double cJ[D::NC][3][D::NV], caref[D::NC][3], cD[D::NC], cu[D::NC][3], cg[D::NC][3], cW[D::NC][5], cjv[D::NC][3];
  double co, si;              // cos / sin of the heading of the current stage
  int ncon, cnt[D::NE], cbeg[D::NE], status, robot_near;
};

// ---- small helpers
template <int NB>
MZP_HD double pl_mass(const PointDev& P, const double M3[3][3], int i, int j) {
  if (i < 3 && j < 3) return M3[i][j];
  if (i != j) return 0.0;
  if (i < 3 + 2 * NB) return P.block_mass;
  return (i - 3 - 2 * NB) % 3 == 2 ? P.ball_izz : P.ball_mass;
}
template <int NB, int NS>
MZP_HD void pl_block_center(const PointDev& P, const PlanarScratch<NB, NS>& s, int b, double* c) {
  c[0] = P.block_pos0[b][0]; c[1] = P.block_pos0[b][1]; c[2] = P.block_pos0[b][2];
  const double q0 = s.q[3 + 2 * b], q1 = s.q[4 + 2 * b];  // the block's two slides, along block_axis[0] < block_axis[1]
  if (P.block_axis[0] == 0) c[0] += q0; else c[1] += q0;
  if (P.block_axis[1] == 1) c[1] += q1; else c[2] += q1;
}

// One contact candidate: dist, position, normal (geom1 -> geom2), bodies (-1 world, 0 robot, 1 + k block k)
// cls 6: joint-limit row of a block slide — a single frictionless row (n = +- the slide axis, pos unused)
struct PlContact { double dist, pos[3], n[3]; int b1, b2, cls; };

// sphere (centre c relative to the box centre) vs axis-aligned box; normal from the sphere to the box
MZP_HD bool pl_sphere_box(const double* c, double r, const double* hb, double margin, double* dist, double* nrm) {
  double cl[3], dd;
  bool inside = true;
  for (int k = 0; k < 3; k++) { cl[k] = fmin(fmax(c[k], -hb[k]), hb[k]); if (cl[k] != c[k]) inside = false; }
  if (!inside) {
    double w[3] = {cl[0] - c[0], cl[1] - c[1], cl[2] - c[2]};
    dd = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (dd - r > margin) return false;
    for (int k = 0; k < 3; k++) nrm[k] = w[k] / dd;
    dd -= r;
  } else {
    int kb = 0; double best = 1e30;
    for (int k = 0; k < 3; k++) { double e = hb[k] - fabs(c[k]); if (e < best) { best = e; kb = k; } }
    nrm[0] = nrm[1] = nrm[2] = 0.0;
    double sg = (kb == 0 ? c[0] : (kb == 1 ? c[1] : c[2])) >= 0.0 ? -1.0 : 1.0;
    if (kb == 0) nrm[0] = sg; else if (kb == 1) nrm[1] = sg; else nrm[2] = sg;
    dd = -best - r;
  }
  *dist = dd;
  return true;
}

// z-rotated box (the arrow: centre bc, axes ex / ey, half sizes ahx / ahy, height az) vs an axis-aligned box (centre
// wc, half sizes wh) [ASSUME-13]; `flip`: report the normal from the arrow to the box (movable block = geom2)
template <class Emit>
MZP_HD void pl_arrow_box(const PointDev& P, const double* bc, double co, double si, const double* wc, const double* wh, double margin,
                         bool flip, int b1, int b2, int cls, Emit&& emit) {
  if (fabs(P.arr_z - wc[2]) > P.arr_hz + wh[2] + margin) return;
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
  if (bestsep > margin) return;
  double nx = best == 0 ? 1.0 : (best == 1 ? 0.0 : (best == 2 ? ex[0] : ey[0])), ny = best == 0 ? 0.0 : (best == 1 ? 1.0 : (best == 2 ? ex[1] : ey[1]));
  double n[3] = {nx * bestsign, ny * bestsign, 0.0};  // box -> arrow
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
      PlContact c;
      c.dist = dep[k];
      c.pos[0] = vx[k] + sg * n[0] * dep[k]; c.pos[1] = vy[k] + sg * n[1] * dep[k]; c.pos[2] = P.arr_z;
      c.n[0] = flip ? -n[0] : n[0]; c.n[1] = flip ? -n[1] : n[1]; c.n[2] = 0.0;
      c.b1 = b1; c.b2 = b2; c.cls = cls;
      emit(c);
    }
}

// axis-aligned box (geom1: centre c1, half h1) vs axis-aligned box (geom2: centre c2, half h2) [ASSUME-12]
template <class Emit>
MZP_HD void pl_box_box(const double* c1, const double* h1, const double* c2, const double* h2, double margin, int b1, int b2, int cls,
                       Emit&& emit) {
  double gap[3];
  int ax = 0;
  for (int k = 0; k < 3; k++) gap[k] = fabs(c2[k] - c1[k]) - (h1[k] + h2[k]);
  if (gap[1] > gap[ax]) ax = 1;
  if (gap[2] > gap[ax]) ax = 2;
  double gmax = ax == 0 ? gap[0] : (ax == 1 ? gap[1] : gap[2]);
  if (gmax > margin) return;
  double lo[3], hi[3];
  for (int k = 0; k < 3; k++) { lo[k] = fmax(c1[k] - h1[k], c2[k] - h2[k]); hi[k] = fmin(c1[k] + h1[k], c2[k] + h2[k]); }
  int u = (ax + 1) % 3, w = (ax + 2) % 3;
  if (!(hi[u] - lo[u] > 1e-6) || !(hi[w] - lo[w] > 1e-6)) return;  // edge / corner touch: no face contact
  double sg = c2[ax] >= c1[ax] ? 1.0 : -1.0;
  for (int iu = 0; iu < 2; iu++)
    for (int iw = 0; iw < 2; iw++) {
      PlContact c;
      c.dist = gmax;
      for (int k = 0; k < 3; k++) c.n[k] = 0.0;
      c.n[ax] = sg;
      c.pos[ax] = c1[ax] + sg * (h1[ax] + 0.5 * gmax);
      c.pos[u] = iu ? hi[u] : lo[u];
      c.pos[w] = iw ? hi[w] : lo[w];
      c.b1 = b1; c.b2 = b2; c.cls = cls;
      emit(c);
    }
}

// wall cell (di, dj) of the 3 x 3 neighbourhood of the cell under (x, y): centre in wc; false when not a BLOCK cell
MZP_HD bool pl_wall_cell(const MazeDev& z, double x, double y, int k9, double* wc) {
  int jc = (int)floor((x + z.tx) / z.scale + 0.5), ic = (int)floor((y + z.ty) / z.scale + 0.5);
  int i = ic + k9 / 3 - 1, j = jc + k9 % 3 - 1;
  if (i < 0 || j < 0 || i >= z.rows || j >= z.cols) return false;
  if (!((maze_row(z, i) >> j) & 1u)) return false;
  wc[0] = j * (double)z.scale - z.tx; wc[1] = i * (double)z.scale - z.ty; wc[2] = z.center_z;
  return true;
}

// platform cell of an elevated maze in the 3 x 3 neighbourhood of the cell under (x, y): every cell that is not a chasm
MZP_HD bool pl_platform_cell(const MazeDev& z, double x, double y, int k9, double* wc) {
  int jc = (int)floor((x + z.tx) / z.scale + 0.5), ic = (int)floor((y + z.ty) / z.scale + 0.5);
  int i = ic + k9 / 3 - 1, j = jc + k9 % 3 - 1;
  if (!z.elevated || i < 0 || j < 0 || i >= z.rows || j >= z.cols) return false;
  uint32_t row = 0u;
#pragma unroll
  for (int r = 0; r < MZ_MAX_GRID; r++) row = (r == i) ? z.platmask[r] : row;
  if (!((row >> j) & 1u)) return false;
  wc[0] = j * (double)z.scale - z.tx; wc[1] = i * (double)z.scale - z.ty; wc[2] = z.half_z;
  return true;
}

// Contacts of enumerator e, in a fixed order (identical in the count and the fill pass)
template <int NB, int NS, class Emit>
MZP_HD void planar_contacts(const PointDev& P, const PlanarScratch<NB, NS>& s, int e, Emit&& emit) {
  const MazeDev& z = P.maze;
  double wh[3] = {z.half_xy, z.half_xy, z.half_z};
  double arrow[2] = {s.q[0] + P.arr_off * s.co, s.q[1] + P.arr_off * s.si};
  if (e < 18) {  // ---- robot vs wall cells
    if (!s.robot_near) return;
    double wc[3];
    if (!pl_wall_cell(z, s.q[0], s.q[1], e % 9, wc)) return;
    const PtPair& pr = P.pair[0];
    if (e < 9) {  // sphere (geom1) vs wall (geom2)
      double c[3] = {s.q[0] - wc[0], s.q[1] - wc[1], P.sph_z - wc[2]}, dd, nrm[3];
      if (!pl_sphere_box(c, P.sph_r, wh, pr.margin, &dd, nrm)) return;
      PlContact ct;
      ct.dist = dd;
      ct.pos[0] = s.q[0] + nrm[0] * (P.sph_r + 0.5 * dd); ct.pos[1] = s.q[1] + nrm[1] * (P.sph_r + 0.5 * dd); ct.pos[2] = P.sph_z + nrm[2] * (P.sph_r + 0.5 * dd);
      for (int k = 0; k < 3; k++) ct.n[k] = nrm[k];
      ct.b1 = 0; ct.b2 = -1; ct.cls = 0;
      emit(ct);
    } else {      // wall (geom1) vs arrow (geom2)
      pl_arrow_box(P, arrow, s.co, s.si, wc, wh, pr.margin, false, -1, 0, 0, emit);
    }
    return;
  }
  if constexpr (NS > 0) {  // ---- object ball (body 1): 9 wall cells, the robot's sphere, the arrow
    int k = e - 18;
    double bc[3] = {P.ball_pos0[0] + s.q[3], P.ball_pos0[1] + s.q[4], P.ball_r};  // sphere centre
    if (k < 9) {  // ball (sphere, geom1) vs wall (geom2)
      double wc[3];
      if (!pl_wall_cell(z, bc[0], bc[1], k, wc)) return;
      double c[3] = {bc[0] - wc[0], bc[1] - wc[1], bc[2] - wc[2]}, dd, nrm[3];
      if (!pl_sphere_box(c, P.ball_r, wh, P.pair[4].margin, &dd, nrm)) return;
      PlContact ct;
      ct.dist = dd;
      for (int q = 0; q < 3; q++) { ct.n[q] = nrm[q]; ct.pos[q] = bc[q] + nrm[q] * (P.ball_r + 0.5 * dd); }
      ct.b1 = 1; ct.b2 = -1; ct.cls = 4;
      emit(ct);
    } else if (k == 9) {  // robot sphere (geom1) vs ball (geom2)
      double dv[3] = {bc[0] - s.q[0], bc[1] - s.q[1], bc[2] - P.sph_z};
      double cd = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]), dd = cd - P.sph_r - P.ball_r;
      if (dd > P.pair[5].margin) return;
      PlContact ct;
      ct.dist = dd;
      if (cd < 1e-15) { ct.n[0] = 1.0; ct.n[1] = 0.0; ct.n[2] = 0.0; }
      else { ct.n[0] = dv[0] / cd; ct.n[1] = dv[1] / cd; ct.n[2] = dv[2] / cd; }
      ct.pos[0] = s.q[0] + ct.n[0] * (P.sph_r + 0.5 * dd); ct.pos[1] = s.q[1] + ct.n[1] * (P.sph_r + 0.5 * dd); ct.pos[2] = P.sph_z + ct.n[2] * (P.sph_r + 0.5 * dd);
      ct.b1 = 0; ct.b2 = 1; ct.cls = 5;
      emit(ct);
    } else {  // ball (sphere, geom1) vs arrow (box rotated about z, geom2): sphere-box in the arrow's frame
      double rel[3] = {bc[0] - arrow[0], bc[1] - arrow[1], bc[2] - P.arr_z};
      double c[3] = {s.co * rel[0] + s.si * rel[1], -s.si * rel[0] + s.co * rel[1], rel[2]}, ah[3] = {P.arr_hx, P.arr_hy, P.arr_hz}, dd, nl[3];
      if (!pl_sphere_box(c, P.ball_r, ah, P.pair[5].margin, &dd, nl)) return;
      PlContact ct;
      ct.dist = dd;
      ct.n[0] = s.co * nl[0] - s.si * nl[1]; ct.n[1] = s.si * nl[0] + s.co * nl[1]; ct.n[2] = nl[2];
      for (int q = 0; q < 3; q++) ct.pos[q] = bc[q] + ct.n[q] * (P.ball_r + 0.5 * dd);
      ct.b1 = 1; ct.b2 = 0; ct.cls = 5;
      emit(ct);
    }
    return;
  }
  if constexpr (NB > 0) {
    int r = e - 18;
    constexpr int EPB = PlanarDims<NB, NS>::EPB;
    if (r < EPB * NB) {
      int b = r / EPB, k = r - EPB * b;
      double bc[3];
      pl_block_center<NB, NS>(P, s, b, bc);
      if (k == 0) {         // sphere (geom1) vs block (geom2)
        const PtPair& pr = P.pair[1];
        double c[3] = {s.q[0] - bc[0], s.q[1] - bc[1], P.sph_z - bc[2]}, dd, nrm[3];
        if (!pl_sphere_box(c, P.sph_r, P.block_half, pr.margin, &dd, nrm)) return;
        PlContact ct;
        ct.dist = dd;
        ct.pos[0] = s.q[0] + nrm[0] * (P.sph_r + 0.5 * dd); ct.pos[1] = s.q[1] + nrm[1] * (P.sph_r + 0.5 * dd); ct.pos[2] = P.sph_z + nrm[2] * (P.sph_r + 0.5 * dd);
        for (int q = 0; q < 3; q++) ct.n[q] = nrm[q];
        ct.b1 = 0; ct.b2 = 1 + b; ct.cls = 1;
        emit(ct);
      } else if (k == 1) {  // arrow (geom1) vs block (geom2)
        pl_arrow_box(P, arrow, s.co, s.si, bc, P.block_half, P.pair[1].margin, true, 0, 1 + b, 1, emit);
      } else if (k < 11) {  // wall (geom1) vs block (geom2)
        double wc[3];
        if (!pl_wall_cell(z, bc[0], bc[1], k - 2, wc)) return;
        pl_box_box(wc, wh, bc, P.block_half, P.pair[2].margin, -1, 1 + b, 2, emit);
      } else if (k < 20) {  // platform of an elevated maze (geom1) vs block (geom2): same box rule, same pair class as a wall
        double wc[3];
        if (!pl_platform_cell(z, bc[0], bc[1], k - 11, wc)) return;
        pl_box_box(wc, wh, bc, P.block_half, P.pair[2].margin, -1, 1 + b, 2, emit);
      } else if (k == 20) {  // floor plane z = 0 (geom1) vs block (geom2): the corners below the plane, normal +z
        if (P.block_axis[1] != 2) return;  // a block without a z slide rests on the floor at dist = 0 exactly: never a contact
        for (int ci = 0; ci < 4; ci++) {
          PlContact c;
          c.dist = bc[2] - P.block_half[2];
          c.n[0] = 0.0; c.n[1] = 0.0; c.n[2] = 1.0;
          c.pos[0] = bc[0] + ((ci & 1) ? P.block_half[0] : -P.block_half[0]); c.pos[1] = bc[1] + ((ci & 2) ? P.block_half[1] : -P.block_half[1]);
          c.pos[2] = 0.5 * c.dist;
          c.b1 = -1; c.b2 = 1 + b; c.cls = 7;
          emit(c);
        }
      } else {              // joint-limit row of slide k - 21 (maze_env.py:607-648: limited slides of falling blocks)
        if (!P.block_limited) return;
        const int a = k - 21;
        const double q = s.q[3 + 2 * b + a];
        for (int side = -1; side <= 1; side += 2) {
          PlContact c;
          c.dist = side < 0 ? q - P.block_lo[a] : P.block_hi[a] - q;
          c.n[0] = c.n[1] = c.n[2] = 0.0;
          c.n[P.block_axis[a]] = -(double)side;  // the row's Jacobian: d dist / d q
          c.pos[0] = c.pos[1] = c.pos[2] = 0.0;
          c.b1 = -1; c.b2 = 1 + b; c.cls = 6;
          emit(c);
        }
      }
      return;
    }
    if constexpr (NB > 1) {  // block pairs (a < b): geom1 = block a, geom2 = block b
      int p = r - EPB * NB, a = 0, b = 1;
      if (p == 1) { a = 0; b = 2; } else if (p == 2) { a = 1; b = 2; }
      if (b < NB) {
        double ca[3], cb[3];
        pl_block_center<NB, NS>(P, s, a, ca);
        pl_block_center<NB, NS>(P, s, b, cb);
        pl_box_box(ca, P.block_half, cb, P.block_half, P.pair[3].margin, 1 + a, 1 + b, 3, emit);
      }
    }
  }
}

// Jacobian row of body `body` for a unit force direction f at the world point p
template <int NB, int NS>
MZP_HD void pl_add_body_row(const PointDev& P, const PlanarScratch<NB, NS>& s, int body, const double* f, const double* p, double sgn, double* J) {
  if (body == 0) {
    double rx = p[0] - s.q[0], ry = p[1] - s.q[1];
    J[0] += sgn * f[0]; J[1] += sgn * f[1]; J[2] += sgn * (-f[0] * ry + f[1] * rx);
  } else if (body > 0) {
    if constexpr (NS > 0) {  // the ball: slide x, slide y, hinge z through the body origin
      double rx = p[0] - (P.ball_pos0[0] + s.q[3]), ry = p[1] - (P.ball_pos0[1] + s.q[4]);
      J[3] += sgn * f[0]; J[4] += sgn * f[1]; J[5] += sgn * (-f[0] * ry + f[1] * rx);
    }
    for (int b = 0; b < NB; b++)
      if (b == body - 1) { J[3 + 2 * b] += sgn * (P.block_axis[0] == 0 ? f[0] : f[1]); J[4 + 2 * b] += sgn * (P.block_axis[1] == 1 ? f[1] : f[2]); }
  }
}

template <int NB, int NS>
MZP_HD void planar_fill_contact(const PointDev& P, PlanarScratch<NB, NS>& s, int slot, const PlContact& c) {
  constexpr int NV = PlanarDims<NB, NS>::NV;
  const PtPair& pr = P.pair[c.cls];
  const double* n = c.n;
  if (c.cls == 6) {
    // joint-limit row: ONE frictionless row  r = J a - aref, cost D/2 min(0, r)^2.  It rides the contact machinery as a
    // pyramid whose tangential Jacobians vanish: the four edge rows then coincide (u0 +- 0), so cD = D / 4 reproduces
    // exactly cost, gradient and curvature of the single row.
    double imp = pt_impedance(pr.solimp, fabs(c.dist - pr.margin));
    double R = fmax(1e-15, (1.0 - imp) / imp * pr.wsum);
    s.cD[slot] = 0.25 / R;
    for (int a = 0; a < 3; a++) {
      double J[NV];
      for (int i = 0; i < NV; i++) J[i] = 0.0;
      if (a == 0) pl_add_body_row<NB, NS>(P, s, c.b2, n, c.pos, 1.0, J);
      double vel = 0.0;
      for (int i = 0; i < NV; i++) { s.cJ[slot][a][i] = J[i]; vel += J[i] * s.v[i]; }
      s.caref[slot][a] = a == 0 ? -pr.B * vel - pr.K * imp * (c.dist - pr.margin) : 0.0;
    }
    return;
  }
  double y[3] = {0.0, (n[1] < 0.5 && n[1] > -0.5) ? 1.0 : 0.0, 0.0};
  y[2] = 1.0 - y[1];
  double dt = n[0] * y[0] + n[1] * y[1] + n[2] * y[2];
  for (int k = 0; k < 3; k++) y[k] -= n[k] * dt;
  double nn = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
  double t1[3] = {y[0] / nn, y[1] / nn, y[2] / nn};
  double t2[3] = {n[1] * t1[2] - n[2] * t1[1], n[2] * t1[0] - n[0] * t1[2], n[0] * t1[1] - n[1] * t1[0]};
  double imp = pt_impedance(pr.solimp, fabs(c.dist - pr.margin));
  double R = fmax(1e-15, (1.0 - imp) / imp * (1.0 + pr.mu * pr.mu) * pr.wsum);
  s.cD[slot] = 1.0 / (2.0 * pr.mu * pr.mu * R);
  for (int a = 0; a < 3; a++) {
    const double* f = a == 0 ? n : (a == 1 ? t1 : t2);
    double sc = a == 0 ? 1.0 : pr.mu;
    double J[NV];
    for (int i = 0; i < NV; i++) J[i] = 0.0;
    pl_add_body_row<NB, NS>(P, s, c.b2, f, c.pos, sc, J);
    pl_add_body_row<NB, NS>(P, s, c.b1, f, c.pos, -sc, J);
    double vel = 0.0;
    for (int i = 0; i < NV; i++) { s.cJ[slot][a][i] = J[i]; vel += J[i] * s.v[i]; }
    s.caref[slot][a] = -pr.B * vel - (a == 0 ? pr.K * imp * (c.dist - pr.margin) : 0.0);
  }
}

// pyramidal contact: gradient block and curvature of the four edge rows (same as pt_contact_eval)
MZP_HD void pl_contact_eval(double D, const double* u, double* g, double* W) {
  double r0 = u[0] + u[1], r1 = u[0] - u[1], r2 = u[0] + u[2], r3 = u[0] - u[2];
  double a0 = r0 < 0 ? 1.0 : 0.0, a1 = r1 < 0 ? 1.0 : 0.0, a2 = r2 < 0 ? 1.0 : 0.0, a3 = r3 < 0 ? 1.0 : 0.0;
  g[0] = D * (a0 * r0 + a1 * r1 + a2 * r2 + a3 * r3); g[1] = D * (a0 * r0 - a1 * r1); g[2] = D * (a2 * r2 - a3 * r3);
  W[0] = D * (a0 + a1 + a2 + a3); W[1] = D * (a0 - a1); W[2] = D * (a2 - a3); W[3] = D * (a0 + a1); W[4] = D * (a2 + a3);
}

// ------------------------------------------------------------------ one forward-dynamics evaluation: s.q, s.v -> s.qacc
template <int NB, int NS, class C>
MZP_HD void planar_forward(const C& cx, const PointDev& P, PlanarScratch<NB, NS>& s, bool warm) {
  using D = PlanarDims<NB, NS>;
  constexpr int NV = D::NV, NC = D::NC, NE = D::NE;
  // Stages 2-4 of RK4 start the Newton solve from the previous stage's solution shifted by the change of qacc_smooth
  // (the optimum is unique: only the iteration count depends on the start).
  MZ_FOR(i, NV) s.wd[i] = warm ? s.qacc[i] - s.qas[i] : 0.0;
  cx.sync();
  MZ_FOR(one, 1) {
    double co = cos(s.q[2]), si = sin(s.q[2]), w2 = s.v[2] * s.v[2], mc = P.mass * P.com_x;
    s.co = co; s.si = si;
    s.qas[0] = P.com_x * w2 * co; s.qas[1] = P.com_x * w2 * si; s.qas[2] = 0.0;
    for (int i = 3; i < NV; i++) s.qas[i] = 0.0;
    if (P.block_axis[1] == 2) { for (int b = 0; b < NB; b++) s.qas[4 + 2 * b] = P.gz; }  // falling blocks: gravity on the z slide
    s.M3[0][0] = P.mass; s.M3[0][1] = 0.0; s.M3[0][2] = -mc * si;
    s.M3[1][0] = 0.0; s.M3[1][1] = P.mass; s.M3[1][2] = mc * co;
    s.M3[2][0] = -mc * si; s.M3[2][1] = mc * co; s.M3[2][2] = P.izz;
    s.robot_near = point_near_wall(P, s.q[0], s.q[1]) ? 1 : 0;
  }
  cx.sync();
  MZ_FOR(i, NV) s.qacc[i] = s.qas[i];
  bool maybe = NB + NS > 0 || s.robot_near != 0;  // group-uniform
  if (!cx.any(maybe)) { cx.sync(); return; }
  // ---- collision: count, prefix, fill
  MZ_FOR(e, NE) {
    int n = 0;
    if (maybe) planar_contacts<NB, NS>(P, s, e, [&](const PlContact& c) { if (c.dist < P.pair[c.cls].margin) n++; });
    s.cnt[e] = n;
  }
  cx.sync();
  MZ_FOR(one, 1) {
    int tot = 0;
    for (int e = 0; e < NE; e++) { s.cbeg[e] = tot; tot += s.cnt[e]; }
    if (tot > NC) { tot = NC; s.status |= MZ_STATUS_CONTACT_OVERFLOW; }
    s.ncon = tot;
  }
  cx.sync();
  const int ncon = s.ncon;
  if (!cx.any(ncon > 0)) return;
  MZ_FOR(e, NE) {
    if (s.cnt[e] > 0) {
      int slot = s.cbeg[e];
      planar_contacts<NB, NS>(P, s, e, [&](const PlContact& c) {
        if (c.dist < P.pair[c.cls].margin) { if (slot < NC) planar_fill_contact<NB, NS>(P, s, slot, c); slot++; }
      });
    }
  }
  if (ncon > 0) { MZ_FOR(i, NV) s.qacc[i] = s.qas[i] + s.wd[i]; }  // envs without contacts keep qacc = qacc_smooth
  cx.sync();
  if constexpr (NB == 0 && NS == 0) {
    // The bare Point: 3 dofs.  Everything per dof (qacc, M qacc - qfrc, gradient, the 3 x 3 Hessian, its Cholesky, the search
    // direction) is carried redundantly in the registers of every lane; the contacts are spread over the lanes and enter
    // through group sums (DPP) — the same Newton iteration, unit-step vote and exact line search as below, without the LDS
    // hand-offs that a per-dof distribution of a 3-dof problem consists of.
    double a[3] = {s.qacc[0], s.qacc[1], s.qacc[2]};
    const double qs[3] = {s.qas[0], s.qas[1], s.qas[2]};
    double M[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i][j] = s.M3[i][j];
    bool done = ncon == 0;
    int it = 0;
    while (cx.any(!done) && it < 50) {
      double Mx[3], pg[3] = {0.0, 0.0, 0.0}, pH[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // H entries 00 01 02 11 12 22
      for (int i = 0; i < 3; i++) { double t = 0.0; for (int j = 0; j < 3; j++) t += M[i][j] * (a[j] - qs[j]); Mx[i] = t; }
      MZ_FOR(c, ncon) {
        double J[3][3], u[3], g3[3], W[5];
        for (int r = 0; r < 3; r++) {
          double t = -s.caref[c][r];
          for (int i = 0; i < 3; i++) { J[r][i] = s.cJ[c][r][i]; t += J[r][i] * a[i]; }
          u[r] = t; s.cu[c][r] = t;
        }
        pl_contact_eval(s.cD[c], u, g3, W);
        int e = 0;
        for (int i = 0; i < 3; i++) {
          pg[i] += J[0][i] * g3[0] + J[1][i] * g3[1] + J[2][i] * g3[2];
          for (int j = i; j < 3; j++, e++)
            pH[e] += W[0] * J[0][i] * J[0][j] + W[1] * (J[0][i] * J[1][j] + J[1][i] * J[0][j]) + W[2] * (J[0][i] * J[2][j] + J[2][i] * J[0][j]) +
                     W[3] * J[1][i] * J[1][j] + W[4] * J[2][i] * J[2][j];
        }
      }
      double g[3], H[3][3];
      for (int i = 0; i < 3; i++) g[i] = Mx[i] + cx.gsum(pg[i]);
      {
        int e = 0;
        for (int i = 0; i < 3; i++) for (int j = i; j < 3; j++, e++) { const double h = M[i][j] + cx.gsum(pH[e]); H[i][j] = h; H[j][i] = h; }
      }
      const double gn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
      if (!done && P.inv_scale * gn < 1e-10) done = true;
      if (!cx.any(!done)) break;
      if (it == 49 && !done) { MZ_FOR(one, 1) s.status |= MZ_STATUS_SOLVER_MAXITER; }
      double L[3][3], y[3], sr[3];
      for (int j = 0; j < 3; j++) {
        double d = H[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
        d = sqrt(fmax(d, 1e-300));
        L[j][j] = d;
        for (int i = j + 1; i < 3; i++) {
          double t = H[i][j];
          for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
          L[i][j] = t / d;
        }
      }
      for (int i = 0; i < 3; i++) { double t = -g[i]; for (int k = 0; k < i; k++) t -= L[i][k] * y[k]; y[i] = t / L[i][i]; }
      for (int i = 2; i >= 0; i--) { double t = y[i]; for (int k = i + 1; k < 3; k++) t -= L[k][i] * y[k]; y[i] = t / L[i][i]; }
      for (int i = 0; i < 3; i++) sr[i] = y[i];
      double p1 = 0.0, p2 = 0.0;
      for (int i = 0; i < 3; i++) {
        double t = 0.0;
        for (int j = 0; j < 3; j++) t += M[i][j] * sr[j];
        p1 += sr[i] * Mx[i]; p2 += sr[i] * t;
      }
      bool changed = false;
      MZ_FOR(c, ncon) {
        double v[3];
        for (int r = 0; r < 3; r++) { v[r] = s.cJ[c][r][0] * sr[0] + s.cJ[c][r][1] * sr[1] + s.cJ[c][r][2] * sr[2]; s.cjv[c][r] = v[r]; }
        const double u0 = s.cu[c][0], u1 = s.cu[c][1], u2 = s.cu[c][2], w0 = u0 + v[0], w1 = u1 + v[1], w2 = u2 + v[2];
        changed = changed || ((u0 + u1 < 0) != (w0 + w1 < 0)) || ((u0 - u1 < 0) != (w0 - w1 < 0)) || ((u0 + u2 < 0) != (w0 + w2 < 0)) ||
                  ((u0 - u2 < 0) != (w0 - w2 < 0));
      }
      changed = cx.gany(changed);
      double lo = 0.0, hi = -1.0, alpha = 1.0, prev_d2 = -1.0;
      for (int ls = 0; ls < 30 && changed; ls++) {
        double d1 = 0.0, d2 = 0.0;
        MZ_FOR(c, ncon) {
          const double Dc = s.cD[c], v0 = s.cjv[c][0], v1 = s.cjv[c][1], v2 = s.cjv[c][2];
          const double u0 = s.cu[c][0] + alpha * v0, u1 = s.cu[c][1] + alpha * v1, u2 = s.cu[c][2] + alpha * v2;
          double r, w;
          r = u0 + u1; w = v0 + v1; if (r < 0) { d1 += Dc * r * w; d2 += Dc * w * w; }
          r = u0 - u1; w = v0 - v1; if (r < 0) { d1 += Dc * r * w; d2 += Dc * w * w; }
          r = u0 + u2; w = v0 + v2; if (r < 0) { d1 += Dc * r * w; d2 += Dc * w * w; }
          r = u0 - u2; w = v0 - v2; if (r < 0) { d1 += Dc * r * w; d2 += Dc * w * w; }
        }
        d1 = cx.gsum(d1) + p1 + alpha * p2;
        d2 = cx.gsum(d2) + p2;
        if (d2 == prev_d2) break;
        prev_d2 = d2;
        if (d1 < 0) lo = alpha; else hi = alpha;
        double next = alpha - d1 / d2;
        if (hi >= 0 && !(next > lo && next < hi)) next = 0.5 * (lo + hi);
        if (!(next > 0)) next = hi >= 0 ? 0.5 * (lo + hi) : 0.0;
        if (fabs(next - alpha) <= 1e-15 * fabs(next)) { alpha = next; break; }
        alpha = next;
      }
      if (!done) { for (int i = 0; i < 3; i++) a[i] += alpha * sr[i]; }
      if (!changed) done = true;
      it++;
    }
    cx.sync();
    if (ncon > 0) { MZ_FOR(i, NV) s.qacc[i] = i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }
    cx.sync();
    return;
  }
  // ---- Newton on the primal problem (dense), exact line search
  bool done = ncon == 0;
  int it = 0;
  while (cx.any(!done) && it < 50) {
    MZ_FOR(i, NV) {
      double t = 0.0;
      for (int j = 0; j < NV; j++) t += pl_mass<NB>(P, s.M3, i, j) * (s.qacc[j] - s.qas[j]);
      s.Mx[i] = t;
    }
    MZ_FOR(c, ncon) {
      double u[3];
      for (int a = 0; a < 3; a++) {
        double t = -s.caref[c][a];
        for (int i = 0; i < NV; i++) t += s.cJ[c][a][i] * s.qacc[i];
        u[a] = t; s.cu[c][a] = t;
      }
      pl_contact_eval(s.cD[c], u, s.cg[c], s.cW[c]);
    }
    cx.sync();
    double gpart = 0.0;
    MZ_FOR(i, NV) {
      double g = s.Mx[i];
      for (int c = 0; c < ncon; c++) g += s.cJ[c][0][i] * s.cg[c][0] + s.cJ[c][1][i] * s.cg[c][1] + s.cJ[c][2][i] * s.cg[c][2];
      s.grad[i] = g;
      gpart += g * g;
    }
    double gn = sqrt(cx.gsum(gpart));
    if (!done && P.inv_scale * gn < 1e-10) done = true;
    if (!cx.any(!done)) break;
    if (it == 49 && !done) { MZ_FOR(one, 1) s.status |= MZ_STATUS_SOLVER_MAXITER; }
    MZ_FOR(e, NV * NV) {
      int i = e / NV, j = e - NV * i;
      double acc = pl_mass<NB>(P, s.M3, i, j);
      for (int c = 0; c < ncon; c++) {
        double ni = s.cJ[c][0][i], pi = s.cJ[c][1][i], qi = s.cJ[c][2][i], nj = s.cJ[c][0][j], pj = s.cJ[c][1][j], qj = s.cJ[c][2][j];
        const double* W = s.cW[c];
        acc += W[0] * ni * nj + W[1] * (ni * pj + pi * nj) + W[2] * (ni * qj + qi * nj) + W[3] * pi * pj + W[4] * qi * qj;
      }
      s.H[i][j] = acc;
    }
    cx.sync();
    MZ_FOR(one, 1) {  // Cholesky H = L L^T and H search = -grad, one lane
      double L[NV][NV], y[NV];
      for (int j = 0; j < NV; j++) {
        double d = s.H[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
        d = sqrt(fmax(d, 1e-300));
        L[j][j] = d;
        for (int i = j + 1; i < NV; i++) {
          double t = s.H[i][j];
          for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
          L[i][j] = t / d;
        }
      }
      for (int i = 0; i < NV; i++) { double t = -s.grad[i]; for (int k = 0; k < i; k++) t -= L[i][k] * y[k]; y[i] = t / L[i][i]; }
      for (int i = NV - 1; i >= 0; i--) { double t = y[i]; for (int k = i + 1; k < NV; k++) t -= L[k][i] * y[k]; y[i] = t / L[i][i]; }
      for (int i = 0; i < NV; i++) s.search[i] = y[i];
    }
    cx.sync();
    double p1p = 0.0, p2p = 0.0;
    MZ_FOR(i, NV) {
      double t = 0.0;
      for (int j = 0; j < NV; j++) t += pl_mass<NB>(P, s.M3, i, j) * s.search[j];
      p1p += s.search[i] * s.Mx[i]; p2p += s.search[i] * t;
    }
    MZ_FOR(c, ncon) {
      for (int a = 0; a < 3; a++) {
        double t = 0.0;
        for (int i = 0; i < NV; i++) t += s.cJ[c][a][i] * s.search[i];
        s.cjv[c][a] = t;
      }
    }
    cx.sync();
    // The Newton direction makes alpha = 1 the exact minimiser whenever the active set at qacc + search equals the one H
    // was built on (the cost is quadratic there): one vote, and then neither the line search nor a verification pass.
    bool changed = false;
    MZ_FOR(c, ncon) {
      double u0 = s.cu[c][0], u1 = s.cu[c][1], u2 = s.cu[c][2];
      double w0 = u0 + s.cjv[c][0], w1 = u1 + s.cjv[c][1], w2 = u2 + s.cjv[c][2];
      changed = changed || ((u0 + u1 < 0) != (w0 + w1 < 0)) || ((u0 - u1 < 0) != (w0 - w1 < 0)) || ((u0 + u2 < 0) != (w0 + w2 < 0)) ||
                ((u0 - u2 < 0) != (w0 - w2 < 0));
    }
    changed = cx.gany(changed);
    double lo = 0.0, hi = -1.0, alpha = 1.0, prev_d2 = -1.0;
    double p1 = 0.0, p2 = 0.0;
    if (changed) { p1 = cx.gsum(p1p); p2 = cx.gsum(p2p); }
    for (int ls = 0; ls < 30 && changed; ls++) {
      double d1 = 0.0, d2 = 0.0;
      MZ_FOR(c, ncon) {
        double Dc = s.cD[c], v0 = s.cjv[c][0], v1 = s.cjv[c][1], v2 = s.cjv[c][2];
        double u0 = s.cu[c][0] + alpha * v0, u1 = s.cu[c][1] + alpha * v1, u2 = s.cu[c][2] + alpha * v2, r, w;
        r = u0 + u1; w = v0 + v1; if (r < 0) { d1 += Dc * r * w; d2 += Dc * w * w; }
        r = u0 - u1; w = v0 - v1; if (r < 0) { d1 += Dc * r * w; d2 += Dc * w * w; }
        r = u0 + u2; w = v0 + v2; if (r < 0) { d1 += Dc * r * w; d2 += Dc * w * w; }
        r = u0 - u2; w = v0 - v2; if (r < 0) { d1 += Dc * r * w; d2 += Dc * w * w; }
      }
      d1 = cx.gsum(d1) + p1 + alpha * p2;
      d2 = cx.gsum(d2) + p2;
      if (d2 == prev_d2) break;
      prev_d2 = d2;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double next = alpha - d1 / d2;
      if (hi >= 0 && !(next > lo && next < hi)) next = 0.5 * (lo + hi);
      if (!(next > 0)) next = hi >= 0 ? 0.5 * (lo + hi) : 0.0;
      if (fabs(next - alpha) <= 1e-15 * fabs(next)) { alpha = next; break; }
      alpha = next;
    }
    if (done) alpha = 0.0;
    cx.sync();
    MZ_FOR(i, NV) s.qacc[i] += alpha * s.search[i];
    cx.sync();
    if (!changed) done = true;
    it++;
  }
  cx.sync();
}

// ------------------------------------------------------------------ One MazeEnv.step.  s.q / s.v hold the state in and out.
template <int NB, int NS, class C>
MZP_HD void planar_env_step(const C& cx, const PointDev& P, PlanarScratch<NB, NS>& s, const double* action) {
  constexpr int NV = PlanarDims<NB, NS>::NV;
  const double PI = 3.141592653589793;
  double old_x = s.q[0], old_y = s.q[1];  // every lane reads the same values
  cx.sync();
  MZ_FOR(one, 1) {  // point.py:45-56
    double th = s.q[2] + action[1];
    if (th < -PI) th += PI * 2;
    else if (PI < th) th -= PI * 2;
    s.q[2] = th;
    s.q[0] += cos(th) * action[0];
    s.q[1] += sin(th) * action[0];
    s.status = 0;
  }
  MZ_FOR(i, NV) s.v[i] = fmin(fmax(s.v[i], -P.vel_limit), P.vel_limit);  // the clip covers the whole qvel (point.py:54-55)
  cx.sync();
  for (int f = 0; f < P.frame_skip; f++) {  // mj_step, RK4 (point.xml:3)
    const double h = P.h;
    MZ_FOR(i, NV) { s.x0[i] = s.q[i]; s.v0[i] = s.v[i]; s.accv[i] = 0.0; s.accf[i] = 0.0; }
    cx.sync();
    for (int st = 0; st < 4; st++) {
      planar_forward<NB, NS>(cx, P, s, st > 0);
      double bw = (st == 0 || st == 3) ? 1.0 / 6 : 1.0 / 3, aw = st == 2 ? 1.0 : 0.5;
      MZ_FOR(i, NV) {
        s.accv[i] += bw * s.v[i]; s.accf[i] += bw * s.qacc[i];
        double nq = s.x0[i] + h * (aw * s.v[i]), nv = s.v0[i] + h * (aw * s.qacc[i]);
        s.q[i] = nq; s.v[i] = nv;
      }
      cx.sync();
    }
    MZ_FOR(i, NV) { s.q[i] = s.x0[i] + h * s.accv[i]; s.v[i] = s.v0[i] + h * s.accf[i]; }
    cx.sync();
  }
  // maze_env.py:454-464: manual wall bounce on the robot's xy
  if (P.nseg > 0) {
    MZ_FOR(one, 1) {
      double old_xy[2] = {old_x, old_y}, new_xy[2] = {s.q[0], s.q[1]}, fin[2];
      int r = point_bounce(P, old_xy, new_xy, fin, nullptr);
      if (r < 0) s.status |= MZ_STATUS_COLLINEAR;
      s.q[0] = fin[0]; s.q[1] = fin[1];
    }
    cx.sync();
  }
}

// coordinate c of movable block b's body origin (get_body_com, maze_env.py:364-368): spawn position + its two slides
template <int NB, int NS>
MZP_HD float planar_block_coord(const PointDev& P, const PlanarScratch<NB, NS>& s, int b, int c) {
  double v = P.block_pos0[b][c];
  if (c == P.block_axis[0]) v += s.q[3 + 2 * b];
  if (c == P.block_axis[1]) v += s.q[4 + 2 * b];
  return (float)v;
}

// observation element i of the returned row: qpos[:3] | ball xyz | block xyz ... | qvel[:3] | t * 0.001  (maze_env.py:351-369)
template <int NB, int NS>
MZP_HD float planar_obs_elem(const PointDev& P, const PlanarScratch<NB, NS>& s, int i, int t) {
  int nb3 = (P.observe_blocks ? 3 * NB : 0) + (P.observe_balls ? 3 * NS : 0);
  if (i < 3) return (float)s.q[i];
  if constexpr (NS > 0) {
    if (i < 3 + nb3) return (float)(i == 5 ? P.ball_pos0[2] : P.ball_pos0[i - 3] + s.q[i]);  // body frame origin: z = 0
  }
  if (i < 3 + nb3) return planar_block_coord<NB, NS>(P, s, (i - 3) / 3, (i - 3) % 3);
  if (i < 6 + nb3) return (float)s.v[i - 3 - nb3];
  return (float)t * 0.001f;
}
