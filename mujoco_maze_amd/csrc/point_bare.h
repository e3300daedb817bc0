// point_bare.h — MazeEnv.step of the bare Point (no movable block, no object ball: PointUMaze, Point4Rooms, ...; BASELINE
// config 2) with the env's state, the contacts' constraint rows and the whole Newton iteration in REGISTERS.
//
// Replaces, per env: PointEnv.step (mujoco_maze/point.py:44-61), the MuJoCo step behind it (RK4, one mj_step) and the manual
// wall bounce of MazeEnv.step (maze_env.py:451-464).  Same physics, contact rules and stopping criteria as the general planar
// path (planar_dyn.h planar_forward, which the Point with blocks / a ball keeps) — restated for what bounds this kernel:
// a launch lasts as long as its slowest wave, and that wave holds ONE env pressed into a wall corner.  So
// everything here shortens the serial path of a single env:
//   * every lane of the group carries q, v, the RK4 accumulators, cos / sin of the heading, qacc and the 3 x 3 system redundantly
//     (a wavefront executes the same instruction for all lanes anyway): no one-lane phases, no LDS hand-offs around them;
//   * sincos by a three-term Cody-Waite reduction and fdlibm's kernels (the heading is wrapped once per step; libm's
//     Payne-Hanek path was a quarter of the median wave);
//   * contacts are staged once per evaluation in the order the lanes find them (slot order only changes the rounding of a sum)
//     and the two contacts a face contact makes at the bottom and the top of the overlap (same xy, same normal, same depth:
//     identical rows for a body that moves in the plane) enter as ONE row set of double weight — the same cost, term for term;
//   * the deep overlap (arrow so far inside a wall cell that the least-penetration axis is vertical — NOT rare: a head-on
//     touch is a tie with it) clips the arrow's rectangle against the cell's with ONE CANDIDATE PER LANE (mjc_BoxBox's 24
//     candidates, in 2-D: both faces are horizontal) instead of a 24-trip loop on the enumerator's lane;
//   * lane l of each 16-lane row owns contacts l and l + 16: its 3 x 3 Jacobian block, reference accelerations and weight
//     live in registers from the fill to the end of the solve; gradient and Hessian are nine 4-step DPP row sums (both rows of
//     a 32-lane group hold the same contacts and arrive at the same bits: no cross-row step, no readlane).
// The one-lane host context (tests/emu) runs this source on the CPU against the oracle.
#pragma once

#define MZ_PB_SLACK 0.25  // how far a stage state may drift from the step's start before the once-per-step broad phase is re-done

// sin and cos of x.  |x| < 1e5 (the heading: wrapped to [-pi, pi] once per step, point.py:47-51): k = nearest multiple of
// pi/2, r = x - k pi/2 by fdlibm's three-term Cody-Waite split (exact first product for |k| < 2^20), then fdlibm's
// __kernel_sin / __kernel_cos minimax polynomials on [-pi/4, pi/4]: below 1 ulp.  Anything else goes to libm.
MZP_HD void pt_sincos(double x, double* sn, double* cs) {
  if (!(fabs(x) < 1e5)) { *sn = sin(x); *cs = cos(x); return; }
  const double k = rint(x * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632673412561417e+00, x);
  r = fma(-k, 6.07710050630396597660e-11, r);
  r = fma(-k, 2.02226624879595063154e-21, r);
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                    z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 +
                    z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double s0 = r + r * z * ps, c0 = 1.0 - 0.5 * z + z * z * pc;
  const int n = (int)k & 3;
  const double ss = (n & 1) ? c0 : s0, cc = (n & 1) ? s0 : c0;
  *sn = (n & 2) ? -ss : ss;
  *cs = (n == 1 || n == 2) ? -cc : cc;
}

// One staged contact of the bare Point: depth, contact point (xy: a planar body's Jacobian does not see z), normal from geom1
// to geom2, and sw = +-weight — the sign says on which side the robot is (sphere: geom1, -; arrow: geom2, +), the magnitude
// how many coincident contacts the entry stands for.  64 bytes.
struct PlBareEntry { double dist, px, py, n[3], sw, pad; };
struct PlDeepJob { double wcx, wcy, dist, sg; };  // deep overlap of the arrow with the wall cell centred at (wcx, wcy)

// constraint rows of one contact: in the registers of the lane that owns it (first slot) or in LDS (later slots)
struct PbRow { double J[3][3], aref[3], D; };

template <>
struct alignas(16) PlanarScratch<0, 0> {
  static constexpr int CAP = 32;  // staged contacts per evaluation (4 sphere-cell + 4 arrow-cell x <= 8; more flags CONTACT_OVERFLOW)
  double q[3], v[3];              // state in and out (planar_step_body, the host emulation)
  double x0[3], v0[3], accv[3], accf[3], old_xy[2];  // RK4 bookkeeping, parked here across the forward evaluations (written by one lane, read by all)
  PlBareEntry stage[CAP];         // between the last evaluation and the next step: the wall detector's hand-off buffer
  PlDeepJob job[4];
  PbRow rowx[CAP];                // constraint rows of the lanes' LATER slots (contact index >= row width): rare, so not worth registers
  int nstage, njobs, status;
  unsigned jmask;
  double wds[3];                  // qacc - qacc_smooth of the step's last evaluation: the next step's first solve starts from it (MuJoCo's qacc_warmstart; round 6: +1.7 %)
#ifdef MZ_EXP_PROF
  unsigned long long prof[12], prof_t0;
#endif
  MZP_HD double* detect_buf() { return &stage[0].dist; }
  static constexpr int detect_buf_doubles = CAP * 8;
};

// BLOCK cell (di, dj) of the 3 x 3 neighbourhood of the cell under (x, y): its centre, or false
MZP_HD bool pb_wall_cell(const MazeDev& z, double x, double y, int k9, double* wcx, double* wcy) {
  const int jc = (int)floor((x + z.tx) / z.scale + 0.5), ic = (int)floor((y + z.ty) / z.scale + 0.5);
  const int i = ic + k9 / 3 - 1, j = jc + k9 % 3 - 1;
  if (i < 0 || j < 0 || i >= z.rows || j >= z.cols) return false;
  if (!((z.rowmask[i] >> j) & 1u)) return false;
  *wcx = j * (double)z.scale - z.tx; *wcy = i * (double)z.scale - z.ty;
  return true;
}

template <class C>
MZP_HD void pb_stage(PlanarScratch<0, 0>& s, double dist, double px, double py, double n0, double n1, double n2, double sw) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int idx = atomicAdd(&s.nstage, 1);
#else
  const int idx = s.nstage++;
#endif
  if (idx < PlanarScratch<0, 0>::CAP) {
    PlBareEntry& e = s.stage[idx];
    e.dist = dist; e.px = px; e.py = py; e.n[0] = n0; e.n[1] = n1; e.n[2] = n2; e.sw = sw;
  }
}

// The arrow's rectangle (centre P, half edges U, V) against the wall cell's |x1|, |x2| <= S, in the cell's frame: candidate c of
// mjc_BoxBox's enumeration as restated in pl_box_box (same order, same tolerances), for two horizontal faces — no third coordinate.
struct PbClip { double P1, P2, U1, U2, V1, V2, S, tol, det; };
MZP_HD bool pb_cand(const PbClip& k, int c, double& x1, double& x2) {
  bool valid = false;
  x1 = 0.0; x2 = 0.0;
  if (c < 4) {  // incident corners inside the reference rectangle
    const double su = (c & 1) ? 1.0 : -1.0, sv = (c & 2) ? 1.0 : -1.0;
    x1 = k.P1 + su * k.U1 + sv * k.V1; x2 = k.P2 + su * k.U2 + sv * k.V2;
    valid = fabs(x1) <= k.S + k.tol && fabs(x2) <= k.S + k.tol;
  } else if (c < 20) {  // incident edges against the four border lines
    const int j = c - 4, e = j >> 2, w = (j >> 1) & 1;
    const double side = (j & 1) ? 1.0 : -1.0, sgn = (e & 1) ? 1.0 : -1.0;
    const double p1 = e < 2 ? k.P1 + sgn * k.V1 : k.P1 + sgn * k.U1, p2 = e < 2 ? k.P2 + sgn * k.V2 : k.P2 + sgn * k.U2;
    const double q1 = e < 2 ? k.U1 : k.V1, q2 = e < 2 ? k.U2 : k.V2;
    const double pc = w ? p2 : p1, qc = w ? q2 : q1, po = w ? p1 : p2, qo = w ? q1 : q2;
    if (fabs(qc) >= 1e-15) {
      const double t = (side * k.S - pc) / qc;
      valid = t >= -1.0 && t <= 1.0 && fabs(po + t * qo) <= k.S + k.tol;
      x1 = p1 + t * q1; x2 = p2 + t * q2;
    }
  } else if (fabs(k.det) > 1e-15) {  // reference corners inside the incident rectangle
    const int j = c - 20;
    const double x = ((j & 1) ? k.S : -k.S) - k.P1, y = ((j & 2) ? k.S : -k.S) - k.P2;
    const double al = (x * k.V2 - y * k.V1) / k.det, be = (k.U1 * y - k.U2 * x) / k.det;
    valid = fabs(al) <= 1.0 + 1e-12 && fabs(be) <= 1.0 + 1e-12;
    x1 = k.P1 + al * k.U1 + be * k.V1; x2 = k.P2 + al * k.U2 + be * k.V2;
  }
  return valid;
}

MZP_HD void pb_zero(PbRow& r) {
  r.D = 0.0;
#pragma unroll
  for (int a = 0; a < 3; a++) { r.aref[a] = 0.0; r.J[a][0] = 0.0; r.J[a][1] = 0.0; r.J[a][2] = 0.0; }
}
MZP_HD void pb_fill(const PointDev& P, const PlBareEntry& e, const double* q, const double* v, PbRow& r) {
  const PtPair& pr = P.pair[0];
  const double n0 = e.n[0], n1 = e.n[1], n2 = e.n[2];
  // tangent frame as mju_makeFrame does (planar_fill_contact)
  double y0 = 0.0, y1 = (n1 < 0.5 && n1 > -0.5) ? 1.0 : 0.0, y2 = 1.0 - y1;
  const double dt = n1 * y1 + n2 * y2;
  y0 -= n0 * dt; y1 -= n1 * dt; y2 -= n2 * dt;
  const double inn = 1.0 / sqrt(y0 * y0 + y1 * y1 + y2 * y2);
  const double t10 = y0 * inn, t11 = y1 * inn, t12 = y2 * inn;
  const double t20 = n1 * t12 - n2 * t11, t21 = n2 * t10 - n0 * t12;
  const double imp = pt_impedance(pr.solimp, fabs(e.dist - pr.margin));
  const double R = fmax(1e-15, (1.0 - imp) / imp * (1.0 + pr.mu * pr.mu) * pr.wsum);
  r.D = fabs(e.sw) / (2.0 * pr.mu * pr.mu * R);
  const double sg = e.sw < 0.0 ? -1.0 : 1.0, rx = e.px - q[0], ry = e.py - q[1];
  const double f0[3] = {n0, t10, t20}, f1[3] = {n1, t11, t21};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double sc = sg * (a == 0 ? 1.0 : pr.mu);
    r.J[a][0] = sc * f0[a]; r.J[a][1] = sc * f1[a]; r.J[a][2] = sc * (-f0[a] * ry + f1[a] * rx);
    const double vel = r.J[a][0] * v[0] + r.J[a][1] * v[1] + r.J[a][2] * v[2];
    r.aref[a] = -pr.B * vel - (a == 0 ? pr.K * imp * (e.dist - pr.margin) : 0.0);
  }
}

// ------------------------------------------------------------------ one forward-dynamics evaluation.  q, v: the stage's state (every
// lane holds it); qacc / qas: in — the previous stage's (warm start), out — this stage's.
template <class C>
MZP_HD void point_forward_bare(const C& cx, const PointDev& P, PlanarScratch<0, 0>& s, const double* q, const double* v, double* qacc, double* qas,
                               bool warm, bool maybe) {
  using S = PlanarScratch<0, 0>;
  constexpr int G = C::nlanes, SW = G < 16 ? G : 16, K = S::CAP / SW, KC = (24 + G - 1) / G;
  const int lane = cx.lane0();
  const double wd[3] = {warm ? qacc[0] - qas[0] : 0.0, warm ? qacc[1] - qas[1] : 0.0, warm ? qacc[2] - qas[2] : 0.0};
  double co, si;
  pt_sincos(q[2], &si, &co);
  const double w2 = v[2] * v[2], mc = P.mass * P.com_x, mcs = mc * si, mcc = mc * co;
  qas[0] = P.com_x * w2 * co; qas[1] = P.com_x * w2 * si; qas[2] = 0.0;
  qacc[0] = qas[0]; qacc[1] = qas[1]; qacc[2] = qas[2];
  // broad phase: `maybe` (once per step, point_env_step_bare: a wall within reach + 1.5 MZ_PB_SLACK of the step's start) or a stage
  // state that has moved by more than the slack sends the env to the exact per-stage test; everybody else is done
  bool near = maybe || fabs(q[0] - s.x0[0]) > MZ_PB_SLACK || fabs(q[1] - s.x0[1]) > MZ_PB_SLACK;
  if (cx.any(near)) near = near && point_near_wall3(P, q[0], q[1], P.reach);
#ifdef MZ_EXP_NOCOLLISION
  near = false;
#endif
  MZB_TICK(0);
  if (!cx.any(near)) return;
  // ---- collision, one pass: 9 arrow-cell enumerators first (they share their code), then 9 sphere-cell ones
  MZ_FOR(one, 1) { s.nstage = 0; s.njobs = 0; s.jmask = 0u; }
  cx.sync();
  const MazeDev& z = P.maze;
  const double margin = P.pair[0].margin, ax = q[0] + P.arr_off * co, ay = q[1] + P.arr_off * si;
  const double wh[3] = {z.half_xy, z.half_xy, z.half_z};
  MZ_FOR(i, 18) {
    double wcx, wcy;
    if (near && pb_wall_cell(z, q[0], q[1], i % 9, &wcx, &wcy)) {
      if (i >= 9) {  // sphere (geom1) vs wall cell (geom2)
        const double c[3] = {q[0] - wcx, q[1] - wcy, P.sph_z - (double)z.center_z};
        double dd, nrm[3];
        if (pl_sphere_box(c, P.sph_r, wh, margin, &dd, nrm) && dd < margin)
          pb_stage<C>(s, dd, q[0] + nrm[0] * (P.sph_r + 0.5 * dd), q[1] + nrm[1] * (P.sph_r + 0.5 * dd), nrm[0], nrm[1], nrm[2], -1.0);
      } else {       // wall cell (geom1) vs arrow (geom2, rotated about z by the heading)
#ifndef MZ_EXP_NOARROW
        const double ex = fmax(fabs(ax - wcx) - wh[0], 0.0), ey = fmax(fabs(ay - wcy) - wh[1], 0.0), rr = P.arr_rxy + margin;
        if (ex * ex + ey * ey <= rr * rr) {
          const double wc[3] = {wcx, wcy, (double)z.center_z}, ac[3] = {ax, ay, P.arr_z}, ah[3] = {P.arr_hx, P.arr_hy, P.arr_hz};
          pl_box_box_upright_t<true>(wc, wh, ac, ah, co, si, false, margin, -1, 0, 0,
            [&](const PlContact& ct, int mult) { if (ct.dist < margin) pb_stage<C>(s, ct.dist, ct.pos[0], ct.pos[1], ct.n[0], ct.n[1], ct.n[2], (double)mult); },
            [&](double dist, double sg) {
              if (dist < margin) {
#if defined(__HIP_DEVICE_COMPILE__)
                const int j = atomicAdd(&s.njobs, 1);
#else
                const int j = s.njobs++;
#endif
                if (j < 4) { s.job[j].wcx = wcx; s.job[j].wcy = wcy; s.job[j].dist = dist; s.job[j].sg = sg; }
              }
            });
        }
#endif
      }
    }
  }
  cx.sync();
  // ---- deep overlaps: one candidate of the rectangle clipping per lane
  {
    const int nj = s.njobs < 4 ? s.njobs : 4;
    if (cx.any(nj > 0)) {
      PbClip k;
      k.U1 = P.arr_hx * co; k.U2 = P.arr_hx * si; k.V1 = -P.arr_hy * si; k.V2 = P.arr_hy * co;
      k.S = z.half_xy; k.tol = 1e-12 * (1.0 + 2.0 * k.S); k.det = k.U1 * k.V2 - k.U2 * k.V1;
      const double dtol = 1e-9 * (1.0 + 2.0 * k.S);
      const bool aligned = fabs(k.U1 * k.U2) + fabs(k.V1 * k.V2) <= 1e-9 * (k.U1 * k.U1 + k.U2 * k.U2 + k.V1 * k.V1 + k.V2 * k.V2);
      for (int j = 0; cx.any(j < nj); j++) {
        const bool mine = j < nj;
        const PlDeepJob jb = s.job[mine ? j : 0];
        k.P1 = ax - jb.wcx; k.P2 = ay - jb.wcy;
        bool skip = !mine;
        if (aligned) {  // no area -> no contact [ASSUME-12] (pl_box_box): only rectangles with parallel edges can share a mere border line
          const double e1 = fabs(k.U1) + fabs(k.V1), e2 = fabs(k.U2) + fabs(k.V2);
          if (fmin(k.S, k.P1 + e1) - fmax(-k.S, k.P1 - e1) <= MZ_BOX_MINOVERLAP || fmin(k.S, k.P2 + e2) - fmax(-k.S, k.P2 - e2) <= MZ_BOX_MINOVERLAP) skip = true;
        }
        double x1[KC], x2[KC];
#pragma unroll
        for (int kk = 0; kk < KC; kk++) {
          const int c = lane + kk * G;
          x1[kk] = 0.0; x2[kk] = 0.0;
          if (!skip && c < 24 && pb_cand(k, c, x1[kk], x2[kk])) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicOr(&s.jmask, 1u << c);
#else
            s.jmask |= 1u << c;
#endif
          }
        }
        cx.sync();
        if (aligned && !skip) {  // aligned rectangles: coincident candidates count once (rare: a heading of exactly k pi/2) — serially
          MZ_FOR(one, 1) {
            const unsigned valid = s.jmask;
            unsigned kept = 0u;
            int nk = 0;
            for (int c = 0; c < 24; c++) {
              if (!((valid >> c) & 1u)) continue;
              double a1, a2;
              pb_cand(k, c, a1, a2);
              bool dup = false;
              for (int e = 0; e < c; e++) {
                if (!((kept >> e) & 1u)) continue;
                double b1, b2;
                pb_cand(k, e, b1, b2);
                if (fabs(a1 - b1) <= dtol && fabs(a2 - b2) <= dtol) dup = true;
              }
              if (dup || nk >= 8) continue;
              kept |= 1u << c;
              nk++;
            }
            s.jmask = kept;
          }
          cx.sync();
        }
        const unsigned mask = s.jmask;
        const int base = s.nstage;
        cx.sync();
#pragma unroll
        for (int kk = 0; kk < KC; kk++) {
          const int c = lane + kk * G;
          if (!skip && c < 24 && ((mask >> c) & 1u)) {
            const int rank = __builtin_popcount(mask & ((1u << c) - 1u));
            if (rank < 8 && base + rank < S::CAP) {
              PlBareEntry& e = s.stage[base + rank];
              e.dist = jb.dist; e.px = x1[kk] + jb.wcx; e.py = x2[kk] + jb.wcy; e.n[0] = 0.0; e.n[1] = 0.0; e.n[2] = jb.sg; e.sw = 1.0;
            }
          }
        }
        MZ_FOR(one, 1) {
          if (!skip) { const int cnt = __builtin_popcount(mask); s.nstage = base + (cnt < 8 ? cnt : 8); }
          s.jmask = 0u;
        }
        cx.sync();
      }
    }
  }
  MZB_TICK(1);
  int ncon = s.nstage;
  if (s.nstage > S::CAP || s.njobs > 4) { ncon = ncon > S::CAP ? S::CAP : ncon; MZ_FOR(one, 1) s.status |= MZ_STATUS_CONTACT_OVERFLOW; }
  if (!near) ncon = 0;
  if (!cx.any(ncon > 0)) return;
  // ---- constraint rows: lane l of a 16-lane row owns the contacts l, l + 16 (groups narrower than a row: l, l + G, ...)
  const int l16 = lane & (SW - 1);
  const bool second = cx.any(ncon > SW);  // wave-uniform: some env needs the lanes' later slots
  PbRow R0;  // slot 0 in registers; slots 1 .. K - 1 in s.rowx[slot] (every row of the group writes the same values)
  pb_zero(R0);
  if (l16 < ncon) pb_fill(P, s.stage[l16], q, v, R0);
  if (second) {
#pragma unroll 1
    for (int kk = 1; kk < K; kk++) {
      const int slot = l16 + kk * SW;
      PbRow r;
      pb_zero(r);
      if (slot < ncon) pb_fill(P, s.stage[slot], q, v, r);
      s.rowx[slot] = r;
    }
    cx.sync();
  }
  MZB_TICK(3);
  // ---- Newton on the primal problem, exact line search (planar_forward's iteration; M = [m 0 -mcs; 0 m mcc; -mcs mcc izz])
  double a[3] = {qas[0] + wd[0], qas[1] + wd[1], qas[2] + wd[2]};
  bool done = ncon == 0;
#ifdef MZ_EXP_NONEWTON
  done = true;
#endif
  int it = 0;
  while (cx.any(!done) && it < 50) {
    const double d0 = a[0] - qas[0], d1 = a[1] - qas[1], d2 = a[2] - qas[2];
    const double Mx[3] = {P.mass * d0 - mcs * d2, P.mass * d1 + mcc * d2, -mcs * d0 + mcc * d1 + P.izz * d2};
    double pg[3] = {0.0, 0.0, 0.0}, pH[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // H entries 00 01 02 11 12 22
    double u[K][3];
#pragma unroll
    for (int kk = 0; kk < K; kk++) {
      u[kk][0] = u[kk][1] = u[kk][2] = 0.0;
      if (kk == 0 || second) {
        const PbRow r = kk == 0 ? R0 : s.rowx[l16 + kk * SW];
        double g3[3], W[5];
#pragma unroll
        for (int c = 0; c < 3; c++) u[kk][c] = r.J[c][0] * a[0] + r.J[c][1] * a[1] + r.J[c][2] * a[2] - r.aref[c];
        pl_contact_eval(r.D, u[kk], g3, W);
        // Y = W J (W: 00, 01, 02, 11, 22), then J^T Y
        double Y[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
          Y[0][i] = W[0] * r.J[0][i] + W[1] * r.J[1][i] + W[2] * r.J[2][i];
          Y[1][i] = W[1] * r.J[0][i] + W[3] * r.J[1][i];
          Y[2][i] = W[2] * r.J[0][i] + W[4] * r.J[2][i];
          pg[i] += r.J[0][i] * g3[0] + r.J[1][i] * g3[1] + r.J[2][i] * g3[2];
        }
        int e = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = i; j < 3; j++, e++) pH[e] += r.J[0][i] * Y[0][j] + r.J[1][i] * Y[1][j] + r.J[2][i] * Y[2][j];
      }
    }
    double g[3], H[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) g[i] = Mx[i] + cx.rowsum(pg[i]);
    H[0][0] = P.mass + cx.rowsum(pH[0]); H[0][1] = cx.rowsum(pH[1]); H[0][2] = -mcs + cx.rowsum(pH[2]);
    H[1][1] = P.mass + cx.rowsum(pH[3]); H[1][2] = mcc + cx.rowsum(pH[4]); H[2][2] = P.izz + cx.rowsum(pH[5]);
    H[1][0] = H[0][1]; H[2][0] = H[0][2]; H[2][1] = H[1][2];
    const double gn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    if (!done && P.inv_scale * gn < 1e-10) done = true;
    if (!cx.any(!done)) break;
    if (it == 49 && !done) { MZ_FOR(one, 1) s.status |= MZ_STATUS_SOLVER_MAXITER; }
    double L[3][3], y[3], inv[3];  // Cholesky with one reciprocal square root per column, no division
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double d = H[j][j];
      for (int kq = 0; kq < j; kq++) d -= L[j][kq] * L[j][kq];
#if defined(__HIP_DEVICE_COMPILE__)
      inv[j] = rsqrt(fmax(d, 1e-300));
#else
      inv[j] = 1.0 / sqrt(fmax(d, 1e-300));
#endif
      for (int i = j + 1; i < 3; i++) {
        double t = H[i][j];
        for (int kq = 0; kq < j; kq++) t -= L[i][kq] * L[j][kq];
        L[i][j] = t * inv[j];
      }
    }
    for (int i = 0; i < 3; i++) { double t = -g[i]; for (int kq = 0; kq < i; kq++) t -= L[i][kq] * y[kq]; y[i] = t * inv[i]; }
    for (int i = 2; i >= 0; i--) { double t = y[i]; for (int kq = i + 1; kq < 3; kq++) t -= L[kq][i] * y[kq]; y[i] = t * inv[i]; }
    const double sr[3] = {y[0], y[1], y[2]};
    const double Ms[3] = {P.mass * sr[0] - mcs * sr[2], P.mass * sr[1] + mcc * sr[2], -mcs * sr[0] + mcc * sr[1] + P.izz * sr[2]};
    const double p1 = sr[0] * Mx[0] + sr[1] * Mx[1] + sr[2] * Mx[2], p2 = sr[0] * Ms[0] + sr[1] * Ms[1] + sr[2] * Ms[2];
    double jv[K][3];
    bool changed = false;
#pragma unroll
    for (int kk = 0; kk < K; kk++) {
      jv[kk][0] = jv[kk][1] = jv[kk][2] = 0.0;
      if (kk == 0 || second) {
        const PbRow r = kk == 0 ? R0 : s.rowx[l16 + kk * SW];
#pragma unroll
        for (int c = 0; c < 3; c++) jv[kk][c] = r.J[c][0] * sr[0] + r.J[c][1] * sr[1] + r.J[c][2] * sr[2];
        const double u0 = u[kk][0], u1 = u[kk][1], u2 = u[kk][2], w0 = u0 + jv[kk][0], w1 = u1 + jv[kk][1], w2 = u2 + jv[kk][2];
        // (an empty slot holds u = jv = 0: no sign changes)
        changed = changed || ((u0 + u1 < 0) != (w0 + w1 < 0)) || ((u0 - u1 < 0) != (w0 - w1 < 0)) || ((u0 + u2 < 0) != (w0 + w2 < 0)) ||
                  ((u0 - u2 < 0) != (w0 - w2 < 0));
      }
    }
    changed = cx.gany(changed);
    double lo = 0.0, hi = -1.0, alpha = 1.0, prev_d2 = -1.0;
    // (round 5, as in the Ant's solvers — ant_newton_rows.h: the first MZ_PL_UNIT_STEPS iterations of a solve take the unit step when
    // the active set changes, the exact search comes behind them: it buys global convergence, not accuracy, and Newton with an exact
    // search converges from wherever the unit steps leave it)
    const bool fast_now = it < P.unit_steps;  // (option "ls_fast_iterations"; default MZ_PL_UNIT_STEPS; custom tasks: 0, maze_env.py)
    for (int ls = 0; ls < 30 && changed && !fast_now; ls++) {
      double e1 = 0.0, e2 = 0.0;
#pragma unroll
      for (int kk = 0; kk < K; kk++) {
        if (kk == 0 || second) {
          const double Dc = kk == 0 ? R0.D : s.rowx[l16 + kk * SW].D, v0 = jv[kk][0], v1 = jv[kk][1], v2 = jv[kk][2];
          const double u0 = u[kk][0] + alpha * v0, u1 = u[kk][1] + alpha * v1, u2 = u[kk][2] + alpha * v2;
          double r, w;
          r = u0 + u1; w = v0 + v1; if (r < 0) { e1 += Dc * r * w; e2 += Dc * w * w; }
          r = u0 - u1; w = v0 - v1; if (r < 0) { e1 += Dc * r * w; e2 += Dc * w * w; }
          r = u0 + u2; w = v0 + v2; if (r < 0) { e1 += Dc * r * w; e2 += Dc * w * w; }
          r = u0 - u2; w = v0 - v2; if (r < 0) { e1 += Dc * r * w; e2 += Dc * w * w; }
        }
      }
      e1 = cx.rowsum(e1) + p1 + alpha * p2;
      e2 = cx.rowsum(e2) + p2;
      if (e2 == prev_d2) break;
      prev_d2 = e2;
      if (e1 < 0) lo = alpha; else hi = alpha;
      double next = alpha - e1 / e2;
      if (hi >= 0 && !(next > lo && next < hi)) next = 0.5 * (lo + hi);
      if (!(next > 0)) next = hi >= 0 ? 0.5 * (lo + hi) : 0.0;
      if (fabs(next - alpha) <= 1e-15 * fabs(next)) { alpha = next; break; }
      alpha = next;
    }
    if (!done) { a[0] += alpha * sr[0]; a[1] += alpha * sr[1]; a[2] += alpha * sr[2]; }
    if (!changed) done = true;
    it++;
  }
  if (ncon > 0) { qacc[0] = a[0]; qacc[1] = a[1]; qacc[2] = a[2]; }
  MZB_TICK(4);
}

template <int NB, int NS, class C>
MZP_HD int point_bounce_group(const C& cx, const PointDev& P, PlanarScratch<NB, NS>& s, const double* old_xy, const double* new_xy, double* fin);

// ------------------------------------------------------------------ One MazeEnv.step.  s.q / s.v hold the state in and out.
template <class C>
MZP_HD void point_env_step_bare(const C& cx, const PointDev& P, PlanarScratch<0, 0>& s, const double* action) {
  const double PI = 3.141592653589793;
  cx.sync();
  double q[3] = {s.q[0], s.q[1], s.q[2]}, v[3] = {s.v[0], s.v[1], s.v[2]};
  cx.sync();  // every lane has read the state
  MZ_FOR(one, 1) { s.old_xy[0] = q[0]; s.old_xy[1] = q[1]; s.status = 0; }
  {  // point.py:45-56
    double th = q[2] + action[1];
    if (th < -PI) th += PI * 2;
    else if (PI < th) th -= PI * 2;
    double sn, cs;
    pt_sincos(th, &sn, &cs);
    q[2] = th; q[0] += cs * action[0]; q[1] += sn * action[0];
#pragma unroll
    for (int i = 0; i < 3; i++) v[i] = fmin(fmax(v[i], -P.vel_limit), P.vel_limit);
  }
  MZB_TICK(5);
  for (int f = 0; f < P.frame_skip; f++) {  // mj_step, RK4 (point.xml:3)
    const double h = P.h;
    MZ_FOR(one, 1) { for (int i = 0; i < 3; i++) { s.x0[i] = q[i]; s.v0[i] = v[i]; s.accv[i] = 0.0; s.accf[i] = 0.0; } }
    cx.sync();
    // (a drift of up to the slack in x AND in y is sqrt(2) slacks: the pre-test's radius grows by 1.5)
    const bool maybe = !(P.reach + 1.5 * MZ_PB_SLACK < (double)P.maze.scale) || point_near_wall3(P, q[0], q[1], P.reach + 1.5 * MZ_PB_SLACK);
    double qacc[3] = {0.0, 0.0, 0.0}, qas[3] = {0.0, 0.0, 0.0};
    if (f == 0) { qacc[0] = s.wds[0]; qacc[1] = s.wds[1]; qacc[2] = s.wds[2]; }  // (carried across steps in the state record: planar_step_body)
    for (int st = 0; st < 4; st++) {
      point_forward_bare(cx, P, s, q, v, qacc, qas, st > 0 || f == 0, maybe);
      if (st == 3) { cx.sync(); MZ_FOR(one, 1) { s.wds[0] = qacc[0] - qas[0]; s.wds[1] = qacc[1] - qas[1]; s.wds[2] = qacc[2] - qas[2]; } }
      MZB_TICK(8);
      const double bw = (st == 0 || st == 3) ? 1.0 / 6 : 1.0 / 3, aw = st == 2 ? 1.0 : 0.5;
      MZ_FOR(one, 1) { for (int i = 0; i < 3; i++) { s.accv[i] += bw * v[i]; s.accf[i] += bw * qacc[i]; } }
      cx.sync();
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double nq = s.x0[i] + h * (aw * v[i]), nv = s.v0[i] + h * (aw * qacc[i]);
        q[i] = nq; v[i] = nv;
      }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { q[i] = s.x0[i] + h * s.accv[i]; v[i] = s.v0[i] + h * s.accf[i]; }
    cx.sync();  // (the next frame's lane 0 overwrites x0 ... accf)
    MZB_TICK(6);
  }
  // maze_env.py:454-464: manual wall bounce on the robot's xy
#ifndef MZ_EXP_NODETECT
  if (P.nseg > 0) {
    const double new_xy[2] = {q[0], q[1]}, old_xy[2] = {s.old_xy[0], s.old_xy[1]};
    double fin[2];
    cx.sync();  // the staging block is free: it is the detector's hand-off buffer now
    const int r = point_bounce_group<0, 0>(cx, P, s, old_xy, new_xy, fin);
    if (r < 0) { MZ_FOR(one, 1) s.status |= MZ_STATUS_COLLINEAR; }
    q[0] = fin[0]; q[1] = fin[1];
    MZB_TICK(7);
  }
#endif
  cx.sync();
  MZ_FOR(one, 1) {
    for (int i = 0; i < 3; i++) { s.q[i] = q[i]; s.v[i] = v[i]; }
  }
  cx.sync();
}
