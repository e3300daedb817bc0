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
// Newton iterations of a solve that take the unit step before the exact line search takes over (round 5; measured on PointUMaze /
// PointPush, tools/gpu_point.sh)
#ifndef MZ_PL_UNIT_STEPS
#define MZ_PL_UNIT_STEPS 5
#endif

template <int NB, int NS>
struct PlanarDims {
  static_assert(NB == 0 || NS == 0, "no registered maze mixes movable blocks and object balls");
  static_assert(NS <= 1, "one object ball");
  static constexpr int NV = 3 + 2 * NB + 3 * NS;  // robot x, y, theta | block x, y ... | ball x, y, spin
    static constexpr int NC = NB == 0 ? 24 + 4 * NS : (NB == 1 ? 40 : (NB == 2 ? 64 : 96));  // contact slots (mjc_BoxBox gives the arrow up to 8 contacts per wall cell)
  // enumerators: 9 sphere-wall, 9 arrow-wall | per block: sphere-block, arrow-block, 9 block-wall, 9 block-platform (elevated
  //              mazes), block-floor, 2 joint-limit rows (limited slides) | block pairs | per ball: 9 ball-wall, robot
  //              sphere-ball, ball-arrow
  static constexpr int EPB = 23;  // enumerators per block
  static constexpr int NE = 18 + EPB * NB + NB * (NB - 1) / 2 + 11 * NS;
  static constexpr int NOBS = 7 + 3 * NB + 3 * NS;
};

// developer aid (tools/exp_build.sh PROF, tools/exp_point_prof.py): -DMZ_EXP_PROF builds an experiment library whose bare-Point kernel (point_bare.h) times its phases
// with s_memtime (lane 0 of a group; planar_kernels.hip prints one workgroup's totals) — compiled out otherwise
#if defined(MZ_EXP_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define MZB_TICK(id) do { if (cx.lane0() == 0) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); s.prof[id] += now_ - s.prof_t0; s.prof_t0 = now_; } } while (0)
#else
#define MZB_TICK(id) do { } while (0)
#endif

template <int NB, int NS>
struct alignas(16) PlanarScratch {  // (the bare Point, <0, 0>: point_bare.h)
  using D = PlanarDims<NB, NS>;
  double q[D::NV], v[D::NV];  // state of the current RK4 stage
  double x0[D::NV], v0[D::NV], accv[D::NV], accf[D::NV];
  double qas[D::NV], qacc[D::NV], grad[D::NV], search[D::NV], Mx[D::NV], Ms[D::NV];
  double wd[D::NV];           // constraint-induced acceleration (qacc - qacc_smooth) of the previous RK4 stage: warm start
  double M3[3][3];            // robot block of the mass matrix (blocks: block_mass on the diagonal)
  double H[D::NV][D::NV];
  double cJ[D::NC][3][D::NV], caref[D::NC][3], cD[D::NC], cu[D::NC][3], cg[D::NC][3], cW[D::NC][5], cjv[D::NC][3];
  double co, si;              // cos / sin of the heading of the current stage
  double old_xy[2];           // robot xy in front of the step (the wall detector's start point): parked here, not carried in registers through the step
  int ncon, cnt[D::NE], cbeg[D::NE], status, robot_near;
  // the wall detector's hand-off buffer: the contact arrays cJ ... cjv (one declaration, contiguous), free between the RK4 step and the next env.step
  MZP_HD double* detect_buf() { return &cJ[0][0][0]; }
  static constexpr int detect_buf_doubles = (int)((sizeof(cJ) + sizeof(caref) + sizeof(cD) + sizeof(cu) + sizeof(cg) + sizeof(cW) + sizeof(cjv)) / sizeof(double));
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
    const double ww = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], rm = r + margin + 1e-9;
    if (ww > rm * rm) return false;  // clearly apart: no square root (the exact test follows)
    dd = sqrt(ww);
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

MZP_HD double sel3d(const double* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }  // no dynamically indexed private array

// box (geom1: centre pos1, axes = columns of the row-major mat1, half sizes size1) vs box (geom2): MuJoCo's mjc_BoxBox as
// restated in oracle/mzo_physics.c box_box, face case: least-penetration face axis in MuJoCo's order
// (per axis index the face of box 1, then of box 2; strict improvement), incident face = the face of the other box most
// anti-parallel to it, contact candidates = the vertices of the intersection of the incident rectangle with the reference
// rectangle (incident corners inside | incident edges against the border lines | reference corners inside, on the incident
// plane; inclusive tests, coincident candidates once), each with its own distance to the reference face, midway position,
// normal from geom1 to geom2, at most 8.  The edge-edge axes are not searched: the boxes of the Point's world (the arrow,
// walls, movable blocks) are rotated about z only, where every edge-edge axis coincides with a face axis and loses the tie
// (point_dev_from_model admits nothing else).  Written with constant array indices throughout (the dynamic choices go
// through selects), so that nothing lands in scratch memory.
template <class Emit>
MZP_HD void pl_box_box(const double* pos1, const double* mat1, const double* size1, const double* pos2, const double* mat2, const double* size2,
                       double margin, int b1id, int b2id, int cls, Emit&& emit) {
  double rot[9], pos21[3], pos12[3];
  const double d0 = pos2[0] - pos1[0], d1 = pos2[1] - pos1[1], d2 = pos2[2] - pos1[2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) rot[3 * i + j] = mat1[i] * mat2[j] + mat1[3 + i] * mat2[3 + j] + mat1[6 + i] * mat2[6 + j];
    pos21[i] = mat1[i] * d0 + mat1[3 + i] * d1 + mat1[6 + i] * d2;
    pos12[i] = -(mat2[i] * d0 + mat2[3 + i] * d1 + mat2[6 + i] * d2);
  }
  double penetration = margin + 3.0 * (size1[0] + size1[1] + size1[2] + size2[0] + size2[1] + size2[2]);
  int code = -1;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double plen2 = fabs(rot[3 * i]) * size2[0] + fabs(rot[3 * i + 1]) * size2[1] + fabs(rot[3 * i + 2]) * size2[2];
    const double plen1 = fabs(rot[i]) * size1[0] + fabs(rot[3 + i]) * size1[1] + fabs(rot[6 + i]) * size1[2];
    const double c1 = -fabs(pos21[i]) + size1[i] + plen2, c2 = -fabs(pos12[i]) + size2[i] + plen1;
    if (c1 < -margin || c2 < -margin) return;
    if (c1 < penetration) { penetration = c1; code = i; }
    if (c2 < penetration) { penetration = c2; code = 3 + i; }
  }
  if (code < 0) return;
  const bool fromB = code >= 3;
  const int a = fromB ? code - 3 : code, a1 = a == 2 ? 0 : a + 1, a2 = a == 0 ? 2 : a - 1;
  double R[9], pBA[3], sA[3], sB[3];  // the reference box A's frame: R[k][j] = axis k of A . axis j of B
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int j = 0; j < 3; j++) R[3 * k + j] = fromB ? rot[3 * j + k] : rot[3 * k + j];
    pBA[k] = fromB ? pos12[k] : pos21[k];
    sA[k] = fromB ? size2[k] : size1[k];
    sB[k] = fromB ? size1[k] : size2[k];
  }
  const double sg = sel3d(pBA, a) < 0.0 ? -1.0 : 1.0;
  const double* posA = fromB ? pos2 : pos1;
  const double* matA = fromB ? mat2 : mat1;
  if (a != 2 && R[8] > 1.0 - 1e-12) {
    // Fast path — both boxes upright (rotated about z only) and a HORIZONTAL reference normal, i.e. every arrow-wall /
    // arrow-block contact short of a deep overlap: the incident face is a vertical rectangle whose projection on the
    // reference face is an axis-aligned rectangle in (h, z) — h the horizontal axis of the reference face — so the
    // intersection the enumeration below would assemble from its 24 candidates is simply [hlo, hhi] x [zlo, zhi], its
    // corners carrying the depth of the incident edge at their h.  Same vertices, same tolerances.
    const int ah = 1 - a;
    const double Ra0 = a == 0 ? R[0] : R[3], Ra1 = a == 0 ? R[1] : R[4];       // row a of R, horizontal part
    const int b = fabs(Ra1) > fabs(Ra0) ? 1 : 0, bh = 1 - b;
    const double sb = (b == 0 ? Ra0 : Ra1) * sg > 0.0 ? -1.0 : 1.0;
    const double sBb = b == 0 ? sB[0] : sB[1], sBh = b == 0 ? sB[1] : sB[0];
    // incident face centre and its horizontal half edge, components along (ah, a)
    const double Rhb = ah == 0 ? (b == 0 ? R[0] : R[1]) : (b == 0 ? R[3] : R[4]), Rab = b == 0 ? Ra0 : Ra1;
    const double Rhh = ah == 0 ? (bh == 0 ? R[0] : R[1]) : (bh == 0 ? R[3] : R[4]), Rah = bh == 0 ? Ra0 : Ra1;
    const double Ph = (ah == 0 ? pBA[0] : pBA[1]) + sb * sBb * Rhb, P3 = (a == 0 ? pBA[0] : pBA[1]) + sb * sBb * Rab, Pz = pBA[2];
    const double Uh = sBh * Rhh, U3 = sBh * Rah;
    const double Sh = ah == 0 ? sA[0] : sA[1], S3 = a == 0 ? sA[0] : sA[1], Sz = sA[2];
    const double dtol = 1e-9 * (1.0 + Sh + Sz);
    const double hlo = fmax(-Sh, Ph - fabs(Uh)), hhi = fmin(Sh, Ph + fabs(Uh)), zlo = fmax(-Sz, Pz - sB[2]), zhi = fmin(Sz, Pz + sB[2]);
    if (hhi - hlo <= MZ_BOX_MINOVERLAP || zhi - zlo <= MZ_BOX_MINOVERLAP) return;  // positive overlap area [ASSUME-12]
    const int nh = hhi - hlo > dtol ? 2 : 1, nz = zhi - zlo > dtol ? 2 : 1;
    for (int ih = 0; ih < nh; ih++) {
      const double hh = ih ? hhi : hlo, x3 = P3 + (hh - Ph) / Uh * U3, dist = sg * x3 - S3;
      if (dist > margin) continue;
      const double x3m = x3 - sg * 0.5 * dist;
      for (int iz = 0; iz < nz; iz++) {
        const double zz = iz ? zhi : zlo;
        const double pl0 = a == 0 ? x3m : hh, pl1 = a == 0 ? hh : x3m, nl0 = a == 0 ? (fromB ? -sg : sg) : 0.0, nl1 = a == 0 ? 0.0 : (fromB ? -sg : sg);
        PlContact c;
        c.dist = dist;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          c.pos[k] = matA[3 * k] * pl0 + matA[3 * k + 1] * pl1 + matA[3 * k + 2] * zz + posA[k];
          c.n[k] = matA[3 * k] * nl0 + matA[3 * k + 1] * nl1;
        }
        c.b1 = b1id; c.b2 = b2id; c.cls = cls;
        emit(c);
      }
    }
    return;
  }
  const double Ra[3] = {a == 0 ? R[0] : (a == 1 ? R[3] : R[6]), a == 0 ? R[1] : (a == 1 ? R[4] : R[7]), a == 0 ? R[2] : (a == 1 ? R[5] : R[8])};
  int b = 0;
  if (fabs(Ra[1]) > fabs(Ra[b])) b = 1;
  if (fabs(Ra[2]) > fabs(sel3d(Ra, b))) b = 2;
  const double sb = sel3d(Ra, b) * sg > 0.0 ? -1.0 : 1.0;
  const int bb1 = b == 2 ? 0 : b + 1, bb2 = b == 0 ? 2 : b - 1;
  double cf[3], u[3], v[3];
  const double sBb = sel3d(sB, b), sB1 = sel3d(sB, bb1), sB2 = sel3d(sB, bb2);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double row[3] = {R[3 * k], R[3 * k + 1], R[3 * k + 2]};
    cf[k] = pBA[k] + sb * sBb * sel3d(row, b);
    u[k] = sB1 * sel3d(row, bb1);
    v[k] = sB2 * sel3d(row, bb2);
  }
  // coordinates permuted to (a1, a2, a): the reference rectangle is |x1| <= S1, |x2| <= S2 in the plane x3 = sg * S3
  const double P1 = sel3d(cf, a1), P2 = sel3d(cf, a2), P3 = sel3d(cf, a), U1 = sel3d(u, a1), U2 = sel3d(u, a2), U3 = sel3d(u, a);
  const double V1 = sel3d(v, a1), V2 = sel3d(v, a2), V3 = sel3d(v, a), S1 = sel3d(sA, a1), S2 = sel3d(sA, a2), S3 = sel3d(sA, a);
  const double tol = 1e-12 * (1.0 + S1 + S2), dtol = 1e-9 * (1.0 + S1 + S2);
  // Candidates can coincide only when an incident corner lies on a border line of the reference rectangle: structurally so
  // when the incident edges run parallel to the reference axes (aligned boxes), otherwise only by a coincidence at the 1e-9
  // level, which is ignored here — the comparison against the kept candidates runs for aligned rectangles only.
  const bool aligned = fabs(U1 * U2) + fabs(V1 * V2) <= 1e-9 * (U1 * U1 + U2 * U2 + V1 * V1 + V2 * V2);
  const double det = U1 * V2 - U2 * V1;
  // no area -> no contact [ASSUME-12]: only rectangles with parallel edges can intersect in a segment (a border line shared);
  // their intersection is the rectangle of the overlapping extents — measured directly (the oracle measures its candidates)
  if (aligned) {
    const double e1 = fabs(U1) + fabs(V1), e2 = fabs(U2) + fabs(V2);
    if (fmin(S1, P1 + e1) - fmax(-S1, P1 - e1) <= MZ_BOX_MINOVERLAP || fmin(S2, P2 + e2) - fmax(-S2, P2 - e2) <= MZ_BOX_MINOVERLAP) return;
  }
  // The 24 candidates in the oracle's order — 4 incident corners inside the reference rectangle | 4 incident edges x 2 border
  // directions x 2 sides | 4 reference corners inside the incident rectangle (on the incident plane) — as ONE rolled loop: this
  // path is rare (a deep overlap, whose least-penetration axis is vertical), and 24 inlined copies of the caller's emit cost every
  // launch a third of its time in instruction fetch alone (measured).
  auto cand = [&](int c, double& x1, double& x2, double& x3) -> bool {
    bool valid = false;
    x1 = 0.0; x2 = 0.0; x3 = 0.0;
    if (c < 4) {
      const double su = (c & 1) ? 1.0 : -1.0, sv = (c & 2) ? 1.0 : -1.0;
      x1 = P1 + su * U1 + sv * V1; x2 = P2 + su * U2 + sv * V2; x3 = P3 + su * U3 + sv * V3;
      valid = fabs(x1) <= S1 + tol && fabs(x2) <= S2 + tol;
    } else if (c < 20) {
      const int k = c - 4, e = k >> 2, w = (k >> 1) & 1;
      const double side = (k & 1) ? 1.0 : -1.0, sgn = (e & 1) ? 1.0 : -1.0;
      const double p1 = e < 2 ? P1 + sgn * V1 : P1 + sgn * U1, p2 = e < 2 ? P2 + sgn * V2 : P2 + sgn * U2, p3 = e < 2 ? P3 + sgn * V3 : P3 + sgn * U3;
      const double q1 = e < 2 ? U1 : V1, q2 = e < 2 ? U2 : V2, q3 = e < 2 ? U3 : V3;
      const double pc = w ? p2 : p1, qc = w ? q2 : q1, po = w ? p1 : p2, qo = w ? q1 : q2, Sc = w ? S2 : S1, So = w ? S1 : S2;
      if (fabs(qc) >= 1e-15) {
        const double t = (side * Sc - pc) / qc;
        valid = t >= -1.0 && t <= 1.0 && fabs(po + t * qo) <= So + tol;
        x1 = p1 + t * q1; x2 = p2 + t * q2; x3 = p3 + t * q3;
      }
    } else if (fabs(det) > 1e-15) {
      const int k = c - 20;
      const double x = ((k & 1) ? S1 : -S1) - P1, y = ((k & 2) ? S2 : -S2) - P2;
      const double al = (x * V2 - y * V1) / det, be = (U1 * y - U2 * x) / det;
      valid = fabs(al) <= 1.0 + 1e-12 && fabs(be) <= 1.0 + 1e-12;
      x1 = P1 + al * U1 + be * V1; x2 = P2 + al * U2 + be * V2; x3 = P3 + al * U3 + be * V3;
    }
    return valid;
  };
  // The candidates kept so far are a BIT SET, not a table of coordinates (round 4): a kept candidate is recomputed from its index
  // where a later one is compared against it.  The table was 24 doubles = 48 registers live through the routine — in a kernel that
  // never runs it for most envs, they were what pushed the bare Point's step kernel into scratch spills (PMC: most of its HBM
  // traffic); the recomputation is paid by aligned rectangles in deep overlap only.
  unsigned kept = 0u;
  int nk = 0, nemit = 0;
#pragma unroll 1
  for (int c = 0; c < 24; c++) {
    double x1, x2, x3;
    if (!cand(c, x1, x2, x3)) continue;
    if (aligned) {
      bool dup = false;
#pragma unroll 1
      for (int e = 0; e < c; e++) {
        if (!((kept >> e) & 1u)) continue;
        double y1, y2, y3;
        cand(e, y1, y2, y3);
        if (fabs(x1 - y1) <= dtol && fabs(x2 - y2) <= dtol && fabs(x3 - y3) <= dtol) dup = true;
      }
      if (dup || nk >= 8) continue;  // (a ninth distinct vertex cannot exist: two rectangles intersect in at most an octagon)
      kept |= 1u << c;
      nk++;
    }
    const double dist = sg * x3 - S3;
    if (dist > margin || nemit >= 8) continue;
    nemit++;
    const double x3m = x3 - sg * 0.5 * dist;
    double pl[3], nl[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { pl[k] = k == a1 ? x1 : (k == a2 ? x2 : x3m); nl[k] = k == a ? (fromB ? -sg : sg) : 0.0; }
    PlContact ct;
    ct.dist = dist;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      ct.pos[k] = matA[3 * k] * pl[0] + matA[3 * k + 1] * pl[1] + matA[3 * k + 2] * pl[2] + posA[k];
      ct.n[k] = matA[3 * k] * nl[0] + matA[3 * k + 1] * nl[1] + matA[3 * k + 2] * nl[2];
    }
    ct.b1 = b1id; ct.b2 = b2id; ct.cls = cls;
    emit(ct);
  }
}

// pl_box_box for the pair every Point step meets: an axis-aligned box and a box rotated about z by the heading (co, si) — the
// arrow against a wall cell (geom1 = the aligned box) or against a movable block (`rot_first`: geom1 = the rotated box).  Same
// routine, written out for these two rotation matrices: the products with the exact zeros and ones of an identity / z rotation
// are dropped (they change no value), which leaves a 2-D separating-axis test, the z overlap, and pl_box_box's fast path in
// scalars — about a quarter of the general routine's instructions.  A vertical least-penetration axis (deep overlap) hands
// over to the general routine.
// BARE (point_bare.h): the two contacts of an overlap's bottom and top edge at one (x, y) are emitted once, with their number —
// emit(contact, mult) — and the deep overlap is reported — deep(dist, sign of the z offset) — instead of clipped here.
template <bool BARE, class Emit, class Deep>
MZP_HD void pl_box_box_upright_t(const double* pos1, const double* size1, const double* pos2, const double* size2, double co, double si, bool rot_first,
                                 double margin, int b1id, int b2id, int cls, Emit&& emit, Deep&& deep) {
  const double d0 = pos2[0] - pos1[0], d1 = pos2[1] - pos1[1], d2 = pos2[2] - pos1[2];
  // rot[i][j] = axis i of box 1 . axis j of box 2 (upper-left 2 x 2; rot[2][2] = 1, the rest 0)
  const double r00 = co, r11 = co, r01 = rot_first ? si : -si, r10 = rot_first ? -si : si;
  const double e0 = co * d0 + si * d1, e1 = -si * d0 + co * d1;  // d in the rotated box's frame
  const double p21x = rot_first ? e0 : d0, p21y = rot_first ? e1 : d1, p12x = rot_first ? -d0 : -e0, p12y = rot_first ? -d1 : -e1;
  double penetration = margin + 3.0 * (size1[0] + size1[1] + size1[2] + size2[0] + size2[1] + size2[2]);
  int code = -1;
  {
    const double plen2 = fabs(r00) * size2[0] + fabs(r01) * size2[1], plen1 = fabs(r00) * size1[0] + fabs(r10) * size1[1];
    const double c1 = -fabs(p21x) + size1[0] + plen2, c2 = -fabs(p12x) + size2[0] + plen1;
    if (c1 < -margin || c2 < -margin) return;
    if (c1 < penetration) { penetration = c1; code = 0; }
    if (c2 < penetration) { penetration = c2; code = 3; }
  }
  {
    const double plen2 = fabs(r10) * size2[0] + fabs(r11) * size2[1], plen1 = fabs(r01) * size1[0] + fabs(r11) * size1[1];
    const double c1 = -fabs(p21y) + size1[1] + plen2, c2 = -fabs(p12y) + size2[1] + plen1;
    if (c1 < -margin || c2 < -margin) return;
    if (c1 < penetration) { penetration = c1; code = 1; }
    if (c2 < penetration) { penetration = c2; code = 4; }
  }
  {
    const double c1 = -fabs(d2) + size1[2] + size2[2], c2 = -fabs(d2) + size2[2] + size1[2];
    if (c1 < -margin || c2 < -margin) return;
    if (c1 < penetration || c2 < penetration) {  // vertical reference normal: the general routine (a deep overlap)
      if constexpr (BARE) { deep(-c1, d2 < 0.0 ? -1.0 : 1.0); return; } else {
#if defined(__HIP_DEVICE_COMPILE__)
      // opaque copy made inside the branch: what the general routine computes then depends on it, so the compiler cannot hoist
      // its loop-invariant parts in front of the branch, onto the path of every arrow near a wall (round 3 measured the general
      // routine's mere presence at 10 % of the launch; ant_dyn.h round_vs_box: same effect, same cure)
      asm volatile("" : "+v"(co), "+v"(si));
#endif
      const double m1[9] = {rot_first ? co : 1.0, rot_first ? -si : 0.0, 0.0, rot_first ? si : 0.0, rot_first ? co : 1.0, 0.0, 0.0, 0.0, 1.0};
      const double m2[9] = {rot_first ? 1.0 : co, rot_first ? 0.0 : -si, 0.0, rot_first ? 0.0 : si, rot_first ? 1.0 : co, 0.0, 0.0, 0.0, 1.0};
      pl_box_box(pos1, m1, size1, pos2, m2, size2, margin, b1id, b2id, cls, emit);
      return;
      }
    }
  }
  if (code < 0) return;
  const bool fromB = code >= 3;
  const int a = fromB ? code - 3 : code;  // 0 or 1
  // the reference box A's frame: R[k][j] = axis k of A . axis j of B
  const double R00 = r00, R01 = fromB ? r10 : r01, R10 = fromB ? r01 : r10, R11 = r11;
  const double pB0 = fromB ? p12x : p21x, pB1 = fromB ? p12y : p21y, pB2 = fromB ? -d2 : d2;
  const double sA0 = fromB ? size2[0] : size1[0], sA1 = fromB ? size2[1] : size1[1], sA2 = fromB ? size2[2] : size1[2];
  const double sB0 = fromB ? size1[0] : size2[0], sB1 = fromB ? size1[1] : size2[1], sB2 = fromB ? size1[2] : size2[2];
  const double sg = (a == 0 ? pB0 : pB1) < 0.0 ? -1.0 : 1.0;
  const double* posA = fromB ? pos2 : pos1;
  const bool rotA = fromB != rot_first;  // A is the rotated box
  const int ah = 1 - a;
  const double Ra0 = a == 0 ? R00 : R10, Ra1 = a == 0 ? R01 : R11;  // row a of R
  const int b = fabs(Ra1) > fabs(Ra0) ? 1 : 0, bh = 1 - b;
  const double sb = (b == 0 ? Ra0 : Ra1) * sg > 0.0 ? -1.0 : 1.0;
  const double sBb = b == 0 ? sB0 : sB1, sBh = b == 0 ? sB1 : sB0;
  const double Rhb = ah == 0 ? (b == 0 ? R00 : R01) : (b == 0 ? R10 : R11), Rab = b == 0 ? Ra0 : Ra1;
  const double Rhh = ah == 0 ? (bh == 0 ? R00 : R01) : (bh == 0 ? R10 : R11), Rah = bh == 0 ? Ra0 : Ra1;
  const double Ph = (ah == 0 ? pB0 : pB1) + sb * sBb * Rhb, P3 = (a == 0 ? pB0 : pB1) + sb * sBb * Rab, Pz = pB2;
  const double Uh = sBh * Rhh, U3 = sBh * Rah;
  const double Sh = ah == 0 ? sA0 : sA1, S3 = a == 0 ? sA0 : sA1, Sz = sA2;
  const double dtol = 1e-9 * (1.0 + Sh + Sz);
  const double hlo = fmax(-Sh, Ph - fabs(Uh)), hhi = fmin(Sh, Ph + fabs(Uh)), zlo = fmax(-Sz, Pz - sB2), zhi = fmin(Sz, Pz + sB2);
  if (hhi - hlo <= MZ_BOX_MINOVERLAP || zhi - zlo <= MZ_BOX_MINOVERLAP) return;  // positive overlap area [ASSUME-12]
  const int nh = hhi - hlo > dtol ? 2 : 1, nz = zhi - zlo > dtol ? 2 : 1;
  const double nsg = fromB ? -sg : sg, nl0 = a == 0 ? nsg : 0.0, nl1 = a == 0 ? 0.0 : nsg;
  const double nx = rotA ? co * nl0 - si * nl1 : nl0, ny = rotA ? si * nl0 + co * nl1 : nl1;
  for (int ih = 0; ih < nh; ih++) {
    const double hh = ih ? hhi : hlo, x3 = P3 + (hh - Ph) / Uh * U3, dist = sg * x3 - S3;
    if (dist > margin) continue;
    const double x3m = x3 - sg * 0.5 * dist;
    const double pl0 = a == 0 ? x3m : hh, pl1 = a == 0 ? hh : x3m;
    const double wx = rotA ? co * pl0 - si * pl1 : pl0, wy = rotA ? si * pl0 + co * pl1 : pl1;
    for (int iz = 0; iz < (BARE ? 1 : nz); iz++) {
      PlContact c;
      c.dist = dist;
      c.pos[0] = wx + posA[0]; c.pos[1] = wy + posA[1]; c.pos[2] = (iz ? zhi : zlo) + posA[2];
      c.n[0] = nx; c.n[1] = ny; c.n[2] = 0.0;
      c.b1 = b1id; c.b2 = b2id; c.cls = cls;
      if constexpr (BARE) emit(c, nz); else emit(c);
    }
  }
}
template <class Emit>
MZP_HD void pl_box_box_upright(const double* pos1, const double* size1, const double* pos2, const double* size2, double co, double si, bool rot_first,
                               double margin, int b1id, int b2id, int cls, Emit&& emit) {
  pl_box_box_upright_t<false>(pos1, size1, pos2, size2, co, si, rot_first, margin, b1id, b2id, cls, emit, [](double, double) {});
}

// axis-aligned box (geom1: centre c1, half h1) vs axis-aligned box (geom2: centre c2, half h2): aligned_box_box (ant_dyn.h)
template <class Emit>
MZP_HD void pl_box_box_aligned(const double* c1, const double* h1, const double* c2, const double* h2, double margin, int b1, int b2, int cls,
                               Emit&& emit) {
  AlignedBB bb;
  if (!aligned_box_box(c1, h1, c2, h2, margin, bb)) return;
  const int ax = bb.ax, u = ax == 2 ? 0 : ax + 1;
  for (int iu = 0; iu < bb.nu; iu++)
    for (int iw = 0; iw < bb.nv; iw++) {
      PlContact c;
      c.dist = bb.dist;
      for (int k = 0; k < 3; k++) { c.n[k] = k == ax ? bb.sg : 0.0; c.pos[k] = k == ax ? bb.pa : (k == u ? (iu ? bb.pu[1] : bb.pu[0]) : (iw ? bb.pv[1] : bb.pv[0])); }
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
#ifdef MZ_EXP_NOARROW
      return;
#endif
      // the arrow lies within its circumscribed circle: a cell farther than that from its centre cannot touch it
      const double ex = fmax(fabs(arrow[0] - wc[0]) - wh[0], 0.0), ey = fmax(fabs(arrow[1] - wc[1]) - wh[1], 0.0), rr = P.arr_rxy + pr.margin;
      if (ex * ex + ey * ey > rr * rr) return;
      const double ac[3] = {arrow[0], arrow[1], P.arr_z}, ah[3] = {P.arr_hx, P.arr_hy, P.arr_hz};
      pl_box_box_upright(wc, wh, ac, ah, s.co, s.si, false, pr.margin, -1, 0, 0, emit);
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
        const double ac[3] = {arrow[0], arrow[1], P.arr_z}, ah[3] = {P.arr_hx, P.arr_hy, P.arr_hz};
        pl_box_box_upright(ac, ah, bc, P.block_half, s.co, s.si, true, P.pair[1].margin, 0, 1 + b, 1, emit);
      } else if (k < 11) {  // wall (geom1) vs block (geom2)
        double wc[3];
        if (!pl_wall_cell(z, bc[0], bc[1], k - 2, wc)) return;
        pl_box_box_aligned(wc, wh, bc, P.block_half, P.pair[2].margin, -1, 1 + b, 2, emit);
      } else if (k < 20) {  // platform of an elevated maze (geom1) vs block (geom2): same box rule, same pair class as a wall
        double wc[3];
        if (!pl_platform_cell(z, bc[0], bc[1], k - 11, wc)) return;
        pl_box_box_aligned(wc, wh, bc, P.block_half, P.pair[2].margin, -1, 1 + b, 2, emit);
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
        pl_box_box_aligned(ca, P.block_half, cb, P.block_half, P.pair[3].margin, 1 + a, 1 + b, 3, emit);
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

#include "point_bare.h"  // PlanarScratch<0, 0>, point_env_step_bare: the bare Point's register-resident step

// ------------------------------------------------------------------ one forward-dynamics evaluation: s.q, s.v -> s.qacc
// (Point with movable blocks or an object ball; the bare Point: point_bare.h)
template <int NB, int NS, class C>
MZP_HD void planar_forward(const C& cx, const PointDev& P, PlanarScratch<NB, NS>& s, bool warm) {
  using D = PlanarDims<NB, NS>;
  constexpr int NV = D::NV, NC = D::NC, NE = D::NE;
  // Stages 2-4 of RK4 start the Newton solve from the previous stage's solution shifted by the change of qacc_smooth
  // (the optimum is unique: only the iteration count depends on the start).
  MZ_FOR(i, NV) s.wd[i] = warm ? s.qacc[i] - s.qas[i] : 0.0;
  cx.sync();
  MZ_FOR(one, 1) {
    double co, si;
    sincos(s.q[2], &si, &co);  // one argument reduction for both
    const double w2 = s.v[2] * s.v[2], mc = P.mass * P.com_x;
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
#ifdef MZ_EXP_NOCOLLISION
  maybe = false;
#endif
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
  if (!cx.any(s.ncon > 0)) return;
  MZ_FOR(e, NE) {
    if (s.cnt[e] > 0) {
      // the slots [cbeg, cbeg + cnt) belong to this enumerator whatever the second enumeration yields (two instantiations of
      // planar_contacts need not round alike): a contact beyond the count is dropped, a missing one becomes an empty row
      int slot = s.cbeg[e];
      const int end = slot + s.cnt[e];
      // the second enumeration only PARKS a contact's geometry in its slot (dist | pos | n in cD / cu / cg, bodies and class in
      // cjv — all rewritten later); the constraint rows are then built one contact per lane (below) instead of up to eight in a
      // row on the enumerator's lane
      planar_contacts<NB, NS>(P, s, e, [&](const PlContact& c) {
        if (c.dist < P.pair[c.cls].margin) {
          if (slot < NC && slot < end) {
            s.cD[slot] = c.dist;
            for (int k = 0; k < 3; k++) { s.cu[slot][k] = c.pos[k]; s.cg[slot][k] = c.n[k]; }
            s.cjv[slot][0] = (double)c.b1; s.cjv[slot][1] = (double)c.b2; s.cjv[slot][2] = (double)c.cls;
          }
          slot++;
        }
      });
      for (; slot < end && slot < NC; slot++) s.cjv[slot][2] = -1.0;  // counted, not found again: an empty row
    }
  }
  cx.sync();
  MZ_FOR(slot, s.ncon) {
    const int cls = (int)s.cjv[slot][2];
    if (cls < 0) {
      s.cD[slot] = 0.0;
      for (int a = 0; a < 3; a++) { s.caref[slot][a] = 0.0; for (int i = 0; i < PlanarDims<NB, NS>::NV; i++) s.cJ[slot][a][i] = 0.0; }
    } else {
      PlContact c;
      c.dist = s.cD[slot];
      for (int k = 0; k < 3; k++) { c.pos[k] = s.cu[slot][k]; c.n[k] = s.cg[slot][k]; }
      c.b1 = (int)s.cjv[slot][0]; c.b2 = (int)s.cjv[slot][1]; c.cls = cls;
      planar_fill_contact<NB, NS>(P, s, slot, c);
    }
  }
  const int ncon = s.ncon;
  if (ncon > 0) { MZ_FOR(i, NV) s.qacc[i] = s.qas[i] + s.wd[i]; }  // envs without contacts keep qacc = qacc_smooth
  cx.sync();
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
    const bool fast_now = it < P.unit_steps;  // (option "ls_fast_iterations"; default MZ_PL_UNIT_STEPS; custom tasks: 0, maze_env.py)
    if (changed && !fast_now) { p1 = cx.gsum(p1p); p2 = cx.gsum(p2p); }
    for (int ls = 0; ls < 30 && changed && !fast_now; ls++) {  // (unit steps first: point_bare.h / ant_newton_rows.h)
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

// CollisionDetector.detect on a lane group: lane l examines the segments l, l + G, ... (point_detect_range), parks its nearest
// collision in LDS (the contact arrays are free between steps), and every lane then picks the group's nearest — the first
// segment among equal distances, as the reference's loop does.  All lanes return the same answer.
template <int NB, int NS, class C>
MZP_HD int point_detect_group(const C& cx, const PointDev& P, PlanarScratch<NB, NS>& s, const double* o, const double* n, double* pt, double* rf) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  static_assert(PlanarScratch<NB, NS>::detect_buf_doubles >= 8 * C::nlanes, "the detector's hand-off buffer lives in arrays that are free between steps");
  const double mvx = n[0] - o[0], mvy = n[1] - o[1];
  if (mz_hypot(mvx, mvy) <= 1e-8) return 0;
  PtCand c;
  c.found = 0; c.degenerate = 0; c.k = 0; c.dist = 0.0; c.pt[0] = c.pt[1] = c.rf[0] = c.rf[1] = 0.0;
  point_detect_range(P, o, n, mvx, mvy, cx.lane0(), C::nlanes, c);
  // which lanes hold a candidate: none (the usual case: no wall between the two positions) ends the detection here, without the
  // hand-off; otherwise only those lanes' entries are read
  unsigned long long holders = cx.gballot(c.found != 0 || c.degenerate != 0);
  if (holders == 0ULL) return 0;
  double* buf = s.detect_buf();
  if (c.found || c.degenerate) {
    double* q = buf + 8 * cx.lane0();
    q[0] = (double)(c.found + 2 * c.degenerate); q[1] = c.dist; q[2] = (double)c.k; q[3] = c.pt[0]; q[4] = c.pt[1]; q[5] = c.rf[0]; q[6] = c.rf[1];
  }
  cx.sync();
  int found = 0, degenerate = 0, bestk = 0;
  double best = 0.0;
  for (; holders != 0ULL; holders &= holders - 1ULL) {
    const int j = __builtin_ctzll(holders);
    const double* q = buf + 8 * j;
    const int fl = (int)q[0], k = (int)q[2];
    if (fl & 2) degenerate = 1;
    if ((fl & 1) && (!found || q[1] < best || (q[1] == best && k < bestk))) {
      found = 1; best = q[1]; bestk = k;
      pt[0] = q[3]; pt[1] = q[4]; rf[0] = q[5]; rf[1] = q[6];
    }
  }
  cx.sync();  // the buffer is reused by the second detection of a bounce
  if (degenerate && !found) return -1;
  return found;
}

// point_bounce (point_dyn.h) on a lane group
template <int NB, int NS, class C>
MZP_HD int point_bounce_group(const C& cx, const PointDev& P, PlanarScratch<NB, NS>& s, const double* old_xy, const double* new_xy, double* fin) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  double pt[2] = {0.0, 0.0}, rf[2] = {0.0, 0.0};
  fin[0] = new_xy[0]; fin[1] = new_xy[1];
  const int hit = point_detect_group<NB, NS>(cx, P, s, old_xy, new_xy, pt, rf);
  if (hit <= 0) return hit;
  const double pos[2] = {pt[0] + P.restitution * (rf[0] - pt[0]), pt[1] + P.restitution * (rf[1] - pt[1])};
  double p2[2], r2[2];
  const int again = point_detect_group<NB, NS>(cx, P, s, old_xy, pos, p2, r2);
  if (again < 0) return -1;
  if (again > 0) { fin[0] = old_xy[0]; fin[1] = old_xy[1]; return 2; }
  fin[0] = pos[0]; fin[1] = pos[1];
  return 1;
}

// ------------------------------------------------------------------ One MazeEnv.step.  s.q / s.v hold the state in and out.
template <int NB, int NS, class C>
MZP_HD void planar_env_step(const C& cx, const PointDev& P, PlanarScratch<NB, NS>& s, const double* action) {
  if constexpr (NB == 0 && NS == 0) { point_env_step_bare(cx, P, s, action); return; } else {
  constexpr int NV = PlanarDims<NB, NS>::NV;
  const double PI = 3.141592653589793;
  cx.sync();
  MZ_FOR(one, 1) {  // point.py:45-56
    s.old_xy[0] = s.q[0]; s.old_xy[1] = s.q[1];
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
#ifndef MZ_EXP_NODETECT
  if (P.nseg > 0) {
    const double old_xy[2] = {s.old_xy[0], s.old_xy[1]}, new_xy[2] = {s.q[0], s.q[1]};
    double fin[2];
    cx.sync();  // every lane has read the new position before the hand-off buffer (and later s.q) is written
    const int r = point_bounce_group<NB, NS>(cx, P, s, old_xy, new_xy, fin);
    MZ_FOR(one, 1) {
      if (r < 0) s.status |= MZ_STATUS_COLLINEAR;
      s.q[0] = fin[0]; s.q[1] = fin[1];
    }
    cx.sync();
  }
#endif
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
