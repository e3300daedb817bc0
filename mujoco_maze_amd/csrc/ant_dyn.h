// ant_dyn.h — Ant rigid-body forward dynamics + constraint solve + RK4, written
// as *lane-group SPMD* code: one environment is advanced by a group of G
// adjacent lanes of a wavefront; every cross-lane hand-off goes through the
// env's LDS scratch block followed by cx.sync().
//
// What it replaces in the reference: the MuJoCo `mj_step` calls issued by
// `AntEnv.step` (mujoco_maze/ant.py:61-73, `do_simulation(action, 5)`), plus the
// surrounding `MazeEnv.step` bookkeeping (maze_env.py:448-481).  Pipeline per
// forward evaluation (20 per env.step: 5 frames x RK4), SURVEY §8a M1-M9:
//   K  kinematics of the 13 bodies in a torso-centred frame (fp32-safe far from the origin)
//   I  spatial inertias, composite-rigid-body mass matrix in *arrow* form: a dense hub block
//      (6 root dofs + 2 slide dofs per movable block), four 2x2 leg blocks and hub x leg couplings
//      (legs never couple with each other)
//   V  velocities and bias forces (recursive Newton-Euler, gravity included)
//   C  collision: floor plane vs sphere/capsule ends; maze wall boxes found through the cell grid
//      (only the <= 2x2 cells under a geom's bounding square); movable XY blocks (axis-aligned boxes:
//      block-floor corners, block-wall, robot-block)
//   J  contact Jacobians (3 x (hub + 2) sparse: hub + own leg), joint-limit rows
//   N  primal Newton solver on the pyramidal-cone soft-constraint cost; the Hessian keeps the arrow
//      sparsity (a contact touches hub + one leg), so each iteration factors four 2x2 blocks and one
//      hub-sized Schur complement
//   R  RK4 stage bookkeeping with manifold quaternion update
//
// Template parameter NB = number of movable XY blocks of the maze (0 for AntUMaze / Ant4Rooms,
// 1 for AntPush / AntBlockMaze / AntBlockCarry; mujoco_maze/maze_env.py:563-660).
//
// Execution contexts (template parameter C):
//   * device: csrc/mazestep.hip — G lanes per env, cx.sync() = wavefront-scope fence,
//     cx.gsum() = DPP/shuffle butterfly inside the group;
//   * host emulation (tests/emu, CPU tests of the kernel logic only — never a
//     product path): nlanes = 1, so every MZ_FOR runs all its items in order.
// Rule that makes both valid: inside one phase (between two cx.sync()) the
// iterations of an MZ_FOR are independent, and nothing but LDS scratch carries
// values from one phase to the next (group-uniform scalars may live in registers).
#pragma once
#include <cstddef>

#include "ant_model.h"

#if defined(__HIPCC__)
#define MZ_HD __host__ __device__ __forceinline__
#else
#define MZ_HD inline
#endif

#define MZ_FOR(i, n) for (int i = cx.lane0(); i < (n); i += C::nlanes)
// items 0..n-1 on the lanes base, base+1, ... (mod group size): lets unrelated work share one phase
#define MZ_FOR_AT(i, n, base) for (int i = mz_first_item(cx.lane0(), (base), C::nlanes); i < (n); i += C::nlanes)

MZ_HD int mz_first_item(int lane, int base, int nl) { return (lane - base) & (nl - 1); }  // group sizes are powers of two

struct HostCtx {
  static constexpr int nlanes = 1;
  static constexpr bool row_solver = false;  // the DPP-row Newton solver (ant_newton_rows.h) exists on the device only
  MZ_HD int lane0() const { return 0; }
  MZ_HD void sync() const {}
  MZ_HD float gsum(float x) const { return x; }
  MZ_HD double gsum(double x) const { return x; }
  MZ_HD double rowsum(double x) const { return x; }
  MZ_HD bool any(bool p) const { return p; }
  MZ_HD bool gany(bool p) const { return p; }
  MZ_HD unsigned long long gballot(bool p) const { return p ? 1ULL : 0ULL; }
  template <class S> MZ_HD void tick(S&, int) const {}
};

// ------------------------------------------------------------------ sizes
// The template parameter NB of everything below is a CONFIGURATION of movable bodies: 0-3 = that many movable blocks with two
// slides each (x y in the Push family, y z / x z falling blocks); 4 = ONE block with three slides (MultiFall's XYZ block);
// 5 = one object ball on a free joint (AntSmallBilliard): six more hub dofs (3 linear world-frame, 3 angular body-frame) and a
// 7-number pose, no blocks.
template <int NB>
struct AntDims {
  static constexpr bool BALL = NB == 5;
  static constexpr int NBLK = NB == 4 ? 1 : (NB == 5 ? 0 : NB);  // movable blocks
  static constexpr int BD = NB == 4 ? 3 : 2;     // slide dofs per block
  static constexpr int NXH = BALL ? 6 : BD * NBLK;  // hub dofs beyond the root's six
  // A movable block's contacts are enumerated by MZ_BSUB lanes (floor corners | the 3 x 3 grid cells under its bounding square,
  // platform and wall each | lower-numbered blocks and its own slide limits), in that order: on one lane the block was ten times
  // the work of any robot geom and the whole phase waited for it (twice: count and fill).
  static constexpr int BSUB = 11;
  static constexpr int NMOV = BSUB * NBLK + (BALL ? 1 : 0);  // enumerators of the movable bodies: the first ones
  static constexpr int NH = 6 + NXH;         // hub dofs: root 6 + the blocks' slides / the ball's six
  static constexpr int NV = 14 + NXH;        // MuJoCo dof order: root 0-5, legs 6-13, movable bodies 14..
  static constexpr int NQ = 15 + (BALL ? 7 : BD * NBLK);
  static constexpr int NCOL = NH + 2;      // contact Jacobian columns: hub, hip, ankle
  // contact slots: a block resting in a corridor holds 4 floor corners + 4 per adjacent wall/block face
  static constexpr int NC = NB == 0 ? 16 : (NB == 1 ? 32 : NB == 5 ? 28 : ((NB == 2 || NB == 4) ? 40 : 72));  // NB = 2: 40 keeps 8 one-env workgroups per CU (20 KB each); NB = 4: a three-slide block between platforms, walls and the floor filled 28 in long rollouts
  // NB = 1: 32 since round 5 (40 before) — the block's own contacts are merged entries now (a block on the floor against two walls: 3,
  // not 12), and 32 slots make an env's LDS block 9.9 KB: FOUR 16-lane waves (16 envs) fit a CU's 160 KB, so that batches beyond
  // 2048 envs run the one-block ant at 16 lanes per env in one round (ant_kernels.hip ant_lanes; AntPush 4096 envs 5.4 -> 8.6 M
  // env-steps/s).  Soak at 32 slots: 2 x (61 M + 15 M) env-steps of AntPush / AntFall, no CONTACT_OVERFLOW (tools/exp_push_nc.sh).
  static constexpr int NGEOM = 13 + NMOV;  // contact enumerators: movable bodies first, then the 13 robot geoms
  static constexpr int NHESS = NH * NH + 8 * NH + 12;
  static constexpr int NTRI = NH * (NH + 1) / 2;
  // coordinates carried as hi + lo pairs INSIDE a step (AntScratchT::qlo): the torso's x, y and the movable bodies' translations —
  // the absolute positions, metres from the origin, that enter contact distances by subtraction (lo index: 0, 1, then 2 + k)
  static constexpr int NLO = 2 + (BALL ? 3 : BD * NBLK);
  static constexpr int REC_T = NQ + 2 * NV;              // state record: qpos | qvel | warm | t | episode
  static constexpr int REC = (REC_T + 2 + 15) / 16 * 16;
};
MZ_HD int hub2dof(int h) { return h < 6 ? h : h + 8; }
// ------------------------------------------------------------------ scratch (LDS) per env
template <int NH>
struct Arrow {        // symmetric matrix with the ant's sparsity
  float rr[NH][NH];   // hub block (full storage)
  float rl[4][2][NH]; // leg l, dof (0 hip, 1 ankle) x hub
  float ll[4][3];     // leg l: hh, ha, aa
};
template <int NH>
struct ArrowFactor {
  float inv[4][3];    // inverse of the 2x2 leg blocks
  float T[4][2][NH];  // inv * rl
  float L[NH][NH];    // Schur complement (lower triangle)
  float rhs[NH];      // reduced right-hand side / hub solution
};

// The per-env scratch block comes in two types.  AntScratchCoreT: what EVERY path uses — and all that the quad forward pass
// (ant_forward_rows.h: plain ant, one two-slide block, >= 16 lanes per env) ever touches.  Its product kernels allocate only this
// type per env (4.7 KB instead of 9.1 KB for the plain ant, so that 32 envs fit a CU's LDS and 8192 envs run in one round) and hand
// the quad functions an AntScratchCoreT& — the members the lane-group formulation adds (AntScratchT below, derived) are out of
// their reach by type, not by convention (round 6).
template <bool BALL>
struct AntBallKinT { float bR[9], bx[3], bc[3]; };  // object ball (config 5): rotation (row-major), body origin and sphere centre relative to the torso origin
template <>
struct AntBallKinT<false> {};                       // (no bytes in the other configurations: an empty base)
template <int NB>
struct alignas(16) AntScratchCoreT : AntBallKinT<AntDims<NB>::BALL> {
  using D = AntDims<NB>;
  // step-persistent
  float qpos[D::NQ + 1], qvel[D::NV], x0q[D::NQ + 1], x0v[D::NV], accv[D::NV], accf[D::NV], warm[D::NV], fact[D::NV];
  float qacc[D::NV], qas[D::NV];
  float grad[D::NV], Mx[D::NV];  // (the RK4 bookkeeping of ant_mj_step borrows them between evaluations)
  // Low-order parts of the absolute positions (AntDims::NLO) within the step: qpos[i] + qlo[..] is the coordinate to ~1e-14.  The
  // state that enters and leaves a step is fp32; between its 20 forward evaluations a coordinate of ~10 m rounded to fp32 is off by
  // up to 5e-7, which a contact row turns into K * 5e-7 * h = 1e-5 of velocity (K ~ 2500-3900 /s^2 for these solref values) — the
  // whole 1e-5 budget, for any contact whose distance is a difference of such coordinates (robot vs wall, block vs wall, robot
  // vs block; measured: AntPush 99.9 % quantile 1.1e-5, round 3).  RK4's position updates are therefore formed in float64 and
  // split (ant_integrate_pos), and the contact code subtracts hi parts first, lo parts after.
  float qlo[D::NLO], x0lo[D::NLO];
  float red[4];
  // kinematics (positions relative to the torso origin c) — the quad path publishes them only for its fall-back
  float R0[9], cz;               // torso rotation (row-major), torso height
  int nearwall;                  // 0: no maze wall within the ant's reach of the torso (wall tests skipped)
  float w[12][3], com[12][3];    // capsule axis and centre of body 1 + 3l + k
  float zw[3];                   // hip axis (world) = R0 * ez
  float Sh[4][3], Sa[4][6];      // hip: linear part (angular = zw); ankle: angular, linear
  // contacts
  int ncon, cnt[D::NGEOM], cbeg[5];  // contacts of leg l occupy slots [cbeg[l], cbeg[l+1]); hub-only contacts [0, cbeg[0])
  int nblkcon;                       // the first nblkcon slots hold the contacts of the movable bodies' own enumerators (no robot geom involved)
  int cleg[D::NC], ccls[D::NC];      // leg (-1 none) and robot body class (-1 none) of the contact
  int csrc[D::NC], con_over;         // single-pass enumeration (plain ant): staging entry of the contact (-1: geometry in cY[c]); a geom overflowed its staging
  alignas(16) float cJ[D::NC][3][D::NCOL];  // [normal, mu*t1, mu*t2] x [hub, hip, ankle]; quad path: the contacts' 24-float wrench records
  alignas(16) float cY[D::NC][3][D::NCOL];  // W * J of the current Newton iterate (also stages contact geometry)
  uint32_t rowmask[MZ_MAX_GRID + 4];  // the maze's cell grid (bit j of word i: BLOCK), copied once per step: one LDS read per row lookup
  uint32_t platmask[MZ_MAX_GRID + 4]; // elevated mazes: bit j of word i: the cell carries a platform (every cell but the chasms)
  int status, iters;
  int ncon_true;  // contacts as MuJoCo counts them (s.ncon counts merged entries of the block's enumerators once: con_enum_item MERGE)
  // quad forward pass, NB = 1: the block's own enumerators are re-run only when the block has moved (ant_forward_rows.h).  bkey[0..3] =
  // bits of its slides (hi, lo) at the enumeration whose results sit in the staging block and in cnt[0 .. NMOV); bkey[4] != 0: valid
  int bkey[5];
};
// The lane-group formulation (8-lane groups, two and more blocks, the ball, the host emulation) and the instrumented builds add:
template <int NB>
struct alignas(16) AntScratchT : AntScratchCoreT<NB> {
  using D = AntDims<NB>;
  alignas(16) float qfs[D::NV];
  float p1[4][3], p2[4][3];      // aux / ankle body origins
  float cin[13][10];             // spatial inertias at c: m, h(3), Ibar(xx yy zz xy xz yz)
  float fbody[13][6], bias[D::NV], Iall[10];  // per-body inertial + velocity-product force (spatial, at c)
  alignas(16) Arrow<D::NH> M, H;
  ArrowFactor<D::NH> F;
  float search[D::NV], Ms[D::NV];
  float caref[D::NC][3], cD[D::NC], cu[D::NC][3], cjv[D::NC][3];
  // joint limits (8 hinges)
  float lsign[8], lD[8], laref[8], ljar[8], ljv[8], lact[8];
  // optional phase timers (device builds with PROF): cycles per phase id, last timestamp
  unsigned long long prof_t0;
  unsigned int prof[16];
  static constexpr size_t slim_bytes() { return sizeof(AntScratchCoreT<NB>); }
};
using AntScratch = AntScratchT<0>;

// ------------------------------------------------------------------ small helpers
MZ_HD float dot3f(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
MZ_HD void cross3f(float* r, const float* a, const float* b) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
MZ_HD void inertia_mulf(float* r, const float* I, const float* v) {  // spatial inertia (compact) times motion vector
  const float* h = I + 1;
  const float* J = I + 4;
  float a[3], b[3];
  cross3f(a, h, v + 3);
  cross3f(b, v, h);
  r[0] = J[0] * v[0] + J[3] * v[1] + J[4] * v[2] + a[0];
  r[1] = J[3] * v[0] + J[1] * v[1] + J[5] * v[2] + a[1];
  r[2] = J[4] * v[0] + J[5] * v[1] + J[2] * v[2] + a[2];
  r[3] = I[0] * v[3] + b[0]; r[4] = I[0] * v[4] + b[1]; r[5] = I[0] * v[5] + b[2];
}
MZ_HD void motion_crossf(float* r, const float* v, const float* s) {
  float a[3], b[3], c[3];
  cross3f(a, v, s); cross3f(b, v, s + 3); cross3f(c, v + 3, s);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
MZ_HD void force_crossf(float* r, const float* v, const float* f) {
  float a[3], b[3], c[3];
  cross3f(a, v, f); cross3f(b, v + 3, f + 3); cross3f(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
MZ_HD float dot6f(const float* a, const float* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
MZ_HD void quat_to_matf(float* m, const float* qin) {
  float n = 1.0f / sqrtf(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
  float w = qin[0] * n, x = qin[1] * n, y = qin[2] * n, z = qin[3] * n;
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
// sin and cos together: one Cody-Waite reduction to [-pi/4, pi/4] (k = nearest multiple of pi/2; three-part constant, exact
// for the |x| < ~1e4 a joint angle or half a rotation step can reach) and the two cephes minimax polynomials (~1 ulp).
// One call replaces a sinf and a cosf of the math library, each of which reduces its argument separately.
MZ_HD void mz_sincosf(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.63661977236758134f);  // 2 / pi
  float y = x - k * 1.5703125f;                      // pi/2 split: 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8
  y -= k * 4.837512969970703125e-4f;
  y -= k * 7.54978995489188e-8f;
  const float z = y * y;
  const float ps = y + y * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
  const float pc = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
  const int q = (int)k & 3;
  const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}
MZ_HD void mat_vecf(float* r, const float* m, const float* v) {
  float x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
        z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
MZ_HD PairDev pair_copy(const PairDev& p) {  // (field by field: plain scalar loads, no address arithmetic on a selected pointer)
  PairDev r;
  r.margin = p.margin; r.mu = p.mu; r.K = p.K; r.B = p.B;
  for (int k = 0; k < 7; k++) r.solimp[k] = p.solimp[k];
  return r;
}
// Impedance d(x) AND its complement 1 - d, which the regulariser R = (1 - d) / d * diagApprox needs.  With the solimp of the block
// mazes (.995 .995: maze_env.py:108-112) the fp32 expression `1.f - imp` is 0.005 with the rounding error of 0.995 in it — 6e-6
// relative, which the stiff rows turn into the 1e-5 tail of AntPush's torso rates (round 3: 99.9 % quantile 1.1e-5, whatever the solver
// tolerance).  si[5], si[6] hold 1 - d0 and 1 - dmax rounded from float64 (ant_model.h pair_from); 1 - d follows from them
// without cancellation: 1 - d = (1 - d0) - y (dmax - d0).
MZ_HD float impedance_pair(const float* si, float x, float* one_minus) {
  const float d0 = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4], od0 = si[5], odm = si[6];
  if (d0 == dmax || width <= 1e-15f) { *one_minus = 0.5f * (od0 + odm); return 0.5f * (d0 + dmax); }
  const float xn = x / width;
  if (xn >= 1.0f) { *one_minus = odm; return dmax; }
  if (xn <= 0.0f) { *one_minus = od0; return d0; }
  float y;
  if (power <= 1.0f) y = xn;
  else if (power == 2.0f) y = xn <= mid ? xn * xn / mid : 1.0f - (1.0f - xn) * (1.0f - xn) / (1.0f - mid);  // MuJoCo default
  else if (xn <= mid) y = powf(xn, power) / powf(mid, power - 1.0f);
  else y = 1.0f - powf(1.0f - xn, power) / powf(1.0f - mid, power - 1.0f);
  *one_minus = od0 - y * (od0 - odm);
  return d0 + y * (dmax - d0);
}
MZ_HD float impedancef(const float* si, float x) {
  float d0 = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
  if (d0 == dmax || width <= 1e-15f) return 0.5f * (d0 + dmax);
  float xn = x / width;
  if (xn >= 1.0f) return dmax;
  if (xn <= 0.0f) return d0;
  float y;
  if (power <= 1.0f) y = xn;
  else if (power == 2.0f) y = xn <= mid ? xn * xn / mid : 1.0f - (1.0f - xn) * (1.0f - xn) / (1.0f - mid);  // MuJoCo default
  else if (xn <= mid) y = powf(xn, power) / powf(mid, power - 1.0f);
  else y = 1.0f - powf(1.0f - xn, power) / powf(1.0f - mid, power - 1.0f);
  return d0 + y * (dmax - d0);
}


// Row bitmask of the cell grid for a per-lane row index.  The grid lives in the kernel-argument block
// (scalar registers); a select chain keeps it there — indexing the array with a vector index would make the
// compiler spill it to scratch memory.
MZ_HD uint32_t maze_row(const MazeDev& z, int i) {
  uint32_t m = 0u;
#pragma unroll
  for (int r = 0; r < MZ_MAX_GRID; r++) m = (r == i) ? z.rowmask[r] : m;
  return m;
}

// the same lookup from the env's LDS copy of the grid (ant_fill_tables): one read instead of a 12-way select
template <class S>
MZ_HD uint32_t maze_row_lds(const S& s, int i) { return (i >= 0 && i < MZ_MAX_GRID) ? s.rowmask[i] : 0u; }
template <class S>
MZ_HD uint32_t plat_row_lds(const S& s, int i) { return (i >= 0 && i < MZ_MAX_GRID) ? s.platmask[i] : 0u; }

// body index b in 0..12: 0 torso, else leg l = (b-1)/3, level k = (b-1)%3 (0 welded leg, 1 aux, 2 ankle)
MZ_HD int body_class(int b) { return b == 0 ? 0 : 1 + (b - 1) % 3; }

// ------------------------------------------------------------------ K + I + V: per-item bodies of the kinematics, inertia,
// mass-matrix and bias-force phases (scheduled onto lanes by ant_forward)
template <int NB>
MZ_HD void kin_item(const AntDev& K, AntScratchT<NB>& s, int l) {
  constexpr int NH = AntDims<NB>::NH; (void)NH;

    if (l == 4) {
      if constexpr (NB == 0) {
        // Plain ant: no torso-level broad phase.  Its lane runs different code than the four leg lanes, i.e. serially with
        // them (1.2 k cycles per evaluation), while the per-geom test it used to save — two grid rows read at once, leave if
        // none of the <= 2 x 2 cells under the geom is a wall (geom_contacts) — costs less than that for all 13 geoms.
        s.nearwall = 1;
        s.con_over = 0;
        return;
      }
      // Wall broad phase for the whole robot (runs beside the four leg lanes): every robot geom lies within
      // ANT_REACH of the torso origin (torso sphere 0.25; leg chain 0.2*sqrt2*2 + 0.4*sqrt2 + capsule radius 0.08
      // + margin 0.01 < 1.25), so when no BLOCK cell comes that close in xy no geom can touch a wall.
      const MazeDev& z = K.maze;
      const float reach = 1.25f;
      float x = s.qpos[0] + z.tx, y = s.qpos[1] + z.ty, inv = 1.0f / z.scale;
      int jc = (int)floorf(x * inv + 0.5f), ic = (int)floorf(y * inv + 0.5f);
      float fx = x - jc * z.scale, fy = y - ic * z.scale;  // offset from the centre of the torso's own cell
      bool inside = ic >= 0 && ic < z.rows && jc >= 0 && jc < z.cols;
      // torso inside a wall cell / off the grid: keep testing; cells smaller than the reach: more than one ring could matter
      int near = (!inside || z.scale < reach) ? 1 : (int)((maze_row_lds(s, ic) >> jc) & 1u);
      // distances to the borders shared with the neighbouring rows / columns (0 for the own row / column)
      const float dlo_y = z.half_xy + fy, dhi_y = z.half_xy - fy, dlo_x = z.half_xy + fx, dhi_x = z.half_xy - fx, r2 = reach * reach;
#pragma unroll
      for (int di = -1; di <= 1; di++) {
        const uint32_t row = (ic + di < z.rows) ? maze_row_lds(s, ic + di) : 0u;  // rows beyond the grid hold no BLOCK cell
        const float dy = di == 0 ? 0.f : (di < 0 ? dlo_y : dhi_y);
#pragma unroll
        for (int dj = -1; dj <= 1; dj++) {
          const int j2 = jc + dj;
          const bool blk = j2 >= 0 && j2 < z.cols && ((row >> (j2 & 31)) & 1u);
          const float dx = dj == 0 ? 0.f : (dj < 0 ? dlo_x : dhi_x);
          if (blk && dx * dx + dy * dy < r2) near = 1;
        }
      }
      s.nearwall = near;
      s.con_over = 0;
      return;
    }
    if constexpr (AntDims<NB>::BALL) {
      if (l == 5) {  // object ball: pose from qpos[15:22] (quaternion normalised as mj_kinematics does), relative to the torso origin
        float R[9];
        quat_to_matf(R, s.qpos + 18);
        for (int k = 0; k < 9; k++) s.bR[k] = R[k];
        s.bx[0] = (s.qpos[15] - s.qpos[0]) + (s.qlo[2] - s.qlo[0]); s.bx[1] = (s.qpos[16] - s.qpos[1]) + (s.qlo[3] - s.qlo[1]); s.bx[2] = (s.qpos[17] - s.qpos[2]) + s.qlo[4];
        for (int k = 0; k < 3; k++) s.bc[k] = s.bx[k] + R[3 * k + 2] * K.ball_h;  // sphere centre = origin + R (0, 0, h)
        return;
      }
    }
    float R0[9];
    quat_to_matf(R0, s.qpos + 3);
    if (l == 0) {
      for (int k = 0; k < 9; k++) s.R0[k] = R0[k];
      s.zw[0] = R0[2]; s.zw[1] = R0[5]; s.zw[2] = R0[8];
      s.cz = s.qpos[2];
    }
    const float isq2 = 0.70710678118654752f;
    float sx = K.sx[l], sy = K.sy[l];
    float u[3] = {sx * isq2, sy * isq2, 0.f}, off[3] = {sx * K.legoff, sy * K.legoff, 0.f};
    float qh = s.qpos[7 + 2 * l], qa = s.qpos[8 + 2 * l];
    float ch, sh, ca, sa;
    mz_sincosf(qh, &sh, &ch);
    mz_sincosf(qa, &sa, &ca);
    float t[3], v[3];
    // level 0: welded leg capsule, frame = torso frame
    mat_vecf(s.w[3 * l], R0, u);
    for (int k = 0; k < 3; k++) s.com[3 * l][k] = s.w[3 * l][k] * K.half_len[1];
    // level 1: aux body at R0*off, rotated about the body z axis by the hip angle
    mat_vecf(s.p1[l], R0, off);
    t[0] = ch * u[0] - sh * u[1]; t[1] = sh * u[0] + ch * u[1]; t[2] = 0.f;  // Rz(qh) u
    mat_vecf(s.w[3 * l + 1], R0, t);
    for (int k = 0; k < 3; k++) s.com[3 * l + 1][k] = s.p1[l][k] + s.w[3 * l + 1][k] * K.half_len[2];
    float zw[3] = {R0[2], R0[5], R0[8]};
    cross3f(s.Sh[l], s.p1[l], zw);  // linear velocity at c of a unit hip rotation: zw x (c - p1)
    // level 2: ankle body at p1 + R1*off, rotated about the local ankle axis
    t[0] = ch * off[0] - sh * off[1]; t[1] = sh * off[0] + ch * off[1]; t[2] = 0.f;
    mat_vecf(v, R0, t);
    for (int k = 0; k < 3; k++) s.p2[l][k] = s.p1[l][k] + v[k];
    const float* a = K.ank_axis[l];
    float au = a[0] * u[0] + a[1] * u[1], axu[3];
    cross3f(axu, a, u);
    float ul[3];  // Rot(a, qa) u  (Rodrigues)
    for (int k = 0; k < 3; k++) ul[k] = u[k] * ca + axu[k] * sa + a[k] * au * (1.f - ca);
    t[0] = ch * ul[0] - sh * ul[1]; t[1] = sh * ul[0] + ch * ul[1]; t[2] = ul[2];
    mat_vecf(s.w[3 * l + 2], R0, t);
    for (int k = 0; k < 3; k++) s.com[3 * l + 2][k] = s.p2[l][k] + s.w[3 * l + 2][k] * K.half_len[3];
    t[0] = ch * a[0] - sh * a[1]; t[1] = sh * a[0] + ch * a[1]; t[2] = a[2];
    mat_vecf(s.Sa[l], R0, t);               // ankle axis (world)
    cross3f(s.Sa[l] + 3, s.p2[l], s.Sa[l]); // aw x (c - p2)
}

template <int NB>
MZ_HD void inertia_item(const AntDev& K, AntScratchT<NB>& s, int b) {

    int c = body_class(b);
    float m = K.mass[c], lat = K.ilat[c], dax = K.iax[c] - K.ilat[c];
    float r[3] = {0, 0, 0}, w[3] = {0, 0, 0};
    if (b > 0) { for (int k = 0; k < 3; k++) { r[k] = s.com[b - 1][k]; w[k] = s.w[b - 1][k]; } }
    float rr = dot3f(r, r);
    float* I = s.cin[b];
    I[0] = m; I[1] = m * r[0]; I[2] = m * r[1]; I[3] = m * r[2];
    I[4] = lat + dax * w[0] * w[0] + m * (rr - r[0] * r[0]);
    I[5] = lat + dax * w[1] * w[1] + m * (rr - r[1] * r[1]);
    I[6] = lat + dax * w[2] * w[2] + m * (rr - r[2] * r[2]);
    I[7] = dax * w[0] * w[1] - m * r[0] * r[1];
    I[8] = dax * w[0] * w[2] - m * r[0] * r[2];
    I[9] = dax * w[1] * w[2] - m * r[1] * r[2];
}

template <int NB>
MZ_HD void crb_leg_item(const AntDev& K, AntScratchT<NB>& s, int l) {
  constexpr int NH = AntDims<NB>::NH;

    float Ia[10], Ih[10], Sh[6], Fa[6], Fh[6];
    for (int k = 0; k < 10; k++) { Ia[k] = s.cin[3 + 3 * l][k]; Ih[k] = Ia[k] + s.cin[2 + 3 * l][k]; }
    for (int k = 0; k < 3; k++) { Sh[k] = s.zw[k]; Sh[3 + k] = s.Sh[l][k]; }
    inertia_mulf(Fa, Ia, s.Sa[l]);
    inertia_mulf(Fh, Ih, Sh);
    s.M.ll[l][0] = dot6f(Sh, Fh) + K.armature;
    s.M.ll[l][1] = dot6f(Sh, Fa);
    s.M.ll[l][2] = dot6f(s.Sa[l], Fa) + K.armature;
    for (int k = 0; k < 3; k++) {
      s.M.rl[l][0][k] = Fh[3 + k];  // root linear dof k: S = [0; e_k]
      s.M.rl[l][1][k] = Fa[3 + k];
      float ax[3] = {s.R0[k], s.R0[3 + k], s.R0[6 + k]};  // root angular dof k: S = [R0[:,k]; 0]
      s.M.rl[l][0][3 + k] = dot3f(ax, Fh);
      s.M.rl[l][1][3 + k] = dot3f(ax, Fa);
    }
    for (int k = 6; k < NH; k++) { s.M.rl[l][0][k] = 0.f; s.M.rl[l][1][k] = 0.f; }  // blocks are separate trees
}

template <int NB>
MZ_HD void iall_item(const AntDev& K, AntScratchT<NB>& s, int k) {
  // whole-body composite inertia about c
    float v = 0.f;
    for (int b = 0; b < ANT_NBODY; b++) v += s.cin[b][k];
    s.Iall[k] = v;
}

template <int NB>
MZ_HD void crb_root_item(const AntDev& K, AntScratchT<NB>& s, int e) {
  constexpr int NH = AntDims<NB>::NH;

    if (e >= 15) {  // rows of the movable bodies in the hub: separate trees, no coupling with the robot
      int q = e - 15, i = 6 + q / NH, j = q - (i - 6) * NH;
      float val = (i == j) ? K.block_mass : 0.f;
      if constexpr (AntDims<NB>::BALL) {
        // Free body whose centre of mass sits c = (0, 0, h) above the frame origin (body frame); dofs: linear velocity of
        // the origin (world frame), angular velocity (body frame).  T = m/2 |v + R (w x c)|^2 + I/2 |w|^2  =>
        //   M = [ m 1,  -m R [c]x ;  sym,  I 1 - m [c]x [c]x ],   -R [c]x = h [ -R[:,1], R[:,0], 0 ],   -[c]x[c]x = h^2 diag(1, 1, 0)
        const int a = i - 6, b = j - 6;
        const float mh = K.ball_mass * K.ball_h;
        val = 0.f;
        if (b >= 0) {
          const int lo = a < b ? a : b, hi = a < b ? b : a;
          if (hi < 3) val = a == b ? K.ball_mass : 0.f;
          else if (lo >= 3) val = a == b ? K.ball_inertia + (a < 5 ? mh * K.ball_h : 0.f) : 0.f;
          else val = hi == 3 ? -mh * s.bR[3 * lo + 1] : (hi == 4 ? mh * s.bR[3 * lo] : 0.f);
        }
      }
      s.M.rr[i][j] = val; s.M.rr[j][i] = val;
      return;
    }
    // root 6x6 block from the whole-body composite inertia: the nine angular-linear entries (e < 9) and the lower triangle of
    // the angular block (mirrored) — 15 items, one pass of a 16-lane group; the linear block is total mass x identity, constant,
    // preset once per step by ant_fill_tables
    int i, j;
    if (e < 9) { i = 3 + e / 3; j = e - 3 * (i - 3); }
    else { const int t = e - 9; i = t < 1 ? 3 : (t < 3 ? 4 : 5); j = 3 + t - ((i - 3) * (i - 2)) / 2; }
    float h[3] = {s.Iall[1], s.Iall[2], s.Iall[3]}, J[6];
    for (int k = 0; k < 6; k++) J[k] = s.Iall[4 + k];
    float val;
    {
      float ai[3] = {s.R0[i - 3], s.R0[3 + i - 3], s.R0[6 + i - 3]};
      if (j < 3) {  // angular i with linear j: ai . (h x e_j)
        float ej[3] = {j == 0 ? 1.f : 0.f, j == 1 ? 1.f : 0.f, j == 2 ? 1.f : 0.f}, hx[3];
        cross3f(hx, h, ej);
        val = dot3f(ai, hx);
      } else {
        float aj[3] = {s.R0[j - 3], s.R0[3 + j - 3], s.R0[6 + j - 3]}, Jv[3];
        Jv[0] = J[0] * aj[0] + J[3] * aj[1] + J[4] * aj[2];
        Jv[1] = J[3] * aj[0] + J[1] * aj[1] + J[5] * aj[2];
        Jv[2] = J[4] * aj[0] + J[5] * aj[1] + J[2] * aj[2];
        val = dot3f(ai, Jv);
      }
    }
    s.M.rr[i][j] = val; s.M.rr[j][i] = val;
}

// Recursive Newton-Euler, outward half, one body per lane: body b = 0 torso, 1 + 3l + k (k = 0 welded leg, 1 aux, 2 ankle).
// Every lane walks its own (at most two-joint) chain from the root, so no hand-off is needed:
//   f_b = I_b a_b + v_b x* (I_b v_b)   with gravity as base acceleration.
// The inward half (sums along the chains, projection on the joint axes) is bias_dof_item.
template <int NB>
MZ_HD void bias_body_item(const AntDev& K, AntScratchT<NB>& s, int b) {
    float v[6], a[6], ww[3];
    mat_vecf(ww, s.R0, s.qvel + 3);  // world angular velocity (root angular dofs are body-frame)
    for (int k = 0; k < 3; k++) { v[k] = ww[k]; v[3 + k] = s.qvel[k]; a[k] = 0.f; }
    cross3f(a + 3, s.qvel, ww);  // sum_k Sdot_k * qvel_k of the root rotation = [0; pdot x w]
    a[5] -= K.gz;                // gravity as base acceleration
    int l = b > 0 ? (b - 1) / 3 : 0, lev = b > 0 ? (b - 1) % 3 : 0;
    if (lev >= 1) {  // through the hip
      float Sh[6], sd[6], qdh = s.qvel[6 + 2 * l];
      for (int k = 0; k < 3; k++) { Sh[k] = s.zw[k]; Sh[3 + k] = s.Sh[l][k]; }
      motion_crossf(sd, v, Sh);
      for (int k = 0; k < 6; k++) { a[k] += sd[k] * qdh; v[k] += Sh[k] * qdh; }
    }
    if (lev >= 2) {  // through the ankle
      float sd[6], qda = s.qvel[7 + 2 * l];
      motion_crossf(sd, v, s.Sa[l]);
      for (int k = 0; k < 6; k++) { a[k] += sd[k] * qda; v[k] += s.Sa[l][k] * qda; }
    }
    float Ia[6], Iv[6], vf[6];
    inertia_mulf(Ia, s.cin[b], a);
    inertia_mulf(Iv, s.cin[b], v);
    force_crossf(vf, v, Iv);
    for (int k = 0; k < 6; k++) s.fbody[b][k] = Ia[k] + vf[k];
}

template <int NB>
MZ_HD void bias_dof_item(const AntDev& K, AntScratchT<NB>& s, int i) {

    float frc;
    if (i < 6) {
      float tot[3] = {0.f, 0.f, 0.f};
      int o = i < 3 ? 3 : 0;  // linear dofs pick the force part, angular dofs the torque part
      for (int b = 0; b < ANT_NBODY; b++) for (int k = 0; k < 3; k++) tot[k] += s.fbody[b][o + k];
      float bb;
      if (i < 3) bb = tot[i];
      else { float ax[3] = {s.R0[i - 3], s.R0[3 + i - 3], s.R0[6 + i - 3]}; bb = dot3f(ax, tot); }
      s.bias[i] = bb;
      frc = -bb;
    } else if (i < 14) {
      int l = (i - 6) >> 1, d = (i - 6) & 1;
      float f[6], bb;
      for (int k = 0; k < 6; k++) f[k] = s.fbody[3 + 3 * l][k];  // ankle body
      if (d == 1) bb = dot6f(s.Sa[l], f);
      else {  // the hip carries aux + ankle
        float Sh[6];
        for (int k = 0; k < 3; k++) { Sh[k] = s.zw[k]; Sh[3 + k] = s.Sh[l][k]; }
        for (int k = 0; k < 6; k++) f[k] += s.fbody[2 + 3 * l][k];
        bb = dot6f(Sh, f);
      }
      s.bias[i] = bb;
      frc = -K.damping * s.qvel[i] - bb + s.fact[i];
    } else if constexpr (AntDims<NB>::BALL) {
      // free ball, undamped and unactuated: force at the centre of mass f = m (w_w x (w_w x R c) - g), no torque about it (a
      // sphere), carried to the frame origin: bias = [ f ;  c x R^T f ]   (w_w = R w, the angular dofs are body-frame)
      const float* R = s.bR;
      const float w[3] = {s.qvel[17], s.qvel[18], s.qvel[19]};
      float ww[3], rc[3], t[3], f[3];
      mat_vecf(ww, R, w);
      for (int k = 0; k < 3; k++) rc[k] = R[3 * k + 2] * K.ball_h;
      cross3f(t, ww, rc);
      cross3f(f, ww, t);
      for (int k = 0; k < 3; k++) f[k] *= K.ball_mass;
      f[2] -= K.ball_mass * K.gz;
      const int k = i - 14;
      float bb;
      if (k < 3) bb = f[k];
      else {
        const float lx = R[0] * f[0] + R[3] * f[1] + R[6] * f[2], ly = R[1] * f[0] + R[4] * f[1] + R[7] * f[2];  // R^T f
        bb = k == 3 ? -K.ball_h * ly : (k == 4 ? K.ball_h * lx : 0.f);
      }
      s.bias[i] = bb;
      frc = -bb;
    } else {  // block slides: undamped, unactuated (maze_env.py:600-648); gravity acts on a z slide (falling blocks)
      const bool zslide = K.block_axis[(i - 14) % AntDims<NB>::BD] == 2;
      s.bias[i] = zslide ? -K.block_mass * K.gz : 0.f;
      frc = -s.bias[i];
    }
    s.qfs[i] = frc;
}




// ------------------------------------------------------------------ arrow linear algebra
// y_i = (A x)_i for one dof in MuJoCo order (called inside an MZ_FOR over dofs)
template <int NH>
MZ_HD float arrow_row_mul(const Arrow<NH>& A, const float* x, int i) {
  float v = 0.f;
  if (i < 6 || i >= 14) {
    int h = i < 6 ? i : i - 8;
    for (int k = 0; k < NH; k++) v += A.rr[h][k] * x[hub2dof(k)];
    for (int l = 0; l < 4; l++) v += A.rl[l][0][h] * x[6 + 2 * l] + A.rl[l][1][h] * x[7 + 2 * l];
  } else {
    int l = (i - 6) >> 1, d = (i - 6) & 1;
    for (int k = 0; k < NH; k++) v += A.rl[l][d][k] * x[hub2dof(k)];
    v += d == 0 ? A.ll[l][0] * x[6 + 2 * l] + A.ll[l][1] * x[7 + 2 * l] : A.ll[l][1] * x[6 + 2 * l] + A.ll[l][2] * x[7 + 2 * l];
  }
  return v;
}

// ---- the four phases of the fused arrow factor + solve, as per-item functions (so that they can share a phase
// ---- with unrelated work on other lanes)
template <int NH>
MZ_HD void factor_leg_item(const Arrow<NH>& A, ArrowFactor<NH>& F, int l) {
  float hh = A.ll[l][0], ha = A.ll[l][1], aa = A.ll[l][2];
  float idet = 1.0f / (hh * aa - ha * ha);
  float ihh = aa * idet, iha = -ha * idet, iaa = hh * idet;
  F.inv[l][0] = ihh; F.inv[l][1] = iha; F.inv[l][2] = iaa;
  for (int k = 0; k < NH; k++) {
    F.T[l][0][k] = ihh * A.rl[l][0][k] + iha * A.rl[l][1][k];
    F.T[l][1][k] = iha * A.rl[l][0][k] + iaa * A.rl[l][1][k];
  }
}
template <int NH>
MZ_HD void factor_schur_item(const Arrow<NH>& A, ArrowFactor<NH>& F, const float* g, int e) {
  constexpr int NTRI = NH * (NH + 1) / 2;
  if (e < NTRI) {
    int i = e < 1 ? 0 : e < 3 ? 1 : e < 6 ? 2 : e < 10 ? 3 : e < 15 ? 4 : e < 21 ? 5 : e < 28 ? 6 : e < 36 ? 7 : e < 45 ? 8 : e < 55 ? 9 : e < 66 ? 10 : 11;
    static_assert(NH <= 12, "triangular index decode covers 12 hub dofs");
    int j = e - (i * (i + 1)) / 2;
    float v = A.rr[i][j];
    for (int l = 0; l < 4; l++) v -= A.rl[l][0][i] * F.T[l][0][j] + A.rl[l][1][i] * F.T[l][1][j];
    F.L[i][j] = v;
  } else {
    int k = e - NTRI;
    float r = g[hub2dof(k)];
    for (int l = 0; l < 4; l++) r -= F.T[l][0][k] * g[6 + 2 * l] + F.T[l][1][k] * g[7 + 2 * l];
    F.rhs[k] = r;
  }
}
template <int NH>
MZ_HD void factor_serial_item(ArrowFactor<NH>& F) {
  float L[NH][NH], y[NH], id[NH];
  for (int i = 0; i < NH; i++) for (int j = 0; j <= i; j++) L[i][j] = F.L[i][j];
  for (int i = 0; i < NH; i++) y[i] = F.rhs[i];
  for (int j = 0; j < NH; j++) {
    float d = L[j][j];
    for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
    float r = 1.0f / sqrtf(fmaxf(d, 1e-30f));
    id[j] = r;
    for (int i = j + 1; i < NH; i++) {
      float t = L[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t * r;
    }
  }
  for (int i = 0; i < NH; i++) { float t = y[i]; for (int k = 0; k < i; k++) t -= L[i][k] * y[k]; y[i] = t * id[i]; }
  for (int i = NH - 1; i >= 0; i--) { float t = y[i]; for (int k = i + 1; k < NH; k++) t -= L[k][i] * y[k]; y[i] = t * id[i]; }
  for (int i = 0; i < NH; i++) F.rhs[i] = y[i];
}
template <int NH>
MZ_HD float factor_back_item(const ArrowFactor<NH>& F, const float* g, float sign, int i) {
  float v;
  if (i < 6) v = F.rhs[i];
  else if (i >= 14) v = F.rhs[i - 8];
  else {
    int l = (i - 6) >> 1, d = (i - 6) & 1;
    float gh = g[6 + 2 * l], ga = g[7 + 2 * l];
    v = d == 0 ? F.inv[l][0] * gh + F.inv[l][1] * ga : F.inv[l][1] * gh + F.inv[l][2] * ga;
    for (int k = 0; k < NH; k++) v -= F.T[l][d][k] * F.rhs[k];
  }
  return sign * v;
}

// Fused factor + solve of an arrow system A x = sign * g (vectors in MuJoCo dof order):
//   phase 1 (4 leg lanes)  2x2 inverses and T = inv * rl
//   phase 2                NH(NH+1)/2 Schur-complement entries + NH reduced right-hand sides
//   phase 3 (1 lane)       NH x NH Cholesky and both substitutions, entirely in registers
//   phase 4 (NV lanes)     back-substitution of the leg dofs
template <int NH, int NV, class C>
MZ_HD void arrow_factor_solve(const C& cx, const Arrow<NH>& A, ArrowFactor<NH>& F, const float* g, float* x, float sign) {
  MZ_FOR(l, 4) factor_leg_item<NH>(A, F, l);
  cx.sync();
  MZ_FOR(e, NH * (NH + 1) / 2 + NH) factor_schur_item<NH>(A, F, g, e);
  cx.sync();
  MZ_FOR(one, 1) factor_serial_item<NH>(F);
  cx.sync();
  MZ_FOR(i, NV) x[i] = factor_back_item<NH>(F, g, sign, i);
  cx.sync();
}


// ------------------------------------------------------------------ C + J: collision and constraint rows
// kind: 0 robot geom vs floor, 1 robot geom vs wall, 2 robot geom vs movable block, 3 block vs floor, 4 block vs wall,
// 5 block `other` (geom1) vs block `blk` (geom2)
struct ContactGeo { float dist, pos[3], n[3], hint[3]; int kind, blk, other; };

MZ_HD void make_tangents(const float* n, const float* hint, float* t1, float* t2) {
  float y[3] = {hint[0], hint[1], hint[2]};
  if (dot3f(y, y) < 0.25f) { y[0] = 0.f; y[1] = (n[1] < 0.5f && n[1] > -0.5f) ? 1.f : 0.f; y[2] = 1.f - y[1]; }
  float d = dot3f(n, y);
  for (int k = 0; k < 3; k++) y[k] -= n[k] * d;
  float nn = sqrtf(dot3f(y, y));
  if (nn < 1e-10f) {
    y[0] = 0.f; y[1] = (n[1] < 0.5f && n[1] > -0.5f) ? 1.f : 0.f; y[2] = 1.f - y[1];
    d = dot3f(n, y);
    for (int k = 0; k < 3; k++) y[k] -= n[k] * d;
    nn = sqrtf(dot3f(y, y));
  }
  for (int k = 0; k < 3; k++) t1[k] = y[k] / nn;
  cross3f(t2, n, t1);
}

// sphere (centre c in box-centred coordinates, radius r) vs axis-aligned box of half sizes bs.
// Normal points from the sphere to the box.  Returns true when dist <= margin.
MZ_HD bool sphere_aabb(const float* c, float r, const float* bs, float margin, float* dist, float* pos, float* n) {
  float q[3];
  bool inside = true;
  for (int k = 0; k < 3; k++) {
    q[k] = fminf(fmaxf(c[k], -bs[k]), bs[k]);
    if (q[k] != c[k]) inside = false;
  }
  float dd;
  if (!inside) {
    float v[3] = {q[0] - c[0], q[1] - c[1], q[2] - c[2]};
    dd = sqrtf(dot3f(v, v));
    if (dd - r > margin) return false;
    float id = 1.0f / dd;
    n[0] = v[0] * id; n[1] = v[1] * id; n[2] = v[2] * id;
    dd -= r;
  } else {
    int kb = 0; float best = 1e30f;
    for (int k = 0; k < 3; k++) { float e = bs[k] - fabsf(c[k]); if (e < best) { best = e; kb = k; } }
    n[0] = n[1] = n[2] = 0.f;
    n[kb] = c[kb] >= 0.f ? -1.f : 1.f;
    dd = -best - r;
  }
  *dist = dd;
  for (int k = 0; k < 3; k++) pos[k] = c[k] + n[k] * (r + 0.5f * dd);
  return dd <= margin;
}
MZ_HD float seg_box_df(const float* a, const float* dir, const float* bs, float t) {
  float g = 0.f;
  for (int k = 0; k < 3; k++) {
    float p = a[k] + t * dir[k];
    if (p > bs[k]) g += (p - bs[k]) * dir[k];
    else if (p < -bs[k]) g += (p + bs[k]) * dir[k];
  }
  return g;
}
MZ_HD float seg_box_t(const float* a, const float* b, const float* bs) {
  // f(t) = dist^2(segment point, box) is convex with a monotone piecewise-linear derivative g(t) whose kinks
  // are the slab crossings.  The minimiser lies between the largest candidate with g < 0 and the smallest
  // with g >= 0 — found without sorting (no dynamically indexed array => no scratch memory on the GPU).
  float dir[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  float tlo = 0.f, glo = seg_box_df(a, dir, bs, 0.f);
  if (glo >= 0.f) return 0.f;
  float thi = 1.f, ghi = seg_box_df(a, dir, bs, 1.f);
  if (ghi < 0.f) return 1.f;
  for (int k = 0; k < 3; k++)
    if (fabsf(dir[k]) > 1e-30f) {
      for (int sgn = 0; sgn < 2; sgn++) {
        float t = ((sgn ? -bs[k] : bs[k]) - a[k]) / dir[k];
        if (t > tlo && t < thi) {
          float g = seg_box_df(a, dir, bs, t);
          if (g < 0.f) { tlo = t; glo = g; } else { thi = t; ghi = g; }
        }
      }
    }
  return tlo - glo * (thi - tlo) / (ghi - glo);
}


MZ_HD float sel3f(const float* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }  // no dynamically indexed private array (scratch)

// Second support point of a capsule against a box — MuJoCo's mjc_CapsuleBox as restated in oracle/mzo_physics.c capsule_box
// (DESIGN.md section 5), in region form: the closest segment point t (in [-1, 1] along MuJoCo's half axis h = geom z * half
// length) and the box feature it faces are known from the closest-point computation instead of the search over two face
// candidates and twelve edges — the same feature wherever the closest point is unique.
//   type 0: a face (axis clface, -1 when the point is inside the box)   1: the interior of an edge (axis cledge)   2: a corner
//   corner: bit k set = the +k side of the box (for an edge: the bits of the two axes across it)
// Returns the offset of the second sphere along the segment (0: none).
MZ_HD float capsule_box_second(const float* cl, const float* h, const float* bs, float t, int type, int clface, int cledge, int corner,
                               float boxpos) {
  const int axisdir = (h[0] > 0.f ? 1 : 0) + (h[1] > 0.f ? 2 : 0) + (h[2] > 0.f ? 4 : 0);
  const float hx = fabsf(h[0]), hy = fabsf(h[1]), hz = fabsf(h[2]), habs[3] = {hx, hy, hz};
  const float n2 = hx * hx + hy * hy + hz * hz;
  if (type == 2) {
    int c1 = axisdir ^ corner;
    if (c1 == 0 || c1 == 7) return 0.f;  // pointing at / away from the corner
    float mul = 1.f;
    if (!(c1 == 1 || c1 == 2 || c1 == 4)) { mul = -1.f; c1 = 7 - c1; }
    const int ax = c1 == 1 ? 0 : (c1 == 2 ? 1 : 2), ax1 = ax == 2 ? 0 : ax + 1, ax2 = ax == 0 ? 2 : ax - 1;
    const float ha = sel3f(habs, ax);
    if (ha * ha > 0.5f * n2) return mul * fminf(1.f - mul * t, 2.f * sel3f(bs, ax) / ha);  // along the edge through the corner
    const float m = fminf(2.f * sel3f(bs, ax1) / sel3f(habs, ax1), 2.f * sel3f(bs, ax2) / sel3f(habs, ax2));
    return -mul * fminf(1.f + mul * t, m);                                               // back across the face
  }
  if (type == 1) {
    const int c1 = (axisdir ^ corner) & (7 - (1 << cledge));
    if (!(c1 == 1 || c1 == 2 || c1 == 4)) return 0.f;  // T configuration
    int ax1 = cledge == 2 ? 0 : cledge + 1, ax2 = cledge == 0 ? 2 : cledge - 1;
    if (sel3f(habs, ax1) > sel3f(habs, ax2)) { const int q = ax1; ax1 = ax2; ax2 = q; }  // ax1: normal of the flatter face, ax2: across it
    const float mul = (c1 & (1 << ax2)) ? 1.f : -1.f;
    float sp = 1.f - mul * t;
    sp = fminf(sp, 2.f * sel3f(bs, ax2) / sel3f(habs, ax2));
    const float e2 = (mul * sel3f(h, cledge) > 0.f) ? 1.f - boxpos : 1.f + boxpos;
    sp = fminf(sp, sel3f(bs, cledge) * e2 / sel3f(habs, cledge));
    return mul * sp;
  }
  // a face: towards the other end, pulled in so that it stays over the face
  const float travel = t <= 0.f ? 1.f - t : -1.f - t;  // (an end: -2 t)
  float frac = 1.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k == clface) continue;
    const float p0 = cl[k] + h[k] * t, v = h[k] * travel;
    if (v > 0.f && p0 + v > bs[k]) frac = fminf(frac, (bs[k] - p0) / v);
    if (v < 0.f && p0 + v < -bs[k]) frac = fminf(frac, (-bs[k] - p0) / v);
  }
  return travel * fmaxf(frac, 0.f);
}

// The feature search of mjc_CapsuleBox as the oracle restates it (two segment ends against the faces, then the twelve edges
// by clamped line-line distance; oracle/mzo_physics.c capsule_box), for the case the region form above cannot decide: the
// segment runs THROUGH the box (distance zero on a stretch — a state only a teleport or a violent impact produces), where
// "the closest point" is not unique and MuJoCo's answer is whatever its search order yields.  Same outputs as the region
// form.  Rarely executed; the closest-point reject in round_vs_box runs first.
MZ_HD void capsule_box_search(const float* cl, const float* h, float hl, const float* bs, float* t_out, int* type, int* clface, int* cledge,
                              int* corner, float* boxpos) {
  float best = 3.4e38f, bt = 0.f, bbp = 0.f;
  int cltype = -4, face = -1, ccorner = 0, cedge = 0;
  for (int i = -1; i <= 1; i += 2) {
    int nout = 0, f = -1;
    float dist = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float e = cl[k] + h[k] * (float)i;
      if (e < -bs[k]) { nout++; f = k; dist += (e + bs[k]) * (e + bs[k]); }
      else if (e > bs[k]) { nout++; f = k; dist += (e - bs[k]) * (e - bs[k]); }
    }
    if (nout > 1) continue;
    if (dist < best) { best = dist; bt = (float)i; cltype = -2 + i; face = f; }
  }
#pragma nounroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (i & (1 << j)) continue;
      float mid[3] = {(i & 1) ? bs[0] : -bs[0], (i & 2) ? bs[1] : -bs[1], (i & 4) ? bs[2] : -bs[2]};
      mid[j] = 0.f;
      const float d0 = mid[0] - cl[0], d1 = mid[1] - cl[1], d2 = mid[2] - cl[2];
      const float dj = j == 0 ? d0 : (j == 1 ? d1 : d2);
      const float u = -bs[j] * dj, v = h[0] * d0 + h[1] * d1 + h[2] * d2;
      const float ma = bs[j] * bs[j], mb = -bs[j] * h[j], mc = hl * hl, det = ma * mc - mb * mb;
      if (fabsf(det) < 1e-15f) continue;
      float x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
      int s1 = 1, s2 = 1;
      if (x1 > 1.f) { x1 = 1.f; s1 = 2; x2 = (v - mb) / mc; }
      else if (x1 < -1.f) { x1 = -1.f; s1 = 0; x2 = (v + mb) / mc; }
      if (x2 > 1.f) { x2 = 1.f; s2 = 2; x1 = (u - mb) / ma; if (x1 > 1.f) { x1 = 1.f; s1 = 2; } else if (x1 < -1.f) { x1 = -1.f; s1 = 0; } else s1 = 1; }
      else if (x2 < -1.f) { x2 = -1.f; s2 = 0; x1 = (u + mb) / ma; if (x1 > 1.f) { x1 = 1.f; s1 = 2; } else if (x1 < -1.f) { x1 = -1.f; s1 = 0; } else s1 = 1; }
      float f0 = d0 - h[0] * x2, f1 = d1 - h[1] * x2, f2 = d2 - h[2] * x2;
      if (j == 0) f0 += bs[0] * x1; else if (j == 1) f1 += bs[1] * x1; else f2 += bs[2] * x1;
      const float dist = f0 * f0 + f1 * f1 + f2 * f2;
      if (dist < best - 1e-15f) {
        best = dist; bt = x2; bbp = x1;
        cltype = 3 * s1 + s2; ccorner = i + (s1 == 2 ? (1 << j) : 0); cedge = j;
      }
    }
  *t_out = bt; *boxpos = bbp; *clface = face; *cledge = cedge; *corner = ccorner;
  *type = cltype < 0 ? 0 : (cltype / 3 == 1 ? 1 : 2);
}

// sphere / capsule (centre ctr, axis ax = the capsule's from -> to direction, half length hl, radius r; torso-relative) against an
// axis-aligned box (centre bc torso-relative, half sizes bs): up to two sphere-box contacts (mjc_CapsuleBox), normal from the
// robot geom to the box
#if defined(MZ_EXP_STAMPS) && defined(__HIPCC__)
__device__ unsigned mz_exp_general_runs;  // (experiment build) capsule-box tests that did not qualify for the face case
#endif
template <class Emit>
MZ_HD void round_vs_box(bool sphere, const float* ctr, const float* ax, float hl, float r, const float* bc, const float* bs,
                        float margin, int kind, int blk, Emit&& emit) {
  float cl[3] = {ctr[0] - bc[0], ctr[1] - bc[1], ctr[2] - bc[2]};  // geom centre in box coordinates
  float dist, pos[3], n[3];
  ContactGeo cg;
  cg.kind = kind; cg.blk = blk; cg.other = 0;
  cg.hint[0] = cg.hint[1] = cg.hint[2] = 0.f;
#ifdef MZ_EXP_SPHEREONLY  // timing experiment (wrong physics): a capsule is tested as a sphere at its centre
  sphere = true;
#endif
  if (sphere) {
    if (sphere_aabb(cl, r, bs, margin, &dist, pos, n) && dist < margin) {
      cg.dist = dist;
      for (int k = 0; k < 3; k++) { cg.n[k] = n[k]; cg.pos[k] = pos[k] + bc[k]; }
      emit(cg);
    }
    return;
  }
  // MuJoCo's half axis: the geom z axis of a fromto capsule points from `to` to `from`, i.e. along -ax
  const float hh[3] = {-ax[0] * hl, -ax[1] * hl, -ax[2] * hl};
  float em[3], ep[3];
  bool in_m = true, in_p = true;
  for (int k = 0; k < 3; k++) {
    em[k] = cl[k] - hh[k]; ep[k] = cl[k] + hh[k];
    in_m = in_m && fabsf(em[k]) <= bs[k]; in_p = in_p && fabsf(ep[k]) <= bs[k];
  }
#ifndef MZ_EXP_NOFACEPATH
  // The FACE case, written out (round 5): both ends of the axis segment beyond the same face of the box and inside its other two
  // slabs — a leg against a wall, which is nearly every wall test an ant ever causes (maze cells are metres wide, capsules
  // centimetres; a concave corner is two boxes, one face each).  Then everything the general code below works out is known: the closest
  // point of the segment is the end nearer to the face (the first end on a tie), the feature is that face, the second support point is
  // the other end (it stays over the face: frac = 1), and each of the two sphere tests sees a centre outside along that one axis —
  // depth = |coordinate| - half size - radius, normal = minus the axis.  Same contacts, same order; ~60 instructions instead of
  // ~1000.  Why it matters: per-wave start / end stamps of the product kernel (tools/exp_launch_stamps.py) put ONE wall test at 1.9 us
  // of its wave — 4 650 cycles — and the slowest wave of a launch at 24 of them per step: 46 of the 61 us it runs longer than the
  // mean wave.  (Round 4 tried a fast path for the closest-point search alone and lost 1 %: the rest of the test still ran.)
  {
    int nin = 0, kf = 0;
    bool beyond = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const bool ink = fabsf(em[k]) <= bs[k] && fabsf(ep[k]) <= bs[k];
      const bool outk = (em[k] > bs[k] && ep[k] > bs[k]) || (em[k] < -bs[k] && ep[k] < -bs[k]);
      nin += ink ? 1 : 0;
      if (outk) { kf = k; beyond = true; }
    }
    if (nin == 2 && beyond) {
      const float emk = sel3f(em, kf), epk = sel3f(ep, kf), bsk = sel3f(bs, kf);
      const bool p_first = fabsf(epk) < fabsf(emk);  // the closest end comes first: end -1 (em) unless end +1 is strictly nearer
      const float sgn = emk > 0.f ? 1.f : -1.f, d0 = fabsf(p_first ? epk : emk) - bsk;
      if (d0 * d0 > (r + margin) * (r + margin)) return;  // (the general code's reject on the closest point)
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        const bool usep = (pass == 0) == p_first;
        const float pk = usep ? epk : emk, dd = (fabsf(pk) - bsk) - r;
        if (dd < margin) {
          cg.dist = dd;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const float nk = k == kf ? -sgn : 0.f;
            cg.n[k] = nk;
            cg.pos[k] = (usep ? ep[k] : em[k]) + nk * (r + 0.5f * dd) + bc[k];
          }
          emit(cg);
        }
      }
      return;
    }
  }
#endif
#if defined(MZ_EXP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  atomicAdd(&mz_exp_general_runs, 1u);
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MZ_EXP_NOLAUNDER)
  // Keep the general case's arithmetic BEHIND its branch (round 5): the wall loop of the forward pass calls this once per candidate
  // cell, and everything below that does not depend on the cell — reciprocals of the axis, its products, two dozen comparison masks —
  // was hoisted in front of that loop by the compiler, i.e. executed by every geom that reached the loop at all, 98.6 % of which
  // leave through the face case above (ISA: ~300 instructions between the candidate scan and the loop's first pass; cycle stamps
  // around the region: 3 500 per narrow-phase run).  Values that pass through an empty asm are defined here, not before the loop.
  float h_[3] = {hh[0], hh[1], hh[2]};
#pragma unroll
  for (int k = 0; k < 3; k++) asm volatile("" : "+v"(cl[k]), "+v"(em[k]), "+v"(ep[k]), "+v"(h_[k]));
  const float* const h = h_;
#else
  const float* const h = hh;
#endif
  float t, boxpos = 0.f;
  int type = 0, clface = -1, cledge = 0, corner = 0;
  if (in_m || in_p) t = in_m ? -1.f : 1.f;  // an end inside the box (end -1 first): the face case without a face
  else {
    t = 2.f * seg_box_t(em, ep, bs) - 1.f;
    int nout = 0, lastout = 0, inaxis = 0;
    float pin = 0.f, d2c = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float pk = cl[k] + t * h[k];
      if (pk > bs[k]) { nout++; corner |= 1 << k; lastout = k; d2c += (pk - bs[k]) * (pk - bs[k]); }
      else if (pk < -bs[k]) { nout++; lastout = k; d2c += (pk + bs[k]) * (pk + bs[k]); }
      else { inaxis = k; pin = pk / bs[k]; }
    }
    // the closest point of the segment is farther than radius + margin from the box: no contact from this pair
    if (d2c > (r + margin) * (r + margin)) return;
    type = nout <= 1 ? 0 : (nout == 2 ? 1 : 2);
    clface = nout == 1 ? lastout : -1;
    cledge = inaxis; boxpos = pin;
    // the segment runs through the box with both ends outside: MuJoCo's answer is its search order's
#ifdef MZ_EXP_NOSEARCH  // timing experiment (wrong physics): never run the twelve-edge search
    if (false) {
#else
    if (nout == 0) {
#endif
      corner = 0;
      float clb[3] = {cl[0], cl[1], cl[2]}, hb[3] = {h[0], h[1], h[2]};
#if defined(__HIP_DEVICE_COMPILE__)
      // Opaque copies made INSIDE the rare branch: everything the search computes then depends on them, so none of it can be
      // hoisted in front of the branch.  (It was: the loop-invariant parts of the twelve-edge search — reciprocals of the 2 x 2
      // determinants, the clamped candidates — ran for every geom that came near a wall, a few hundred instructions and a dozen
      // spilled condition masks on the path of the slowest waves.)
      asm volatile("" : "+v"(clb[0]), "+v"(clb[1]), "+v"(clb[2]), "+v"(hb[0]), "+v"(hb[1]), "+v"(hb[2]));
#endif
      capsule_box_search(clb, hb, hl, bs, &t, &type, &clface, &cledge, &corner, &boxpos);
    }
  }
#ifdef MZ_EXP_NOSECOND  // timing experiment (wrong physics): no second support point
  const float second = 0.f;
#else
  const float second = capsule_box_second(cl, h, bs, t, type, clface, cledge, corner, boxpos);
#endif
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1 && !(fabsf(second) > 1e-12f)) break;
    const float tt = t + (pass ? second : 0.f);
    float p[3] = {cl[0] + tt * h[0], cl[1] + tt * h[1], cl[2] + tt * h[2]};
    if (sphere_aabb(p, r, bs, margin, &dist, pos, n) && dist < margin) {
      cg.dist = dist;
      for (int k = 0; k < 3; k++) { cg.n[k] = n[k]; cg.pos[k] = pos[k] + bc[k]; }
      emit(cg);
    }
  }
}

// Two axis-aligned boxes (movable blocks never rotate, maze cells are grid-aligned): MuJoCo's mjc_BoxBox as restated in
// oracle/mzo_physics.c box_box, specialised to parallel axes, in float64 on WORLD coordinates (grid-aligned boxes sit on exact
// ties — a block at its spawn position shares border lines with the diagonal wall cells, its z extent equals the walls' —
// which fp32 torso-relative coordinates would decide at random).  Separating axis = the face axis of least penetration
// (first of x, y, z on ties; an edge-edge axis never wins between parallel boxes), dist = -penetration; contact points = the
// corners of the intersection of the two facing faces (inclusive border tests); an intersection without area — boxes that share
// only a border line: a block at its spawn position and the diagonal wall cells — makes no contact [ASSUME-12].
// Box 1 = geom1: the normal points from box 1 to box 2.
#define MZ_BOX_MINOVERLAP 1e-6
struct AlignedBB { int ax, nu, nv; double dist, sg, pa, pu[2], pv[2]; };  // (pu / pv: read through selects, never by a run-time index — an indexed read put the struct into scratch memory)
MZ_HD bool aligned_box_box(const double* c1, const double* h1, const double* c2, const double* h2, double margin, AlignedBB& o) {
  double pen[3];
  for (int k = 0; k < 3; k++) { pen[k] = h1[k] + h2[k] - fabs(c2[k] - c1[k]); if (pen[k] < -margin) return false; }
  int ax = 0;
  if (pen[1] < pen[ax]) ax = 1;
  if (pen[2] < pen[ax]) ax = 2;
  const int u = ax == 2 ? 0 : ax + 1, v = ax == 0 ? 2 : ax - 1;
  double lo[3], hi[3];
  for (int k = 0; k < 3; k++) { lo[k] = fmax(c1[k] - h1[k], c2[k] - h2[k]); hi[k] = fmin(c1[k] + h1[k], c2[k] + h2[k]); }
  const double h1u = u == 0 ? h1[0] : (u == 1 ? h1[1] : h1[2]), h1v = v == 0 ? h1[0] : (v == 1 ? h1[1] : h1[2]);
  const double dtol = 1e-9 * (1.0 + h1u + h1v);  // coincident candidates (oracle: same)
  const double lou = u == 0 ? lo[0] : (u == 1 ? lo[1] : lo[2]), hiu = u == 0 ? hi[0] : (u == 1 ? hi[1] : hi[2]);
  const double lov = v == 0 ? lo[0] : (v == 1 ? lo[1] : lo[2]), hiv = v == 0 ? hi[0] : (v == 1 ? hi[1] : hi[2]);
  if (hiu - lou <= MZ_BOX_MINOVERLAP || hiv - lov <= MZ_BOX_MINOVERLAP) return false;  // the faces must overlap by a positive area (oracle: MZO_BOX_MINOVERLAP)
  const double c1a = ax == 0 ? c1[0] : (ax == 1 ? c1[1] : c1[2]), c2a = ax == 0 ? c2[0] : (ax == 1 ? c2[1] : c2[2]);
  const double h1a = ax == 0 ? h1[0] : (ax == 1 ? h1[1] : h1[2]), pa = ax == 0 ? pen[0] : (ax == 1 ? pen[1] : pen[2]);
  o.ax = ax; o.sg = c2a >= c1a ? 1.0 : -1.0; o.dist = -pa;
  o.pa = c1a + o.sg * (h1a + 0.5 * o.dist);
  o.nu = hiu - lou > dtol ? 2 : 1; o.nv = hiv - lov > dtol ? 2 : 1;
  o.pu[0] = lou; o.pu[1] = hiu; o.pv[0] = lov; o.pv[1] = hiv;
  return true;
}

// torso-relative centre of movable block k
template <int NB>
MZ_HD void block_center(const AntDev& K, const AntScratchCoreT<NB>& s, int k, float* bc) {
  using D = AntDims<NB>;
  float p0[3] = {0.f, 0.f, 0.f}, q[3] = {0.f, 0.f, 0.f}, ql[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < (D::NBLK ? D::NBLK : 1); j++)
    if (j == k && j < D::NBLK) {
      p0[0] = K.block_pos0[j][0]; p0[1] = K.block_pos0[j][1]; p0[2] = K.block_pos0[j][2];
#pragma unroll
      for (int a = 0; a < D::BD; a++) { q[a] = s.qpos[15 + D::BD * j + a]; ql[a] = s.qlo[2 + D::BD * j + a]; }
    }
  // slide a runs along coordinate axis K.block_axis[a] (increasing): (x, y), (y, z) / (x, z) for falling blocks, (x, y, z)
  float d[3] = {0.f, 0.f, 0.f}, dl[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < D::BD; a++)
#pragma unroll
    for (int c = 0; c < 3; c++) { d[c] += K.block_axis[a] == c ? q[a] : 0.f; dl[c] += K.block_axis[a] == c ? ql[a] : 0.f; }
  // hi parts first (the large coordinates cancel), the low-order parts after (AntScratchT::qlo)
  bc[0] = ((p0[0] - s.qpos[0]) + d[0]) + (dl[0] - s.qlo[0]); bc[1] = ((p0[1] - s.qpos[1]) + d[1]) + (dl[1] - s.qlo[1]); bc[2] = ((p0[2] - s.cz) + d[2]) + dl[2];
}

// Enumerate the contacts of enumerator e: e < NMOV -> part e % BSUB of movable block e / BSUB (floor corners | one grid cell |
// other blocks and slide limits) or the object ball; else robot geom (= body) b = e - NMOV (floor, walls, blocks / ball).  `emit` is called once per contact, in a fixed
// order, identical in the count and fill passes.
template <int NB, class Emit>
MZ_HD void geom_contacts(const AntDev& K, const AntScratchCoreT<NB>& s, int e, Emit&& emit) {
  const MazeDev& z = K.maze;
  float inv = 1.0f / z.scale;
  float bs[3] = {z.half_xy, z.half_xy, z.half_z};
  ContactGeo cg;
  if (e < AntDims<NB>::BSUB * AntDims<NB>::NBLK) {  // ---- movable block e / BSUB, part e % BSUB of its enumeration
    const int sub = e % AntDims<NB>::BSUB;
    e = e / AntDims<NB>::BSUB;
    float bc[3];
    block_center<NB>(K, s, e, bc);
    const float* hb = K.block_half;
    double bwd[3] = {0.0, 0.0, 0.0};  // world position of the block centre in float64: spawn position + its slides
    {
      using D = AntDims<NB>;
#pragma unroll
      for (int j = 0; j < (D::NBLK ? D::NBLK : 1); j++)
        if (j == e && j < D::NBLK) {
          for (int c = 0; c < 3; c++) bwd[c] = K.d_block_pos0[j][c];
#pragma unroll
          for (int a = 0; a < D::BD; a++)
            for (int c = 0; c < 3; c++) if (K.block_axis[a] == c) bwd[c] += (double)s.qpos[15 + D::BD * j + a] + (double)s.qlo[2 + D::BD * j + a];
        }
    }
    float bottom = (bc[2] + s.cz) - hb[2];  // absolute height of the bottom face
    if (sub == 0 && bottom < K.floor.margin)
      for (int ci = 0; ci < 4; ci++) {  // plane-box: the four bottom corners
        cg.kind = 3; cg.blk = e; cg.other = 0; cg.dist = bottom;
        cg.n[0] = 0.f; cg.n[1] = 0.f; cg.n[2] = 1.f;
        cg.hint[0] = cg.hint[1] = cg.hint[2] = 0.f;
        cg.pos[0] = bc[0] + ((ci & 1) ? hb[0] : -hb[0]); cg.pos[1] = bc[1] + ((ci & 2) ? hb[1] : -hb[1]);
        cg.pos[2] = 0.5f * bottom - s.cz;
        emit(cg);
      }
    float reach = sqrtf(hb[0] * hb[0] + hb[1] * hb[1] + hb[2] * hb[2]) + K.wall.margin;
    float gx = s.qpos[0] + bc[0], gy = s.qpos[1] + bc[1];
    int j0 = (int)floorf((gx - reach + z.tx) * inv + 0.5f), j1 = (int)floorf((gx + reach + z.tx) * inv + 0.5f);
    int i0 = (int)floorf((gy - reach + z.ty) * inv + 0.5f), i1 = (int)floorf((gy + reach + z.ty) * inv + 0.5f);
    // parts 1..9: cell (i0 + (sub - 1) / 3, j0 + (sub - 1) % 3) — a block is one cell wide (or less), its bounding square with the
    // margin spans at most three cells per axis (ant_dev_from_model checks the sizes); row-major like the loop it replaces
    const int ci_ = i0 + (sub - 1) / 3, cj_ = j0 + (sub - 1) % 3;
    for (int i = ci_; i <= ci_; i++)
      for (int j = cj_; j <= cj_; j++) {
        if (sub < 1 || sub > 9 || i > i1 || j > j1) continue;
        if (i < 0 || j < 0 || i >= z.rows || j >= z.cols) continue;
        // per cell: the platform of an elevated maze (z from 0 to 2 half_z), then the wall block standing on it
        for (int layer = 0; layer < 2; layer++) {
        if (!(((layer ? maze_row_lds(s, i) : plat_row_lds(s, i)) >> j) & 1u)) continue;
        // aligned box-box (aligned_box_box above): geom1 = wall / platform, geom2 = block; decided in float64 on world
        // coordinates, as the reference's arithmetic has them (AntDev: a falling block at maze scale 2 sits exactly `margin` away
        // from its neighbours' faces; a block at its spawn position shares border lines with the diagonal cells)
        const double cw[3] = {j * K.d_scale - K.d_tx, i * K.d_scale - K.d_ty, layer ? K.d_center_z : K.d_half_z};
        const double hw[3] = {K.d_half_xy, K.d_half_xy, K.d_half_z};
        AlignedBB bb;
        if (!aligned_box_box(cw, hw, bwd, K.d_block_half, K.d_wall_margin, bb)) continue;
        if (!(bb.dist < K.d_wall_margin)) continue;  // not active: no row (mj_instantiateContact)
        const int ax = bb.ax, u = ax == 2 ? 0 : ax + 1;
        const double org[3] = {(double)s.qpos[0] + (double)s.qlo[0], (double)s.qpos[1] + (double)s.qlo[1], (double)s.cz};  // torso origin: positions go back to torso-relative fp32
        for (int iu = 0; iu < bb.nu; iu++)
          for (int iv = 0; iv < bb.nv; iv++) {
            cg.kind = 4; cg.blk = e; cg.other = 0; cg.dist = (float)bb.dist;
            for (int k = 0; k < 3; k++) {
              const double pw = k == ax ? bb.pa : (k == u ? (iu ? bb.pu[1] : bb.pu[0]) : (iv ? bb.pv[1] : bb.pv[0]));
              cg.n[k] = k == ax ? (float)bb.sg : 0.f; cg.hint[k] = 0.f;
              cg.pos[k] = (float)(pw - (k == 0 ? org[0] : (k == 1 ? org[1] : org[2])));
            }
            emit(cg);
          }
        }
      }
    if (sub != AntDims<NB>::BSUB - 1) return;
    // lower-numbered movable blocks: aligned box-box, geom1 = block k, geom2 = block e
    for (int k = 0; k < e; k++) {
      double c1w[3] = {0.0, 0.0, 0.0};
      {
        using D = AntDims<NB>;
#pragma unroll
        for (int j = 0; j < (D::NBLK ? D::NBLK : 1); j++)
          if (j == k && j < D::NBLK) {
            for (int c = 0; c < 3; c++) c1w[c] = K.d_block_pos0[j][c];
#pragma unroll
            for (int a = 0; a < D::BD; a++)
              for (int c = 0; c < 3; c++) if (K.block_axis[a] == c) c1w[c] += (double)s.qpos[15 + D::BD * j + a] + (double)s.qlo[2 + D::BD * j + a];
          }
      }
      AlignedBB bb;
      if (!aligned_box_box(c1w, K.d_block_half, bwd, K.d_block_half, K.d_wall_margin, bb)) continue;
      if (!(bb.dist < K.d_wall_margin)) continue;
      const int ax = bb.ax, u = ax == 2 ? 0 : ax + 1;
      const double org[3] = {(double)s.qpos[0] + (double)s.qlo[0], (double)s.qpos[1] + (double)s.qlo[1], (double)s.cz};
      for (int iu = 0; iu < bb.nu; iu++)
        for (int iv = 0; iv < bb.nv; iv++) {
          cg.kind = 5; cg.blk = e; cg.other = k; cg.dist = (float)bb.dist;
          for (int q = 0; q < 3; q++) {
            const double pw = q == ax ? bb.pa : (q == u ? (iu ? bb.pu[1] : bb.pu[0]) : (iv ? bb.pv[1] : bb.pv[0]));
            cg.n[q] = q == ax ? (float)bb.sg : 0.f; cg.hint[q] = 0.f;
            cg.pos[q] = (float)(pw - (q == 0 ? org[0] : (q == 1 ? org[1] : org[2])));
          }
          emit(cg);
        }
    }
    // joint-limit rows of the block's own slides (falling blocks, maze_env.py:607-648): kind 6, `other` = slide index, n = the
    // row's Jacobian direction d dist / d q along the slide axis; a single frictionless row (see con_row_item)
    if (K.block_limited) {
      using D = AntDims<NB>;
      float qs[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < (D::NBLK ? D::NBLK : 1); j++)
        if (j == e && j < D::NBLK) {
#pragma unroll
          for (int a = 0; a < D::BD; a++) qs[a] = s.qpos[15 + D::BD * j + a];  // (limits sit within a cell of zero: the fp32 part is exact enough)
        }
#pragma unroll
      for (int a = 0; a < D::BD; a++) {
        const float q = qs[a], lo = K.block_lo[a], hi = K.block_hi[a];
        const int axis = K.block_axis[a];
        for (int side = -1; side <= 1; side += 2) {
          const float dist = side < 0 ? q - lo : hi - q;
          if (!(dist < K.blim_margin)) continue;
          cg.kind = 6; cg.blk = e; cg.other = a; cg.dist = dist;
          for (int k = 0; k < 3; k++) { cg.n[k] = k == axis ? -(float)side : 0.f; cg.hint[k] = 0.f; cg.pos[k] = 0.f; }
          emit(cg);
        }
      }
    }
    return;
  }
  if constexpr (AntDims<NB>::BALL) {
    if (e == AntDims<NB>::NBLK) {  // ---- object ball (kinds 7: floor -> ball, 8: ball -> wall box)
      const float rb = K.ball_r;
      const float dist = (s.cz + s.bc[2]) - rb;
      if (dist < K.ball_floor.margin) {  // plane-sphere: geom1 = floor, normal +z
        cg.dist = dist; cg.kind = 7; cg.blk = 0; cg.other = 0;
        cg.n[0] = 0.f; cg.n[1] = 0.f; cg.n[2] = 1.f;
        cg.pos[0] = s.bc[0]; cg.pos[1] = s.bc[1]; cg.pos[2] = s.bc[2] - (rb + 0.5f * dist);
        cg.hint[0] = cg.hint[1] = cg.hint[2] = 0.f;
        emit(cg);
      }
      // maze walls: the cells under the sphere's bounding square (sphere-box: geom1 = the sphere)
      const float reach = rb + K.ball_wall.margin;
      const float gx = s.qpos[0] + s.bc[0], gy = s.qpos[1] + s.bc[1], gz = s.cz + s.bc[2];
      if (gz - reach > z.center_z + z.half_z) return;
      const int j0 = (int)floorf((gx - reach + z.tx) * inv + 0.5f), j1 = (int)floorf((gx + reach + z.tx) * inv + 0.5f);
      const int i0 = (int)floorf((gy - reach + z.ty) * inv + 0.5f), i1 = (int)floorf((gy + reach + z.ty) * inv + 0.5f);
      const float zero[3] = {0.f, 0.f, 0.f};
      for (int i = i0; i <= i1; i++)
        for (int j = j0; j <= j1; j++) {
          if (i < 0 || j < 0 || i >= z.rows || j >= z.cols) continue;
          if (!((maze_row_lds(s, i) >> j) & 1u)) continue;
          float wc[3] = {((j * z.scale - z.tx) - s.qpos[0]) - s.qlo[0], ((i * z.scale - z.ty) - s.qpos[1]) - s.qlo[1], z.center_z - s.cz};
          round_vs_box(true, s.bc, zero, 0.f, rb, wc, bs, K.ball_wall.margin, 8, 0, emit);
        }
      return;
    }
  }
  // ---- robot geom
  int b = e - AntDims<NB>::NMOV;
  int c = body_class(b);
  float r = K.radius[c], hl = K.half_len[c];
  float ctr[3] = {0, 0, 0}, ax[3] = {0, 0, 0};
  if (b > 0) for (int k = 0; k < 3; k++) { ctr[k] = s.com[b - 1][k]; ax[k] = s.w[b - 1][k]; }
  // floor plane z = 0, normal +z; capsule ends in MuJoCo's geom-frame order [ASSUME-5]: the geom z axis of a
  // fromto capsule points from `to` to `from`, i.e. along -w, so "+axis" is the end at the body origin
  int nend = b == 0 ? 1 : 2;
  for (int k2 = 0; k2 < nend; k2++) {
    float sg = k2 == 0 ? -1.f : 1.f, p[3];
    for (int k = 0; k < 3; k++) p[k] = ctr[k] + sg * ax[k] * hl;
    float dist = (s.cz + p[2]) - r;
    if (dist < K.floor.margin) {
      cg.dist = dist; cg.kind = 0; cg.blk = 0; cg.other = 0;
      cg.n[0] = 0.f; cg.n[1] = 0.f; cg.n[2] = 1.f;
      cg.pos[0] = p[0]; cg.pos[1] = p[1]; cg.pos[2] = p[2] - (r + 0.5f * dist);
      for (int k = 0; k < 3; k++) cg.hint[k] = b == 0 ? 0.f : ax[k];
      emit(cg);
    }
  }
  // movable blocks
  for (int k = 0; k < AntDims<NB>::NBLK; k++) {
    float bc[3];
    block_center<NB>(K, s, k, bc);
    float d2 = 0.f;
    for (int q = 0; q < 3; q++) { float dd = fmaxf(fabsf(ctr[q] - bc[q]) - K.block_half[q], 0.f); d2 += dd * dd; }
    float reachb = r + hl + K.wall.margin;
    if (d2 < reachb * reachb) round_vs_box(b == 0, ctr, ax, hl, r, bc, K.block_half, K.wall.margin, 2, k, emit);
  }
  if constexpr (AntDims<NB>::BALL) {
    // object ball.  MuJoCo orders a pair by geom type, then by geom id: torso sphere (geom1) -> ball sphere, kind 9; ball sphere
    // (geom1) -> leg capsule, kind 10 (mjc_SphereCapsule: the point of the capsule's axis segment nearest to the ball, then
    // sphere-sphere).  The normal points from geom1 to geom2.
    float x = 0.f, pt[3], dv[3];
    for (int k = 0; k < 3; k++) x += ax[k] * (s.bc[k] - ctr[k]);
    x = fminf(fmaxf(x, -hl), hl);
    for (int k = 0; k < 3; k++) { pt[k] = ctr[k] + ax[k] * x; dv[k] = b == 0 ? s.bc[k] - pt[k] : pt[k] - s.bc[k]; }
    const float cd = sqrtf(dot3f(dv, dv)), r1 = b == 0 ? r : K.ball_r, dist = cd - r - K.ball_r;
    if (!(dist > K.ball_robot.margin)) {
      const float icd = cd < 1e-14f ? 0.f : 1.0f / cd;
      cg.dist = dist; cg.kind = b == 0 ? 9 : 10; cg.blk = 0; cg.other = 0;
      for (int k = 0; k < 3; k++) {
        cg.n[k] = cd < 1e-14f ? (k == 0 ? 1.f : 0.f) : dv[k] * icd;
        cg.pos[k] = (b == 0 ? pt[k] : s.bc[k]) + cg.n[k] * (r1 + 0.5f * dist);
        cg.hint[k] = 0.f;
      }
      emit(cg);
    }
  }
  // maze walls: cells under the bounding square of the geom (skipped when the torso-level broad phase is clear); in an
  // elevated maze the platforms under those cells as well — they are what the robot stands on
  // (the platform code exists in the movable-block instantiations only: every registered elevated maze has a falling block,
  // ant_dev_from_model refuses an elevated maze without one — the plain ant's kernel stays free of it)
  const bool elevated = NB > 0 && z.elevated;
  if (!s.nearwall && !elevated) return;
  float reach = r + hl + K.wall.margin;
  float gx = s.qpos[0] + ctr[0], gy = s.qpos[1] + ctr[1], gz = s.cz + ctr[2];
  if (gz - reach > z.center_z + z.half_z) return;
  int j0 = (int)floorf((gx - reach + z.tx) * inv + 0.5f), j1 = (int)floorf((gx + reach + z.tx) * inv + 0.5f);
  int i0 = (int)floorf((gy - reach + z.ty) * inv + 0.5f), i1 = (int)floorf((gy + reach + z.ty) * inv + 0.5f);
  if constexpr (NB == 0) {
    // Most geoms of an ant that stands near a wall are still clear of it: both grid rows under the bounding square are read
    // at once (one LDS wait instead of one per cell) and the geom leaves when none of its <= 2 x 2 cells is a wall.
    if (i1 - i0 <= 1 && j1 - j0 <= 1) {
      const uint32_t rows2 = maze_row_lds(s, i0) | (i1 != i0 ? maze_row_lds(s, i1) : 0u);
      const uint32_t cols2 = ((j0 >= 0 && j0 < z.cols) ? 1u << j0 : 0u) | ((j1 != j0 && j1 >= 0 && j1 < z.cols) ? 1u << j1 : 0u);
      if (!(rows2 & cols2)) return;
    }
  }
  for (int i = i0; i <= i1; i++)
    for (int j = j0; j <= j1; j++) {
      if (i < 0 || j < 0 || i >= z.rows || j >= z.cols) continue;
      for (int layer = elevated ? 0 : 1; layer < 2; layer++) {
        if (layer == 1 && !s.nearwall) continue;
        if (!(((layer ? maze_row_lds(s, i) : plat_row_lds(s, i)) >> j) & 1u)) continue;
        const float cz1 = layer ? z.center_z : z.half_z;
        if (gz - reach > cz1 + z.half_z || gz + reach < cz1 - z.half_z) continue;
        // box centre relative to the torso origin, computed so that the large world coordinates cancel first
        float bc[3] = {((j * z.scale - z.tx) - s.qpos[0]) - s.qlo[0], ((i * z.scale - z.ty) - s.qpos[1]) - s.qlo[1], cz1 - s.cz};
        // no point of the geom is farther than hl from its centre: a centre at least hl + r + margin away from the box cannot
        // give a contact (dist < margin) — skips the segment-box minimisation for cells that only the bounding square touches
        float d2 = 0.f;
        for (int q = 0; q < 3; q++) { const float dd = fmaxf(fabsf(ctr[q] - bc[q]) - bs[q], 0.f); d2 += dd * dd; }
        if (d2 >= reach * reach) continue;
        round_vs_box(b == 0, ctr, ax, hl, r, bc, bs, K.wall.margin, 1, 0, emit);
      }
    }
}

// ---- single-pass enumeration (plain ant on the device).  The two-pass count / fill scheme below runs every narrow-phase test
// twice; measured, the second pass is 5 % of a wave's cycles and — being paid only by the envs that touch something — 14 % of
// a launch (it feeds the slowest waves).  Here a geom's lane keeps the first MZ_STAGE contacts it finds in a staging entry of
// its own (8 floats: position, normal, distance, kind code; the tangent hint of a capsule-floor contact is the capsule axis
// and is rebuilt from the geom) inside the not-yet-used cY block, the second phase only maps compact contact slots to staging
// entries (prefix sums over the counts).  A geom with more contacts than MZ_STAGE (a foot in a wall corner) flags the env,
// which then runs the two-pass fill: same contacts, same order, always.
#define MZ_NEWTON_STALL 2e-6f  // relative step below which a line-searched Newton step counts as no step (fp32: 6e-8 per operation)
#define MZ_STAGE_OF(NB) ((NB) == 0 ? 3 : 4)  // a block's cell enumerator finds up to 2 x 2 contacts
template <int NB>
MZ_HD float* con_stage(AntScratchCoreT<NB>& s, int entry) { return &s.cY[0][0][0] + 8 * entry; }
template <int NB>
MZ_HD const float* con_stage(const AntScratchCoreT<NB>& s, int entry) { return &s.cY[0][0][0] + 8 * entry; }

// MERGE (the movable block's own enumerators in the quad forward pass, ant_forward_rows.h): the contact points of one face pair —
// four floor corners, the up to four corners of a block / wall overlap rectangle — arrive one after the other with the same
// kind, normal and distance, and the block has no rotational dof: their constraint rows are IDENTICAL.  They are staged as one
// entry with a multiplicity (code + 2048 (mult - 1)), which block_rows_direct turns into mult times the row's weight D — the same
// cost function term for term, a third of the block's rows.  s.cnt[e] = entries | emitted contacts << 8.
template <int NB, bool MERGE = false>
MZ_HD void con_enum_item(const AntDev& K, AntScratchCoreT<NB>& s, int e) {
  constexpr int MZ_STAGE = MZ_STAGE_OF(NB);
  static_assert(8 * MZ_STAGE * AntDims<NB>::NGEOM <= 3 * AntDims<NB>::NC * AntDims<NB>::NCOL, "staging lives in the cY block");
  int n = 0, emitted = 0;
  float ld = 0.f, ln[3] = {0.f, 0.f, 0.f};
  int lk = -1, lmult = 0, lblk = -1, lother = -1;
  geom_contacts<NB>(K, s, e, [&](const ContactGeo& g) {
    emitted++;
    if constexpr (MERGE) {
      // (same block AND same partner body: two partner blocks touching one face at the same depth have different reaction rows)
      if (n > 0 && g.kind == lk && g.kind != 6 && g.blk == lblk && g.other == lother && g.dist == ld && g.n[0] == ln[0] && g.n[1] == ln[1] && g.n[2] == ln[2] && lmult < 8) {
        lmult++;
        if (n <= MZ_STAGE) con_stage<NB>(s, MZ_STAGE * e + n - 1)[7] = (float)(g.kind + 16 * g.blk + 128 * g.other + 2048 * (lmult - 1));
        return;
      }
      lk = g.kind; ld = g.dist; lmult = 1; lblk = g.blk; lother = g.other;
      for (int k = 0; k < 3; k++) ln[k] = g.n[k];
    }
    if (n < MZ_STAGE) {
      float* q = con_stage<NB>(s, MZ_STAGE * e + n);
      for (int k = 0; k < 3; k++) { q[k] = g.pos[k]; q[3 + k] = g.n[k]; }
      q[6] = g.dist; q[7] = (float)(g.kind + 16 * g.blk + 128 * g.other);
    }
    n++;
  });
  s.cnt[e] = MERGE ? (n | (emitted << 8)) : n;
  if (n > MZ_STAGE) s.con_over = 1;
}

// MERGED: the movable bodies' enumerators (e < NMOV) ran con_enum_item<NB, true> — cnt = merged entries | contacts emitted << 8
template <int NB, bool MERGED = false>
MZ_HD void con_map_item(const AntDev& K, AntScratchCoreT<NB>& s, int e) {
  using D = AntDims<NB>;
  constexpr int NC = D::NC, NG = D::NGEOM, MZ_STAGE = MZ_STAGE_OF(NB);
  auto cnt_of = [&](int g) { return (MERGED && g < D::NMOV) ? (s.cnt[g] & 255) : s.cnt[g]; };
  int off = 0;
  for (int g = 0; g < e; g++) off += cnt_of(g);
  if (e == NG - 1) {
    int tot = off + cnt_of(e);
    int nrep = 0;  // contact points folded into merged entries: what MuJoCo would count on top
    if constexpr (MERGED) { for (int g = 0; g < D::NMOV; g++) nrep += (s.cnt[g] >> 8) - (s.cnt[g] & 255); }
    if (tot > NC) { tot = NC; s.status |= MZ_STATUS_CONTACT_OVERFLOW; }
    s.ncon = tot;
    s.ncon_true = tot + nrep;
    s.cbeg[4] = tot;
  }
  const int b = e - D::NMOV;
  if (b == 0) s.nblkcon = off < NC ? off : NC;  // the torso's enumerator follows those of the movable bodies
  if (b > 0 && (b - 1) % 3 == 0) s.cbeg[(b - 1) / 3] = off < NC ? off : NC;
  if (s.con_over) return;  // the env re-enumerates with con_fill_item
  const int cls = b >= 0 ? body_class(b) : -1, leg = b > 0 ? (b - 1) / 3 : -1, n = cnt_of(e);
  for (int i = 0; i < n; i++) {
    const int slot = off + i;
    if (slot < NC) { s.csrc[slot] = MZ_STAGE * e + i; s.cleg[slot] = leg; s.ccls[slot] = cls; }
  }
}

// per-item bodies of the collision / constraint-row phases
template <int NB>
MZ_HD void con_count_item(const AntDev& K, AntScratchCoreT<NB>& s, int e) {

    int n = 0;
    geom_contacts<NB>(K, s, e, [&](const ContactGeo&) { n++; });
    s.cnt[e] = n;
}

template <int NB>
MZ_HD void con_fill_item(const AntDev& K, AntScratchCoreT<NB>& s, int e) {
  using D = AntDims<NB>;
  constexpr int NC = D::NC, NG = D::NGEOM;

    int off = 0;
    for (int g = 0; g < e; g++) off += s.cnt[g];
    if (e == NG - 1) {
      int tot = off + s.cnt[e];
      if (tot > NC) { tot = NC; s.status |= MZ_STATUS_CONTACT_OVERFLOW; }
      s.ncon = tot;
      s.cbeg[4] = tot;
    }
    int b = e - D::NMOV;
    if (b == 0) s.nblkcon = off < NC ? off : NC;
    if (b > 0 && (b - 1) % 3 == 0) s.cbeg[(b - 1) / 3] = off < NC ? off : NC;
    int cls = b >= 0 ? body_class(b) : -1, leg = b > 0 ? (b - 1) / 3 : -1, slot = off;
    if (s.cnt[e] == 0) return;  // nothing to store: skip the second enumeration
    // The slots [off, end) belong to this enumerator, whatever the second enumeration yields: the two passes are two
    // instantiations of geom_contacts, and under the relaxed floating-point flags of the Ant build the compiler is free to
    // round an intermediate differently in each — a capsule whose closest box feature sits on a face / edge border can then
    // emit its second support point in one pass only (soak, round 3: a slot left unwritten turned into NaN).  A contact
    // beyond the count is dropped, a counted one that does not come is replaced by an inert one (below).
    const int end = off + s.cnt[e];
    geom_contacts<NB>(K, s, e, [&](const ContactGeo& g) {
      if (slot >= NC || slot >= end) { slot++; return; }
      float* q = &s.cY[slot][0][0];
      for (int k = 0; k < 3; k++) { q[k] = g.pos[k]; q[3 + k] = g.n[k]; q[8 + k] = g.hint[k]; }
      q[6] = g.dist; q[7] = (float)(g.kind + 16 * g.blk + 128 * g.other);
      s.cleg[slot] = leg;
      s.ccls[slot] = cls;
      s.csrc[slot] = -1;
      slot++;
    });
    // inert contact: a separation of a kilometre — its rows are never active (r = J a - aref > 0 for any acceleration the
    // solver can produce), so cost, gradient and Hessian of the Newton problem do not see it
    for (; slot < end && slot < NC; slot++) {
      float* q = &s.cY[slot][0][0];
      for (int k = 0; k < 3; k++) { q[k] = 0.f; q[3 + k] = k == 2 ? 1.f : 0.f; q[8 + k] = 0.f; }
      const int kind = b >= 0 ? 1 : (D::BALL && e == D::NMOV - 1 ? 7 : 3), blk = (b >= 0 || D::NBLK == 0) ? 0 : (e / D::BSUB < D::NBLK ? e / D::BSUB : 0);
      q[6] = 1e3f; q[7] = (float)(kind + 16 * blk);
      s.cleg[slot] = leg;
      s.ccls[slot] = cls;
      s.csrc[slot] = -1;
    }
}

template <int NB>
MZ_HD void con_row_item(const AntDev& K, AntScratchT<NB>& s, int item) {
  using D = AntDims<NB>;
  constexpr int NH = D::NH;

    int c = item / 3, a = item - 3 * c;
    const int src = s.csrc[c];
    const float* q = src >= 0 ? con_stage<NB>(s, src) : &s.cY[c][0][0];
    float r[3] = {q[0], q[1], q[2]}, n[3] = {q[3], q[4], q[5]}, hint[3], dist = q[6];
    // code = kind + 16 blk + 128 other + 2048 (multiplicity - 1): a merged entry of a movable block's own enumerators stands for
    // `mult` contact points with identical rows (con_enum_item MERGE) — mult times the weight, the same cost function term for term
    int code = (int)q[7], kind = code & 15, blk = (code >> 4) & 7, other = (code >> 7) & 15;
    const float mult = (float)((code >> 11) + 1);
    if (src >= 0) {  // staged contact: the hint of a capsule-floor contact is the capsule's axis (geom_contacts), nothing else has one
      const int b = src / MZ_STAGE_OF(NB) - D::NMOV;
      for (int k = 0; k < 3; k++) hint[k] = (kind == 0 && b > 0) ? s.w[b > 0 ? b - 1 : 0][k] : 0.f;
    } else {
      for (int k = 0; k < 3; k++) hint[k] = q[8 + k];
    }
    if (kind == 6) {
      // joint-limit row of slide `other` of block `blk`: ONE frictionless row  r = J a - aref, cost D/2 min(0, r)^2.  It rides
      // the contact machinery as a pyramid whose tangential rows vanish: the four edge rows coincide (u0 +- 0), so
      // cD = D / 4 reproduces cost, gradient and curvature of the single row exactly.
      float J[D::NCOL];
      for (int k = 0; k < D::NCOL; k++) J[k] = 0.f;
      float sg = n[0] + n[1] + n[2], vel = 0.f;  // +-1 on the slide axis
#pragma unroll
      for (int k = 0; k < D::NBLK; k++)
#pragma unroll
        for (int sl = 0; sl < D::BD; sl++)
          if (k == blk && sl == other) { if (a == 0) J[6 + D::BD * k + sl] = sg; vel = sg * s.qvel[14 + D::BD * k + sl]; }
      float aref = 0.f;
      if (a == 0) {
        float omi, imp = impedance_pair(K.blim_solimp, fabsf(dist - K.blim_margin), &omi);
        float R = fmaxf(1e-15f, omi / imp * K.blim_w);
        s.cD[c] = 0.25f / R;
        aref = -K.blim_B * vel - K.blim_K * imp * (dist - K.blim_margin);
      }
      for (int k = 0; k < D::NCOL; k++) s.cJ[c][a][k] = J[k];
      s.caref[c][a] = aref;
      return;
    }
    // contact kinds: 0 floor -> robot geom, 1 robot geom -> wall box, 2 robot geom -> block, 3 floor -> block, 4 wall box -> block,
    // 5 block -> block, (6 block slide limit, above), 7 floor -> ball, 8 ball -> wall box, 9 torso -> ball, 10 ball -> leg capsule
    // the pair's parameters by VALUE: every candidate set is read with scalar loads and the nine numbers are selected — a reference
    // chosen by `kind` compiles to an address select followed by dependent vector-memory loads, per contact row and evaluation
    PairDev P = (kind == 0 || kind == 3) ? pair_copy(K.floor) : pair_copy(K.wall);
    if constexpr (D::BALL) {
      const PairDev bf = pair_copy(K.ball_floor), bw = pair_copy(K.ball_wall), br = pair_copy(K.ball_robot);
      P = kind == 7 ? bf : (kind == 8 ? bw : (kind >= 9 ? br : P));
    }
    int leg = s.cleg[c], cls = s.ccls[c];
    float t1[3], t2[3], f[3];
    make_tangents(n, hint, t1, t2);
    // sign with which the contact force f (on geom2) acts on the robot / on the movable block / on the ball
    float sr = (kind == 0 || kind == 10) ? 1.f : ((kind == 1 || kind == 2 || kind == 9) ? -1.f : 0.f), sb = (kind >= 2 && kind <= 5) ? 1.f : 0.f;
    const float sball = (kind == 7 || kind == 9) ? 1.f : ((kind == 8 || kind == 10) ? -1.f : 0.f);
    float sc = a == 0 ? 1.f : P.mu;
    for (int k = 0; k < 3; k++) f[k] = sc * (a == 0 ? n[k] : (a == 1 ? t1[k] : t2[k]));
    float m[3];
    cross3f(m, r, f);  // (axis x r) . f = axis . (r x f)
    float J[D::NCOL];
    for (int k = 0; k < 3; k++) { J[k] = sr * f[k]; J[3 + k] = sr * (s.R0[k] * m[0] + s.R0[3 + k] * m[1] + s.R0[6 + k] * m[2]); }
    for (int k = 6; k < NH; k++) J[k] = 0.f;
    float fb[3];  // force components along the block's slides
#pragma unroll
    for (int sl = 0; sl < D::BD; sl++) fb[sl] = K.block_axis[sl] == 0 ? f[0] : (K.block_axis[sl] == 1 ? f[1] : f[2]);
#pragma unroll
    for (int k = 0; k < D::NBLK; k++)
#pragma unroll
      for (int sl = 0; sl < D::BD; sl++) {
        if (k == blk && kind >= 2 && kind <= 5) J[6 + D::BD * k + sl] = sb * fb[sl];
        if (kind == 5 && k == other) J[6 + D::BD * k + sl] = -fb[sl];  // geom1 of a block-block pair
      }
    if constexpr (D::BALL) {  // the ball's six columns: linear (world) = f, angular (body frame) = R^T ((r - x_ball) x f)
      float rb[3] = {r[0] - s.bx[0], r[1] - s.bx[1], r[2] - s.bx[2]}, mb[3];
      cross3f(mb, rb, f);
      for (int k = 0; k < 3; k++) {
        J[6 + k] = sball * f[k];
        J[9 + k] = sball * (s.bR[k] * mb[0] + s.bR[3 + k] * mb[1] + s.bR[6 + k] * mb[2]);
      }
    }
    J[NH] = cls >= 2 ? sr * (dot3f(s.zw, m) + dot3f(s.Sh[leg < 0 ? 0 : leg], f)) : 0.f;
    J[NH + 1] = cls == 3 ? sr * (dot3f(s.Sa[leg < 0 ? 0 : leg], m) + dot3f(s.Sa[leg < 0 ? 0 : leg] + 3, f)) : 0.f;
    float vel = 0.f;
    for (int k = 0; k < NH; k++) vel += J[k] * s.qvel[hub2dof(k)];
    if (leg >= 0) vel += J[NH] * s.qvel[6 + 2 * leg] + J[NH + 1] * s.qvel[7 + 2 * leg];
    float aref = -P.B * vel;
    if (a == 0) {
      float omi, imp = impedance_pair(P.solimp, fabsf(dist - P.margin), &omi);
      float tran = (cls >= 0 ? K.bw_tran[cls] : 0.f) + ((kind >= 2 && kind <= 5) ? K.block_bw_tran : 0.f) + (kind == 5 ? K.block_bw_tran : 0.f) +
                   (sball != 0.f ? K.ball_bw_tran : 0.f);
      float R = fmaxf(1e-15f, omi / imp * (tran + P.mu * P.mu * tran));
      s.cD[c] = mult / (2.f * P.mu * P.mu * R);  // [ASSUME-3]
      aref -= P.K * imp * (dist - P.margin);
    }
    for (int k = 0; k < D::NCOL; k++) s.cJ[c][a][k] = J[k];
    s.caref[c][a] = aref;
}

template <int NB>
MZ_HD void limit_item(const AntDev& K, AntScratchT<NB>& s, int j) {

    int l = j >> 1;
    float q = s.qpos[7 + j], lo = (j & 1) ? K.ank_lo[l] : K.hip_lo, hi = (j & 1) ? K.ank_hi[l] : K.hip_hi;
    float sg = 0.f, pos = 0.f;
    if (q - lo < 0.f) { sg = 1.f; pos = q - lo; }
    else if (hi - q < 0.f) { sg = -1.f; pos = hi - q; }
    float Dl = 0.f, aref = 0.f;
    if (sg != 0.f) {
      float omi, imp = impedance_pair(K.lim_solimp, fabsf(pos), &omi);
      float R = fmaxf(1e-15f, omi / imp * ((j & 1) ? K.dofw_ank : K.dofw_hip));
      Dl = 1.0f / R;
      aref = -K.lim_B * (sg * s.qvel[6 + j]) - K.lim_K * imp * pos;
    }
    s.lsign[j] = sg; s.lD[j] = Dl; s.laref[j] = aref;
}


// ------------------------------------------------------------------ N: Newton solver
// contact c: u = J qacc - aref (3), pyramid rows r = (u0+u1, u0-u1, u0+u2, u0-u2), active where r < 0
MZ_HD float contact_eval(float D, const float* u, float* g, float* W) {
  float r0 = u[0] + u[1], r1 = u[0] - u[1], r2 = u[0] + u[2], r3 = u[0] - u[2];
  float a0 = r0 < 0.f ? 1.f : 0.f, a1 = r1 < 0.f ? 1.f : 0.f, a2 = r2 < 0.f ? 1.f : 0.f, a3 = r3 < 0.f ? 1.f : 0.f;
  if (g) { g[0] = D * (a0 * r0 + a1 * r1 + a2 * r2 + a3 * r3); g[1] = D * (a0 * r0 - a1 * r1); g[2] = D * (a2 * r2 - a3 * r3); }
  if (W) { W[0] = D * (a0 + a1 + a2 + a3); W[1] = D * (a0 - a1); W[2] = D * (a2 - a3); W[3] = D * (a0 + a1); W[4] = D * (a2 + a3); }
  return 0.5f * D * (a0 * r0 * r0 + a1 * r1 * r1 + a2 * r2 * r2 + a3 * r3 * r3);
}
template <int NB>
MZ_HD float contact_Jdot(const AntScratchT<NB>& s, int c, int a, const float* x) {
  constexpr int NH = AntDims<NB>::NH;
  const float* J = s.cJ[c][a];
  float v = 0.f;
  for (int k = 0; k < NH; k++) v += J[k] * x[hub2dof(k)];
  int leg = s.cleg[c];
  if (leg >= 0) v += J[NH] * x[6 + 2 * leg] + J[NH + 1] * x[7 + 2 * leg];
  return v;
}

template <int NB, class C>
MZ_HD void ant_solve(const C& cx, const AntDev& K, AntScratchT<NB>& s, bool compare) {
  using D = AntDims<NB>;
  constexpr int NH = D::NH, NV = D::NV, NCOL = D::NCOL;
  bool has = false;
  MZ_FOR(one, 1) { bool h = s.ncon > 0; for (int j = 0; j < 8; j++) h = h || s.lsign[j] != 0.f; s.red[0] = h ? 1.f : 0.f; }
  cx.sync();
  // Initial guess.  `compare` (first evaluation of a step): MuJoCo's rule — the better of the warm-start
  // acceleration and the unconstrained one, by cost (one fused pass; the smooth part vanishes at qacc_smooth).
  // Later evaluations start from the previous evaluation's solution shifted by the change of the
  // unconstrained acceleration (s.warm was prepared by the caller); the optimum is unique, so the start only
  // changes the iteration count.
  if (compare) {
    MZ_FOR(i, NV) s.grad[i] = s.warm[i] - s.qas[i];
    cx.sync();
    float cw = 0.f, cs = 0.f;
    MZ_FOR(i, NV) cw += 0.5f * arrow_row_mul<NH>(s.M, s.grad, i) * s.grad[i];
    MZ_FOR(c, s.ncon) {
      float uw[3], us[3];
      for (int a = 0; a < 3; a++) { uw[a] = contact_Jdot<NB>(s, c, a, s.warm) - s.caref[c][a]; us[a] = contact_Jdot<NB>(s, c, a, s.qas) - s.caref[c][a]; }
      cw += contact_eval(s.cD[c], uw, nullptr, nullptr);
      cs += contact_eval(s.cD[c], us, nullptr, nullptr);
    }
    MZ_FOR(j, 8) {
      if (s.lsign[j] != 0.f) {
        float jw = s.lsign[j] * s.warm[6 + j] - s.laref[j], js = s.lsign[j] * s.qas[6 + j] - s.laref[j];
        if (jw < 0.f) cw += 0.5f * s.lD[j] * jw * jw;
        if (js < 0.f) cs += 0.5f * s.lD[j] * js * js;
      }
    }
    cw = cx.gsum(cw); cs = cx.gsum(cs);
    cx.sync();
    MZ_FOR(i, NV) s.qacc[i] = (cw < cs) ? s.warm[i] : s.qas[i];
  } else {
    MZ_FOR(i, NV) s.qacc[i] = s.warm[i];
  }
  cx.sync();
  has = s.red[0] != 0.f;
  if (!has) { MZ_FOR(i, NV) s.qacc[i] = s.qas[i]; cx.sync(); }
  cx.tick(s, 4);
  bool done = !has;
  int it = 0;
  // An accepted unit step is exact in exact arithmetic only.  In the mazes with movable bodies every geom sits at solimp .995
  // (maze_env.py:108-112): H = M + J^T D J then has a condition number of 1e3 .. 1e4 and an fp32 factorisation leaves the step 1e-4 of
  // its size off — hinge accelerations of thousands of rad/s^2 next to torso entries of tens (round 5: the whole error tail of the
  // Push / Fall families; ant_newton_rows.h carries the one-block ant's cheap form of the cure).  Here: the first accepted unit step
  // of a solve is followed by ONE more iteration from the stepped point — the affine residuals follow the step as after a line
  // search, the gradient is re-evaluated, and unless it already sits at the round-off floor a second Newton step (the refinement)
  // is taken, whose acceptance ends the solve.
  constexpr bool REFINE = NB >= 2;
  bool refined = false;
  while (cx.any(!done) && it < K.max_iter) {
    // (a) one dot product per lane: rows of M (qacc - qas) = M qacc - qfrc_smooth | contact residuals u[c][a] | limit residuals.
    // Only the first iteration computes them from qacc; afterwards they follow the step: every one of them is affine in
    // qacc, and M search / J search are already at hand from the line search (see the update at the end of the loop).
    if (it == 0) {
      MZ_FOR(i, NV) s.Mx[i] = arrow_row_mul<NH>(s.M, s.qacc, i) - s.qfs[i];
      MZ_FOR_AT(e, 3 * s.ncon, NV) {
        int c = e / 3, a = e - 3 * c;
        s.cu[c][a] = contact_Jdot<NB>(s, c, a, s.qacc) - s.caref[c][a];
      }
      MZ_FOR_AT(j, 8, NV + 3 * s.ncon) {
        float jar = 0.f, act = 0.f;
        if (s.lsign[j] != 0.f) { jar = s.lsign[j] * s.qacc[6 + j] - s.laref[j]; act = jar < 0.f ? s.lD[j] : 0.f; }
        s.ljar[j] = jar; s.lact[j] = act;
      }
      cx.sync();
    }
    // (b) gradient rows (contacts of leg l are the slots [cbeg[l], cbeg[l+1])) | Y = W J, one (contact, column) per lane
    float gpart = 0.f, apart = 0.f;  // |grad|^2 and the squared magnitude of the terms that cancel in it
    MZ_FOR(i, NV) {
      float g = s.Mx[i], ga = fabsf(g);
      int c0 = 0, c1 = s.ncon, col = i < 6 ? i : i - 8;
      if (i >= 6 && i < 14) { int l = (i - 6) >> 1; c0 = s.cbeg[l]; c1 = s.cbeg[l + 1]; col = NH + ((i - 6) & 1); }
#pragma unroll 2
      for (int c = c0; c < c1; c++) {  // (two contacts' loads in flight together)
        float g3[3];
        contact_eval(s.cD[c], s.cu[c], g3, nullptr);
        float t = s.cJ[c][0][col] * g3[0] + s.cJ[c][1][col] * g3[1] + s.cJ[c][2][col] * g3[2];
        g += t; ga += fabsf(t);
      }
      if (i >= 6 && i < 14 && s.lsign[i - 6] != 0.f) { float t = s.lsign[i - 6] * s.lact[i - 6] * s.ljar[i - 6]; g += t; ga += fabsf(t); }
      s.grad[i] = g;
      gpart += g * g;
      apart += ga * ga;
    }
    MZ_FOR_AT(e, NCOL * s.ncon, NV) {
      int c = e / NCOL, k = e - NCOL * c;
      float W[5];
      contact_eval(s.cD[c], s.cu[c], nullptr, W);
      float n_ = s.cJ[c][0][k], p_ = s.cJ[c][1][k], q_ = s.cJ[c][2][k];
      s.cY[c][0][k] = W[0] * n_ + W[1] * p_ + W[2] * q_;
      s.cY[c][1][k] = W[1] * n_ + W[3] * p_;
      s.cY[c][2][k] = W[2] * n_ + W[4] * q_;
    }
    float gnorm = sqrtf(cx.gsum(gpart)), anorm = sqrtf(cx.gsum(apart));
    // converged: MuJoCo's scaled-gradient test, or the gradient is at the fp32 cancellation floor
    if (!done && (K.inv_scale * gnorm < K.tol || gnorm <= K.rtol * anorm)) done = true;
    if (!cx.any(!done)) { cx.sync(); cx.tick(s, 5); break; }
    cx.sync();
    MZ_FOR(e, D::NHESS) {  // Hessian: NH x NH hub (full square) + 8 NH hub-leg + 12 leg-leg arrow entries
      int ci, cj, c0 = 0, c1 = s.ncon;
      float* dst;
      float acc;
      if (e < NH * NH) { ci = e / NH; cj = e - NH * ci; dst = &s.H.rr[ci][cj]; acc = s.M.rr[ci][cj]; }
      else if (e < NH * NH + 8 * NH) {
        int q = e - NH * NH, l = q / (2 * NH), r = q - 2 * NH * l, d = r / NH;
        cj = r - NH * d; ci = NH + d;
        c0 = s.cbeg[l]; c1 = s.cbeg[l + 1];
        dst = &s.H.rl[l][d][cj]; acc = s.M.rl[l][d][cj];
      } else {
        int q = e - NH * NH - 8 * NH, l = q / 3, t = q - 3 * l;
        ci = NH + (t == 2 ? 1 : 0); cj = NH + (t >= 1 ? 1 : 0);
        c0 = s.cbeg[l]; c1 = s.cbeg[l + 1];
        dst = &s.H.ll[l][t]; acc = s.M.ll[l][t];
        if (t != 1) acc += s.lact[2 * l + (t == 2 ? 1 : 0)];
      }
      {  // (round 6) four contacts per round trip, into independent partial sums: rolled, with one accumulator, every contact of the
         // range cost the lane a full LDS latency — and a three-block maze holds thirty of them per hub entry
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int c = c0;
        for (; c + 4 <= c1; c += 4) {
          acc += s.cJ[c][0][ci] * s.cY[c][0][cj] + s.cJ[c][1][ci] * s.cY[c][1][cj] + s.cJ[c][2][ci] * s.cY[c][2][cj];
          a1 += s.cJ[c + 1][0][ci] * s.cY[c + 1][0][cj] + s.cJ[c + 1][1][ci] * s.cY[c + 1][1][cj] + s.cJ[c + 1][2][ci] * s.cY[c + 1][2][cj];
          a2 += s.cJ[c + 2][0][ci] * s.cY[c + 2][0][cj] + s.cJ[c + 2][1][ci] * s.cY[c + 2][1][cj] + s.cJ[c + 2][2][ci] * s.cY[c + 2][2][cj];
          a3 += s.cJ[c + 3][0][ci] * s.cY[c + 3][0][cj] + s.cJ[c + 3][1][ci] * s.cY[c + 3][1][cj] + s.cJ[c + 3][2][ci] * s.cY[c + 3][2][cj];
        }
        for (; c < c1; c++) acc += s.cJ[c][0][ci] * s.cY[c][0][cj] + s.cJ[c][1][ci] * s.cY[c][1][cj] + s.cJ[c][2][ci] * s.cY[c][2][cj];
        acc = (acc + a1) + (a2 + a3);
      }
      *dst = acc;
    }
    cx.sync();
    cx.tick(s, 5);
    // (c) Newton direction
    arrow_factor_solve<NH, NV>(cx, s.H, s.F, s.grad, s.search, -1.f);
    cx.tick(s, 6);
    // (d) exact line search on phi(alpha) = cost(qacc + alpha * search).  The Newton direction makes
    // alpha = 1 the exact minimiser whenever the active set at qacc + search equals the one H was built on:
    // test that first with one ballot; only otherwise find the root of the piecewise-linear phi'.
    MZ_FOR(e, 3 * s.ncon) { int c = e / 3, a = e - 3 * c; s.cjv[c][a] = contact_Jdot<NB>(s, c, a, s.search); }
    MZ_FOR_AT(j, 8, 3 * s.ncon) s.ljv[j] = s.lsign[j] * s.search[6 + j];
    cx.sync();
    bool changed = false;
    MZ_FOR(c, s.ncon) {
      float u0 = s.cu[c][0], u1 = s.cu[c][1], u2 = s.cu[c][2];
      float w0 = u0 + s.cjv[c][0], w1 = u1 + s.cjv[c][1], w2 = u2 + s.cjv[c][2];
      changed = changed || ((u0 + u1 < 0.f) != (w0 + w1 < 0.f)) || ((u0 - u1 < 0.f) != (w0 - w1 < 0.f)) ||
                ((u0 + u2 < 0.f) != (w0 + w2 < 0.f)) || ((u0 - u2 < 0.f) != (w0 - w2 < 0.f));
    }
    MZ_FOR(j, 8) {
      if (s.lsign[j] != 0.f) changed = changed || ((s.ljar[j] < 0.f) != (s.ljar[j] + s.ljv[j] < 0.f));
    }
    changed = cx.gany(changed);
    float alpha = 1.f, sn = 1.f, qn = 0.f;
    bool exact = !changed;
    const bool follow = changed || (REFINE && !refined);  // the affine residuals are needed again after this step
    if (follow && !changed) { MZ_FOR(i, NV) s.Ms[i] = arrow_row_mul<NH>(s.M, s.search, i); }
    if (changed) {
      float p1 = 0.f, p2 = 0.f;
      MZ_FOR(i, NV) { float ms = arrow_row_mul<NH>(s.M, s.search, i); s.Ms[i] = ms; p1 += s.search[i] * s.Mx[i]; p2 += s.search[i] * ms; }
      p1 = cx.gsum(p1); p2 = cx.gsum(p2);
      {
        float sp = 0.f, qp = 0.f;
        MZ_FOR(i, NV) { sp += s.search[i] * s.search[i]; qp += s.qacc[i] * s.qacc[i]; }
        sn = cx.gsum(sp); qn = cx.gsum(qp);
      }
      float lo = 0.f, hi = -1.f, prev_d2 = -1.f;  // phi'(0) < 0 (descent direction); hi < 0: no upper bracket yet
      const int ls_max = it < K.ls_fast_iters ? K.ls_fast : K.ls_iter;  // (the rule of ant_newton_rows.h: unit steps first, the exact search behind them)
      for (int ls = 0; ls < ls_max; ls++) {
        float d1 = 0.f, d2 = 0.f;
        MZ_FOR(c, s.ncon) {
          float Dc = s.cD[c];
          float v0 = s.cjv[c][0], v1 = s.cjv[c][1], v2 = s.cjv[c][2];
          float u0 = s.cu[c][0] + alpha * v0, u1 = s.cu[c][1] + alpha * v1, u2 = s.cu[c][2] + alpha * v2;
          float r, v;
          r = u0 + u1; v = v0 + v1; if (r < 0.f) { d1 += Dc * r * v; d2 += Dc * v * v; }
          r = u0 - u1; v = v0 - v1; if (r < 0.f) { d1 += Dc * r * v; d2 += Dc * v * v; }
          r = u0 + u2; v = v0 + v2; if (r < 0.f) { d1 += Dc * r * v; d2 += Dc * v * v; }
          r = u0 - u2; v = v0 - v2; if (r < 0.f) { d1 += Dc * r * v; d2 += Dc * v * v; }
        }
        MZ_FOR(j, 8) {
          if (s.lsign[j] != 0.f) { float r = s.ljar[j] + alpha * s.ljv[j]; if (r < 0.f) { d1 += s.lD[j] * r * s.ljv[j]; d2 += s.lD[j] * s.ljv[j] * s.ljv[j]; } }
        }
        d1 = cx.gsum(d1) + p1 + alpha * p2;
        d2 = cx.gsum(d2) + p2;
        if (d2 == prev_d2) break;  // same slope as at the previous iterate: same linear piece, alpha is its root
        prev_d2 = d2;
        if (d1 < 0.f) lo = alpha; else hi = alpha;
        float next = alpha - d1 / d2;                                           // Newton step on phi'
        if (hi >= 0.f && !(next > lo && next < hi)) next = 0.5f * (lo + hi);  // safeguard
        if (!(next > 0.f)) next = hi >= 0.f ? 0.5f * (lo + hi) : 0.f;
        if (fabsf(next - alpha) <= 1e-7f * fabsf(next)) { alpha = next; break; }
        alpha = next;
      }
    }
    if (done) alpha = 0.f;
    cx.sync();
    // step, and the affine quantities of phase (a) along with it (an env whose step was exact is done: its M search is
    // not computed and its residuals are not used again)
    if (alpha != 0.f) {  // group-uniform; a finished env touches nothing (0 * stale values could still poison qacc)
      MZ_FOR(i, NV) s.qacc[i] += alpha * s.search[i];
      if (follow) {
        MZ_FOR(i, NV) s.Mx[i] += alpha * s.Ms[i];
        MZ_FOR_AT(e, 3 * s.ncon, NV) { int c = e / 3, a = e - 3 * c; s.cu[c][a] += alpha * s.cjv[c][a]; }
        MZ_FOR_AT(j, 8, NV + 3 * s.ncon) {
          if (s.lsign[j] != 0.f) { float jar = s.ljar[j] + alpha * s.ljv[j]; s.ljar[j] = jar; s.lact[j] = jar < 0.f ? s.lD[j] : 0.f; }
        }
      }
    }
    cx.sync();
    // The full Newton step stayed inside one active set: the cost is exactly quadratic there, so the new
    // point is its minimiser — no verification pass needed.
    if (exact && K.trust_exact && (!REFINE || refined)) done = true;
    if (exact) refined = true;
    // stationary at fp32 resolution: a line-searched step that moves qacc by less than MZ_NEWTON_STALL of its norm.  Active-set
    // flips of rows whose residual is zero within round-off otherwise keep the iteration alive until the cap (soak, round 3:
    // such envs jittered by 3e-7 |qacc| per iteration for 50 iterations, 1e-7 from the oracle's answer all along).
    if (changed && alpha * alpha * sn <= MZ_NEWTON_STALL * MZ_NEWTON_STALL * qn) done = true;
    cx.tick(s, 7);
    it++;
  }
  MZ_FOR(one, 1) { s.iters = it; if (it >= K.max_iter && !done) s.status |= MZ_STATUS_SOLVER_MAXITER; s.prof[15] += (unsigned)it; }
  cx.sync();
  cx.tick(s, 8);
}

// ------------------------------------------------------------------ one forward-dynamics evaluation: qacc from (qpos, qvel, fact)
// `first`: first evaluation of an env.step (warm = MuJoCo's qacc_warmstart, compared by cost against
// qacc_smooth); otherwise s.warm holds the previous evaluation's solution and s.qas its qacc_smooth.
//
// Schedule: independent pieces of work share a phase on different lanes (MZ_FOR_AT(item, count, first lane)),
// so one evaluation needs 8 phase boundaries before the solver instead of 13:
//   P0  leg kinematics (4) | wall broad phase (1)
//   P1  body inertias (13) | contact count per geom (13 + NB)
//   P2  leg mass-matrix blocks (4) | composite inertia (10) | per-body bias forces (13) | contact geometry fill (13 + NB)
//   P3  hub mass-matrix entries (21 + ..) | bias / smooth force per dof (NV)
//   P4  2x2 leg inverses of M (4) | contact Jacobian rows (3 ncon) | joint-limit rows (8)
//   P5  Schur entries + reduced rhs    P6  hub Cholesky (1 lane)    P7  back-substitution -> qacc_smooth
// On the device at >= 16 lanes per env contacts are enumerated once (P1 con_enum_item stages them, P2 con_map_item assigns the
// slots).  The plain ant and the ant with one two-slide block do not come here at all at those widths: their whole evaluation is
// ant_forward_rows (ant_forward_rows.h) + the row solver of ant_newton_rows.h.
// (the quad layout's evaluation, ant_forward_rows.h: device builds only — declared for the host emulation's parser, as ant_mj_step_rows below)
template <int NB, class C, class S>
MZ_HD float ant_forward_rows(const C& cx, const AntDev& K, S& s, bool first);
template <int NB, class C>
MZ_HD void ant_forward(const C& cx, const AntDev& K, AntScratchT<NB>& s, bool first) {
  using D = AntDims<NB>;
  constexpr int NH = D::NH, NV = D::NV, NG = D::NGEOM, NROOT = 15 + (NH - 6) * NH;
  if constexpr (NB <= 1 && C::row_solver) {
    // the plain ant, and the ant with one two-slide block, on the device at >= 16 lanes per env: the whole evaluation in the registers
    // of the row's leg quads (ant_forward_rows.h) — no LDS hand-off before the contact records
    ant_forward_rows<NB>(cx, K, s, first);
    return;
  }
  cx.tick(s, 9);
  MZ_FOR(l, 5 + (D::BALL ? 1 : 0)) kin_item<NB>(K, s, l);
  cx.sync();
  cx.tick(s, 0);
  // single-pass contact enumeration (con_enum_item) on the device: the narrow phase runs ONCE per evaluation, so count and
  // geometry cannot disagree; only an env whose enumerator overflows its staging re-enumerates (con_fill_item)
  constexpr bool one_pass = C::row_solver;
  MZ_FOR_AT(b, ANT_NBODY, 0) inertia_item<NB>(K, s, b);
  // Round 5: the movable blocks' own enumerators merge the contact points of one face pair into one entry with a multiplicity, as the
  // one-block ant's quad path does (con_enum_item MERGE): blocks only translate, the points' rows are identical.  Two blocks in a
  // corner are 7 entries instead of 28 — fewer rows for the solver, and the 40 / 72 slots stop overflowing under an ant lying across
  // blocks and walls (AntMultiPush soak, round 4: 2 flagged envs per bench sweep; round 5 before this: 8 of 1024 in 2000 steps).
  constexpr bool merged = one_pass && D::NBLK > 0;
  if constexpr (one_pass) {
    MZ_FOR_AT(e, NG, ANT_NBODY) { if (merged && e < D::NMOV) con_enum_item<NB, true>(K, s, e); else con_enum_item<NB, false>(K, s, e); }
  } else { MZ_FOR_AT(e, NG, ANT_NBODY) con_count_item<NB>(K, s, e); }
  cx.sync();
  cx.tick(s, 1);
  MZ_FOR_AT(l, 4, 0) crb_leg_item<NB>(K, s, l);
  MZ_FOR_AT(k, 10, 4) iall_item<NB>(K, s, k);
  MZ_FOR_AT(b, ANT_NBODY, 14) bias_body_item<NB>(K, s, b);
  if constexpr (one_pass) { MZ_FOR_AT(e, NG, 14 + ANT_NBODY) con_map_item<NB, merged>(K, s, e); }
  else { MZ_FOR_AT(e, NG, 14 + ANT_NBODY) con_fill_item<NB>(K, s, e); }
  cx.sync();
  if constexpr (one_pass) {
    if (s.con_over) {  // some geom found more contacts than its staging holds: this env counts and fills the two-pass way (unmerged)
      if constexpr (merged) { MZ_FOR(e, NG) con_count_item<NB>(K, s, e); cx.sync(); }
      MZ_FOR(e, NG) con_fill_item<NB>(K, s, e);
      cx.sync();
      MZ_FOR(one, 1) s.ncon_true = s.ncon;
      cx.sync();
    }
  }
  cx.tick(s, 2);
  MZ_FOR_AT(e, NROOT, 0) crb_root_item<NB>(K, s, e);
  MZ_FOR_AT(i, NV, NROOT) bias_dof_item<NB>(K, s, i);
  if (!first) { MZ_FOR(i, NV) s.warm[i] -= s.qas[i]; }  // previous qacc_smooth: still intact until P7 (shifted start)
  cx.sync();
  cx.tick(s, 3);
  {
    // the lane-group formulation (two and more movable bodies, 8-lane groups, the host emulation)
    MZ_FOR_AT(l, 4, 0) factor_leg_item<NH>(s.M, s.F, l);
    MZ_FOR_AT(item, 3 * s.ncon, 4) con_row_item<NB>(K, s, item);
    MZ_FOR_AT(j, 8, 4 + 3 * s.ncon) limit_item<NB>(K, s, j);
    cx.sync();
    cx.tick(s, 11);
    MZ_FOR(e, NH * (NH + 1) / 2 + NH) factor_schur_item<NH>(s.M, s.F, s.qfs, e);
    cx.sync();
    MZ_FOR(one, 1) factor_serial_item<NH>(s.F);
    cx.sync();
    MZ_FOR(i, NV) {
      float v = factor_back_item<NH>(s.F, s.qfs, 1.f, i);
      s.qas[i] = v;
      if (!first) s.warm[i] += v;
    }
    cx.sync();
    cx.tick(s, 12);
    ant_solve<NB>(cx, K, s, first);
  }
}


// qpos <- integrate(qpos, vel, h): free joint on the manifold, hinges and block slides linear (mj_integratePos)
// quaternion of a free joint advanced by the body-frame angular velocity w over h (mju_quatIntegrate), normalised before and after
MZ_HD void quat_integratef(const float* base, const float* w, float h, float* out) {
  float n = sqrtf(dot3f(w, w));
  float q[4] = {base[0], base[1], base[2], base[3]};
  float qn = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; k++) q[k] *= qn;
  if (n > 1e-15f) {
    float ang = 0.5f * h * n, sn, c0;
    mz_sincosf(ang, &sn, &c0);
    sn /= n;
    float r[4] = {c0, w[0] * sn, w[1] * sn, w[2] * sn}, o[4];
    o[0] = q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3];
    o[1] = q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2];
    o[2] = q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1];
    o[3] = q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0];
    float on = 1.0f / sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    for (int k = 0; k < 4; k++) q[k] = o[k] * on;
  }
  for (int k = 0; k < 4; k++) out[k] = q[k];
}

// base + base_lo + h v in float64, split into the fp32 coordinate and its remainder
MZ_HD void mz_step_split(float base, float base_lo, float h, float v, float* hi, float* lo) {
  const double t = ((double)base + (double)base_lo) + (double)h * (double)v;
  const float fh = (float)t;
  *hi = fh; *lo = (float)(t - (double)fh);
}
template <int NB, class C>
MZ_HD void ant_integrate_pos(const C& cx, AntScratchCoreT<NB>& s, const float* base, const float* vel, float h) {
  using D = AntDims<NB>;
  MZ_FOR(i, 12 + (D::BALL ? 4 : D::BD * D::NBLK)) {
    if (i < 2) mz_step_split(base[i], s.x0lo[i], h, vel[i], &s.qpos[i], &s.qlo[i]);
    else if (i < 3) s.qpos[i] = base[i] + h * vel[i];
    else if (i == 3) {
      float w[3] = {vel[3], vel[4], vel[5]};
      quat_integratef(base + 3, w, h, s.qpos + 3);
    } else if (D::BALL && i >= 12) {  // the ball's free joint: qpos[15:18] += h v, quaternion qpos[18:22] on the manifold
      if (i < 15) mz_step_split(base[15 + (i - 12)], s.x0lo[2 + (i - 12)], h, vel[14 + (i - 12)], &s.qpos[15 + (i - 12)], &s.qlo[2 + (i - 12)]);
      else {
        float w[3] = {vel[17], vel[18], vel[19]};
        quat_integratef(base + 18, w, h, s.qpos + 18);
      }
    } else {
      int j = i - 4;  // hinges 0..7, then block slides
      if (j < 8) s.qpos[7 + j] = base[7 + j] + h * vel[6 + j];
      else mz_step_split(base[7 + j], s.x0lo[2 + (j - 8)], h, vel[6 + j], &s.qpos[7 + j], &s.qlo[2 + (j - 8)]);
    }
  }
}

// (the quad layout's step, ant_forward_rows.h: device builds only — declared here so that the host emulation, which never takes that
// branch, can parse the call with its explicit template arguments)
template <int NB, bool XREG, class C, class S>
MZ_HD void ant_mj_step_rows(const C& cx, const AntDev& K, S& s, bool first_frame);
// one mj_step with RK4.  State in s.qpos / s.qvel / s.warm, actuator forces in s.fact.
// S: AntScratchT<NB>, or AntScratchCoreT<NB> where the caller allocated no more (the quad layout's product kernels)
template <int NB, class C, class S>
MZ_HD void ant_mj_step(const C& cx, const AntDev& K, S& s, bool first_frame) {
  using D = AntDims<NB>;
  if constexpr (NB <= 1 && C::row_solver) {  // the quad layout: RK4 bookkeeping in the dof lanes' registers (ant_forward_rows.h)
    if (cx.mfma) ant_mj_step_rows<NB, false>(cx, K, s, first_frame);  // (a constant after inlining: one of the two survives)
    else ant_mj_step_rows<NB, true>(cx, K, s, first_frame);
    return;
  } else {
  const float h = K.h;
  MZ_FOR(i, D::NQ) s.x0q[i] = s.qpos[i];
  MZ_FOR(i, D::NLO) s.x0lo[i] = s.qlo[i];
  MZ_FOR(i, D::NV) { s.x0v[i] = s.qvel[i]; s.accv[i] = 0.f; s.accf[i] = 0.f; }
  cx.sync();
  for (int st = 0; st < 4; st++) {
    ant_forward<NB>(cx, K, s, first_frame && st == 0);
    const float bw = (st == 0 || st == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
    const float aw = st == 2 ? 1.0f : 0.5f;  // Butcher A: diag(1/2, 1/2, 1)
    // accumulate B-weighted sums, form the next stage state from this stage's (qvel, qacc)
    MZ_FOR(i, D::NV) {
      s.accv[i] += bw * s.qvel[i];
      s.accf[i] += bw * s.qacc[i];
      s.grad[i] = aw * s.qvel[i];   // dX velocity for the position update
      s.Mx[i] = s.x0v[i] + h * aw * s.qacc[i];
      // initial guess of the next stage's constraint solve = this stage's solution (MuJoCo re-uses the
      // previous step's; the optimum is unique, only the iteration count changes); after the 4th stage this
      // is exactly MuJoCo's qacc_warmstart
      s.warm[i] = s.qacc[i];
    }
    cx.sync();
    if (st < 3) {
      ant_integrate_pos<NB>(cx, s, s.x0q, s.grad, h);
      MZ_FOR(i, D::NV) s.qvel[i] = s.Mx[i];
      cx.sync();
    }
  }
  ant_integrate_pos<NB>(cx, s, s.x0q, s.accv, h);
  MZ_FOR(i, D::NV) s.qvel[i] = s.x0v[i] + h * s.accf[i];
  cx.sync();
  }
}

// ------------------------------------------------------------------ MazeTask reward / termination on the fp32 observation
// that is returned to the caller (obs[0:3] agent slot, obs[3:6] object slot).  Flags and goal index are the reference's
// float64 predicate (maze_task.py:43-44 `np.linalg.norm(obs[:dim] - pos) <= threshold`, :77-81 any goal, :403-407 first
// match) evaluated on float64(obs): differences and squares in fp64, summed in index order without contraction, compared
// with the squared-threshold bound of TaskDev (bit-exact whatever the build flags of the translation unit).
// `env`: the env slot whose row of TaskDev::env_goals holds its own goal positions (per-episode resampling); -1 or no table bound:
// the batch's shared goal table.
MZ_HD void task_eval_dev(const TaskDev& T, const float* obs, float* reward, int* term, int* goal_idx, int env = -1) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  // (two loads, not one pointer select: the shared table stays a scalar load of the constant block)
  const double* eg = (T.env_goals && env >= 0) ? T.env_goals + (size_t)env * (3 * MZ_MAX_GOAL) : nullptr;
  const double slot_a[3] = {(double)obs[0], (double)obs[1], (double)obs[2]}, slot_o[3] = {(double)obs[3], (double)obs[4], (double)obs[5]};
  int tm = 0, first = -1, first_t = -1;
  for (int g = 0; g < T.ngoal; g++) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < 3; k++)
      if (k < T.goal_dim[g]) {
        const double gk = eg ? eg[3 * g + k] : T.goal_pos[g][k];
        double e = (T.term_slot == MZ_SLOT_OBJECT ? slot_o[k] : slot_a[k]) - gk; a += e * e;
        double f = (T.reward_slot == MZ_SLOT_OBJECT ? slot_o[k] : slot_a[k]) - gk; b += f * f;
      }
    if (!tm && a <= T.thr_sq[g]) { tm = 1; first_t = g; }
    if (first < 0 && b <= T.thr_sq[g]) first = g;
  }
  double r = 0.0;
  if (T.reward_kind == MZ_REWARD_FIRST_MATCH) r = T.reward_binary ? (tm ? 1.0 : T.penalty) : (first >= 0 ? T.rscale[first] : T.penalty);
  else if (T.reward_kind == MZ_REWARD_NEG_DIST && T.ngoal > 0) {
    double a = 0.0;
    for (int k = 0; k < 3; k++)
      if (k < T.goal_dim[0]) { double e = (T.reward_slot == MZ_SLOT_OBJECT ? slot_o[k] : slot_a[k]) - (eg ? eg[k] : T.goal_pos[0][k]); a += e * e; }
    r = -sqrt(a) / T.task_scale;
  }
  // goal index: the goal that set the reward where the reward is a goal's (first match on the reward's slot, maze_task.py:403-407);
  // for the other reward kinds (zero, distance) the first goal that ends the episode (termination's slot, maze_task.py:77-81,599,653)
  *reward = (float)r; *term = tm; *goal_idx = T.reward_kind == MZ_REWARD_FIRST_MATCH ? first : first_t;
}

// coordinate c of movable block k's body origin (get_body_com, maze_env.py:364-368): spawn position + its slides
template <int NB>
MZ_HD float ant_block_coord(const AntDev& K, const AntScratchCoreT<NB>& s, int k, int c) {
  using D = AntDims<NB>;
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < (D::NBLK ? D::NBLK : 1); j++)
    if (j == k && j < D::NBLK) {
      v = K.block_pos0[j][c];
#pragma unroll
      for (int a = 0; a < D::BD; a++) v += K.block_axis[a] == c ? s.qpos[15 + D::BD * j + a] : 0.f;
    }
  return v;
}

// entries the maze adds to the robot's observation: body positions of the object ball / of the movable blocks, when observed
template <int NB>
MZ_HD int ant_obs_extra(const AntDev& K) {
  using D = AntDims<NB>;
  return D::BALL ? (K.observe_balls ? 3 : 0) : (K.observe_blocks ? 3 * D::NBLK : 0);
}

// observation element i (maze_env.py:351-369): qpos[:3] | ball xpos / block xpos (3 each, if observed) | qpos[3:15] | qvel[:14] | t/1000
template <int NB>
MZ_HD float ant_obs_elem(const AntDev& K, const AntScratchCoreT<NB>& s, int i, int t) {
  using D = AntDims<NB>;
  int nb3 = ant_obs_extra<NB>(K);
  if (i < 3) return s.qpos[i];
  if (i < 3 + nb3) {
    if constexpr (D::BALL) return s.qpos[15 + (i - 3)];  // get_body_com of a free-joint body: the frame origin = qpos[15:18]
    int k = (i - 3) / 3, c = (i - 3) - 3 * k;
    return ant_block_coord<NB>(K, s, k, c);
  }
  int q = i - nb3;
  if (q < ANT_NQ) return s.qpos[q];
  if (q < ANT_NQ + ANT_NV) return s.qvel[q - ANT_NQ];
  return (float)t * 0.001f;
}

// constant tables of the scratch block, once per step (to be followed by a cx.sync() before the first forward evaluation)
template <int NB, class C, class S>
MZ_HD void ant_fill_tables(const C& cx, const AntDev& K, S& s) {
  MZ_FOR(i, AntDims<NB>::NLO) s.qlo[i] = 0.f;  // the state that enters a step is fp32: no low-order parts yet
  MZ_FOR(one, 1) s.bkey[4] = 0;                  // (LDS does not survive the launch: nothing is staged yet)
  constexpr bool quad = NB <= 1 && C::row_solver;  // the quad forward pass keeps M in registers — and its kernels do not even allocate part 2 of the scratch block
  if constexpr (!quad) {
  cx.sync();
  MZ_FOR(e, 9) {  // linear block of the root's mass matrix: total mass x identity (summed in body order, as the composite inertia is)
    float m = 0.f;
    for (int b = 0; b < ANT_NBODY; b++) m += K.mass[body_class(b)];
    const int i = e / 3, j = e - 3 * i;
    s.M.rr[i][j] = i == j ? m : 0.f;
  }
  }
  MZ_FOR(i, MZ_MAX_GRID) {
    s.rowmask[i] = maze_row(K.maze, i);
    if constexpr (NB > 0) {
      uint32_t pm = 0u;
#pragma unroll
      for (int r = 0; r < MZ_MAX_GRID; r++) pm = (r == i) ? K.maze.platmask[r] : pm;
      s.platmask[i] = K.maze.elevated ? pm : 0u;
    }
  }
}

// ------------------------------------------------------------------ MazeEnv.step for the Ant (maze_env.py:448-481, ant.py:61-73)
// in: s.qpos/qvel/warm loaded, action[8], *t_io = steps so far.  out: obs[obs_dim], reward, done, goal_idx, info[4], *t_io + 1
// (t travels through memory so that it does not occupy a register across the whole step)
template <int NB, class C, class S>
MZ_HD void ant_env_step(const C& cx, const AntDev& K, S& s, const float* action, float* obs, float* reward,
                        uint8_t* done, int* goal_idx, float* info, int* t_io) {
  using D = AntDims<NB>;
  MZ_FOR(i, D::NV) s.fact[i] = 0.f;
  ant_fill_tables<NB>(cx, K, s);
  MZ_FOR(one, 1) { s.status = 0; s.red[1] = s.qpos[0]; s.red[2] = s.qpos[1]; }
  cx.sync();
  MZ_FOR(u, ANT_NU) s.fact[K.act_dof[u]] = K.gear * fminf(fmaxf(action[u], K.ctrl_lo), K.ctrl_hi);
  cx.sync();
  for (int f = 0; f < K.frame_skip; f++) ant_mj_step<NB>(cx, K, s, f == 0);
  int t = *t_io + 1;
  int obs_dim = ANT_OBS + ant_obs_extra<NB>(K);
  MZ_FOR(i, obs_dim) obs[i] = ant_obs_elem<NB>(K, s, i, t);
  MZ_FOR(one, 1) {
    float dt = K.h * (float)K.frame_skip;
    float vx = ((s.qpos[0] - s.red[1]) + s.qlo[0]) / dt, vy = ((s.qpos[1] - s.red[2]) + s.qlo[1]) / dt;
    float fwd = sqrtf(vx * vx + vy * vy), cc = 0.f;
    for (int u = 0; u < ANT_NU; u++) cc += action[u] * action[u];
    cc *= (float)K.task.ctrl_w;
    float o6[6];
    for (int k = 0; k < 6; k++) o6[k] = ant_obs_elem<NB>(K, s, k, t);
    float outer; int tm, gi;
    // (per-env goals: the step kernel parks the env slot in *goal_idx — through LDS, so that it does not occupy a register across the step)
    task_eval_dev(K.task, o6, &outer, &tm, &gi, (K.task.env_goals && goal_idx) ? *goal_idx : -1);
    *reward = (float)K.task.inner_scale * ((float)K.task.fwd_w * fwd - cc) + outer;
    *done = (uint8_t)((tm ? 1 : 0) | (t >= K.task.max_steps ? 2 : 0));
    if (goal_idx) *goal_idx = gi;
    if (info) { info[0] = s.qpos[0]; info[1] = s.qpos[1]; info[2] = fwd; info[3] = -cc; }
    bool badv = false;
    for (int i = 0; i < D::NQ; i++) badv = badv || !(fabsf(s.qpos[i]) < 1e10f);
    for (int i = 0; i < D::NV; i++) badv = badv || !(fabsf(s.qvel[i]) < 1e10f);
    if (badv) s.status |= MZ_STATUS_BAD_STATE;
  }
  cx.sync();
  MZ_FOR(one, 1) *t_io = t;  // after every lane has read the old value
  cx.sync();
}
