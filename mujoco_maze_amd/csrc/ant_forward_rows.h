// ant_forward_rows.h — the plain ant's forward dynamics up to the constraint solve (kinematics, composite
// inertias, mass matrix, bias forces, collision, constraint rows) in the REGISTERS of the row's four leg quads.  Device only;
// included by ant_kernels.hip after ant_newton_rows.h, called by ant_forward (ant_dyn.h) for the plain ant at >= 16 lanes per env.
//
// Why (round 4).  After the Newton iteration had moved into one DPP row (ant_newton_rows.h) the phases in
// front of it were 43 % of a wave's cycles: five LDS hand-offs (kinematics -> inertias -> leg blocks / composite inertia / body
// forces -> hub block / dof forces -> contact rows), each phase a different instruction stream for a quarter to three quarters
// of the lanes.  Here the 16 lanes of a row are the four legs' quads (rows::pos2dof):
//
//     lane 4l + j      as a BODY / GEOM lane             as a DOF lane (= the solver's position)
//     j = 0            aux body of leg l                 hip l
//     j = 1            ankle body of leg l               ankle l
//     j = 2            welded leg capsule of leg l       root dof 2l       (l < 3; quad 3: spare)
//     j = 3            torso (quad 0 only)               root dof 2l + 1   (l < 3; quad 3: spare)
//
// and one instruction stream serves all of them:
//   K  every lane of quad l evaluates leg l's chain (two sincos, the capsule axes, joint axes at the torso origin) — redundantly,
//      which costs what one lane per leg cost before, without the hand-off that followed;
//   I  every lane the spatial inertia of its own body; the whole-body composite is ten row butterflies (rows::rsum), a leg's
//      composites are `quad_perm` moves;
//   M  composite-rigid-body rule in its general form: M[p][q] = S_p . (I_q^c S_q) for q in the subtree of p.  Every dof lane forms
//      F_p = I_p^c S_p with ITS composite inertia (whole body / aux + ankle / ankle) and ITS motion axis; its root columns are dot
//      products with the root's axes (known everywhere), its hinge columns S_p . F_q come through `row_newbcast:q` operands fused
//      into the multiply-adds (48 instructions for the 8 hinge columns of all 14 rows) — row p of M lands in the registers of
//      lane p, which is where the solver wants it (no dense copy in LDS any more);
//   V  recursive Newton-Euler with per-lane masks instead of per-level branches; subtree force sums by quad_perm / rsum;
//   C  a geom's lane enumerates its contacts from its own registers (same narrow phase: round_vs_box of ant_dyn.h), the compact
//      slot order — geom order, as MuJoCo's — comes from ONE packed-count butterfly (2 bits per geom), and the lane that found a
//      contact builds its three constraint rows on the spot: the only LDS hand-off left before the solver is the contact rows
//      it reads (cJ, caref, cD).
// A geom that finds more than three contacts (a foot in a wall corner) sends its env down the lane-group path of ant_dyn.h once:
// the kinematics are published to LDS and con_count / con_fill / con_row run as for the block mazes — same contacts, same order.
#pragma once
#include "ant_newton_rows.h"

// experiment build -DMZ_EXP_SUBTICK2: the forward pass's timers all book on slot 3, slots 0 .. 2 time the solver's set-up (ant_newton_rows.h)
#ifdef MZ_EXP_SUBTICK2
#define MZ_FT(k) 3
#else
#define MZ_FT(k) (k)
#endif
namespace rows {

template <int J>
__device__ __forceinline__ float qbcast(float x) {  // value of lane J of this quad, on every lane of the quad
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), J | (J << 2) | (J << 4) | (J << 6), 0xF, 0xF, true));
}
__device__ __forceinline__ unsigned rsum_u(unsigned x) {  // integer all-reduce over the row (fields must not carry into each other)
  x += (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);
  x += (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true);
  return x;
}
__device__ __forceinline__ float sel4(const float (&v)[4], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); }
__device__ __forceinline__ unsigned sum2bit(unsigned w) { return (unsigned)__popc(w & 0x55555555u) + 2u * (unsigned)__popc(w & 0xAAAAAAAAu); }

struct GeomHit { float pos[3], n[3], dist; int kind; };  // one staged contact of a geom lane (registers)

// fall-back (an env whose geom overflowed its staging): the record of compact slot c from the geometry con_fill_item left in
// s.cY[c] and the kinematics the forward pass published to LDS
template <int NB, class S>
__device__ __forceinline__ void con_record_item(const AntDev& K, S& s, int c) {
  const float* q = &s.cY[c][0][0];
  const float pos[3] = {q[0], q[1], q[2]}, n[3] = {q[3], q[4], q[5]}, hint[3] = {q[8], q[9], q[10]}, dist = q[6];
  const int kind = (int)q[7] & 15, leg = s.cleg[c], cls = s.ccls[c], lg = leg < 0 ? 0 : leg;
  float ww[3], vb[6];
  mat_vecf(ww, s.R0, s.qvel + 3);
  const float qdh = cls >= 2 ? s.qvel[6 + 2 * lg] : 0.f, qda = cls == 3 ? s.qvel[7 + 2 * lg] : 0.f;
  for (int k = 0; k < 3; k++) { vb[k] = ww[k] + s.zw[k] * qdh + s.Sa[lg][k] * qda; vb[3 + k] = s.qvel[k] + s.Sh[lg][k] * qdh + s.Sa[lg][3 + k] * qda; }
  const float vblk[2] = {NB == 1 ? s.qvel[14] : 0.f, NB == 1 ? s.qvel[14 + (NB == 1 ? 1 : 0)] : 0.f};
  contact_raw_store(wr_record(s, c), pos, n, dist, kind, hint, cls, leg, vb, K.bw_tran[cls < 0 ? 0 : cls]);
  (void)vblk;
}

// contacts of one robot geom against the floor plane, the movable block (NB = 1) and the maze's cells (walls; in an elevated maze the
// platform under a cell first) — the robot-geom part of geom_contacts (ant_dyn.h), same order, on register inputs: centre / axis
// relative to the torso origin, torso origin (x0 + x0l, y0 + y0l, cz) in the world
template <int NB, class S, class Emit>
__device__ __forceinline__ void robot_geom_contacts(const AntDev& K, const AntU& z, const S& s, bool sphere, const float* ctr, const float* ax, float hl, float r,
                                                    float x0, float y0, float x0l, float y0l, float cz, Emit&& emit) {
  // (z: the maze's geometry and the two margins — register-held constants of the step in the one-wave kernel, ant_newton_rows.h ant_u)
  const float floor_margin = z.floor_margin, wall_margin = z.wall_margin;
  const float inv = z.inv_scale_xy;
  const float bs[3] = {z.half_xy, z.half_xy, z.half_z};
  ContactGeo cg;
  // floor plane z = 0, normal +z; capsule ends in MuJoCo's geom-frame order [ASSUME-5]: "+axis" is the end at the body origin
#pragma unroll
  for (int k2 = 0; k2 < 2; k2++) {
    if (k2 == 1 && sphere) break;
    const float sg = k2 == 0 ? -1.f : 1.f;
    float p[3];
    for (int k = 0; k < 3; k++) p[k] = ctr[k] + sg * ax[k] * hl;
    const float dist = (cz + p[2]) - r;
    if (dist < floor_margin) {
      cg.dist = dist; cg.kind = 0; cg.blk = 0; cg.other = 0;
      cg.n[0] = 0.f; cg.n[1] = 0.f; cg.n[2] = 1.f;
      cg.pos[0] = p[0]; cg.pos[1] = p[1]; cg.pos[2] = p[2] - (r + 0.5f * dist);
      for (int k = 0; k < 3; k++) cg.hint[k] = 0.f;
      emit(cg);
    }
  }
  const float reach = r + hl + wall_margin;
  if constexpr (NB == 1) {
    // the movable block: spawn position + its two slides, hi parts first, low-order parts after (block_center of ant_dyn.h)
    float d[3] = {0.f, 0.f, 0.f}, dl[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < 2; sl++)
#pragma unroll
      for (int c = 0; c < 3; c++) { d[c] += K.block_axis[sl] == c ? s.qpos[15 + sl] : 0.f; dl[c] += K.block_axis[sl] == c ? s.qlo[2 + sl] : 0.f; }
    const float bc[3] = {((K.block_pos0[0][0] - x0) + d[0]) + (dl[0] - x0l), ((K.block_pos0[0][1] - y0) + d[1]) + (dl[1] - y0l), ((K.block_pos0[0][2] - cz) + d[2]) + dl[2]};
    float d2 = 0.f;
#pragma unroll
    for (int q = 0; q < 3; q++) { const float dd = fmaxf(fabsf(ctr[q] - bc[q]) - K.block_half[q], 0.f); d2 += dd * dd; }
    if (d2 < reach * reach) round_vs_box(sphere, ctr, ax, hl, r, bc, K.block_half, wall_margin, 2, 0, emit);
  }
  const bool elevated = NB > 0 && z.elevated;
  const float gx = x0 + ctr[0], gy = y0 + ctr[1], gz = cz + ctr[2];
  if (gz - reach > z.center_z + z.half_z) return;
  const int j0 = (int)floorf((gx - reach + z.tx) * inv + 0.5f), j1 = (int)floorf((gx + reach + z.tx) * inv + 0.5f);
  const int i0 = (int)floorf((gy - reach + z.ty) * inv + 0.5f), i1 = (int)floorf((gy + reach + z.ty) * inv + 0.5f);
  const float lim2 = (r + wall_margin) * (r + wall_margin) * 1.0001f;
  // can the box of cell (i, j) — layer 1: the wall, layer 0: the platform of an elevated maze — give a contact at all?  In the grid
  // and present; the geom's z extent meets the box's; and the geom's axis segment — its bounding box, axis by axis — comes within
  // radius + margin of the box: a lower bound of the true distance that is exact whenever the nearest feature is a face, i.e. for
  // every wall a leg merely passes (cells are metres wide, capsules centimetres).  Only what survives pays for the
  // closest-feature search of mjc_CapsuleBox (round_vs_box).
  auto candidate = [&](int i, int j, int layer) {
    if (i < 0 || j < 0 || i >= z.rows || j >= z.cols) return false;
    if (!(((layer ? maze_row_lds(s, i) : plat_row_lds(s, i)) >> j) & 1u)) return false;
    const float cz1 = layer ? z.center_z : z.half_z;
    if (gz - reach > cz1 + z.half_z || gz + reach < cz1 - z.half_z) return false;
    const float bc[3] = {((j * z.scale - z.tx) - x0) - x0l, ((i * z.scale - z.ty) - y0) - y0l, cz1 - cz};
    float g2 = 0.f;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const float c = ctr[q] - bc[q], e = fabsf(ax[q]) * hl;  // the segment spans [c - e, c + e] on this axis
      const float gap = fmaxf(fabsf(c) - e - bs[q], 0.f);
      g2 += gap * gap;
    }
    return g2 <= lim2;
  };
  auto test = [&](int i, int j, int layer) {
    const float cz1 = layer ? z.center_z : z.half_z;
    const float bc[3] = {((j * z.scale - z.tx) - x0) - x0l, ((i * z.scale - z.ty) - y0) - y0l, cz1 - cz};  // large world coordinates cancel first, then the low-order parts (AntScratchT::qlo)
    round_vs_box(sphere, ctr, ax, hl, r, bc, bs, wall_margin, 1, 0, emit);
  };
  if (i1 - i0 <= 1 && j1 - j0 <= 1) {
    // the usual case (a geom's bounding square covers at most 2 x 2 cells): both grid rows at once, leave when none of the cells holds a box
    uint32_t rows2 = maze_row_lds(s, i0) | (i1 != i0 ? maze_row_lds(s, i1) : 0u);
    if (elevated) rows2 |= plat_row_lds(s, i0) | (i1 != i0 ? plat_row_lds(s, i1) : 0u);
    const uint32_t cols2 = ((j0 >= 0 && j0 < z.cols) ? 1u << j0 : 0u) | ((j1 != j0 && j1 >= 0 && j1 < z.cols) ? 1u << j1 : 0u);
    if (!(rows2 & cols2)) return;
  }
#ifdef MZ_EXP_STAMPS  // cycles the wave spends between here and the end of the narrow phase (booked by the first active lane, on its env)
  const unsigned long long exp_t0 = __builtin_amdgcn_s_memtime();
#endif
  // the surviving boxes as bit sets (bit 4 (i - i0) + (j - j0): up to 8 x 4 cells, row-major = MuJoCo's geom order of the maze's
  // boxes; per cell the platform before the wall), then one cell per pass — lanes with one candidate each meet in the same pass
  // instead of each waiting for the other's position in a loop nest
  unsigned candw = 0u, candp = 0u;
  if (i1 - i0 <= 1 && j1 - j0 <= 1) {
    // the usual 2 x 2 block in a straight line (round 5): the same test as `candidate` — a cell's gap is (x gap of its column)^2 +
    // (y gap of its row)^2 + (z gap of its layer)^2, six numbers for eight boxes — without the two nested loops over lane-dependent
    // cell ranges (every geom of an ant near a wall ran them, four to eight `candidate` calls each, whether or not it touched anything)
    const float ex = fabsf(ax[0]) * hl, ey = fabsf(ax[1]) * hl, ez = fabsf(ax[2]) * hl;
    float xg2[2], yg2[2];
    bool jin[2], iin[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int j = q ? j1 : j0, i = q ? i1 : i0;
      const float bx = ((j * z.scale - z.tx) - x0) - x0l, by = ((i * z.scale - z.ty) - y0) - y0l;
      const float gx_ = fmaxf(fabsf(ctr[0] - bx) - ex - bs[0], 0.f), gy_ = fmaxf(fabsf(ctr[1] - by) - ey - bs[1], 0.f);
      xg2[q] = gx_ * gx_; yg2[q] = gy_ * gy_;
      jin[q] = j >= 0 && j < z.cols && (q == 0 || j1 != j0);
      iin[q] = i >= 0 && i < z.rows && (q == 0 || i1 != i0);
    }
#pragma unroll
    for (int layer = 1; layer >= 0; layer--) {
      if (layer == 0 && !elevated) break;
      const float cz1 = layer ? z.center_z : z.half_z;
      const bool zok = !(gz - reach > cz1 + z.half_z || gz + reach < cz1 - z.half_z);
      const float gz_ = fmaxf(fabsf(ctr[2] - (cz1 - cz)) - ez - bs[2], 0.f), zg2 = gz_ * gz_;
      unsigned m = 0u;
#pragma unroll
      for (int qi = 0; qi < 2; qi++) {
        const int i = qi ? i1 : i0;
        const uint32_t row = layer ? maze_row_lds(s, i) : plat_row_lds(s, i);  // (kept from the early-out's reads, tried: neutral)
#pragma unroll
        for (int qj = 0; qj < 2; qj++) {
          const int j = qj ? j1 : j0;
          const bool present = iin[qi] && jin[qj] && ((row >> (j & 31)) & 1u);
          if (present && zok && xg2[qj] + yg2[qi] + zg2 <= lim2) m |= 1u << (4 * qi + qj);
        }
      }
      if (layer) candw = m; else candp = m;
    }
  } else
  for (int i = i0; i <= i1 && i < i0 + 8; i++)
    for (int j = j0; j <= j1 && j < j0 + 4; j++) {
      if (candidate(i, j, 1)) candw |= 1u << (4 * (i - i0) + (j - j0));
      if (elevated && candidate(i, j, 0)) candp |= 1u << (4 * (i - i0) + (j - j0));
    }
#ifdef MZ_EXP_NOWALL  // timing experiment (wrong physics): no wall narrow phase at all
  candw = 0u; candp = 0u;
#endif
#ifdef MZ_EXP_STAMPS  // cycles of the candidate scan alone (same booking as below)
  {
    const unsigned long long act = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(const_cast<int*>(&s.bkey[3]), (int)(__builtin_amdgcn_s_memtime() - exp_t0));
  }
#endif
  while (candw | candp) {
#ifdef MZ_EXP_STAMPS  // (NB = 0 only: bkey is free there) narrow-phase runs of this env over the step
    atomicAdd(const_cast<int*>(&s.bkey[0]), 1);
#endif
    const int b = __ffs((int)(candw | candp)) - 1;
    const unsigned bit = 1u << b;
    if (candp & bit) test(i0 + (b >> 2), j0 + (b & 3), 0);
    if (candw & bit) test(i0 + (b >> 2), j0 + (b & 3), 1);
    candw &= ~bit; candp &= ~bit;
#ifdef MZ_EXP_ONECAND  // timing experiment (wrong physics): at most one narrow-phase run per geom — what would dealing the candidates to idle lanes save?
    candw = 0u; candp = 0u;
#endif
  }
#ifdef MZ_EXP_STAMPS
  {
    const unsigned long long act = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(const_cast<int*>(&s.bkey[2]), (int)(__builtin_amdgcn_s_memtime() - exp_t0));
  }
#endif
}

}  // namespace rows

// Per-lane constants of the quad layout, loaded ONCE per step into DevCtx::lc (registers; run-time indexed loads of the constant
// block inside the 20 evaluations came out as address selects + vector-memory loads): leg signs, ankle axis, and the own body's
// mass / inertia / capsule size / contact weight — zero mass for the torso's duplicates in quads 1..3, so that row sums count it once.
enum { LC_SX, LC_SY, LC_AX0, LC_AX1, LC_AX2, LC_MASS, LC_ILAT, LC_DAX, LC_HLEN, LC_RAD, LC_TRAN, LC_N };  // (+ LC_LO, LC_HI, LC_DOFW: ant_newton_rows.h)
template <int G, bool PROF>
__device__ __forceinline__ void ant_lane_consts(const AntDev& K, DevCtx<G, PROF>& cx) {
  static_assert(LC_N <= DevCtx<G, PROF>::NLC, "DevCtx::lc too small");
  const int p = cx.l & 15, l = p >> 2, j = p & 3;
  const int cls = j == 0 ? 2 : (j == 1 ? 3 : (j == 2 ? 1 : 0));
  const bool primary = !(j == 3 && l > 0);
  cx.lc[LC_SX] = K.sx[l]; cx.lc[LC_SY] = K.sy[l];
  cx.lc[LC_AX0] = K.ank_axis[l][0]; cx.lc[LC_AX1] = K.ank_axis[l][1]; cx.lc[LC_AX2] = K.ank_axis[l][2];
  cx.lc[LC_MASS] = primary ? K.mass[cls] : 0.f; cx.lc[LC_ILAT] = primary ? K.ilat[cls] : 0.f; cx.lc[LC_DAX] = primary ? K.iax[cls] - K.ilat[cls] : 0.f;
  cx.lc[LC_HLEN] = K.half_len[cls]; cx.lc[LC_RAD] = K.radius[cls]; cx.lc[LC_TRAN] = K.bw_tran[cls];
  ant_limit_consts(K, cx);
}

// One forward-dynamics evaluation of the plain ant (NB = 0) or of the ant with ONE movable block on two slides (NB = 1: AntPush,
// AntFall, ...): qacc from (s.qpos, s.qvel, s.fact); `first`: first evaluation of an env.step (s.warm = MuJoCo's qacc_warmstart),
// otherwise s.warm = the previous evaluation's solution.  Same outputs as ant_forward's lane-group path: s.qacc, s.qas (where
// computed), s.ncon, s.iters, status bits.
// NB = 1: the block's two slides are positions 14, 15 of the row (a diagonal block of M, gravity on a z slide).  Its OWN contacts
// (floor corners, maze boxes, slide limits) are enumerated by the first eleven lanes of the group in a second pass — the
// sub-enumerators of geom_contacts, staged as before — and take the first slots; their rows are built inside the solver by the
// lanes that own them (block_rows_direct).  A robot geom touching the block is a record like any other, flagged so that the slide
// lanes see its reaction.
// SCR: AntScratchCoreT<NB> (product kernels: all they allocate) or AntScratchT<NB> (instrumented builds: the phase timers live there)
template <int NB, int G, bool PROF, class SCR>
__device__ __forceinline__ float ant_forward_rows(const DevCtx<G, PROF>& cx_step, const AntDev& K, SCR& s, bool first) {
  static_assert(G >= 16, "one DPP row per env at least");
  static_assert(NB <= 1, "row layout: 14 robot dofs + one block's two slides");
  using namespace rows;
  // The two-waves-per-SIMD instantiation (cx.mfma; 256 registers in all) re-reads its per-lane constants from the constant block at
  // every evaluation instead of carrying them through the step: carried, they were what the compiler spilled — 36 registers to
  // scratch, stored by every wave and re-read per evaluation (PMC, 8192 envs: WRITE_SIZE 14.5 MB per launch against 2.8 MB of
  // algorithmic writes).  The opaque lane index keeps the loads inside the evaluation (they are loop-invariant otherwise).
  DevCtx<G, PROF> cx = cx_step;
#ifdef MZ_EXP_LCRELOAD  // A / B build: the one-wave kernel as well
  if (true) {
#else
  if (cx_step.mfma) {
#endif
    int ll = cx_step.l;
    asm volatile("" : "+v"(ll));
    cx.l = ll;
    ant_lane_consts(K, cx);
  }
  else if constexpr (NB == 0) {
    // (round 5) the plain ant's one-wave kernel: the lane index opaque per evaluation, so that the lane predicates (role masks, `r == P`
    // ...) are compared again inside each evaluation instead of being carried through the step in scalar-register pairs the kernel does
    // not have — v_readlane 371 -> 130, v_writelane 196 -> 85 in the kernel, 0.2455 -> 0.2443 ms.  (The one-block kernel at 32 lanes
    // loses 1.7 % with it: not there.)
    int ll = cx_step.l;
    asm volatile("" : "+v"(ll));
    cx.l = ll;
  }
  using C = DevCtx<G, PROF>;  // (MZ_FOR)
  using D = AntDims<NB>;
  constexpr int NC = D::NC, NR = 14 + 2 * NB;
  cx.tick(s, 9);
  const int p = cx.l & 15, l = p >> 2, j = p & 3;
  const bool hinge = j < 2, isroot = j >= 2 && l < 3;
  const int iroot = 2 * l + j - 2;                 // root dof of a root lane
  const bool isgeom = j < 3 || p == 3;
  const int m0 = -(int)(j == 0), m1 = -(int)(j == 1), m2 = -(int)(j == 2), mh = -(int)hinge;  // role masks (bsel)
  auto role = [&](float a0, float a1, float a2, float a3) { return bsel(m0, a0, bsel(m1, a1, bsel(m2, a2, a3))); };

  // ---- K: leg l's chain, on every lane of its quad (kin_item of ant_dyn.h)
  float R0[9];
  quat_to_matf(R0, s.qpos + 3);
  const float cz = s.qpos[2], x0 = s.qpos[0], y0 = s.qpos[1];
  const float zw[3] = {R0[2], R0[5], R0[8]};
  const float sx = cx.lc[LC_SX], sy = cx.lc[LC_SY];
  const float aax[3] = {cx.lc[LC_AX0], cx.lc[LC_AX1], cx.lc[LC_AX2]};
  const float qh = s.qpos[7 + 2 * l], qa = s.qpos[8 + 2 * l];
  float w0[3], w1[3], w2[3], com0[3], com1[3], com2[3], p1[3], p2[3], ShL[3], Sa[6];
  {
    const float isq2 = 0.70710678118654752f;
    const float u[3] = {sx * isq2, sy * isq2, 0.f}, off[3] = {sx * K.legoff, sy * K.legoff, 0.f};
    float ch, sh, ca, sa;
    mz_sincosf(qh, &sh, &ch);
    mz_sincosf(qa, &sa, &ca);
    float t[3], v[3];
    mat_vecf(w0, R0, u);                                                    // level 0: welded leg capsule, torso frame
    for (int k = 0; k < 3; k++) com0[k] = w0[k] * K.half_len[1];
    mat_vecf(p1, R0, off);                                                  // level 1: aux body, rotated about body z by the hip angle
    t[0] = ch * u[0] - sh * u[1]; t[1] = sh * u[0] + ch * u[1]; t[2] = 0.f;
    mat_vecf(w1, R0, t);
    for (int k = 0; k < 3; k++) com1[k] = p1[k] + w1[k] * K.half_len[2];
    cross3f(ShL, p1, zw);                                                   // linear velocity at c of a unit hip rotation: zw x (c - p1)
    t[0] = ch * off[0] - sh * off[1]; t[1] = sh * off[0] + ch * off[1]; t[2] = 0.f;
    mat_vecf(v, R0, t);
    for (int k = 0; k < 3; k++) p2[k] = p1[k] + v[k];                       // level 2: ankle body, rotated about its local axis
    const float au = aax[0] * u[0] + aax[1] * u[1];
    float axu[3], ul[3];
    cross3f(axu, aax, u);
    for (int k = 0; k < 3; k++) ul[k] = u[k] * ca + axu[k] * sa + aax[k] * au * (1.f - ca);  // Rodrigues
    t[0] = ch * ul[0] - sh * ul[1]; t[1] = sh * ul[0] + ch * ul[1]; t[2] = ul[2];
    mat_vecf(w2, R0, t);
    for (int k = 0; k < 3; k++) com2[k] = p2[k] + w2[k] * K.half_len[3];
    t[0] = ch * aax[0] - sh * aax[1]; t[1] = sh * aax[0] + ch * aax[1]; t[2] = aax[2];
    mat_vecf(Sa, R0, t);                                                    // ankle axis (world)
    cross3f(Sa + 3, p2, Sa);                                                // aw x (c - p2)
  }
  // this lane's own body: j = 0 aux (class 2), 1 ankle (3), 2 welded leg (1), 3 torso (0)
  float com[3], w[3];
  for (int k = 0; k < 3; k++) { com[k] = role(com1[k], com2[k], com0[k], 0.f); w[k] = role(w1[k], w2[k], w0[k], 0.f); }
  const int cls = j == 0 ? 2 : (j == 1 ? 3 : (j == 2 ? 1 : 0));
  const float hlen = cx.lc[LC_HLEN], rad = cx.lc[LC_RAD];
  cx.tick(s, MZ_FT(0));
  if constexpr (NB == 1) { if (p == 0) { s.cz = cz; s.con_over = 0; } }  // (the block's enumerators below read them)
  // ---- C: contacts of the own geom, staged in registers (the first three; more: the env takes the fall-back below).  Right after the
  // kinematics: the narrow phase is the branchiest code of the evaluation, and here little else is live across it
  GeomHit hit[3];
  int nfound = 0;
  if (isgeom) {
    robot_geom_contacts<NB>(K, ant_u(cx, K), s, j == 3, com, w, hlen, rad, x0, y0, s.qlo[0], s.qlo[1], cz, [&](const ContactGeo& g) {
#pragma unroll
      for (int q = 0; q < 3; q++)
        if (nfound == q) {
          for (int k = 0; k < 3; k++) { hit[q].pos[k] = g.pos[k]; hit[q].n[k] = g.n[k]; }
          hit[q].dist = g.dist; hit[q].kind = g.kind;
        }
      nfound++;
    });
  }
#ifdef MZ_EXP_SUBTICK  // experiment build: the robot geoms' narrow phase is booked on slot 0, the block's enumerators on slot 3
  cx.tick(s, MZ_FT(0));
#endif
  // the movable block's own enumerators (floor corners | the 3 x 3 cells under it, platform and wall each | slide limits), one per
  // lane of the group's first eleven, staged in the cY block as in the lane-group path; their contacts take the first slots
  int nblk = 0, nrep = 0;  // nrep: contact points folded into merged entries (for the count MuJoCo would report)
  bool bover = false;
  if constexpr (NB == 1) {
    cx.sync();
    // Round 5: enumerate only when the block has MOVED since the enumeration whose results are still staged.  A block nobody touches
    // is at rest to the bit (no force along its slides: zero acceleration, zero velocity), which is most envs at most times — and its
    // eleven float64 enumerators were 14 % of a wave (24 % of the slow ones: profiles/r04/tail_phases_AntPush-v0_2048.txt), re-run
    // for each of the step's 20 evaluations.  What they stage — kind, normal, distance, multiplicity of each entry; block_rows_direct
    // reads nothing else — is a function of the block's position and the maze alone; the rows' velocity terms are rebuilt from the
    // staged geometry at every evaluation as before.  An env that took the lane-group fall-back (its staging was overwritten) or whose
    // enumeration overflowed re-enumerates.
#ifndef MZ_EXP_NOBLOCKCACHE
    const int k0 = __float_as_int(s.qpos[15]), k1 = __float_as_int(s.qpos[16]), k2 = __float_as_int(s.qlo[2]), k3 = __float_as_int(s.qlo[3]);
    const bool staged = s.bkey[4] != 0 && s.bkey[0] == k0 && s.bkey[1] == k1 && s.bkey[2] == k2 && s.bkey[3] == k3;
#else
    const int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    const bool staged = false;
#endif
    if (cx.any(!staged)) {
      cx.sync();  // (every lane has read the key)
      if (!staged) { MZ_FOR(e, D::NMOV) con_enum_item<NB, true>(K, s, e); }
      cx.sync();
      if (!staged && cx.l == 0) { s.bkey[0] = k0; s.bkey[1] = k1; s.bkey[2] = k2; s.bkey[3] = k3; s.bkey[4] = s.con_over == 0; }
    }
    int mine = 0, before = 0;
#pragma unroll
    for (int e = 0; e < D::NMOV; e++) {
      const int ce = s.cnt[e], c = ce & 255;  // merged entries | contacts emitted << 8
      nblk += c; nrep += (ce >> 8) - c; before += e < cx.l ? c : 0; mine = e == cx.l ? c : mine;
    }
    bover = s.con_over != 0;
    if (cx.l < D::NMOV && !bover)
      for (int i = 0; i < mine; i++) { const int slot = before + i; if (slot < NC) { s.csrc[slot] = MZ_STAGE_OF(NB) * cx.l + i; s.cleg[slot] = -1; s.ccls[slot] = -1; } }
    if (nblk > NC) nblk = NC;
  }
#ifdef MZ_EXP_SUBTICK
  cx.tick(s, 3);
#endif
  const bool over = cx.gany(nfound > 3) || bover;
  // compact slots in geom order (torso, then per leg: welded capsule, aux, ankle): one packed-count butterfly
  const int rank = j == 3 ? 0 : 1 + 3 * l + (j == 2 ? 0 : j + 1);
  const unsigned word = rsum_u((unsigned)(nfound > 3 ? 3 : nfound) << (2 * rank));
  const int off = nblk + (int)sum2bit(word & ((1u << (2 * rank)) - 1u));
  int ncon = nblk + (int)sum2bit(word);
  cx.tick(s, MZ_FT(2));

  // ---- I: spatial inertia of the own body about the torso origin (inertia_item); composites
  float cin[10];
  {
    const float m = cx.lc[LC_MASS], lat = cx.lc[LC_ILAT], dax = cx.lc[LC_DAX];
    const float rr = dot3f(com, com);
    cin[0] = m; cin[1] = m * com[0]; cin[2] = m * com[1]; cin[3] = m * com[2];
    cin[4] = lat + dax * w[0] * w[0] + m * (rr - com[0] * com[0]);
    cin[5] = lat + dax * w[1] * w[1] + m * (rr - com[1] * com[1]);
    cin[6] = lat + dax * w[2] * w[2] + m * (rr - com[2] * com[2]);
    cin[7] = dax * w[0] * w[1] - m * com[0] * com[1];
    cin[8] = dax * w[0] * w[2] - m * com[0] * com[2];
    cin[9] = dax * w[1] * w[2] - m * com[1] * com[2];
  }
  float Ip[10];  // composite inertia of this lane's dof: whole body (root), aux + ankle (hip), ankle (ankle)
#pragma unroll
  for (int k = 0; k < 10; k++) {
    const float all = rsum(cin[k]), ank = qbcast<1>(cin[k]);
    Ip[k] = role(cin[k] + ank, cin[k], all, all);
  }
  // ---- motion axis of this lane's dof at the torso origin: [angular; linear]
  float S[6];
  {
    const int c = iroot - 3;  // column of R0 for a root angular dof
    const int mang = -(int)(isroot && iroot >= 3), mc0 = -(int)(c == 0), mc1 = -(int)(c == 1);
    for (int k = 0; k < 3; k++) {
      const float rc = bsel(mc0, R0[3 * k], bsel(mc1, R0[3 * k + 1], R0[3 * k + 2]));
      S[k] = bsel(m0, zw[k], bsel(m1, Sa[k], bsel(mang, rc, 0.f)));
      S[3 + k] = bsel(m0, ShL[k], bsel(m1, Sa[3 + k], bsel(-(int)(isroot && iroot == k), 1.f, 0.f)));
    }
  }
  float Sax[6];  // what the solver's contact columns use: S, and for the block's slide lanes MINUS their world axis (linear part)
#pragma unroll
  for (int k = 0; k < 6; k++) Sax[k] = S[k];
  if constexpr (NB == 1) {
    const int sl = p - 14;  // slide index of a block lane
    const int ax = sl == 0 ? K.block_axis[0] : K.block_axis[1];
#pragma unroll
    for (int k = 0; k < 3; k++) Sax[3 + k] = p >= 14 ? (ax == k ? -1.f : 0.f) : S[3 + k];
  }
  // ---- M: row p of the mass matrix, position order
  float Mrow[NR];
  {
    float F[6];
    inertia_mulf(F, Ip, S);
    // root columns: S_i . F with the root's axes (linear e_i; angular the columns of R0)
    const float mr[6] = {F[3], F[4], F[5], R0[0] * F[0] + R0[3] * F[1] + R0[6] * F[2], R0[1] * F[0] + R0[4] * F[1] + R0[7] * F[2],
                         R0[2] * F[0] + R0[5] * F[1] + R0[8] * F[2]};
#pragma unroll
    for (int i = 0; i < 6; i++) Mrow[dof2pos(i)] = mr[i];
    // hinge columns q: S_mine . F_q — valid where q lies in the subtree of this lane's dof (root: every hinge; hip: itself and its
    // ankle; ankle: itself).  The ankle's hip column is the transposed entry S_hip . F_mine (the hip's axis is known in its quad).
    const float g0 = dot6_from<0>(F, S), g1 = dot6_from<1>(F, S), g4 = dot6_from<4>(F, S), g5 = dot6_from<5>(F, S);
    const float g8 = dot6_from<8>(F, S), g9 = dot6_from<9>(F, S), g12 = dot6_from<12>(F, S), g13 = dot6_from<13>(F, S);
    const float hipT = zw[0] * F[0] + zw[1] * F[1] + zw[2] * F[2] + ShL[0] * F[3] + ShL[1] * F[4] + ShL[2] * F[5];
    const float gq[8] = {g0, g1, g4, g5, g8, g9, g12, g13};
#pragma unroll
    for (int l2 = 0; l2 < 4; l2++) {
      const int mine = -(int)(l2 == l);
      const float vh = gq[2 * l2], va = gq[2 * l2 + 1];
      const float hh = bsel(mine, bsel(m1, hipT, vh + ant_u(cx, K).armature), 0.f), ha = bsel(mine, bsel(m1, va + ant_u(cx, K).armature, va), 0.f);  // hinge lanes
      Mrow[4 * l2] = bsel(mh, hh, vh); Mrow[4 * l2 + 1] = bsel(mh, ha, va);
    }
    if constexpr (NB == 1) {  // the block: a separate tree — its mass on the diagonal of its two slides, no coupling with the robot
      Mrow[14] = p == 14 ? K.block_mass : 0.f; Mrow[15] = p == 15 ? K.block_mass : 0.f;
    }
  }
  cx.tick(s, 3);
  // ---- V: recursive Newton-Euler (bias_body_item / bias_dof_item), masks instead of branches
  const float qv[6] = {s.qvel[0], s.qvel[1], s.qvel[2], s.qvel[3], s.qvel[4], s.qvel[5]};
  const float qdh = s.qvel[6 + 2 * l], qda = s.qvel[7 + 2 * l];
  float qfs, vb[6];  // vb: spatial velocity of the own body at the torso origin (the contact rows use it too)
  {
    float a[6], ww[3];
    mat_vecf(ww, R0, qv + 3);  // world angular velocity (root angular dofs are body-frame)
    for (int k = 0; k < 3; k++) { vb[k] = ww[k]; vb[3 + k] = qv[k]; a[k] = 0.f; }
    cross3f(a + 3, qv, ww);
    a[5] -= ant_u(cx, K).gz;   // gravity as base acceleration
    const float vh = bsel(mh, qdh, 0.f), va = bsel(m1, qda, 0.f);  // aux and ankle move with the hip, the ankle body with the ankle as well
    const float Sh6[6] = {zw[0], zw[1], zw[2], ShL[0], ShL[1], ShL[2]};
    float sd[6];
    motion_crossf(sd, vb, Sh6);
    for (int k = 0; k < 6; k++) { a[k] += sd[k] * vh; vb[k] += Sh6[k] * vh; }
    motion_crossf(sd, vb, Sa);
    for (int k = 0; k < 6; k++) { a[k] += sd[k] * va; vb[k] += Sa[k] * va; }
    float Ia[6], Iv[6], vf[6], f[6];
    inertia_mulf(Ia, cin, a);
    inertia_mulf(Iv, cin, vb);
    force_crossf(vf, vb, Iv);
    for (int k = 0; k < 6; k++) f[k] = Ia[k] + vf[k];
    // force on the subtree of this lane's dof: ankle = its body; hip = aux + ankle; root = all bodies
    float bias = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const float all = rsum(f[k]), ank = qbcast<1>(f[k]);
      bias += S[k] * role(f[k] + ank, f[k], all, all);
    }
    const float fact = s.fact[hinge ? pos2dof(p) : 6];  // motors sit on the hinges only
    qfs = bsel(mh, -ant_u(cx, K).damping * bsel(m0, qdh, qda) - bias + fact, bsel(-(int)isroot, -bias, 0.f));
    if constexpr (NB == 1) {  // block slides: undamped, unactuated; gravity acts on a z slide (falling blocks)
      const int ax = p == 14 ? K.block_axis[0] : K.block_axis[1];
      if (p >= 14) qfs = ax == 2 ? K.block_mass * K.gz : 0.f;
    }
  }
  cx.tick(s, MZ_FT(1));
  if (!over) {
    if (ncon > NC) { ncon = NC; if (p == 0) s.status |= MZ_STATUS_CONTACT_OVERFLOW; }
    if (p == 0) { s.ncon = ncon; s.nblkcon = nblk; s.ncon_true = ncon + nrep; }
    // Round 5: the geom's lane only STAGES its contacts (position, normal, distance, kind, tangent hint, body class / leg, the body's
    // spatial velocity, the body's contact weight: 18 numbers); the three wrenches, the reference accelerations and D are built by the
    // lane that OWNS the slot in the solver (contact_record in ant_solve_rows_core), one contact per lane in one pass — a geom with two
    // or three contacts (a capsule lying on the floor, a foot in a wall corner: the envs a launch waits for) built them one after
    // the other here (`C records` 72 k against 33 k cycles in the slowest wave of a step, profiles/r05/tail_phases.txt).
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int slot = off + q;
      if (q < nfound && slot < NC) {
        const GeomHit& h = hit[q];
        const float hint[3] = {(h.kind == 0 && j < 3) ? w[0] : 0.f, (h.kind == 0 && j < 3) ? w[1] : 0.f, (h.kind == 0 && j < 3) ? w[2] : 0.f};
        contact_raw_store(wr_record(s, slot), h.pos, h.n, h.dist, h.kind, hint, cls, j == 3 ? -1 : l, vb, cx.lc[LC_TRAN]);
      }
    }
    cx.sync();
  }
#ifdef MZ_EXP_STAMPS
  if (over && p == 0) s.bkey[1] += 1;
#endif
  if (cx.any(over)) {  // (compiled out as a timing experiment, round 5: no faster — nothing of this block is paid by the waves that skip it)
    // Fall-back (rare: some geom of some env of this wave found more than three contacts): publish the kinematics the lane-group
    // contact code of ant_dyn.h reads, and let the envs concerned enumerate the two-pass way.
    if (isgeom && j < 3) { const int b = 3 * l + (j == 2 ? 0 : j + 1); for (int k = 0; k < 3; k++) { s.com[b][k] = com[k]; s.w[b][k] = w[k]; } }
    if (j == 0) { for (int k = 0; k < 3; k++) s.Sh[l][k] = ShL[k]; for (int k = 0; k < 6; k++) s.Sa[l][k] = Sa[k]; }
    if (p == 0) { for (int k = 0; k < 9; k++) s.R0[k] = R0[k]; for (int k = 0; k < 3; k++) s.zw[k] = zw[k]; s.cz = cz; s.nearwall = 1; s.con_over = 1; s.bkey[4] = 0; }
    cx.sync();
    if (over) {
      constexpr int NG = D::NGEOM;
      MZ_FOR(e, NG) con_count_item<NB>(K, s, e);
      cx.sync();
      MZ_FOR(e, NG) con_fill_item<NB>(K, s, e);
      cx.sync();
      const int nb0 = NB ? s.nblkcon : 0;  // (the block's own contacts keep their staged geometry: block_rows_direct reads it)
      MZ_FOR(c, s.ncon - nb0) con_record_item<NB>(K, s, nb0 + c);
      if (p == 0) s.ncon_true = s.ncon;  // (nothing is merged on this path)
      cx.sync();
    }
  }
  cx.tick(s, 11);
  return ant_solve_rows_core<NB, G, PROF, true>(cx, K, s, first, Mrow, qfs, Sax, bsel(m0, qh, qa), bsel(m0, qdh, qda));  // qacc of this lane's dof
}

// One mj_step with RK4 (ant_mj_step of ant_dyn.h) on the quad layout: every dof lane keeps its own velocity, the RK4
// accumulators and the stage's acceleration in registers — the forward pass hands qacc back in the register of the lane that owns
// it — and writes only what the next evaluation reads (qpos, qvel, the solver's warm start) to LDS: one hand-off per stage instead
// of three.  The free joint's quaternion is advanced redundantly by every lane (the three body-frame rates come by row
// broadcast) and stored by one.
template <int NB, bool XREG, int G, bool PROF, class SCR>
__device__ __forceinline__ void ant_mj_step_rows(const DevCtx<G, PROF>& cx, const AntDev& K, SCR& s, bool first_frame) {
  using namespace rows;
  using C = DevCtx<G, PROF>;  // (MZ_FOR)
  using D = AntDims<NB>;
  constexpr int NR = 14 + 2 * NB;
  const float h = K.h;
  const int p = cx.l & 15;
  const bool isdof = p < NR;
  const int i = isdof ? pos2dof(p) : 0;              // this lane's dof
  // (round 5) the frame's start state stays in registers — every dof lane its own coordinate and that coordinate's low part, every
  // lane the quaternion — instead of the x0q / x0lo copies in LDS: 17 loads + 17 stores + a hand-off per frame and two waits per RK4
  // stage less (A / B: 0.2403 -> 0.2365 ms)
  // XREG = false: the two-waves-per-SIMD instantiation keeps the LDS copies (six more live registers sent its 256-register budget to
  // scratch: 8192 envs, WRITE + FETCH 5.3 -> 11.8 MB per launch).  A compile-time switch: the same choice behind the run-time constant
  // `cx.mfma` left the one-wave kernel 2 % slower than the plain register form (A / B in one call: 0.2421 against 0.2367 ms).
  const int qi_ = !isdof ? 0 : (i < 3 ? i : (i >= 6 ? i + 1 : 0)), li_ = !isdof ? 0 : (i < 2 ? i : (i >= 14 ? 2 + (i - 14) : 0));
  float x0c = 0.f, x0l = 0.f, q0[4] = {1.f, 0.f, 0.f, 0.f};
  if constexpr (XREG) {
    x0c = s.qpos[qi_]; x0l = s.qlo[li_ < D::NLO ? li_ : 0];
    q0[0] = s.qpos[3]; q0[1] = s.qpos[4]; q0[2] = s.qpos[5]; q0[3] = s.qpos[6];
  } else {
    MZ_FOR(k, D::NQ) s.x0q[k] = s.qpos[k];
    MZ_FOR(k, D::NLO) s.x0lo[k] = s.qlo[k];
  }
  const float x0v = isdof ? s.qvel[i] : 0.f;
  float qvel = x0v, accv = 0.f, accf = 0.f;
  cx.sync();
  // position update of this lane's coordinate from the frame's start: qpos <- integrate(x0, vel, h) (mj_integratePos)
  auto integrate = [&](float vel) {
    const float w0 = bcast<7>(vel), w1 = bcast<10>(vel), w2 = bcast<11>(vel);  // root angular dofs 3, 4, 5 sit on lanes 7, 10, 11
    float quat[4];
    const float w[3] = {w0, w1, w2};
    if constexpr (XREG) {
      quat_integratef(q0, w, h, quat);
      if (p == 7) { s.qpos[3] = quat[0]; s.qpos[4] = quat[1]; s.qpos[5] = quat[2]; s.qpos[6] = quat[3]; }
      if (isdof) {
        if (i < 2) mz_step_split(x0c, x0l, h, vel, &s.qpos[i], &s.qlo[i]);                        // absolute x, y: hi + lo (AntScratchT::qlo)
        else if (i == 2) s.qpos[2] = x0c + h * vel;
        else if (i >= 6 && i < 14) s.qpos[i + 1] = x0c + h * vel;                                 // hinges: qpos index = dof + 1
        else if (i >= 14) mz_step_split(x0c, x0l, h, vel, &s.qpos[i + 1], &s.qlo[2 + (i - 14)]);  // block slides
      }
    } else {
      quat_integratef(s.x0q + 3, w, h, quat);
      if (p == 7) { s.qpos[3] = quat[0]; s.qpos[4] = quat[1]; s.qpos[5] = quat[2]; s.qpos[6] = quat[3]; }
      if (isdof) {
        if (i < 2) mz_step_split(s.x0q[i], s.x0lo[i], h, vel, &s.qpos[i], &s.qlo[i]);
        else if (i == 2) s.qpos[2] = s.x0q[2] + h * vel;
        else if (i >= 6 && i < 14) s.qpos[i + 1] = s.x0q[i + 1] + h * vel;
        else if (i >= 14) mz_step_split(s.x0q[i + 1], s.x0lo[2 + (i - 14)], h, vel, &s.qpos[i + 1], &s.qlo[2 + (i - 14)]);
      }
    }
  };
  for (int st = 0; st < 4; st++) {
    const float qacc = ant_forward_rows<NB>(cx, K, s, first_frame && st == 0);
    const float bw = (st == 0 || st == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
    const float aw = st == 2 ? 1.0f : 0.5f;  // Butcher A: diag(1/2, 1/2, 1)
    accv += bw * qvel;
    accf += bw * qacc;
    // the next stage's constraint solve starts from this stage's solution (after the 4th stage: MuJoCo's qacc_warmstart)
    if (isdof) s.warm[i] = qacc;
    if (st < 3) {
      integrate(aw * qvel);
      qvel = x0v + h * aw * qacc;
      if (isdof) s.qvel[i] = qvel;
#ifdef MZ_EXP_RK4TICK  // (tools/tail_phases.py: what the "rk4" timer slot of the slowest waves is made of)
      cx.tick(s, 13);
#endif
      cx.sync();
#ifdef MZ_EXP_RK4TICK
      cx.tick(s, 14);
#endif
    }
  }
  integrate(accv);
  if (isdof) s.qvel[i] = x0v + h * accf;
#ifdef MZ_EXP_RK4TICK
  cx.tick(s, 13);
#endif
  cx.sync();
#ifdef MZ_EXP_RK4TICK
  cx.tick(s, 14);
#endif
}
