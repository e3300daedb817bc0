// ant_newton_rows.h — the plain ant's constraint solve with the whole Newton iteration resident in the
// registers of one 16-lane DPP row.  Device only (CDNA row operations); included by ant_kernels.hip after ant_dyn.h.
//
// Why.  The problem is tiny — 14 dofs, 1-3 contacts (3 rows each), a few joint-limit rows — and the generic lane-group code
// (ant_solve in ant_dyn.h: every cross-lane hand-off through LDS) spends it almost entirely in `s_waitcnt`: ~127 waits per
// Newton iteration for ~900 vector instructions (profiles/r01: SQ_WAIT_ANY 44 % of the wave's life, 12 k cycles per
// iteration).  Here lane r of a row owns dof r (MuJoCo order: root 0-5, then hip / ankle of the four legs) and keeps in
// registers: row r of M and of the Hessian H, its entries of grad / qacc / search / M qacc, and — dofs 6-13 being the
// hinges — the joint-limit row of its own dof.  Lane c (< ncon <= 16) doubles as the owner of contact c: residuals u[3],
// J search, curvature.  Cross-lane traffic is DPP only:
//   * `row_newbcast:p` hands the pivot row / pivot entry to all lanes — a Gauss-Jordan elimination on the row-distributed
//     matrix (leg dofs first, so the arrow sparsity costs nothing: rows of other legs have zero multipliers) solves
//     H x = -grad in ~230 vector instructions with no memory access and no back substitution;
//   * sums over dofs or over contacts are 4-step row butterflies.
// LDS is touched per iteration only to publish the 8 curvature numbers of each contact, to read the contact Jacobians
// (broadcast reads) and to hand `search` to the contact lanes: ~4 waits instead of ~127.
//
// Same mathematics and stopping rule as ant_solve: primal Newton on the pyramidal soft-constraint cost, unit step accepted
// by a vote when the active set is unchanged, otherwise exact line search (safeguarded Newton on phi'), MuJoCo's
// scaled-gradient tolerance or the fp32 cancellation floor.  Rows beyond the first of a lane group (32 / 64 lanes per env)
// mirror row 0 (same r, same values, duplicate stores): no masking, no divergence.
#pragma once
#include <type_traits>

#include "ant_dyn.h"
#include "mz_device.h"

namespace rows {

// Position p of the 16-lane row <-> MuJoCo dof (round 4).  Quad l (lanes 4l .. 4l + 3) belongs to leg l: lane 4l owns its hip
// (dof 6 + 2l), lane 4l + 1 its ankle (dof 7 + 2l); lanes 2, 3 of quads 0 .. 2 own the root's six dofs (0 .. 5) and lanes 14, 15
// the movable block's two slides (dofs 14, 15; NB = 1) or nothing.  Everything the solver keeps per row — Mrow / Hrow entries,
// `row_newbcast:k`, the pivot order — is indexed by POSITION; only the LDS vectors shared with the rest of the step (qpos, qvel,
// qfs, warm, qas, qacc, the columns of cJ) are in dof order.  The point of the layout: a leg's kinematics, composite inertias and
// bias forces are computed by the four lanes of its own quad (ant_forward_rows.h), and its two rows of M come out on the lanes
// that own them — `quad_perm` moves instead of LDS hand-offs.
__host__ __device__ __forceinline__ constexpr int pos2dof(int p) { return (p & 3) < 2 ? 6 + 2 * (p >> 2) + (p & 1) : (p < 12 ? 2 * (p >> 2) + (p & 3) - 2 : p); }
__host__ __device__ __forceinline__ constexpr int dof2pos(int d) { return d < 6 ? 4 * (d >> 1) + 2 + (d & 1) : (d < 14 ? 4 * ((d - 6) >> 1) + ((d - 6) & 1) : d); }
static_assert(dof2pos(0) == 2 && dof2pos(5) == 11 && dof2pos(6) == 0 && dof2pos(13) == 13 && pos2dof(7) == 3 && pos2dof(9) == 11 && pos2dof(15) == 15, "row layout");

template <int P>
__device__ __forceinline__ float bcast(float x) {  // value of lane P of this 16-lane row, on every lane of the row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + P, 0xF, 0xF, false));
}
template <int P>
__device__ __forceinline__ int bcast_i(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x150 + P, 0xF, 0xF, false); }
// Every lane must end up with the SAME BITS: alpha, the convergence tests and the active-set votes are computed redundantly by the
// lanes of a row from these sums, and a row whose lanes disagree in the last bit can part ways at a branch (round 3 soak: one env
// in 2.5e8 env-steps — a line search whose Newton step landed exactly on its bracket on one lane and one ulp inside it on the
// others; that lane bisected, took alpha = 3e-7, declared itself done, and the row iterated to the cap on an inconsistent
// qacc).  The butterfly is symmetric — partners add the same two partial sums, and a + b == b + a — but only if the additions
// stay additions: with contraction on, `rsum(p * q)` became fma(p, q, partner's ROUNDED product) on each lane, which is not
// symmetric.  Hence contract(off) here and in DevCtx's group sums (mz_device.h).
__device__ __forceinline__ float rsum(float x) {  // all-reduce over the 16 lanes of the row
#pragma clang fp contract(off)
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xF, 0xF, true));  // row_half_mirror
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x140, 0xF, 0xF, true));  // row_mirror
  return x;
}

// ---- DPP operands fused into the arithmetic (hand-placed: the compiler's DPP combiner does not fuse a `row_newbcast` move into
// an FMA on gfx950 — V_FMA_F32 is still three-address when the combiner runs — and emits  v_mov 0 ; s_nop 1 ; v_mov_dpp ; v_fma,
// five issue slots, for what the hardware does in one:  v_fmac_f32_dpp acc, x, m row_newbcast:P  =  acc += x[lane P of my row] * m).
// Hazard rule kept by construction and checked on the built code by tools/check_dpp_hazards.py (tests/test_capi_and_emu.py):
// a VGPR read by a DPP instruction — any operand — must not have been written by the two preceding VALU issue slots.  Every
// block below therefore opens with `s_nop 1` (its inputs may be fresh) and never reads a register it wrote less than three
// instructions earlier.
#define MZ_DPP_TAIL " row_mask:0xf bank_mask:0xf\n\t"
// matvec: three accumulators round-robin (a dependent fmac_dpp chain would need two idle slots per link)
#define MZ_MV_MUL(acc, k) "v_mul_f32_dpp %" #acc ", %[x], %[a" #k "] row_newbcast:" #k MZ_DPP_TAIL
#define MZ_MV_FMA(acc, k) "v_fmac_f32_dpp %" #acc ", %[x], %[a" #k "] row_newbcast:" #k MZ_DPP_TAIL

// y_r = sum_k A[r][k] x_k with row r of A in registers (Arow) and x distributed one entry per lane: 14 / 16 instructions
__device__ __forceinline__ float matvec(const float (&A)[14], float x) {
  float s0, s1, s2;
  asm("s_nop 1\n\t" MZ_MV_MUL(0, 0) MZ_MV_MUL(1, 1) MZ_MV_MUL(2, 2) MZ_MV_FMA(0, 3) MZ_MV_FMA(1, 4) MZ_MV_FMA(2, 5) MZ_MV_FMA(0, 6) MZ_MV_FMA(1, 7)
      MZ_MV_FMA(2, 8) MZ_MV_FMA(0, 9) MZ_MV_FMA(1, 10) MZ_MV_FMA(2, 11) MZ_MV_FMA(0, 12) MZ_MV_FMA(1, 13)
      : "=&v"(s0), "=&v"(s1), "=&v"(s2)
      : [x] "v"(x), [a0] "v"(A[0]), [a1] "v"(A[1]), [a2] "v"(A[2]), [a3] "v"(A[3]), [a4] "v"(A[4]), [a5] "v"(A[5]), [a6] "v"(A[6]), [a7] "v"(A[7]),
        [a8] "v"(A[8]), [a9] "v"(A[9]), [a10] "v"(A[10]), [a11] "v"(A[11]), [a12] "v"(A[12]), [a13] "v"(A[13]));
  return (s0 + s1) + s2;
}
__device__ __forceinline__ float matvec(const float (&A)[16], float x) {
  float s0, s1, s2;
  asm("s_nop 1\n\t" MZ_MV_MUL(0, 0) MZ_MV_MUL(1, 1) MZ_MV_MUL(2, 2) MZ_MV_FMA(0, 3) MZ_MV_FMA(1, 4) MZ_MV_FMA(2, 5) MZ_MV_FMA(0, 6) MZ_MV_FMA(1, 7)
      MZ_MV_FMA(2, 8) MZ_MV_FMA(0, 9) MZ_MV_FMA(1, 10) MZ_MV_FMA(2, 11) MZ_MV_FMA(0, 12) MZ_MV_FMA(1, 13) MZ_MV_FMA(2, 14) MZ_MV_FMA(0, 15)
      : "=&v"(s0), "=&v"(s1), "=&v"(s2)
      : [x] "v"(x), [a0] "v"(A[0]), [a1] "v"(A[1]), [a2] "v"(A[2]), [a3] "v"(A[3]), [a4] "v"(A[4]), [a5] "v"(A[5]), [a6] "v"(A[6]), [a7] "v"(A[7]),
        [a8] "v"(A[8]), [a9] "v"(A[9]), [a10] "v"(A[10]), [a11] "v"(A[11]), [a12] "v"(A[12]), [a13] "v"(A[13]), [a14] "v"(A[14]), [a15] "v"(A[15]));
  return (s0 + s1) + s2;
}

// elimination step of one pivot: h_k += nli * h_k[lane P] for the right-hand side and every column that can still be non-zero
// in the pivot row — one instruction per column, the columns independent of each other (overloads by column count)
#define MZ_EL(i) "v_fmac_f32_dpp %" #i ", %" #i ", %[li] row_newbcast:%[p]" MZ_DPP_TAIL
#define MZ_EL_IN [li] "v"(nli), [p] "n"(P)
template <int P> __device__ __forceinline__ void elim(float nli, float& b) { asm("s_nop 1\n\t" MZ_EL(0) : "+v"(b) : MZ_EL_IN); }
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0) { asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) : "+v"(b), "+v"(h0) : MZ_EL_IN); }
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0, float& h1) {
  asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) MZ_EL(2) : "+v"(b), "+v"(h0), "+v"(h1) : MZ_EL_IN);
}
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0, float& h1, float& h2) {
  asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) MZ_EL(2) MZ_EL(3) : "+v"(b), "+v"(h0), "+v"(h1), "+v"(h2) : MZ_EL_IN);
}
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0, float& h1, float& h2, float& h3) {
  asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) MZ_EL(2) MZ_EL(3) MZ_EL(4) : "+v"(b), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : MZ_EL_IN);
}
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0, float& h1, float& h2, float& h3, float& h4) {
  asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) MZ_EL(2) MZ_EL(3) MZ_EL(4) MZ_EL(5) : "+v"(b), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4) : MZ_EL_IN);
}
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0, float& h1, float& h2, float& h3, float& h4, float& h5) {
  asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) MZ_EL(2) MZ_EL(3) MZ_EL(4) MZ_EL(5) MZ_EL(6)
      : "+v"(b), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5) : MZ_EL_IN);
}
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0, float& h1, float& h2, float& h3, float& h4, float& h5, float& h6) {
  asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) MZ_EL(2) MZ_EL(3) MZ_EL(4) MZ_EL(5) MZ_EL(6) MZ_EL(7)
      : "+v"(b), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6) : MZ_EL_IN);
}
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0, float& h1, float& h2, float& h3, float& h4, float& h5, float& h6, float& h7) {
  asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) MZ_EL(2) MZ_EL(3) MZ_EL(4) MZ_EL(5) MZ_EL(6) MZ_EL(7) MZ_EL(8)
      : "+v"(b), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : MZ_EL_IN);
}
template <int P> __device__ __forceinline__ void elim(float nli, float& b, float& h0, float& h1, float& h2, float& h3, float& h4, float& h5, float& h6, float& h7, float& h8) {
  asm("s_nop 1\n\t" MZ_EL(0) MZ_EL(1) MZ_EL(2) MZ_EL(3) MZ_EL(4) MZ_EL(5) MZ_EL(6) MZ_EL(7) MZ_EL(8) MZ_EL(9)
      : "+v"(b), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7), "+v"(h8) : MZ_EL_IN);
}

// contribution of contact C (owner: lane P of the row) to row r of the Hessian and to the gradient.  cg[8] = the contact's gradient
// block (3) and curvature block (5), read on the owner lane; (j0, j1, j2) = this lane's own column of the contact's Jacobian.
//   fold_t:  t = g . j,  (t0, t1, t2) = (W j)         ten instructions, four independent chains
//   fold_h:  Hrow[k] += t0 j0[lane k] + t1 j1[lane k] + t2 j2[lane k]   three instructions per column, the columns independent
#define MZ_FA_MUL(o, c, j) "v_mul_f32_dpp %" #o ", %[" #c "], %[" #j "] row_newbcast:%[p]" MZ_DPP_TAIL
#define MZ_FA_FMA(o, c, j) "v_fmac_f32_dpp %" #o ", %[" #c "], %[" #j "] row_newbcast:%[p]" MZ_DPP_TAIL
template <int P>
__device__ __forceinline__ void fold_t(const float (&cg)[8], float j0, float j1, float j2, float& t, float& t0, float& t1, float& t2) {
  asm("s_nop 1\n\t" MZ_FA_MUL(0, c0, j0) MZ_FA_MUL(1, c3, j0) MZ_FA_MUL(2, c4, j0) MZ_FA_MUL(3, c5, j0) MZ_FA_FMA(0, c1, j1) MZ_FA_FMA(1, c4, j1)
      MZ_FA_FMA(2, c6, j1) MZ_FA_FMA(3, c7, j2) MZ_FA_FMA(0, c2, j2) MZ_FA_FMA(1, c5, j2)
      : "=&v"(t), "=&v"(t0), "=&v"(t1), "=&v"(t2)
      : [c0] "v"(cg[0]), [c1] "v"(cg[1]), [c2] "v"(cg[2]), [c3] "v"(cg[3]), [c4] "v"(cg[4]), [c5] "v"(cg[5]), [c6] "v"(cg[6]), [c7] "v"(cg[7]),
        [j0] "v"(j0), [j1] "v"(j1), [j2] "v"(j2), [p] "n"(P));
}
// gradient part alone (the refinement step of ant_solve_rows_core): t = g . j with g = three registers of the owner lane P
template <int P>
__device__ __forceinline__ float fold_g(float g0, float g1, float g2, float j0, float j1, float j2) {
  float a0, a1, a2;  // three independent products (a dependent fmac_dpp chain would read its accumulator inside the DPP hazard window)
  asm("s_nop 1\n\t" MZ_FA_MUL(0, c0, j0) MZ_FA_MUL(1, c1, j1) MZ_FA_MUL(2, c2, j2)
      : "=&v"(a0), "=&v"(a1), "=&v"(a2) : [c0] "v"(g0), [c1] "v"(g1), [c2] "v"(g2), [j0] "v"(j0), [j1] "v"(j1), [j2] "v"(j2), [p] "n"(P));
  return (a0 + a1) + a2;
}
#define MZ_FH(k, j, t) "v_fmac_f32_dpp %" #k ", %[" #j "], %[" #t "] row_newbcast:" #k MZ_DPP_TAIL
#define MZ_FH14(j, t) MZ_FH(0, j, t) MZ_FH(1, j, t) MZ_FH(2, j, t) MZ_FH(3, j, t) MZ_FH(4, j, t) MZ_FH(5, j, t) MZ_FH(6, j, t) MZ_FH(7, j, t) \
                      MZ_FH(8, j, t) MZ_FH(9, j, t) MZ_FH(10, j, t) MZ_FH(11, j, t) MZ_FH(12, j, t) MZ_FH(13, j, t)
#define MZ_FH16(j, t) MZ_FH14(j, t) MZ_FH(14, j, t) MZ_FH(15, j, t)
#define MZ_FH_IN [j0] "v"(j0), [j1] "v"(j1), [j2] "v"(j2), [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2)
__device__ __forceinline__ void fold_h(float (&H)[14], float j0, float j1, float j2, float t0, float t1, float t2) {
  asm("s_nop 1\n\t" MZ_FH14(j0, t0) MZ_FH14(j1, t1) MZ_FH14(j2, t2)
      : "+v"(H[0]), "+v"(H[1]), "+v"(H[2]), "+v"(H[3]), "+v"(H[4]), "+v"(H[5]), "+v"(H[6]), "+v"(H[7]), "+v"(H[8]), "+v"(H[9]), "+v"(H[10]),
        "+v"(H[11]), "+v"(H[12]), "+v"(H[13])
      : MZ_FH_IN);
}
__device__ __forceinline__ void fold_h(float (&H)[16], float j0, float j1, float j2, float t0, float t1, float t2) {
  asm("s_nop 1\n\t" MZ_FH16(j0, t0) MZ_FH16(j1, t1) MZ_FH16(j2, t2)
      : "+v"(H[0]), "+v"(H[1]), "+v"(H[2]), "+v"(H[3]), "+v"(H[4]), "+v"(H[5]), "+v"(H[6]), "+v"(H[7]), "+v"(H[8]), "+v"(H[9]), "+v"(H[10]),
        "+v"(H[11]), "+v"(H[12]), "+v"(H[13]), "+v"(H[14]), "+v"(H[15])
      : MZ_FH_IN);
}

// ---- the Hessian update on the matrix cores (round 5).  `v_mfma_f32_16x16x1_4b_f32` is FOUR independent 16 x 16 rank-1 updates,
// D_b += A_b (x) B_b, block b = lanes 16 b .. 16 b + 15 of the wave — exactly the solver's layout: the wave's four 16-lane rows (four
// envs at 16 lanes per env; mirrored rows of one env at 32 / 64), lane 16 b + i = position i.  With A = (W j)_a and B = j_a of a contact
// slot (what fold_t leaves in every lane) one instruction is  H_b[i][k] += t_a[i] j_a[k]  for all four rows at once: three per slot
// where fold_h spends 3 x 16 fused DPP multiply-adds — on another pipe, next to which the vector ALU goes on with the next slot's
// fold_t.  The accumulator comes back in the MFMA's own layout — register 4 b + v of lane 16 g + j = D_b[4 g + v][j] (probed:
// tools/mfma_fold_bench.hip) — i.e., H_b being symmetric once a slot's three rows are in, lane 16 g + j holds H_b[j][4 g .. 4 g + 3]:
// a 4 x 4 transpose of register groups across the wave's four rows (8 v_permlane32_swap + 8 v_permlane16_swap) puts row j of block
// b on lane 16 b + j, where the elimination wants it.  Measured in isolation (profiles/r05/mfma_fold_bench.txt, one wave per SIMD,
// fold_t included): 0.97 / 0.81 / 0.75 / 0.73 of the DPP fold's cycles at 2 / 4 / 6 / 8 slots.
// Hazard: fold_t is inline asm — the compiler's hazard recognizer does not see its VALU writes in front of the MFMA that reads them
// (the benchmark's first version read stale operands, erratically); `mfma_operands_ready` puts the wait states there by hand.
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void mfma_operands_ready(float& t0, float& t1, float& t2) { asm("s_nop 3" : "+v"(t0), "+v"(t1), "+v"(t2)); }
__device__ __forceinline__ void fold_h_mfma(v16f& acc, float j0, float j1, float j2, float t0, float t1, float t2) {
  mfma_operands_ready(t0, t1, t2);
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32(t0, j0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32(t1, j1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32(t2, j2, acc, 0, 0, 0);
}
// accumulator (MFMA layout) -> this lane's row: out[4 g + v] = H_b[row of this lane][4 g + v]
__device__ __forceinline__ void mfma_rows(const v16f& acc, float (&out)[16]) {
  float G[4][4];
#pragma unroll
  for (int b = 0; b < 4; b++)
#pragma unroll
    for (int v = 0; v < 4; v++) G[b][v] = acc[4 * b + v];
#pragma unroll
  for (int v = 0; v < 4; v++) {  // halves of the wave: register groups (0, 2) and (1, 3) change places between lanes 0-31 and 32-63
    auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_int(G[0][v]), __float_as_int(G[2][v]), false, false);
    G[0][v] = __int_as_float(r0[0]); G[2][v] = __int_as_float(r0[1]);
    auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_int(G[1][v]), __float_as_int(G[3][v]), false, false);
    G[1][v] = __int_as_float(r1[0]); G[3][v] = __int_as_float(r1[1]);
  }
#pragma unroll
  for (int v = 0; v < 4; v++) {  // 16-lane rows inside each half: groups (0, 1) and (2, 3)
    auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_int(G[0][v]), __float_as_int(G[1][v]), false, false);
    G[0][v] = __int_as_float(r0[0]); G[1][v] = __int_as_float(r0[1]);
    auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_int(G[2][v]), __float_as_int(G[3][v]), false, false);
    G[2][v] = __int_as_float(r1[0]); G[3][v] = __int_as_float(r1[1]);
  }
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int v = 0; v < 4; v++) out[4 * g + v] = G[g][v];
}

// one Gauss-Jordan pivot P on the row-distributed system (Hrow | b): every other row gets rid of column P.
// COLS... = the columns that can still be non-zero in the pivot row (compile-time list: the arrow structure).
template <int P, int N, int... COLS>
__device__ __forceinline__ void pivot(int r, float (&Hrow)[N], float& b, float& dinv, float (&mlt)[N]) {
  // `r == P` is compared HERE, next to its two selects (one v_cmp into VCC): left to itself the compiler hoists the 14-16 lane masks of
  // a solve out of every loop, runs out of scalar registers, spills them to VGPR lanes and restores each with two v_readlane per
  // pivot (round 4: 387 -> 315 v_readlane in the kernel, 48 -> 42 accumulation registers, 0.2949 -> 0.2937 ms)
  asm("" : "+v"(r));
  const float d = bcast<P>(Hrow[P]);
  const float ri = 1.0f / fmaxf(d, 1e-30f);
  const float nli = (r == P) ? 0.f : -(Hrow[P] * ri);
  elim<P>(nli, b, Hrow[COLS]...);
  dinv = (r == P) ? ri : dinv;
  mlt[P] = nli;  // this row's multiplier of pivot P: kept for a second right-hand side (resolve_rows); dead code where nobody asks
}

// H x = b, arrow-structured SPD H, row-distributed in POSITION order (pos2dof above): the hinges (positions 0 1 | 4 5 | 8 9 | 12 13)
// are eliminated first, each touching its partner and the hub columns (root: positions 2 3 6 7 10 11) only
__device__ __forceinline__ float solve_rows(int r, float (&Hrow)[14], float b, float (&mlt)[14], float& dinv) {
  dinv = 0.f;
  // the four legs do not couple: their hip pivots (then their ankle pivots) are independent chains — issued next to each
  // other so that the reciprocal / broadcast latencies of one hide behind the others
  pivot<0, 14, 1, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<4, 14, 5, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<8, 14, 9, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<12, 14, 13, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<1, 14, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<5, 14, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<9, 14, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<13, 14, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<2, 14, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<3, 14, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<6, 14, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<7, 14, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<10, 14, 11>(r, Hrow, b, dinv, mlt);
  pivot<11, 14>(r, Hrow, b, dinv, mlt);
  return b * dinv;
}
__device__ __forceinline__ float solve_rows(int r, float (&Hrow)[14], float b) {
  float mlt[14], dinv;
  return solve_rows(r, Hrow, b, mlt, dinv);
}
// the same with one movable block: its two slides (positions 14, 15) belong to the hub (a robot-block contact couples them with the
// root and with one leg), eliminated between the legs and the root
__device__ __forceinline__ float solve_rows(int r, float (&Hrow)[16], float b, float (&mlt)[16], float& dinv) {
  dinv = 0.f;
  pivot<0, 16, 1, 2, 3, 6, 7, 10, 11, 14, 15>(r, Hrow, b, dinv, mlt);
  pivot<4, 16, 5, 2, 3, 6, 7, 10, 11, 14, 15>(r, Hrow, b, dinv, mlt);
  pivot<8, 16, 9, 2, 3, 6, 7, 10, 11, 14, 15>(r, Hrow, b, dinv, mlt);
  pivot<12, 16, 13, 2, 3, 6, 7, 10, 11, 14, 15>(r, Hrow, b, dinv, mlt);
  pivot<1, 16, 2, 3, 6, 7, 10, 11, 14, 15>(r, Hrow, b, dinv, mlt);
  pivot<5, 16, 2, 3, 6, 7, 10, 11, 14, 15>(r, Hrow, b, dinv, mlt);
  pivot<9, 16, 2, 3, 6, 7, 10, 11, 14, 15>(r, Hrow, b, dinv, mlt);
  pivot<13, 16, 2, 3, 6, 7, 10, 11, 14, 15>(r, Hrow, b, dinv, mlt);
  pivot<14, 16, 15, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<15, 16, 2, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<2, 16, 3, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<3, 16, 6, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<6, 16, 7, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<7, 16, 10, 11>(r, Hrow, b, dinv, mlt);
  pivot<10, 16, 11>(r, Hrow, b, dinv, mlt);
  pivot<11, 16>(r, Hrow, b, dinv, mlt);
  return b * dinv;
}

__device__ __forceinline__ float solve_rows(int r, float (&Hrow)[16], float b) {
  float mlt[16], dinv;
  return solve_rows(r, Hrow, b, mlt, dinv);
}
// A second right-hand side through the SAME elimination (iterative refinement of the Newton step, ant_solve_rows_core): the pivots
// of solve_rows in their order, each one instruction on the right-hand side alone — b_r += mlt_P(r) * b[lane P] — then the diagonal.
template <int P>
__device__ __forceinline__ void repivot(float m, float& b) { elim<P>(m, b); }
__device__ __forceinline__ float resolve_rows(const float (&mlt)[16], float dinv, float b) {
  repivot<0>(mlt[0], b); repivot<4>(mlt[4], b); repivot<8>(mlt[8], b); repivot<12>(mlt[12], b);
  repivot<1>(mlt[1], b); repivot<5>(mlt[5], b); repivot<9>(mlt[9], b); repivot<13>(mlt[13], b);
  repivot<14>(mlt[14], b); repivot<15>(mlt[15], b);
  repivot<2>(mlt[2], b); repivot<3>(mlt[3], b); repivot<6>(mlt[6], b); repivot<7>(mlt[7], b); repivot<10>(mlt[10], b); repivot<11>(mlt[11], b);
  return b * dinv;
}
__device__ __forceinline__ float resolve_rows(const float (&mlt)[14], float dinv, float b) {
  repivot<0>(mlt[0], b); repivot<4>(mlt[4], b); repivot<8>(mlt[8], b); repivot<12>(mlt[12], b);
  repivot<1>(mlt[1], b); repivot<5>(mlt[5], b); repivot<9>(mlt[9], b); repivot<13>(mlt[13], b);
  repivot<2>(mlt[2], b); repivot<3>(mlt[3], b); repivot<6>(mlt[6], b); repivot<7>(mlt[7], b); repivot<10>(mlt[10], b); repivot<11>(mlt[11], b);
  return b * dinv;
}

// S . F_q with F_q = the six F registers of lane Q of this row: sum_k F[k](lane Q) * S[k](mine) — six fused DPP multiply-adds on
// three accumulators (the DPP read-after-write rule: ant_newton_rows.h; checked by tools/check_dpp_hazards.py)
#define MZ_D6_MUL(o, k) "v_mul_f32_dpp %" #o ", %[f" #k "], %[s" #k "] row_newbcast:%[q]" MZ_DPP_TAIL
#define MZ_D6_FMA(o, k) "v_fmac_f32_dpp %" #o ", %[f" #k "], %[s" #k "] row_newbcast:%[q]" MZ_DPP_TAIL
template <int Q>
__device__ __forceinline__ float dot6_from(const float (&F)[6], const float (&S)[6]) {
  float a0, a1, a2;
  asm("s_nop 1\n\t" MZ_D6_MUL(0, 0) MZ_D6_MUL(1, 1) MZ_D6_MUL(2, 2) MZ_D6_FMA(0, 3) MZ_D6_FMA(1, 4) MZ_D6_FMA(2, 5)
      : "=&v"(a0), "=&v"(a1), "=&v"(a2)
      : [f0] "v"(F[0]), [f1] "v"(F[1]), [f2] "v"(F[2]), [f3] "v"(F[3]), [f4] "v"(F[4]), [f5] "v"(F[5]), [s0] "v"(S[0]), [s1] "v"(S[1]), [s2] "v"(S[2]),
        [s3] "v"(S[3]), [s4] "v"(S[4]), [s5] "v"(S[5]), [q] "n"(Q));
  return (a0 + a1) + a2;
}
// pyramidal contact: cost, and optionally gradient block g[3] / curvature block W[5] (same as contact_eval)
__device__ __forceinline__ float ceval(float D, float u0, float u1, float u2) {
  float r0 = u0 + u1, r1 = u0 - u1, r2 = u0 + u2, r3 = u0 - u2;
  float c = 0.f;
  if (r0 < 0.f) c += r0 * r0;
  if (r1 < 0.f) c += r1 * r1;
  if (r2 < 0.f) c += r2 * r2;
  if (r3 < 0.f) c += r3 * r3;
  return 0.5f * D * c;
}

// bit select: m = all ones -> a, m = 0 -> b.  One v_bfi_b32 — and, unlike `cond ? a : b` on values only one role of lanes
// needs, nothing the compiler can turn into a divergent branch with the operand's computation sunk into it (it did: the first
// version of this file ran its role selects as ~60 exec-mask branches per evaluation)
__device__ __forceinline__ float bsel(int m, float a, float b) { return __int_as_float((__float_as_int(a) & m) | (__float_as_int(b) & ~m)); }

// The 24-float record of one contact of a robot geom (ant_solve_rows_core, WR mode): rec[8 a + 0..5] = wrench of row a about the
// torso origin, sr [r x f_a; f_a] (f_0 = n, f_1 = mu t1, f_2 = mu t2; sr: +1 floor -> geom, -1 geom -> wall / geom -> movable block:
// the side of the pair the robot is on), rec[8 a + 6] = reference acceleration of row a, rec[7] = D,
// rec[15] = leg (7: none) | body class << 3 | 64 when the partner is the movable block (its slide lanes then see the reaction).
// Same arithmetic as con_row_item (ant_dyn.h): the row's velocity J qvel is wrench . (spatial velocity of the touching body) —
// minus, for the block, the force direction times the block's slide velocities.  kind: 0 floor, 1 wall / platform, 2 block.
template <int NB>
__device__ __forceinline__ void contact_record(const AntDev& K, const float* pos, const float* n, float dist, int kind, const float* hint, int cls, int leg,
                                               const float* vb, const float* vblk, float tran, float* rec) {
  // the pair's parameters: BOTH sets by scalar loads, then value selects (`const PairDev& P = kind == 0 ? K.floor : K.wall` came out as
  // an address select followed by dependent vector-memory loads: three round trips to the cache on the contact's critical path)
  const int mk = -(int)(kind == 0);
  struct { float margin, mu, K, B; } P = {bsel(mk, K.floor.margin, K.wall.margin), bsel(mk, K.floor.mu, K.wall.mu), bsel(mk, K.floor.K, K.wall.K),
                                         bsel(mk, K.floor.B, K.wall.B)};
  float si[7];
#pragma unroll
  for (int k = 0; k < 7; k++) si[k] = bsel(mk, K.floor.solimp[k], K.wall.solimp[k]);
  float t1[3], t2[3];
  make_tangents(n, hint, t1, t2);
  const float sr = bsel(mk, 1.f, -1.f);
  float omi;
  const float imp = impedance_pair(si, fabsf(dist - P.margin), &omi);
  if (NB == 1 && kind == 2) tran += K.block_bw_tran;
  const float Rr = fmaxf(1e-15f, omi / imp * (tran + P.mu * P.mu * tran));
  rec[7] = 1.0f / (2.f * P.mu * P.mu * Rr);  // [ASSUME-3]
  rec[15] = __int_as_float((leg < 0 ? 7 : leg) | (cls << 3) | ((NB == 1 && kind == 2) ? 64 : 0));
  rec[23] = 0.f;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float sc = (a == 0 ? 1.f : P.mu) * sr;
    const float* dir = a == 0 ? n : (a == 1 ? t1 : t2);
    const float f[3] = {sc * dir[0], sc * dir[1], sc * dir[2]};
    float m[3];
    cross3f(m, pos, f);
    float* w = rec + 8 * a;
    w[0] = m[0]; w[1] = m[1]; w[2] = m[2]; w[3] = f[0]; w[4] = f[1]; w[5] = f[2];
    float vel = m[0] * vb[0] + m[1] * vb[1] + m[2] * vb[2] + f[0] * vb[3] + f[1] * vb[4] + f[2] * vb[5];
    if constexpr (NB == 1) {
      if (kind == 2) {  // the block moves along its slides: its point velocity enters with the opposite sign
#pragma unroll
        for (int sl = 0; sl < 2; sl++) vel -= (K.block_axis[sl] == 0 ? f[0] : (K.block_axis[sl] == 1 ? f[1] : f[2])) * vblk[sl];
      }
    }
    float aref = -P.B * vel;
    if (a == 0) aref -= P.K * imp * (dist - P.margin);
    w[6] = aref;
  }
}

// A staged robot contact (round 5): what the geom's lane knows, for the slot's owner lane to turn into the contact's rows.
//   q[0..2] position (torso-relative), q[3..5] normal, q[6] distance, q[7] kind | (cls + 1) << 4 | (leg + 1) << 8 as an int,
//   q[8..10] tangent hint, q[11..16] spatial velocity of the touching body at the torso origin, q[17] the body's contact weight
__device__ __forceinline__ void contact_raw_store(float* q, const float* pos, const float* n, float dist, int kind, const float* hint, int cls, int leg,
                                                  const float* vb, float tran) {
  float4* dst = reinterpret_cast<float4*>(q);
  dst[0] = make_float4(pos[0], pos[1], pos[2], n[0]);
  dst[1] = make_float4(n[1], n[2], dist, __int_as_float(kind | ((cls + 1) << 4) | ((leg + 1) << 8)));
  dst[2] = make_float4(hint[0], hint[1], hint[2], vb[0]);
  dst[3] = make_float4(vb[1], vb[2], vb[3], vb[4]);
  dst[4] = make_float4(vb[5], tran, 0.f, 0.f);
}
}  // namespace rows

// wrench record of contact slot c (WR mode of ant_solve_rows_core): 24 floats, packed one after the other in the cJ block (for the
// plain ant, NCOL = 8, that IS cJ[c]; with a movable block cJ's slots are 30 floats wide and would leave every other record off a
// 16-byte boundary)
template <class S>
__device__ __forceinline__ float* wr_record(S& s, int c) { return &s.cJ[0][0][0] + 24 * c; }
template <class S>
__device__ __forceinline__ const float* wr_record(const S& s, int c) { return &s.cJ[0][0][0] + 24 * c; }

// Per-lane constants of the hinge lanes' joint-limit rows, loaded once per step into DevCtx::lc (run-time indexed reads of the
// constant block inside the 20 evaluations were dependent vector-memory loads): range and inverse weight of the lane's own hinge.
enum { LC_LO = 11, LC_HI = 12, LC_DOFW = 13 };
// UNIFORM constants of the model that every forward evaluation reads — margins, the maze grid's geometry, armature / damping /
// gravity, the solver's settings — held in vector registers next to the per-lane ones (round 5).  Left in the constant block they
// were scalar loads INSIDE the evaluation (the step's 100-odd spilled scalar registers leave the compiler no room to keep them):
// ten wait groups per evaluation, each the scalar cache's latency in front of the instruction that needs the value, with one wave
// per SIMD and nothing else to issue.  The empty asm makes a value opaque: it cannot be rematerialised by loading it again.
// Measured (A / B, AntUMaze-v0 4096 envs): 0.2516 -> 0.2483..0.2499 ms.  NOT in the two-waves-per-SIMD instantiation (cx.mfma), whose
// 256 registers have no room for 18 more: it reads the constant block where it needs a value, as before (`ant_u` below hands out one
// or the other; held there as well, 8192 envs lose 1 %).
enum { LU_FLOORM = 14, LU_WALLM, LU_SCALE, LU_INVS, LU_TX, LU_TY, LU_HXY, LU_HZ, LU_CZ, LU_GRID, LU_ARM, LU_DAMP, LU_GZ, LU_TOL, LU_RTOL, LU_ISC,
       LU_ITERS, LU_LS, LU_N };
struct AntU {
  float floor_margin, wall_margin, scale, inv_scale_xy, tx, ty, half_xy, half_z, center_z;
  int rows, cols;
  bool elevated;
  float armature, damping, gz, tol, rtol, inv_scale;
  int max_iter, ls_iter, ls_fast_iters, ls_fast, trust_exact;  // (scalar registers: wave-uniform loop bounds)
};
template <int G, bool PROF>
__device__ __forceinline__ AntU ant_u(const DevCtx<G, PROF>& cx, const AntDev& K) {
  const MazeDev& z = K.maze;
  if (cx.mfma)
    return {K.floor.margin, K.wall.margin, z.scale, 1.0f / z.scale, z.tx, z.ty, z.half_xy, z.half_z, z.center_z, z.rows, z.cols, z.elevated != 0,
            K.armature, K.damping, K.gz, K.tol, K.rtol, K.inv_scale, K.max_iter, K.ls_iter, K.ls_fast_iters, K.ls_fast, K.trust_exact};
  const int g = __float_as_int(cx.lc[LU_GRID]);
  const int a = __builtin_amdgcn_readfirstlane(__float_as_int(cx.lc[LU_ITERS])), b = __builtin_amdgcn_readfirstlane(__float_as_int(cx.lc[LU_LS]));
  return {cx.lc[LU_FLOORM], cx.lc[LU_WALLM], cx.lc[LU_SCALE], cx.lc[LU_INVS], cx.lc[LU_TX], cx.lc[LU_TY], cx.lc[LU_HXY], cx.lc[LU_HZ], cx.lc[LU_CZ],
          g & 255, (g >> 8) & 255, ((g >> 16) & 1) != 0, cx.lc[LU_ARM], cx.lc[LU_DAMP], cx.lc[LU_GZ], cx.lc[LU_TOL], cx.lc[LU_RTOL], cx.lc[LU_ISC],
          a, b & 1023, (b >> 10) & 1023, (b >> 20) & 1023, (b >> 30) & 1};
}
__device__ __forceinline__ float lu_hold(float x) { asm("" : "+v"(x)); return x; }
template <int G, bool PROF>
__device__ __forceinline__ void ant_limit_consts(const AntDev& K, DevCtx<G, PROF>& cx) {
  static_assert(LU_N <= DevCtx<G, PROF>::NLC, "DevCtx::lc too small");
  const int p = cx.l & 15, l = p >> 2, d = p & 1;
  cx.lc[LC_LO] = d ? K.ank_lo[l] : K.hip_lo; cx.lc[LC_HI] = d ? K.ank_hi[l] : K.hip_hi; cx.lc[LC_DOFW] = d ? K.dofw_ank : K.dofw_hip;
  if (cx.mfma) return;
  const MazeDev& z = K.maze;
  cx.lc[LU_FLOORM] = lu_hold(K.floor.margin); cx.lc[LU_WALLM] = lu_hold(K.wall.margin);
  cx.lc[LU_SCALE] = lu_hold(z.scale); cx.lc[LU_INVS] = lu_hold(1.0f / z.scale); cx.lc[LU_TX] = lu_hold(z.tx); cx.lc[LU_TY] = lu_hold(z.ty);
  cx.lc[LU_HXY] = lu_hold(z.half_xy); cx.lc[LU_HZ] = lu_hold(z.half_z); cx.lc[LU_CZ] = lu_hold(z.center_z);
  cx.lc[LU_GRID] = lu_hold(__int_as_float(z.rows | (z.cols << 8) | ((z.elevated ? 1 : 0) << 16)));
  cx.lc[LU_ARM] = lu_hold(K.armature); cx.lc[LU_DAMP] = lu_hold(K.damping); cx.lc[LU_GZ] = lu_hold(K.gz);
  cx.lc[LU_TOL] = lu_hold(K.tol); cx.lc[LU_RTOL] = lu_hold(K.rtol); cx.lc[LU_ISC] = lu_hold(K.inv_scale);
  auto c10 = [](int v) { return v < 0 ? 0 : (v > 1023 ? 1023 : v); };
  cx.lc[LU_ITERS] = lu_hold(__int_as_float(K.max_iter < 0 ? 0 : K.max_iter));
  cx.lc[LU_LS] = lu_hold(__int_as_float(c10(K.ls_iter) | (c10(K.ls_fast_iters) << 10) | (c10(K.ls_fast) << 20) | ((K.trust_exact ? 1 : 0) << 30)));
}

// Constraint rows of a contact of the block's OWN enumerators (floor -> block, maze box -> block, slide limit; kinds 3, 4, 6 of
// con_row_item in ant_dyn.h — the same arithmetic) straight from the contact's staged geometry into the owner lane's registers:
// only the block's two columns of the Jacobian exist, so the generic 3 x (hub + 2) row builder and its LDS round trip are skipped.
template <int NB>
__device__ __forceinline__ void block_rows_direct(const AntDev& K, const AntScratchCoreT<NB>& s, int c, float (&jb)[3][2], float (&ar)[3], float& Dout) {
  using D = AntDims<NB>;
  const int src = s.csrc[c];
  const float* q = src >= 0 ? con_stage<NB>(s, src) : &s.cY[c][0][0];
  const float n[3] = {q[3], q[4], q[5]}, dist = q[6];
  // code = kind + 16 blk + 128 other + 2048 (multiplicity - 1): a merged entry stands for `mult` identical contact points (MERGE)
  const int code = (int)q[7], kind = code & 15, other = (code >> 7) & 15;
  const float mult = (float)((code >> 11) + 1);
  const float v0 = s.qvel[14], v1 = s.qvel[15];
  if (kind == 6) {  // slide limit: one frictionless row riding as a pyramid with vanishing tangents (cD = D / 4)
    const float sg = n[0] + n[1] + n[2];
    jb[0][0] = other == 0 ? sg : 0.f; jb[0][1] = other == 1 ? sg : 0.f;
    jb[1][0] = jb[1][1] = jb[2][0] = jb[2][1] = 0.f;
    const float vel = sg * (other == 0 ? v0 : v1);
    float omi;
    const float imp = impedance_pair(K.blim_solimp, fabsf(dist - K.blim_margin), &omi);
    const float R = fmaxf(1e-15f, omi / imp * K.blim_w);
    Dout = 0.25f / R;
    ar[0] = -K.blim_B * vel - K.blim_K * imp * (dist - K.blim_margin); ar[1] = 0.f; ar[2] = 0.f;
    return;
  }
  // both pairs' parameters by scalar loads, then value selects (a reference chosen by `kind` became an address select followed
  // by dependent vector-memory loads on every block contact of every evaluation)
  const bool fl = kind == 3;
  struct { float margin, mu, K, B; } P = {fl ? K.floor.margin : K.wall.margin, fl ? K.floor.mu : K.wall.mu, fl ? K.floor.K : K.wall.K, fl ? K.floor.B : K.wall.B};
  float psi[7];
#pragma unroll
  for (int k = 0; k < 7; k++) { const float fv = K.floor.solimp[k], wv = K.wall.solimp[k]; psi[k] = fl ? fv : wv; }
  const float hint[3] = {0.f, 0.f, 0.f};
  float t1[3], t2[3];
  make_tangents(n, hint, t1, t2);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float sc = a == 0 ? 1.f : P.mu;
    const float* dir = a == 0 ? n : (a == 1 ? t1 : t2);
#pragma unroll
    for (int sl = 0; sl < 2; sl++) jb[a][sl] = sc * (K.block_axis[sl] == 0 ? dir[0] : (K.block_axis[sl] == 1 ? dir[1] : dir[2]));
    ar[a] = -P.B * (jb[a][0] * v0 + jb[a][1] * v1);
  }
  float omi;
  const float imp = impedance_pair(psi, fabsf(dist - P.margin), &omi);
  const float tran = K.block_bw_tran;
  const float R = fmaxf(1e-15f, omi / imp * (tran + P.mu * P.mu * tran));
  Dout = mult / (2.f * P.mu * P.mu * R);  // `mult` identical rows = one row with mult times the weight: same cost, gradient, curvature
  ar[0] -= P.K * imp * (dist - P.margin);
  (void)sizeof(D);
}

// The solve.  In: `Mrow` / `qfs` (below), s.warm (first evaluation of a step: MuJoCo's qacc_warmstart; later: the previous
// evaluation's solution), contacts (s.cJ, s.caref, s.cD, s.cleg, s.ncon, s.nblkcon); the joint-limit rows of the robot's hinges
// are built here from s.qpos / s.qvel.  Out: s.qas = M^-1 qfs (where needed), s.qacc, s.iters, status bits.  Requires G >= 16.
//
// NB = 1 (one movable block, 16 dofs: the row is full).  The contact list starts with the block's OWN contacts — floor corners,
// maze cells, slide limits: s.nblkcon of them, typically 10-25, none touching the robot — followed by the robot's (<= 16, owned
// by the row's lanes as for the plain ant; a robot-block contact has the block's two columns in its hub part).  A block-own
// contact sees two dofs only: its 3 x 2 Jacobian, residuals and J search live in the registers of ONE lane of the group (lane l
// owns the contacts l, l + G, ...), its gradient (2) and curvature (3 numbers) enter the rows of dofs 14 / 15 through group
// sums, and the line search adds its terms the same way.  Twenty contacts of the block thus cost five group sums per
// iteration instead of twenty folds into sixteen Hessian rows.
// `Mrow` = row r of M in position order and `qfs` = entry r of qfrc_smooth, in registers (the plain ant's forward pass builds them
// there: ant_forward_rows.h — the only caller since round 5; the wrapper that fed this solver from LDS copies written by the
// lane-group forward pass, and the WR = false paths it took, are history).
//
// WR ("wrench records", the plain ant's forward pass of ant_forward_rows.h): a contact arrives as the 24-float record its geom's
// lane wrote into s.cJ[c] — three wrenches sr [r x f_a; f_a] about the torso origin (normal, mu t1, mu t2), the three reference
// accelerations, D and the (leg, body class) of the touching geom — read by ITS owner lane only.  No Jacobian is ever stored:
//   * this lane's column of contact C is  S_p . wrench_a(C)  with S_p = the lane's own motion axis (`Sax`, in registers since the
//     mass matrix) and the wrench fetched from lane C by `row_newbcast:C` operands (rows::dot6_from: 18 fused instructions per
//     contact), masked by "dof p moves the touching body";
//   * J x, which the owner needs, is the transposed product: one row butterfly per (contact, row) over the lanes' columns.
// The staged Jacobian rows (cJ written by con_row_item, read back as jown / Jf: 16 + 6 LDS reads per lane and evaluation, on top
// of the phase that built them) are gone for the plain ant; the ant with a movable block (WR = false) keeps them.
// Per-lane conditions of the Newton loop that only skip work whose result is zero anyway (a lane that owns no contact holds D = 0,
// u = v = 0; a dof without an active limit row holds lsign = lD = 0; a row without contact C holds zero columns): as `if`s they are
// exec-mask regions — two scalar instructions, a branch and the exec hazards, per region — around a handful of vector instructions
// that the wave executes anyway as long as ONE of its lanes needs them.  With one wave per SIMD nobody fills those bubbles: the
// conditions are gone (the arithmetic adds its zeros), round 4: 0.2941 -> 0.2813 ms.  MZ_IF_OWNER marks the places (-DMZ_EXP_BRANCHY
// brings the `if`s back for an A / B run).
#ifdef MZ_EXP_BRANCHY
#define MZ_IF_OWNER(c) if (c)
#else
#define MZ_IF_OWNER(c) if (true)
#endif
template <int NB, int G, bool PROF, bool WR = false, class SCR>
__device__ __forceinline__ float ant_solve_rows_core(const DevCtx<G, PROF>& cx, const AntDev& K, SCR& s, bool compare,
                                                    const float (&Mrow)[14 + 2 * NB], const float qfs, const float (&Sax)[6], const float hq, const float hv) {
  // (hq, hv: angle and velocity of this lane's own hinge, from the forward pass's registers — the limit row reads no LDS)
  static_assert(G >= 16, "one DPP row per env at least");
  static_assert(NB <= 1, "one 16-lane row holds 14 dofs + one block's two slides");
  using namespace rows;
  using D = AntDims<NB>;
  constexpr int NR = 14 + 2 * NB;                 // positions of the row that own a dof
  constexpr int NHC = D::NH;                      // hub columns of a contact Jacobian: root 6 (+ block 2); then hip, ankle
  // block-own contacts per lane.  WR (the quad forward pass): the enumerators merge the identical rows of a face's contact points
  // (con_enum_item MERGE: a block resting on the floor against two walls is 3 entries instead of 12), one per lane of the group
  // holds them; more than G of them: flagged, the surplus dropped
  // Round 5 (WR): the block's entries live on the 16 lanes of a ROW — lane l of every row of the group owns entry l, every row holds
  // the same ones and arrives at the same bits — so their sums are 4-step row butterflies like everything else in the solver (the
  // 32-lane group sums cost a cross-row step, two v_readlane and a select each: seven per iteration plus two per line-search step), and
  // the line search's two block sums ride inside the robot contacts' (one butterfly for both).  Sixteen merged entries are ample: the
  // floor is one (four corners, one row set), a wall or platform face one each, a slide limit one — eight at the very most; more
  // flags CONTACT_OVERFLOW.  (A second entry per lane, tried: 55 more accumulation registers and spills — slower than the group sums.)
  constexpr int BW = WR ? 16 : G;                 // lanes the block's entries are dealt over
  constexpr int MB = NB ? (WR ? 1 : (D::NC + G - 1) / G) : 0;
  constexpr int MA = NB ? 2 : 1;                   // robot contacts per lane of the row (16 MA in all; an ant on its back next to the block: > 16)
#ifdef MZ_EXP_NOREFINE
  constexpr bool REFINE = false;
#else
  constexpr bool REFINE = NB == 1;                 // iterative refinement of an accepted unit step (below): the stiff (solimp .995) mazes
#endif
  const int r = cx.l & 15;                       // position of this lane (rows::pos2dof); beyond NR: spare lanes (zero rows, never pivots)
  const bool isdof = r < NR, ishinge = (r & 3) < 2 && r < 14;
  const int leg = r >> 2, d = r & 1;              // hinge lanes: own leg, 0 hip / 1 ankle
  const int nB = NB ? s.nblkcon : 0;             // block-own contacts: slots [0, nB)
  int nA = s.ncon - nB;                          // robot contacts: slots [nB, ncon), owned by the lanes 0 .. nA - 1 of the row
  if (nA > 16 * MA) { nA = 16 * MA; if (cx.l == 0) s.status |= MZ_STATUS_CONTACT_OVERFLOW; }
  if (NB && MB * BW < nB && cx.l == 0) s.status |= MZ_STATUS_CONTACT_OVERFLOW;  // block contacts without an owner lane (robot slots still start at nB)
  auto bsum = [&](float x) { if constexpr (WR) return rsum(x); else return cx.gsum(x); };
  const int ncon = nA;
  const bool any2 = MA > 1 && cx.any(nA > 16);   // some env of the wave uses the second contact slot of its lanes (wave-uniform: the slot's code is skipped otherwise)
  bool iscon[MA];                                // this lane owns robot contact r (+ 16: second slot)
  int cr[MA], cl[MA];
#pragma unroll
  for (int m = 0; m < MA; m++) { iscon[m] = r + 16 * m < ncon; cr[m] = nB + (iscon[m] ? r + 16 * m : 0); cl[m] = (!WR && iscon[m]) ? s.cleg[cr[m]] : -1; }

  // ---- per-dof vectors, own limit row
  const int ri = isdof ? pos2dof(r) : 0;          // this lane's index into the dof-ordered LDS vectors
  cx.tick(s, 12);
  // joint-limit row of this lane's own hinge (limit_item of ant_dyn.h, in registers: the row never leaves its lane)
  float lsign = 0.f, lD = 0.f, laref = 0.f;
  if (ishinge) {
    const float q = hq, lo = cx.lc[LC_LO], hi = cx.lc[LC_HI];
    float pos = 0.f;
    if (q - lo < 0.f) { lsign = 1.f; pos = q - lo; }
    else if (hi - q < 0.f) { lsign = -1.f; pos = hi - q; }
    if (lsign != 0.f) {
      float omi;
      const float imp = impedance_pair(K.lim_solimp, fabsf(pos), &omi);
      const float R = fmaxf(1e-15f, omi / imp * cx.lc[LC_DOFW]);
      lD = 1.0f / R;
      laref = -K.lim_B * (lsign * hv) - K.lim_K * imp * pos;  // (the limit's constants held in registers as well, tried: 0.2453 -> 0.2477 ms)
    }
  }
  const bool has = s.ncon > 0 || cx.gany(lsign != 0.f);
  // qacc_smooth = M^-1 qfrc_smooth by the same row elimination — only where it is used: on the first evaluation of a step
  // (MuJoCo's warm-start rule compares against it) and for an env without any constraint (then it is the answer).  The
  // Newton iteration itself works on M qacc - qfrc_smooth and never needs it.  (Round 5, tried: an unconstrained env through the
  // Newton loop instead — H = M there, the first unit step lands on the answer — so that one airborne ant does not buy its wave a
  // second elimination: nothing in the settled rollout of the bench, and the different register allocation cost 1 %.  Not taken.)
  float qas = 0.f;
  if (compare || cx.any(!has)) {
    float Hq[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) Hq[k] = Mrow[k];
    qas = solve_rows(r, Hq, qfs);
    if (isdof) s.qas[ri] = qas;
  }
  const float warm = isdof ? s.warm[ri] : 0.f;  // later evaluations start from the previous evaluation's solution
#ifdef MZ_EXP_SUBTICK2
  cx.tick(s, 0);  // limit row, qacc_smooth where needed
#endif
  // own robot contact: 3 x (NHC + 2) Jacobian rows stay in LDS (row-major, read as needed); constants in registers
  float cD[MA], ar[MA][3];
  float wr[MA][3][6];  // WR: the own contacts' three wrenches each
  int wmeta[MA];       // WR: leg (7: none) | body class << 3 | 64 if the touching geom's partner is the movable block
#pragma unroll
  for (int m = 0; m < MA; m++) wmeta[m] = 0;
  if constexpr (WR) {
#pragma unroll
    for (int m = 0; m < MA; m++) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int k = 0; k < 6; k++) wr[m][a][k] = 0.f;
        ar[m][a] = 0.f;
      }
      cD[m] = 0.f;
      if (m == 1 && !any2) continue;
      // the slot's staged contact (contact_raw_store) -> its 24-float record, here on the lane that owns it
      const float4* q = reinterpret_cast<const float4*>(wr_record(s, cr[m]));
      float raw[20];
#pragma unroll
      for (int k = 0; k < 5; k++) { const float4 v4 = q[k]; raw[4 * k] = v4.x; raw[4 * k + 1] = v4.y; raw[4 * k + 2] = v4.z; raw[4 * k + 3] = v4.w; }
      const int rmeta = __float_as_int(raw[7]);
      const float vblk[2] = {NB == 1 ? s.qvel[14] : 0.f, NB == 1 ? s.qvel[14 + (NB == 1 ? 1 : 0)] : 0.f};
      float rec[24];
      contact_record<NB>(K, raw, raw + 3, raw[6], rmeta & 15, raw + 8, ((rmeta >> 4) & 15) - 1, ((rmeta >> 8) & 15) - 1, raw + 11, vblk, raw[17], rec);
      const int on = -(int)iscon[m];
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int k = 0; k < 6; k++) wr[m][a][k] = __int_as_float(__float_as_int(rec[8 * a + k]) & on);
        ar[m][a] = __int_as_float(__float_as_int(rec[8 * a + 6]) & on);
      }
      cD[m] = __int_as_float(__float_as_int(rec[7]) & on);
      wmeta[m] = __float_as_int(rec[15]) & on;
    }
  } else {
#pragma unroll
    for (int m = 0; m < MA; m++) {
      cD[m] = iscon[m] ? s.cD[cr[m]] : 0.f;
#pragma unroll
      for (int a = 0; a < 3; a++) ar[m][a] = iscon[m] ? s.caref[cr[m]][a] : 0.f;
    }
  }
#ifdef MZ_EXP_SUBTICK2
  cx.tick(s, 1);  // the slots' records
#endif
  // own block contacts (NB = 1): slots cx.l + m G < nB
  float bj[MB ? MB : 1][3][2], bar[MB ? MB : 1][3], bD[MB ? MB : 1], bu[MB ? MB : 1][3], bv[MB ? MB : 1][3];
  if constexpr (NB == 1) {
#pragma unroll
    for (int m = 0; m < MB; m++) {
      const int c = (WR ? (cx.l & 15) : cx.l) + m * BW;
      const bool on = c < nB;
      bD[m] = 0.f;
#pragma unroll
      for (int a = 0; a < 3; a++) { bj[m][a][0] = 0.f; bj[m][a][1] = 0.f; bar[m][a] = 0.f; bu[m][a] = 0.f; bv[m][a] = 0.f; }
      if (on) block_rows_direct<NB>(K, s, c, bj[m], bar[m], bD[m]);
    }
  }

  // ---- the robot contacts' Jacobians, in registers for the whole solve (read from LDS once per evaluation):
  //   jown[C][a]  this lane's own COLUMN of contact C (zero when the contact does not see this dof) — the Hessian folds;
  //   Jf[m][a][k] the ROW a of this lane's own contact over all dofs k (zero outside hub + its leg) — J x as three DPP matvecs.
  // With them a Newton iteration touches no memory at all.
  auto each_contact = [&](auto&& f) {  // f(C) for the contact slots some env of the wave uses (wave-uniform guards)
    if (cx.any(ncon > 0)) {
      f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{}); f(std::integral_constant<int, 3>{});
      if (cx.any(ncon > 4)) {
        // (slots 4 .. 7 in pairs: the waves a launch waits for are those whose ants lean on a wall — five or six contacts — and two
        // skipped slots are 160 instructions per Newton iteration; measured 0.2971 -> 0.2949 ms.  The same guard between slots 1 and 2
        // gains nothing: the waves with at most two contacts per env are not the ones the launch waits for)
        f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});
        if (cx.any(ncon > 6)) { f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{}); }
        if (cx.any(ncon > 8)) {
          f(std::integral_constant<int, 8>{}); f(std::integral_constant<int, 9>{}); f(std::integral_constant<int, 10>{}); f(std::integral_constant<int, 11>{});
          if (cx.any(ncon > 12)) { f(std::integral_constant<int, 12>{}); f(std::integral_constant<int, 13>{}); f(std::integral_constant<int, 14>{}); f(std::integral_constant<int, 15>{}); }
          if constexpr (MA > 1) {
            if (any2) {  // second slot of the contact lanes: an ant on its back, a robot lying against block and walls at once —
              // in groups of four as well: such an env decides how long its whole launch lasts (a launch waits for its slowest wave)
              f(std::integral_constant<int, 16>{}); f(std::integral_constant<int, 17>{}); f(std::integral_constant<int, 18>{}); f(std::integral_constant<int, 19>{});
              if (cx.any(ncon > 20)) {
                f(std::integral_constant<int, 20>{}); f(std::integral_constant<int, 21>{}); f(std::integral_constant<int, 22>{}); f(std::integral_constant<int, 23>{});
                if (cx.any(ncon > 24)) {
                  f(std::integral_constant<int, 24>{}); f(std::integral_constant<int, 25>{}); f(std::integral_constant<int, 26>{}); f(std::integral_constant<int, 27>{});
                  f(std::integral_constant<int, 28>{}); f(std::integral_constant<int, 29>{}); f(std::integral_constant<int, 30>{}); f(std::integral_constant<int, 31>{});
                }
              }
            }
          }
        }
      }
    }
  };
  // WR: the columns of the SECOND slot's contacts (C >= 16: an ant on its back next to the block, rare, behind the wave-uniform
  // `any2`) are not kept — 48 registers live through the whole solve for a path hardly any wave takes — but recomputed from the
  // owner's wrenches where they are used (own_col below)
  constexpr int NJ = WR ? 16 : 16 * MA;
  float jown[NJ][3];
  auto wr_col = [&](auto Cc, float (&o)[3]) {  // S_p . wrench_a of contact C, masked by "dof p moves the touching body"
    constexpr int C = decltype(Cc)::value;
    // S_p . wrench_a of contact C (its owner: lane C of the row), for the dofs that move the touching body: the root always, a hip
    // when the geom sits on its leg below it (aux, ankle body), an ankle for its own ankle body.  (The movable block's two slide
    // lanes carry MINUS their axis in Sax: a robot -> block contact pushes the block with the reaction of what the record holds
    // for the robot.)
    const int mc = bcast_i<(C & 15)>(wmeta[C / 16]), lc = mc & 7, cc = (mc >> 3) & 7;
    const bool sees = C < ncon && (ishinge ? (leg == lc && (d == 0 ? cc >= 2 : cc == 3)) : (r < 12 || (NB == 1 && r >= 14 && (mc & 64) != 0)));
    const int on = -(int)sees;
#pragma unroll
    for (int a = 0; a < 3; a++) o[a] = __int_as_float(__float_as_int(dot6_from<(C & 15)>(wr[C / 16][a], Sax)) & on);
  };
  auto own_col = [&](auto Cc, float (&o)[3]) {
    constexpr int C = decltype(Cc)::value;
    if constexpr (WR && C >= 16) wr_col(Cc, o);
    else { o[0] = jown[C < NJ ? C : 0][0]; o[1] = jown[C < NJ ? C : 0][1]; o[2] = jown[C < NJ ? C : 0][2]; }
  };
  each_contact([&](auto Cc) {
    constexpr int C = decltype(Cc)::value;
    if constexpr (WR && C >= 16) return;
    else {
    jown[C][0] = jown[C][1] = jown[C][2] = 0.f;
    if constexpr (WR) {
      wr_col(Cc, jown[C]);
    } else
    if (C < ncon) {  // (uniform within the env's row)
      const int lc = s.cleg[nB + C];
      // column of this lane's dof in the contact's 3 x (hub + 2) Jacobian: root dof i -> i, the block's slides -> 6, 7, the hinges
      // of the contact's own leg -> NHC, NHC + 1
      const int col = ishinge ? (leg == lc ? NHC + d : -1) : (r < 12 ? ri : ((NB == 1 && r >= 14 && r < 16) ? r - 8 : -1));
      if (col >= 0) { jown[C][0] = s.cJ[nB + C][0][col]; jown[C][1] = s.cJ[nB + C][1][col]; jown[C][2] = s.cJ[nB + C][2][col]; }
    }
    }
  });
  float Jf[MA][3][NR];
  if constexpr (!WR)
#pragma unroll
  for (int m = 0; m < MA; m++) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int k = 0; k < NR; k++) Jf[m][a][k] = 0.f;
    if (m == 1 && !any2) continue;
    if (iscon[m]) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float* Jr = s.cJ[cr[m]][a];
#pragma unroll
        for (int k = 0; k < 6; k++) Jf[m][a][dof2pos(k)] = Jr[k];
        if constexpr (NB == 1) { Jf[m][a][14] = Jr[6]; Jf[m][a][15] = Jr[7]; }
        const float jh = Jr[NHC], ja = Jr[NHC + 1];
#pragma unroll
        for (int l2 = 0; l2 < 4; l2++) { Jf[m][a][4 * l2] = cl[m] == l2 ? jh : 0.f; Jf[m][a][4 * l2 + 1] = cl[m] == l2 ? ja : 0.f; }
      }
    }
  }
  // J[c][a] . x for the lane's own robot contact(s), x row-distributed (one entry per dof lane; spare lanes are never read)
  auto jdot3 = [&](float x, float (&o)[MA][3]) {
    if constexpr (WR) {
      // transposed: (J x)[C][a] = sum over the lanes of their column entry times their x — one row butterfly per (contact, row);
      // every lane gets every sum, the owner keeps its own
#pragma unroll
      for (int m = 0; m < MA; m++) o[m][0] = o[m][1] = o[m][2] = 0.f;
      each_contact([&](auto Cc) {
        constexpr int C = decltype(Cc)::value;
        {  // (no per-row guard `C < ncon`: a row without contact C holds zero columns and adds zeros; one exec-mask region less per slot)
          float jc[3];
          own_col(Cc, jc);
          const float t0 = rsum(jc[0] * x), t1 = rsum(jc[1] * x), t2 = rsum(jc[2] * x);
          if (r == (C & 15)) { o[C / 16][0] = t0; o[C / 16][1] = t1; o[C / 16][2] = t2; }
        }
      });
    } else {
#pragma unroll
      for (int m = 0; m < MA; m++) {
        o[m][0] = o[m][1] = o[m][2] = 0.f;
        if (m == 1 && !any2) continue;
        o[m][0] = matvec(Jf[m][0], x); o[m][1] = matvec(Jf[m][1], x); o[m][2] = matvec(Jf[m][2], x);
      }
    }
  };
  // block contacts: residual rows from the two block entries (b0, b1) of a row-distributed vector
  auto bdot = [&](float b0, float b1, float (&o)[MB ? MB : 1][3]) {
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
      for (int a = 0; a < 3; a++) o[m][a] = bj[m][a][0] * b0 + bj[m][a][1] * b1;
  };

#ifdef MZ_EXP_SUBTICK2
  cx.tick(s, 2);  // block rows, own columns / rows of the contacts
#endif
  // ---- initial guess
  float qacc = warm;
  if (compare) {  // MuJoCo's rule on the first evaluation of a step: the better of warm start and qacc_smooth, by cost
    const float dw = warm - qas;
    float cw = 0.5f * dw * matvec(Mrow, dw), cs = 0.f;
    float jw3[MA][3], jq3[MA][3];
    jdot3(warm, jw3);
    jdot3(qas, jq3);
#pragma unroll
    for (int m = 0; m < MA; m++)
      if (iscon[m]) {
        cw += ceval(cD[m], jw3[m][0] - ar[m][0], jw3[m][1] - ar[m][1], jw3[m][2] - ar[m][2]);
        cs += ceval(cD[m], jq3[m][0] - ar[m][0], jq3[m][1] - ar[m][1], jq3[m][2] - ar[m][2]);
      }
    if (lsign != 0.f) {
      const float jw = lsign * warm - laref, js = lsign * qas - laref;
      if (jw < 0.f) cw += 0.5f * lD * jw * jw;
      if (js < 0.f) cs += 0.5f * lD * js * js;
    }
    cw = rsum(cw); cs = rsum(cs);
    if constexpr (NB == 1) {
      float bw[MB][3], bs[MB][3], cwb = 0.f, csb = 0.f;
      bdot(bcast<14>(warm), bcast<15>(warm), bw);
      bdot(bcast<14>(qas), bcast<15>(qas), bs);
#pragma unroll
      for (int m = 0; m < MB; m++) {
        cwb += ceval(bD[m], bw[m][0] - bar[m][0], bw[m][1] - bar[m][1], bw[m][2] - bar[m][2]);
        csb += ceval(bD[m], bs[m][0] - bar[m][0], bs[m][1] - bar[m][1], bs[m][2] - bar[m][2]);
      }
      cw += bsum(cwb); cs += bsum(csb);
    }
    qacc = cw < cs ? warm : qas;
  }
  if (!has) qacc = qas;
  cx.tick(s, 4);
  bool done = !has;
  int it = 0;
  float Mx = 0.f, ljar = 0.f, lact = 0.f, u[MA][3], v[MA][3];
#pragma unroll
  for (int m = 0; m < MA; m++) { u[m][0] = u[m][1] = u[m][2] = 0.f; v[m][0] = v[m][1] = v[m][2] = 0.f; }
  if (cx.any(!done)) {  // residuals at the starting point; afterwards they follow the step
    Mx = matvec(Mrow, qacc) - qfs;
    jdot3(qacc, u);
#pragma unroll
    for (int m = 0; m < MA; m++) { u[m][0] -= ar[m][0]; u[m][1] -= ar[m][1]; u[m][2] -= ar[m][2]; }
    if (lsign != 0.f) { ljar = lsign * qacc - laref; lact = ljar < 0.f ? lD : 0.f; }
    if constexpr (NB == 1) {
      bdot(bcast<14>(qacc), bcast<15>(qacc), bu);
#pragma unroll
      for (int m = 0; m < MB; m++)
#pragma unroll
        for (int a = 0; a < 3; a++) bu[m][a] -= bar[m][a];
    }
  }
  // Unit steps and robustness (round 6).  The first ls_fast_iters iterations of a solve take the unit step across an
  // active-set change without a line search; such a step may raise the cost, and unit steps alone can cycle.  Two guards were built
  // and measured (A / B libraries, profiles/r06/unit_guard_ab.txt, AntUMaze-v0 4096 envs): an overshoot test on the next gradient
  // (phi'(1) > 0 ends the solve's unit steps: 0.2366 -> 0.2577 ms, it fires often) and ADVICE's rule — take the unit step only if
  // cost(qacc + search) <= cost(qacc), else search the line in that iteration (0.2570 ms: ~100 vector instructions per iteration
  // with one wave per SIMD; as a run-time switch it still cost 1.2 % when off).  Either way a guarded unit step costs what the
  // exact search it replaces costs (ls_fast_iters = 0: 0.2555 ms), so there is no guard: the registered mazes — soaked for 1e8
  // env-steps without one solve at the iteration cap, error quantiles against the float64 oracle identical to the search in every
  // iteration — take unit steps, and a CUSTOM task, maze or robot variant, whose stiffness nobody has soaked, runs with
  // ls_fast_iterations = 0 by default (maze_env.py): MuJoCo's line search in every iteration, monotone on any maze.
  while (cx.any(!done) && it < ant_u(cx, K).max_iter) {
    // ---- contact lanes: gradient block g3 and curvature block W of their contact, in registers; every lane of the row reads
    // them with `row_newbcast:c` (fold_contact<c>): no LDS publish, no hand-off wait
    float mycg[MA][8];
#pragma unroll
    for (int m = 0; m < MA; m++) {
#pragma unroll
      for (int k = 0; k < 8; k++) mycg[m][k] = 0.f;
      if (m == 1 && !any2) continue;
      MZ_IF_OWNER(iscon[m]) {
        const float u0 = u[m][0], u1 = u[m][1], u2 = u[m][2], Dm = cD[m];
        const float r0 = u0 + u1, r1 = u0 - u1, r2 = u0 + u2, r3 = u0 - u2;
        const float a0 = r0 < 0.f ? 1.f : 0.f, a1 = r1 < 0.f ? 1.f : 0.f, a2 = r2 < 0.f ? 1.f : 0.f, a3 = r3 < 0.f ? 1.f : 0.f;
        mycg[m][0] = Dm * (a0 * r0 + a1 * r1 + a2 * r2 + a3 * r3); mycg[m][1] = Dm * (a0 * r0 - a1 * r1); mycg[m][2] = Dm * (a2 * r2 - a3 * r3);
        mycg[m][3] = Dm * (a0 + a1 + a2 + a3); mycg[m][4] = Dm * (a0 - a1); mycg[m][5] = Dm * (a2 - a3); mycg[m][6] = Dm * (a0 + a1); mycg[m][7] = Dm * (a2 + a3);
      }
    }
    // ---- the block's own contacts: gradient (2), |terms| (2) and curvature (3) on the block's two dofs, summed over the group
    float bg0 = 0.f, bg1 = 0.f, bga0 = 0.f, bga1 = 0.f, bh00 = 0.f, bh01 = 0.f, bh11 = 0.f;
    if constexpr (NB == 1) {
#pragma unroll
      for (int m = 0; m < MB; m++) {
        const float x0 = bu[m][0], x1 = bu[m][1], x2 = bu[m][2], Dm = bD[m];
        const float r0 = x0 + x1, r1 = x0 - x1, r2 = x0 + x2, r3 = x0 - x2;
        const float a0 = r0 < 0.f ? 1.f : 0.f, a1 = r1 < 0.f ? 1.f : 0.f, a2 = r2 < 0.f ? 1.f : 0.f, a3 = r3 < 0.f ? 1.f : 0.f;
        const float g0 = Dm * (a0 * r0 + a1 * r1 + a2 * r2 + a3 * r3), g1 = Dm * (a0 * r0 - a1 * r1), g2 = Dm * (a2 * r2 - a3 * r3);
        const float W0 = Dm * (a0 + a1 + a2 + a3), W1 = Dm * (a0 - a1), W2 = Dm * (a2 - a3), W3 = Dm * (a0 + a1), W4 = Dm * (a2 + a3);
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const float j0 = bj[m][0][c], j1 = bj[m][1][c], j2 = bj[m][2][c];
          const float t = j0 * g0 + j1 * g1 + j2 * g2;
          if (c == 0) { bg0 += t; bga0 += fabsf(t); } else { bg1 += t; bga1 += fabsf(t); }
          const float t0 = j0 * W0 + j1 * W1 + j2 * W2, t1 = j0 * W1 + j1 * W3, t2 = j0 * W2 + j2 * W4;  // (W J)[:, c]
          if (c == 0) { bh00 += t0 * j0 + t1 * j1 + t2 * j2; bh01 += t0 * bj[m][0][1] + t1 * bj[m][1][1] + t2 * bj[m][2][1]; }
          else bh11 += t0 * j0 + t1 * j1 + t2 * j2;
        }
      }
      bg0 = bsum(bg0); bg1 = bsum(bg1); bga0 = bsum(bga0); bga1 = bsum(bga1);
      bh00 = bsum(bh00); bh01 = bsum(bh01); bh11 = bsum(bh11);
    }
    // ---- row r of H = M + sum_c Jc^T Wc Jc + limit curvature; gradient entry r
    float Hrow[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) Hrow[k] = Mrow[k];
    float g = Mx, ga = fabsf(Mx);
    // Where the matrix cores take the fold (cx.mfma: the two-waves-per-SIMD instantiation, ant_kernels.hip): measured on the whole
    // kernel (A / B libraries, AntUMaze-v0), the MFMA fold is 1.4 % SLOWER with one wave per SIMD (4096 envs: 0.2804 -> 0.2846 ms — the
    // dependent MFMA chain, its drain and the accumulator's way back through 16 more accumulation registers cost what the 42-instruction
    // DPP fold of the usual four slots costs) and 2 % FASTER with two (8192 envs: 0.3839 -> 0.3765 ms — the second wave's vector
    // instructions fill the matrix pipe's latency).  The constant folds after inlining: each instantiation carries one of the two.
    if (cx.mfma) {
     if (cx.any(ncon > 0)) {  // (wave-uniform) the contacts' J^T W J through the matrix cores (fold_h_mfma above)
      v16f hacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      each_contact([&](auto Cc) {
        constexpr int C = decltype(Cc)::value;
        float t, t0, t1, t2;
        float jc[3];
        own_col(Cc, jc);
        fold_t<(C & 15)>(mycg[C / 16], jc[0], jc[1], jc[2], t, t0, t1, t2);
        g += t; ga += fabsf(t);
        fold_h_mfma(hacc, jc[0], jc[1], jc[2], t0, t1, t2);
      });
      float hc[16];
      mfma_rows(hacc, hc);
#pragma unroll
      for (int k = 0; k < NR; k++) Hrow[k] += hc[k];
     }
    } else
    each_contact([&](auto Cc) {
      constexpr int C = decltype(Cc)::value;
      {  // (no per-row guard `C < ncon`, as in jdot3)
        float t, t0, t1, t2;
        float jc[3];
        own_col(Cc, jc);
        fold_t<(C & 15)>(mycg[C / 16], jc[0], jc[1], jc[2], t, t0, t1, t2);
        g += t; ga += fabsf(t);
        fold_h(Hrow, jc[0], jc[1], jc[2], t0, t1, t2);
      }
    });
    if constexpr (NB == 1) {
      if (r == 14) { g += bg0; ga += bga0; Hrow[14] += bh00; Hrow[15] += bh01; }
      if (r == 15) { g += bg1; ga += bga1; Hrow[14] += bh01; Hrow[15] += bh11; }
    }
    MZ_IF_OWNER(lsign != 0.f) { const float t = lsign * lact * ljar; g += t; ga += fabsf(t); }
#pragma unroll
    for (int k = 0; k < 14; k++)
      if ((k & 3) < 2) Hrow[k] += (r == k) ? lact : 0.f;  // hinge positions: the limit row's curvature on its own diagonal
    if (!isdof) { g = 0.f; ga = 0.f; }
    // converged: MuJoCo's scaled-gradient test, or the gradient is at the fp32 cancellation floor.  Not in the first iteration of a
    // solve (round 5): the state has moved since the warm start was a solution, so the test can hardly pass there, and an env
    // that did sit on its optimum takes a Newton step of length ~0 and leaves as "exact" — two row reductions and two square roots
    // less per evaluation (A / B: 0.2470 -> 0.2456 ms).
    float gnorm = 0.f, anorm = 0.f;
    if (it > 0) {
      gnorm = sqrtf(rsum(g * g)); anorm = sqrtf(rsum(ga * ga));
      if (!done && (ant_u(cx, K).inv_scale * gnorm < ant_u(cx, K).tol || gnorm <= ant_u(cx, K).rtol * anorm)) done = true;
    }
    if (!cx.any(!done)) { cx.tick(s, 5); break; }
    cx.tick(s, 5);
    // ---- Newton direction: H search = -grad
    float mlt[NR], dinv;
    const float search = solve_rows(r, Hrow, -g, mlt, dinv);
    cx.tick(s, 6);
    // ---- J search on the contact lanes, limit rows on their own dofs; vote on the active set
    jdot3(search, v);
    const float ljv = lsign * search;
    bool changed = false;
#pragma unroll
    for (int m = 0; m < MA; m++)
      if (m == 0 || any2) MZ_IF_OWNER(iscon[m]) {
        const float u0 = u[m][0], u1 = u[m][1], u2 = u[m][2], w0 = u0 + v[m][0], w1 = u1 + v[m][1], w2 = u2 + v[m][2];
        changed = changed || ((u0 + u1 < 0.f) != (w0 + w1 < 0.f)) || ((u0 - u1 < 0.f) != (w0 - w1 < 0.f)) || ((u0 + u2 < 0.f) != (w0 + w2 < 0.f)) ||
                  ((u0 - u2 < 0.f) != (w0 - w2 < 0.f));
      }
    MZ_IF_OWNER(lsign != 0.f) changed = changed || ((ljar < 0.f) != (ljar + ljv < 0.f));
    if constexpr (NB == 1) {
      bdot(bcast<14>(search), bcast<15>(search), bv);
#pragma unroll
      for (int m = 0; m < MB; m++) {
        const float x0 = bu[m][0], x1 = bu[m][1], x2 = bu[m][2], w0 = x0 + bv[m][0], w1 = x1 + bv[m][1], w2 = x2 + bv[m][2];
        changed = changed || ((x0 + x1 < 0.f) != (w0 + w1 < 0.f)) || ((x0 - x1 < 0.f) != (w0 - w1 < 0.f)) || ((x0 + x2 < 0.f) != (w0 + w2 < 0.f)) ||
                  ((x0 - x2 < 0.f) != (w0 - w2 < 0.f));
      }
    }
    changed = cx.gany(changed);
    float alpha = 1.f, Ms = 0.f, sn = 1.f, qn = 0.f;
    const bool exact = !changed;
    if (changed) {  // exact line search on phi(alpha) = cost(qacc + alpha search): safeguarded Newton on the piecewise-linear phi'
      Ms = matvec(Mrow, search);
      const float p1 = rsum(search * Mx), p2 = rsum(search * Ms);
      sn = rsum(isdof ? search * search : 0.f); qn = rsum(isdof ? qacc * qacc : 0.f);
      float lo = 0.f, hi = -1.f, prev_d2 = -1.f;  // phi'(0) < 0 (descent direction); hi < 0: no upper bracket yet
      // Round 5: the first K.ls_fast_iters (5) iterations of an evaluation take the UNIT step when the active set changes (K.ls_fast = 0
      // evaluations of phi'), later ones search the line exactly as before.  What the line search buys is global convergence, not
      // accuracy: the solve ends with a unit step inside the final active set (or on the gradient test) whichever way it got there, so
      // the answer is the same to fp32 round-off (profiles/r05/ls_parity.txt: identical error quantiles against the float64 oracle), and
      // Newton with an exact search converges from ANY point — the unit steps only move where it starts.  Unit steps alone do cycle
      // (AntPush with 50 fast iterations: 2024 of 2048 envs hit the iteration cap within 500 steps; profiles/r05/ls_iters.txt), a
      // few of them (3 .. 8 measure alike on the plain ant, 5 is best on the one-block mazes: ls_k.txt) do not cost an iteration (lock-step iterations per step 55 -> 55) and save the search's matvec + ~3 evaluations, two
      // reductions each: AntUMaze-v0 0.2754 -> 0.2555 ms per step, AntPush-v0 0.5009 -> 0.4551.  `it` is wave-uniform: no divergence.
      const int ls_max = it < ant_u(cx, K).ls_fast_iters ? ant_u(cx, K).ls_fast : ant_u(cx, K).ls_iter;
      for (int ls = 0; ls < ls_max; ls++) {
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int m = 0; m < MA; m++)
          if (m == 0 || any2) MZ_IF_OWNER(iscon[m]) {
            const float v0 = v[m][0], v1 = v[m][1], v2 = v[m][2], Dm = cD[m];
            const float x0 = u[m][0] + alpha * v0, x1 = u[m][1] + alpha * v1, x2 = u[m][2] + alpha * v2;
            float rr, vv;
            rr = x0 + x1; vv = v0 + v1; if (rr < 0.f) { d1 += Dm * rr * vv; d2 += Dm * vv * vv; }
            rr = x0 - x1; vv = v0 - v1; if (rr < 0.f) { d1 += Dm * rr * vv; d2 += Dm * vv * vv; }
            rr = x0 + x2; vv = v0 + v2; if (rr < 0.f) { d1 += Dm * rr * vv; d2 += Dm * vv * vv; }
            rr = x0 - x2; vv = v0 - v2; if (rr < 0.f) { d1 += Dm * rr * vv; d2 += Dm * vv * vv; }
          }
        MZ_IF_OWNER(lsign != 0.f) { const float rr = ljar + alpha * ljv; if (rr < 0.f) { d1 += lD * rr * ljv; d2 += lD * ljv * ljv; } }
        if constexpr (NB != 1 || !WR) {
          d1 = rsum(d1) + p1 + alpha * p2;
          d2 = rsum(d2) + p2;
        }
        if constexpr (NB == 1) {
          float e1 = 0.f, e2 = 0.f;
#pragma unroll
          for (int m = 0; m < MB; m++) {
            const float y0 = bv[m][0], y1 = bv[m][1], y2 = bv[m][2], Dm = bD[m];
            const float x0 = bu[m][0] + alpha * y0, x1 = bu[m][1] + alpha * y1, x2 = bu[m][2] + alpha * y2;
            float rr, vv;
            rr = x0 + x1; vv = y0 + y1; if (rr < 0.f) { e1 += Dm * rr * vv; e2 += Dm * vv * vv; }
            rr = x0 - x1; vv = y0 - y1; if (rr < 0.f) { e1 += Dm * rr * vv; e2 += Dm * vv * vv; }
            rr = x0 + x2; vv = y0 + y2; if (rr < 0.f) { e1 += Dm * rr * vv; e2 += Dm * vv * vv; }
            rr = x0 - x2; vv = y0 - y2; if (rr < 0.f) { e1 += Dm * rr * vv; e2 += Dm * vv * vv; }
          }
          if constexpr (WR) { d1 = rsum(d1 + e1) + p1 + alpha * p2; d2 = rsum(d2 + e2) + p2; }
          else { d1 += cx.gsum(e1); d2 += cx.gsum(e2); }
        }
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
    // ---- step; the affine quantities follow it (an env whose step was exact is done: nothing of it is used again)
    if (alpha != 0.f) {
      qacc += alpha * search;
      if (changed) {
        Mx += alpha * Ms;
#pragma unroll
        for (int m = 0; m < MA; m++) { u[m][0] += alpha * v[m][0]; u[m][1] += alpha * v[m][1]; u[m][2] += alpha * v[m][2]; }
        MZ_IF_OWNER(lsign != 0.f) { ljar += alpha * ljv; lact = ljar < 0.f ? lD : 0.f; }
        if constexpr (NB == 1) {
#pragma unroll
          for (int m = 0; m < MB; m++)
#pragma unroll
            for (int a = 0; a < 3; a++) bu[m][a] += alpha * bv[m][a];
        }
      }
    }
    // ---- one step of iterative refinement behind an ACCEPTED unit step (round 5; mazes with a movable block).  The unit step is
    // taken as exact when the active set does not change — true of the arithmetic, not of an fp32 elimination: with every geom at
    // solimp .995 (maze_env.py:108-112) H = M + J^T D J has a condition number of 1e3 .. 1e4, and the step's error is relative to
    // the WHOLE step — hinge accelerations of thousands of rad/s^2 — so the torso's angular entries (tens of rad/s^2) came out 3e-4
    // off, 4e-6 in their velocity per evaluation: the whole of AntPush's error tail (tools/exp_forward_err.py; the float64 oracle,
    // fed inputs perturbed at fp32 round-off, moves by 1e-5 at most).  The residual gradient at qacc + search (same active set, the
    // affine quantities follow the step) goes through the SAME elimination — its multipliers were kept (resolve_rows) — and the
    // correction is added: ~120 instructions, no second Hessian.
    if constexpr (REFINE) {
      if (cx.any(exact && !done)) {
        const float Ms2 = changed ? Ms : matvec(Mrow, search);
        float g2 = Mx + Ms2;
        float cg2[MA][3];
#pragma unroll
        for (int m = 0; m < MA; m++) {
          cg2[m][0] = cg2[m][1] = cg2[m][2] = 0.f;
          if (m == 1 && !any2) continue;
          const float u0 = u[m][0] + v[m][0], u1 = u[m][1] + v[m][1], u2 = u[m][2] + v[m][2], Dm = cD[m];
          const float r0 = u0 + u1, r1 = u0 - u1, r2 = u0 + u2, r3 = u0 - u2;
          const float a0 = r0 < 0.f ? r0 : 0.f, a1 = r1 < 0.f ? r1 : 0.f, a2 = r2 < 0.f ? r2 : 0.f, a3 = r3 < 0.f ? r3 : 0.f;
          cg2[m][0] = Dm * (a0 + a1 + a2 + a3); cg2[m][1] = Dm * (a0 - a1); cg2[m][2] = Dm * (a2 - a3);
        }
        each_contact([&](auto Cc) {
          constexpr int C = decltype(Cc)::value;
          float jc[3];
          own_col(Cc, jc);
          g2 += fold_g<(C & 15)>(cg2[C / 16][0], cg2[C / 16][1], cg2[C / 16][2], jc[0], jc[1], jc[2]);
        });
        {
          const float lj2 = ljar + ljv;
          g2 += lsign * (lj2 < 0.f ? lD : 0.f) * lj2;
        }
        if constexpr (NB == 1) {
          float c0 = 0.f, c1 = 0.f;
#pragma unroll
          for (int m = 0; m < MB; m++) {
            const float x0 = bu[m][0] + bv[m][0], x1 = bu[m][1] + bv[m][1], x2 = bu[m][2] + bv[m][2], Dm = bD[m];
            const float r0 = x0 + x1, r1 = x0 - x1, r2 = x0 + x2, r3 = x0 - x2;
            const float a0 = r0 < 0.f ? r0 : 0.f, a1 = r1 < 0.f ? r1 : 0.f, a2 = r2 < 0.f ? r2 : 0.f, a3 = r3 < 0.f ? r3 : 0.f;
            const float g0 = Dm * (a0 + a1 + a2 + a3), g1 = Dm * (a0 - a1), gg2 = Dm * (a2 - a3);
            c0 += bj[m][0][0] * g0 + bj[m][1][0] * g1 + bj[m][2][0] * gg2;
            c1 += bj[m][0][1] * g0 + bj[m][1][1] * g1 + bj[m][2][1] * gg2;
          }
          c0 = bsum(c0); c1 = bsum(c1);
          if (r == 14) g2 += c0;
          if (r == 15) g2 += c1;
        }
        if (!isdof) g2 = 0.f;
        const float corr = resolve_rows(mlt, dinv, -g2);
        if (exact && !done) qacc += corr;
      }
    }
#ifdef MZ_EXP_TRACE  // developer aid (tools/replay_trace.py): one line per Newton iteration of the first env of a wave
    {
      const float g14 = NR > 14 ? bcast<14>(g) : 0.f, g15 = NR > 14 ? bcast<15>(g) : 0.f, s14 = NR > 14 ? bcast<14>(search) : 0.f, s15 = NR > 14 ? bcast<15>(search) : 0.f;
      if (cx.l == 0) printf("TRACE it %d ncon %d nB %d gnorm %g anorm %g gblk %g %g sblk %g %g changed %d alpha %g sn %g qn %g exact %d done %d u %g %g %g v %g %g %g D %g\n", it, ncon, nB, gnorm, anorm, g14, g15, s14, s15, (int)changed, alpha, sn, qn, (int)exact, (int)done, u[0][0], u[0][1], u[0][2], v[0][0], v[0][1], v[0][2], cD[0]);
    }
#endif
    if (exact && ant_u(cx, K).trust_exact) done = true;
    if (changed && alpha * alpha * sn <= MZ_NEWTON_STALL * MZ_NEWTON_STALL * qn) done = true;  // stationary at fp32 resolution (ant_dyn.h ant_solve)
    cx.tick(s, 7);
    it++;
  }
  if (isdof) s.qacc[ri] = qacc;
  if (cx.l == 0) {
    s.iters = it;
    if (it >= ant_u(cx, K).max_iter && !done) s.status |= MZ_STATUS_SOLVER_MAXITER;
    if constexpr (PROF) s.prof[15] += (unsigned)it;  // (part 2 of the scratch block: instrumented builds only)
#ifdef MZ_EXP_STAMPS  // (tools/exp_launch_stamps.py) lock-step iterations of the wave | contact-evaluations of this env, over the step
    s.red[2] += (float)it; s.red[3] += (float)ncon;
#endif
  }
  cx.sync();
  cx.tick(s, 8);
  return qacc;  // this lane's entry (position order; 0 on lanes without a dof)
}
