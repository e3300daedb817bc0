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

template <int P>
__device__ __forceinline__ float bcast(float x) {  // value of lane P of this 16-lane row, on every lane of the row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + P, 0xF, 0xF, false));
}
__device__ __forceinline__ float rsum(float x) {  // all-reduce over the 16 lanes of the row
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xF, 0xF, true));  // row_half_mirror
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x140, 0xF, 0xF, true));  // row_mirror
  return x;
}

// y_r = sum_k A[r][k] x_k with row r of A in registers (Arow) and x distributed one entry per lane
template <int K = 0>
__device__ __forceinline__ float matvec(const float (&Arow)[14], float x) {
  if constexpr (K == 14) return 0.f;
  else return Arow[K] * bcast<K>(x) + matvec<K + 1>(Arow, x);
}

// one Gauss-Jordan pivot P on the row-distributed system (Hrow | b): every other row gets rid of column P.
// COLS... = the columns that can still be non-zero in the pivot row (compile-time list: the arrow structure).
template <int P, int... COLS>
__device__ __forceinline__ void pivot(int r, float (&Hrow)[14], float& b, float& dinv) {
  const float d = bcast<P>(Hrow[P]);
  const float ri = 1.0f / fmaxf(d, 1e-30f);
  const float li = (r == P) ? 0.f : Hrow[P] * ri;
  ((Hrow[COLS] -= li * bcast<P>(Hrow[COLS])), ...);
  b -= li * bcast<P>(b);
  dinv = (r == P) ? ri : dinv;
}

// H x = b, arrow-structured SPD H: leg dofs (6..13) are eliminated first, each touching its partner and the hub columns only
__device__ __forceinline__ float solve14(int r, float (&Hrow)[14], float b) {
  float dinv = 0.f;
  // the four legs do not couple: their hip pivots (then their ankle pivots) are independent chains — issued next to each
  // other so that the reciprocal / broadcast latencies of one hide behind the others
  pivot<6, 7, 0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<8, 9, 0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<10, 11, 0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<12, 13, 0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<7, 0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<9, 0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<11, 0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<13, 0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<0, 1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<1, 2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<2, 3, 4, 5>(r, Hrow, b, dinv);
  pivot<3, 4, 5>(r, Hrow, b, dinv);
  pivot<4, 5>(r, Hrow, b, dinv);
  pivot<5>(r, Hrow, b, dinv);
  return b * dinv;
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

}  // namespace rows

// The solve.  In: s.M, s.qfs, s.warm (first evaluation of a step: MuJoCo's qacc_warmstart; later: the previous evaluation's
// solution minus its qacc_smooth, see ant_forward), contacts (s.cJ, s.caref, s.cD, s.cleg, s.ncon); the joint-limit rows are built here
// from s.qpos / s.qvel.  Out: s.qas = M^-1 qfs, s.qacc, s.iters, status bits.  Requires G >= 16.
template <int G, bool PROF>
__device__ __forceinline__ void ant_solve_rows(const DevCtx<G, PROF>& cx, const AntDev& K, AntScratchT<0>& s, bool compare) {
  static_assert(G >= 16, "one DPP row per env at least");
  using namespace rows;
  const int r = cx.l & 15;                       // dof of this lane; 14, 15: spare lanes (zero rows, never pivots)
  const bool isdof = r < 14, ishinge = r >= 6 && r < 14;
  const int leg = (r - 6) >> 1, d = (r - 6) & 1;  // hinge lanes: own leg, 0 hip / 1 ankle
  const int ncon = s.ncon;
  const bool iscon = r < ncon;                   // this lane owns contact r
  const int cl = iscon ? s.cleg[iscon ? r : 0] : -1;

  // ---- row r of M, per-dof vectors, own limit row
  float Mrow[14];
#pragma unroll
  for (int k = 0; k < 14; k++) Mrow[k] = s.Md[r][k];  // dense rows written beside the arrow form (crb_leg_item / crb_root_item); rows 14, 15 are zero
  const int ri = isdof ? r : 0;
  const float qfs = isdof ? s.qfs[ri] : 0.f;
  cx.tick(s, 12);
  // joint-limit row of this lane's own hinge (limit_item of ant_dyn.h, in registers: the row never leaves its lane)
  float lsign = 0.f, lD = 0.f, laref = 0.f;
  if (ishinge) {
    const int j = r - 6, l = j >> 1;
    const float q = s.qpos[7 + j], lo = (j & 1) ? K.ank_lo[l] : K.hip_lo, hi = (j & 1) ? K.ank_hi[l] : K.hip_hi;
    float pos = 0.f;
    if (q - lo < 0.f) { lsign = 1.f; pos = q - lo; }
    else if (hi - q < 0.f) { lsign = -1.f; pos = hi - q; }
    if (lsign != 0.f) {
      const float imp = impedancef(K.lim_solimp, fabsf(pos));
      const float R = fmaxf(1e-15f, (1.f - imp) / imp * ((j & 1) ? K.dofw_ank : K.dofw_hip));
      lD = 1.0f / R;
      laref = -K.lim_B * (lsign * s.qvel[6 + j]) - K.lim_K * imp * pos;
    }
  }
  const bool has = ncon > 0 || cx.gany(lsign != 0.f);
  // qacc_smooth = M^-1 qfrc_smooth by the same row elimination — only where it is used: on the first evaluation of a step
  // (MuJoCo's warm-start rule compares against it) and for an env without any constraint (then it is the answer).  The
  // Newton iteration itself works on M qacc - qfrc_smooth and never needs it.
  float qas = 0.f;
  if (compare || cx.any(!has)) {
    float Hq[14];
#pragma unroll
    for (int k = 0; k < 14; k++) Hq[k] = Mrow[k];
    qas = solve14(r, Hq, qfs);
    if (isdof) s.qas[ri] = qas;
  }
  const float warm = isdof ? s.warm[ri] : 0.f;  // later evaluations start from the previous evaluation's solution
  // own contact: 3 x 8 Jacobian rows stay in LDS (row-major, read as needed); constants in registers
  const int cr = iscon ? r : 0;
  const float cD = iscon ? s.cD[cr] : 0.f;
  const float ar0 = iscon ? s.caref[cr][0] : 0.f, ar1 = iscon ? s.caref[cr][1] : 0.f, ar2 = iscon ? s.caref[cr][2] : 0.f;

  // J[c][a] . x for the lane's own contact, x read from an LDS vector in MuJoCo dof order
  auto jdot3 = [&](const float* x, float& o0, float& o1, float& o2) {
    o0 = o1 = o2 = 0.f;
    if (iscon) {
      float xv[8];
#pragma unroll
      for (int k = 0; k < 6; k++) xv[k] = x[k];
      xv[6] = cl >= 0 ? x[6 + 2 * (cl >= 0 ? cl : 0)] : 0.f;
      xv[7] = cl >= 0 ? x[7 + 2 * (cl >= 0 ? cl : 0)] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; k++) { o0 += s.cJ[cr][0][k] * xv[k]; o1 += s.cJ[cr][1][k] * xv[k]; o2 += s.cJ[cr][2][k] * xv[k]; }
    }
  };

  // ---- initial guess
  float qacc = warm;
  if (compare) {  // MuJoCo's rule on the first evaluation of a step: the better of warm start and qacc_smooth, by cost
    cx.sync();  // s.qas is read back by the contact lanes
    const float dw = warm - qas;
    float cw = 0.5f * dw * matvec(Mrow, dw), cs = 0.f;
    float w0, w1, w2, q0, q1, q2;
    jdot3(s.warm, w0, w1, w2);
    jdot3(s.qas, q0, q1, q2);
    if (iscon) { cw += ceval(cD, w0 - ar0, w1 - ar1, w2 - ar2); cs += ceval(cD, q0 - ar0, q1 - ar1, q2 - ar2); }
    if (lsign != 0.f) {
      const float jw = lsign * warm - laref, js = lsign * qas - laref;
      if (jw < 0.f) cw += 0.5f * lD * jw * jw;
      if (js < 0.f) cs += 0.5f * lD * js * js;
    }
    cw = rsum(cw); cs = rsum(cs);
    qacc = cw < cs ? warm : qas;
  }
  if (!has) qacc = qas;
  cx.tick(s, 4);
  bool done = !has;
  int it = 0;
  float Mx = 0.f, u0 = 0.f, u1 = 0.f, u2 = 0.f, ljar = 0.f, lact = 0.f;
  if (cx.any(!done)) {  // residuals at the starting point; afterwards they follow the step
    if (isdof) s.qacc[ri] = qacc;
    cx.sync();
    Mx = matvec(Mrow, qacc) - qfs;
    jdot3(s.qacc, u0, u1, u2);
    u0 -= ar0; u1 -= ar1; u2 -= ar2;
    if (lsign != 0.f) { ljar = lsign * qacc - laref; lact = ljar < 0.f ? lD : 0.f; }
  }
  while (cx.any(!done) && it < K.max_iter) {
    // ---- contact lanes: gradient block g3 and curvature block W of their contact, in registers; every lane of the row reads
    // them with `row_newbcast:c` (fold_contact<c>): no LDS publish, no hand-off wait
    float mycg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (iscon) {
      const float r0 = u0 + u1, r1 = u0 - u1, r2 = u0 + u2, r3 = u0 - u2;
      const float a0 = r0 < 0.f ? 1.f : 0.f, a1 = r1 < 0.f ? 1.f : 0.f, a2 = r2 < 0.f ? 1.f : 0.f, a3 = r3 < 0.f ? 1.f : 0.f;
      mycg[0] = cD * (a0 * r0 + a1 * r1 + a2 * r2 + a3 * r3); mycg[1] = cD * (a0 * r0 - a1 * r1); mycg[2] = cD * (a2 * r2 - a3 * r3);
      mycg[3] = cD * (a0 + a1 + a2 + a3); mycg[4] = cD * (a0 - a1); mycg[5] = cD * (a2 - a3); mycg[6] = cD * (a0 + a1); mycg[7] = cD * (a2 + a3);
    }
    // ---- row r of H = M + sum_c Jc^T Wc Jc + limit curvature; gradient entry r
    float Hrow[14];
#pragma unroll
    for (int k = 0; k < 14; k++) Hrow[k] = Mrow[k];
    float g = Mx, ga = fabsf(Mx);
    auto fold = [&](auto Cc) {
      constexpr int C = decltype(Cc)::value;
      if (C < ncon) {  // (uniform within the env's row)
        float cg[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cg[k] = bcast<C>(mycg[k]);
        const int lc = __builtin_amdgcn_update_dpp(0, cl, 0x150 + C, 0xF, 0xF, false);
        const int col = r < 6 ? r : ((ishinge && leg == lc) ? 6 + d : -1);
        const int cc = col >= 0 ? col : 0;
        const float j0 = col >= 0 ? s.cJ[C][0][cc] : 0.f, j1 = col >= 0 ? s.cJ[C][1][cc] : 0.f, j2 = col >= 0 ? s.cJ[C][2][cc] : 0.f;
        const float t = j0 * cg[0] + j1 * cg[1] + j2 * cg[2];
        g += t; ga += fabsf(t);
        const float t0 = j0 * cg[3] + j1 * cg[4] + j2 * cg[5], t1 = j0 * cg[4] + j1 * cg[6], t2 = j0 * cg[5] + j2 * cg[7];  // (W Jc)[:, col]
#pragma unroll
        for (int k = 0; k < 6; k++) Hrow[k] += t0 * s.cJ[C][0][k] + t1 * s.cJ[C][1][k] + t2 * s.cJ[C][2][k];
        const float h6 = t0 * s.cJ[C][0][6] + t1 * s.cJ[C][1][6] + t2 * s.cJ[C][2][6];
        const float h7 = t0 * s.cJ[C][0][7] + t1 * s.cJ[C][1][7] + t2 * s.cJ[C][2][7];
#pragma unroll
        for (int l2 = 0; l2 < 4; l2++) { Hrow[6 + 2 * l2] += lc == l2 ? h6 : 0.f; Hrow[7 + 2 * l2] += lc == l2 ? h7 : 0.f; }
      }
    };
    if (cx.any(ncon > 0)) {
      fold(std::integral_constant<int, 0>{}); fold(std::integral_constant<int, 1>{}); fold(std::integral_constant<int, 2>{}); fold(std::integral_constant<int, 3>{});
      if (cx.any(ncon > 4)) {
        fold(std::integral_constant<int, 4>{}); fold(std::integral_constant<int, 5>{}); fold(std::integral_constant<int, 6>{}); fold(std::integral_constant<int, 7>{});
        if (cx.any(ncon > 8)) {
          fold(std::integral_constant<int, 8>{}); fold(std::integral_constant<int, 9>{}); fold(std::integral_constant<int, 10>{}); fold(std::integral_constant<int, 11>{});
          fold(std::integral_constant<int, 12>{}); fold(std::integral_constant<int, 13>{}); fold(std::integral_constant<int, 14>{}); fold(std::integral_constant<int, 15>{});
        }
      }
    }
    if (lsign != 0.f) { const float t = lsign * lact * ljar; g += t; ga += fabsf(t); }
#pragma unroll
    for (int k = 6; k < 14; k++) Hrow[k] += (r == k) ? lact : 0.f;
    if (!isdof) { g = 0.f; ga = 0.f; }
    const float gnorm = sqrtf(rsum(g * g)), anorm = sqrtf(rsum(ga * ga));
    // converged: MuJoCo's scaled-gradient test, or the gradient is at the fp32 cancellation floor
    if (!done && (K.inv_scale * gnorm < K.tol || gnorm <= K.rtol * anorm)) done = true;
    if (!cx.any(!done)) { cx.tick(s, 5); break; }
    cx.tick(s, 5);
    // ---- Newton direction: H search = -grad
    const float search = solve14(r, Hrow, -g);
    if (isdof) s.search[ri] = search;
    cx.sync();
    cx.tick(s, 6);
    // ---- J search on the contact lanes, limit rows on their own dofs; vote on the active set
    float v0, v1, v2;
    jdot3(s.search, v0, v1, v2);
    const float ljv = lsign * search;
    bool changed = false;
    if (iscon) {
      const float w0 = u0 + v0, w1 = u1 + v1, w2 = u2 + v2;
      changed = ((u0 + u1 < 0.f) != (w0 + w1 < 0.f)) || ((u0 - u1 < 0.f) != (w0 - w1 < 0.f)) || ((u0 + u2 < 0.f) != (w0 + w2 < 0.f)) ||
                ((u0 - u2 < 0.f) != (w0 - w2 < 0.f));
    }
    if (lsign != 0.f) changed = changed || ((ljar < 0.f) != (ljar + ljv < 0.f));
    changed = cx.gany(changed);
    float alpha = 1.f, Ms = 0.f;
    const bool exact = !changed;
    if (changed) {  // exact line search on phi(alpha) = cost(qacc + alpha search): safeguarded Newton on the piecewise-linear phi'
      Ms = matvec(Mrow, search);
      const float p1 = rsum(search * Mx), p2 = rsum(search * Ms);
      float lo = 0.f, hi = -1.f, prev_d2 = -1.f;  // phi'(0) < 0 (descent direction); hi < 0: no upper bracket yet
      for (int ls = 0; ls < K.ls_iter; ls++) {
        float d1 = 0.f, d2 = 0.f;
        if (iscon) {
          const float x0 = u0 + alpha * v0, x1 = u1 + alpha * v1, x2 = u2 + alpha * v2;
          float rr, vv;
          rr = x0 + x1; vv = v0 + v1; if (rr < 0.f) { d1 += cD * rr * vv; d2 += cD * vv * vv; }
          rr = x0 - x1; vv = v0 - v1; if (rr < 0.f) { d1 += cD * rr * vv; d2 += cD * vv * vv; }
          rr = x0 + x2; vv = v0 + v2; if (rr < 0.f) { d1 += cD * rr * vv; d2 += cD * vv * vv; }
          rr = x0 - x2; vv = v0 - v2; if (rr < 0.f) { d1 += cD * rr * vv; d2 += cD * vv * vv; }
        }
        if (lsign != 0.f) { const float rr = ljar + alpha * ljv; if (rr < 0.f) { d1 += lD * rr * ljv; d2 += lD * ljv * ljv; } }
        d1 = rsum(d1) + p1 + alpha * p2;
        d2 = rsum(d2) + p2;
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
        u0 += alpha * v0; u1 += alpha * v1; u2 += alpha * v2;
        if (lsign != 0.f) { ljar += alpha * ljv; lact = ljar < 0.f ? lD : 0.f; }
      }
    }
    if (exact && K.trust_exact) done = true;
    cx.tick(s, 7);
    it++;
  }
  if (isdof) s.qacc[ri] = qacc;
  if (cx.l == 0) { s.iters = it; if (it >= K.max_iter && !done) s.status |= MZ_STATUS_SOLVER_MAXITER; s.prof[15] += (unsigned)it; }
  cx.sync();
  cx.tick(s, 8);
}
