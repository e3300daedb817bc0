// mfma_fold_bench.hip — the measurement VERDICT r04 #9 asked for: the Hessian update of the plain ant's row solver,
//     H_b += sum over contact slots C, rows a of  t_a^C (x) j_a^C          (b = the wave's four envs, 16 lanes each),
// (A) as it runs today — rows::fold_h: 3 x 16 fused `v_fmac_f32_dpp ... row_newbcast:k` per slot, row r of H_b on lane 16 b + r — against
// (B) `v_mfma_f32_16x16x1_4b_f32`: one instruction = one rank-1 update of ALL FOUR envs' 16 x 16 blocks (A / B operands are exactly the
//     row-distributed t_a, j_a the solver holds), three per slot, followed by the move back into the solver's layout: the MFMA leaves
//     block b in accumulator registers 4 b .. 4 b + 3 with lane 16 g + j holding column j of rows 4 g .. 4 g + 3; H_b is symmetric once
//     all three rows of a contact are in, so lane 16 g + j holds H_b[j][4 g .. 4 g + 3], and a 4 x 4 transpose of register groups across
//     the wave's four 16-lane rows (8 v_permlane32_swap + 8 v_permlane16_swap) puts row j of env b on lane 16 b + j.
// Built and run on the GPU box (tools/mfma_fold_bench.sh); prints shader cycles per Hessian build for S = 2, 4, 6, 8 slots, one wave per
// SIMD (the metric's regime), with the fold_t part (identical in both) included, and checks B against A.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../mujoco_maze_amd/csrc/ant_newton_rows.h"

typedef float v16f __attribute__((ext_vector_type(16)));

template <int S, int MODE>
__global__ __launch_bounds__(64) void bench(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  using namespace rows;
  const int lane = threadIdx.x;
  float cg[8], j[S][3], M[16];
  for (int k = 0; k < 8; k++) cg[k] = in[(blockIdx.x * 64 + lane) * 64 + k];
  for (int c = 0; c < S; c++) for (int a = 0; a < 3; a++) j[c][a] = in[(blockIdx.x * 64 + lane) * 64 + 8 + 3 * c + a];
  for (int k = 0; k < 16; k++) M[k] = in[(blockIdx.x * 64 + lane) * 64 + 40 + k];
  float acc = 0.f;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    float H[16];
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 16; k++) H[k] = M[k];
#pragma unroll
      for (int c = 0; c < S; c++) {
        float t, t0_, t1_, t2_;
        fold_t<0>(cg, j[c][0], j[c][1], j[c][2], t, t0_, t1_, t2_);
        acc += t;
        fold_h(H, j[c][0], j[c][1], j[c][2], t0_, t1_, t2_);
      }
    } else {
      v16f d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < S; c++) {
        float t, t0_, t1_, t2_;
        fold_t<0>(cg, j[c][0], j[c][1], j[c][2], t, t0_, t1_, t2_);
        acc += t;
#ifdef PAD_PRE  // fold_t is inline asm: the compiler's hazard recognizer does not see its VALU writes in front of the MFMA that reads them
        asm volatile("s_nop 3" : "+v"(t0_), "+v"(t1_), "+v"(t2_));
#endif
        d = __builtin_amdgcn_mfma_f32_16x16x1f32(t0_, j[c][0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x1f32(t1_, j[c][1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x1f32(t2_, j[c][2], d, 0, 0, 0);
#ifdef PAD_NOP  // does the next slot's fold_t (inline asm: invisible to the compiler's hazard recognizer) overwrite an operand of an MFMA in flight?
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(t0_), "+v"(t1_), "+v"(t2_));
#endif
      }
#ifdef PAD_READ  // the accumulator is read back too early?  (the compiler pads the MFMA -> v_accvgpr_read distance with `s_nop 9`)
      asm volatile("s_nop 15" : "+a"(d));
#endif
      // register group b (4 registers) on lane-row g  ->  lane-row b, columns 4 g .. 4 g + 3
      float G[4][4];
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int v = 0; v < 4; v++) G[b][v] = d[4 * b + v];
#pragma unroll
      for (int v = 0; v < 4; v++) {  // halves: groups (0, 2) and (1, 3) exchange between lanes 0-31 and 32-63
        auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_int(G[0][v]), __float_as_int(G[2][v]), false, false);
        G[0][v] = __int_as_float(r0[0]); G[2][v] = __int_as_float(r0[1]);
        auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_int(G[1][v]), __float_as_int(G[3][v]), false, false);
        G[1][v] = __int_as_float(r1[0]); G[3][v] = __int_as_float(r1[1]);
      }
#pragma unroll
      for (int v = 0; v < 4; v++) {  // rows inside each half: groups (0, 1) and (2, 3)
        auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_int(G[0][v]), __float_as_int(G[1][v]), false, false);
        G[0][v] = __int_as_float(r0[0]); G[1][v] = __int_as_float(r0[1]);
        auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_int(G[2][v]), __float_as_int(G[3][v]), false, false);
        G[2][v] = __int_as_float(r1[0]); G[3][v] = __int_as_float(r1[1]);
      }
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int v = 0; v < 4; v++) H[4 * g + v] = M[4 * g + v] + G[g][v];
    }
    // consume H the way the solver would start to: the pivot entry of row 0 broadcast, a reduction
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) s += H[k] * (float)(k + 1);
    acc += s * 1e-6f;
    cg[0] += acc * 1e-30f;  // loop-carried dependence: the next build waits for this one, as a Newton iteration does
    if (it == iters - 1) { for (int k = 0; k < 16; k++) out[(blockIdx.x * 64 + lane) * 16 + k] = H[k]; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 1.2345f) out[0] = acc;
}

// layout probe: D_b[i][k] = A[16 b + i] * B[16 b + k] for one instruction; where does each product land?
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  v16f d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(lane + 1), (float)(100 * (lane + 1)), d, 0, 0, 0);
  for (int v = 0; v < 16; v++) out[lane * 16 + v] = d[v];
}
static void run_probe(float* out) {
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out);
  float h[64 * 16];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; lane++)
    for (int v = 0; v < 16; v++) {
      const int b = v / 4, g = lane / 16, jj = lane % 16, i = 4 * g + v % 4;  // assumed: register 4 b + v', lane 16 g + j  <->  D_b[4 g + v'][j]
      const float want = (float)(16 * b + i + 1) * (float)(100 * (16 * b + jj + 1));
      if (h[lane * 16 + v] != want) { if (bad < 4) printf("   probe: lane %d reg %d holds %.0f, layout says %.0f\n", lane, v, h[lane * 16 + v], want); bad++; }
    }
  printf("layout probe of v_mfma_f32_16x16x1_4b_f32 (register 4 b + v, lane 16 g + j = D_b[4 g + v][j]): %d of 1024 entries differ\n", bad);
}

template <int S>
static void run(const float* in, float* out, unsigned long long* cyc, int nb, float* hA, float* hB, const float* hin) {
  const int iters = 2000;
  double c[2];
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) hipLaunchKernelGGL((bench<S, 0>), dim3(nb), dim3(64), 0, 0, in, out, cyc, iters);
      else hipLaunchKernelGGL((bench<S, 1>), dim3(nb), dim3(64), 0, 0, in, out, cyc, iters);
      hipDeviceSynchronize();
    }
    unsigned long long* h = (unsigned long long*)malloc(nb * 8);
    hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < nb; i++) sum += (double)h[i];
    c[mode] = sum / nb / iters;
    hipMemcpy(mode == 0 ? hA : hB, out, (size_t)nb * 64 * 16 * 4, hipMemcpyDeviceToHost);
    free(h);
  }
  // host reference for block 0 (float64): which of the two is right when they differ
  {
    double eA = 0, eB = 0; int nprint = 0;
    for (int b = 0; b < 4; b++)
      for (int i = 0; i < 16; i++)
        for (int k = 0; k < 16; k++) {
          const float* own = hin + (size_t)(16 * b) * 64;  // lane 0 of the row owns the curvature block cg[3..7] = W00 W01 W02 W11 W22
          const double W[3][3] = {{own[3], own[4], own[5]}, {own[4], own[6], 0.0}, {own[5], 0.0, own[7]}};
          double h = hin[(size_t)(16 * b + i) * 64 + 40 + k];
          for (int c = 0; c < S; c++) {
            const float* ji = hin + (size_t)(16 * b + i) * 64 + 8 + 3 * c;
            const float* jk = hin + (size_t)(16 * b + k) * 64 + 8 + 3 * c;
            for (int a = 0; a < 3; a++) { double t = 0; for (int q = 0; q < 3; q++) t += W[a][q] * ji[q]; h += t * jk[a]; }
          }
          eA = fmax(eA, fabs(h - hA[(size_t)(16 * b + i) * 16 + k])); eB = fmax(eB, fabs(h - hB[(size_t)(16 * b + i) * 16 + k]));

        }
    printf("   against a float64 host reference (block 0): DPP %.2e, MFMA %.2e; MFMA rows off:", eA, eB);
    for (int b = 0; b < 4; b++)
      for (int i = 0; i < 16; i++) {
        int off = 0;
        for (int k = 0; k < 16; k++) {
          const float* own = hin + (size_t)(16 * b) * 64;
          const double W[3][3] = {{own[3], own[4], own[5]}, {own[4], own[6], 0.0}, {own[5], 0.0, own[7]}};
          double h = hin[(size_t)(16 * b + i) * 64 + 40 + k];
          for (int c = 0; c < S; c++) {
            const float* ji = hin + (size_t)(16 * b + i) * 64 + 8 + 3 * c;
            const float* jk = hin + (size_t)(16 * b + k) * 64 + 8 + 3 * c;
            for (int a = 0; a < 3; a++) { double t = 0; for (int q = 0; q < 3; q++) t += W[a][q] * ji[q]; h += t * jk[a]; }
          }
          if (fabs(h - hB[(size_t)(16 * b + i) * 16 + k]) > 1e-3) off++;
        }
        if (off) printf(" (%d,%d:%d)", b, i, off);
      }
    printf("\n");
  }
  double maxd = 0, maxv = 0;
  for (size_t i = 0; i < (size_t)nb * 64 * 16; i++) { double d = fabs((double)hA[i] - hB[i]); if (d > maxd) maxd = d; if (fabs(hA[i]) > maxv) maxv = fabs(hA[i]); }
  printf("S = %d contact slots: DPP fold %7.1f cycles per Hessian build, MFMA + transpose %7.1f (x %.2f); max |H_A - H_B| = %.2e of %.1f\n", S, c[0], c[1], c[1] / c[0], maxd, maxv);
}

int main() {
  const int nb = 1024;  // one wave per SIMD
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, (size_t)nb * 64 * 64 * 4); hipMalloc(&out, (size_t)nb * 64 * 16 * 4); hipMalloc(&cyc, nb * 8);
  float* h = (float*)malloc((size_t)nb * 64 * 64 * 4);
  srand(1);
  for (size_t i = 0; i < (size_t)nb * 64 * 64; i++) h[i] = (float)rand() / RAND_MAX - 0.5f;
  // the curvature block W (cg[3..7]: W00 W01 W02 W11 W22) must make J^T W J symmetric — it is, for any numbers; nothing to do
  hipMemcpy(in, h, (size_t)nb * 64 * 64 * 4, hipMemcpyHostToDevice);
  float* hA = (float*)malloc((size_t)nb * 64 * 16 * 4); float* hB = (float*)malloc((size_t)nb * 64 * 16 * 4);
  run_probe(out);
  run<2>(in, out, cyc, nb, hA, hB, h); run<4>(in, out, cyc, nb, hA, hB, h); run<6>(in, out, cyc, nb, hA, hB, h); run<8>(in, out, cyc, nb, hA, hB, h);
  return 0;
}
