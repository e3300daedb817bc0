// mazestep.hip — HIP kernels (gfx950 / CDNA4) and the C-ABI of include/mazestep.h.
//
// Kernels
//   ant_step_kernel<NB,G>       one MazeEnv.step for the Ant (+ NB movable blocks): G lanes per environment, 64/G
//                               environments per one-wavefront workgroup, the env's whole working set (7.7 KB for the
//                               plain ant, AntScratchT<NB>) resident in LDS across the 20 forward-dynamics evaluations
//                               of the step; HBM is touched once per step (192-B state record in, record +
//                               obs / reward / done out).
//   planar_step_kernel<NB,NS,G> one MazeEnv.step for the Point (+ NB movable blocks or NS object balls): lane group
//                               per env, PlanarScratch in LDS, fp64.
//   swimmer_step_kernel<NL,NB>  one MazeEnv.step for the Swimmer (NL = 3) / Reacher (NL = 2): one env per lane, fp64.
//   *_reset / state copy / debug kernels.
//
// Data layout in HBM
//   Ant:   state[N][REC] fp32 record = qpos[nq] | qvel[nv] | qacc_warmstart[nv] | t | episode | pad
//          (REC = 48 words for the plain ant, 64 with one movable block: AntDims<NB>::REC)
//          (a lane group reads 48 consecutive words: coalesced for lane-group-per-env kernels;
//          SoA would scatter a group's loads over 48 cache lines)
//   Point / Swimmer / Reacher: SoA  q_0..q_{NV-1} | v_0..v_{NV-1}, each [N] fp32; t[N], episode[N] i32
//   API arrays are row-major [N, k] as in include/mazestep.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "ant_dyn.h"
#include "point_dyn.h"
#include "planar_dyn.h"
#include "swimmer_dyn.h"


// ------------------------------------------------------------------ device context of a lane group
template <int G, bool PROF = false>
struct DevCtx {
  static constexpr int nlanes = G;
  int l;
  // phase timer (PROF builds only): lane 0 of the group accumulates shader cycles since the previous tick
  template <class S>
  __device__ __forceinline__ void tick(S& s, int id) const {
    if constexpr (PROF) {
      if (l == 0) {
        unsigned long long now = __builtin_amdgcn_s_memtime();
        s.prof[id] += (unsigned)(now - s.prof_t0);
        s.prof_t0 = now;
      }
    }
  }
  __device__ __forceinline__ int lane0() const { return l; }
  // A lane group never spans wavefronts and LDS operations of one wavefront execute in order, so a
  // hand-off between lanes of a group needs no s_barrier: a wavefront-scope fence (no instruction, it only
  // pins the compiler's ordering of the LDS stores before and loads after) is a complete phase boundary.
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // All-reduce (sum) inside the lane group.  Rows of 16 lanes reduce with four DPP moves (quad_perm xor 1,
  // xor 2, row_half_mirror, row_mirror: VALU-rate, no LDS crossbar); only the cross-row steps use ds_bpermute.
  static __device__ __forceinline__ float dpp_add(float x, const int ctrl_sel) {
    int xi = __float_as_int(x), yi;
    switch (ctrl_sel) {
      case 0: yi = __builtin_amdgcn_mov_dpp(xi, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
      case 1: yi = __builtin_amdgcn_mov_dpp(xi, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
      case 2: yi = __builtin_amdgcn_mov_dpp(xi, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
      default: yi = __builtin_amdgcn_mov_dpp(xi, 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return x + __int_as_float(yi);
  }
  __device__ __forceinline__ float gsum(float x) const {
    if constexpr (G >= 2) x = dpp_add(x, 0);
    if constexpr (G >= 4) x = dpp_add(x, 1);
    if constexpr (G >= 8) x = dpp_add(x, 2);
    if constexpr (G >= 16) x = dpp_add(x, 3);
    if constexpr (G >= 32) x += __shfl_xor(x, 16, 64);
    if constexpr (G >= 64) x += __shfl_xor(x, 32, 64);
    return x;
  }
  __device__ __forceinline__ double gsum(double x) const {  // fp64 paths (Point): plain butterfly
#pragma unroll
    for (int o = 1; o < G; o <<= 1) x += __shfl_xor(x, o, 64);
    return x;
  }
  __device__ __forceinline__ bool any(bool p) const { return __any(p) != 0; }
  // any() restricted to this lane group: one ballot, no shuffles
  __device__ __forceinline__ bool gany(bool p) const {
    unsigned long long b = __ballot(p);
    unsigned lane = __lane_id();
    unsigned long long gm = (G >= 64) ? ~0ULL : (((1ULL << (G & 63)) - 1ULL) << (lane - (unsigned)l));
    return (b & gm) != 0ULL;
  }
};

// ------------------------------------------------------------------ RNG (same definition as the oracle's mzo_rng_u32)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t env, uint32_t counter) {
  uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (env + 1));
  z = mix64(z + 0x9E3779B97F4A7C15ULL * ((uint64_t)counter + 1));
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float rng_u01(uint64_t seed, uint64_t env, uint32_t c) { return (float)(rng_u32(seed, env, c) >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float rng_normal(uint64_t seed, uint64_t env, uint32_t c) {
  float u1 = ((float)(rng_u32(seed, env, c) >> 8) + 1.0f) * (1.0f / 16777216.0f);
  float u2 = rng_u01(seed, env, c + 1);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
// reset distribution (ant.py:84-96, point.py:71-81): qpos0 + U(-.1,.1); qvel by kind
__device__ __forceinline__ float reset_qpos(float q0, uint64_t seed, uint64_t env, int i) { return q0 + (-0.1f + 0.2f * rng_u01(seed, env, (uint32_t)i)); }
__device__ __forceinline__ float reset_qvel(int kind, int nq, uint64_t seed, uint64_t env, int i) {
  uint32_t c = (uint32_t)(nq + 2 * i);
  if (kind == 0) return 0.1f * rng_normal(seed, env, c);
  if (kind == 1) return 0.1f * rng_u01(seed, env, c);
  return -0.1f + 0.2f * rng_u01(seed, env, c);
}
__host__ __device__ __forceinline__ uint64_t episode_seed(uint64_t seed, uint32_t episode) {
  return episode == 0 ? seed : mix64(seed ^ (0xD6E8FEB86659FD93ULL * (uint64_t)episode));
}

// ------------------------------------------------------------------ Ant kernels
template <int NB, int G, bool P>
__device__ __forceinline__ void ant_load(const DevCtx<G, P>& cx, AntScratchT<NB>& s, const float* rec) {
  using D = AntDims<NB>;
  for (int i = cx.l; i < D::REC_T; i += G) {
    float v = rec[i];
    if (i < D::NQ) s.qpos[i] = v;
    else if (i < D::NQ + D::NV) s.qvel[i - D::NQ] = v;
    else s.warm[i - D::NQ - D::NV] = v;
  }
}
template <int NB, int G, bool P>
__device__ __forceinline__ void ant_store(const DevCtx<G, P>& cx, const AntScratchT<NB>& s, float* rec) {
  using D = AntDims<NB>;
  for (int i = cx.l; i < D::REC_T; i += G) {
    float v = i < D::NQ ? s.qpos[i] : (i < D::NQ + D::NV ? s.qvel[i - D::NQ] : s.warm[i - D::NQ - D::NV]);
    rec[i] = v;
  }
}

struct AntIO {  // per-env staging of the step's inputs / outputs next to the scratch block
  float act[ANT_NU], obs[MZ_MAX_OBS], out[8];
  int iout[4];
};
template <int NB>
struct alignas(16) AntEnvLDS { AntScratchT<NB> s; AntIO io; };

// Register budget per lane-group width: the batch is fixed (4096 envs/GPU), so the wave count is 64*N/G and
// the kernel must fit  N*G/64 / 1024 SIMDs  waves per SIMD to be resident in one round.
template <int G>
constexpr int ant_waves_per_simd() { return G >= 64 ? 4 : (G == 32 ? 2 : 1); }

template <int NB, int G, bool PROF>
__global__ __launch_bounds__(256, (NB ? (G >= 64 ? 2 : 1) : ant_waves_per_simd<G>())) void ant_step_kernel(const AntDev* __restrict__ Kp, int n, float* __restrict__ state, const float* __restrict__ actions,
                                                       float* __restrict__ obs, float* __restrict__ reward,
                                                       uint8_t* __restrict__ done, int* __restrict__ goal_idx,
                                                       float* __restrict__ info, int* __restrict__ status, int auto_reset,
                                                       uint64_t seed, uint64_t env0, unsigned long long* __restrict__ prof) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  using D = AntDims<NB>;
  const AntDev& K = *Kp;  // model constants: scalar loads from a device-resident block (L2 / scalar-cache hits)
  AntEnvLDS<NB>* lds = reinterpret_cast<AntEnvLDS<NB>*>(lds_raw);
  const int EPB = blockDim.x / G;  // envs per workgroup (blockDim.x = 64 * waves per workgroup)
  DevCtx<G, PROF> cx{(int)threadIdx.x % G};
  const int slot = threadIdx.x / G;
  int env = blockIdx.x * EPB + slot;
  const bool live = env < n;
  if (!live) env = n - 1;  // surplus groups shadow the last env (no stores)
  AntScratchT<NB>& s = lds[slot].s;
  float* act_s = lds[slot].io.act;
  float* obs_s = lds[slot].io.obs;
  float* out_s = lds[slot].io.out;
  int* iout_s = lds[slot].io.iout;
  float* rec = state + (size_t)env * D::REC;
  ant_load<NB>(cx, s, rec);
  for (int i = cx.l; i < ANT_NU; i += G) act_s[i] = actions[(size_t)env * ANT_NU + i];
  if (cx.l == 0) { iout_s[2] = ((const int*)rec)[D::REC_T]; iout_s[3] = ((const int*)rec)[D::REC_T + 1]; }  // t, episode: parked in LDS for the step
  if constexpr (PROF) { if (cx.l == 0) { for (int k = 0; k < 16; k++) s.prof[k] = 0; s.prof_t0 = __builtin_amdgcn_s_memtime(); } }
  cx.sync();
  ant_env_step<NB>(cx, K, s, act_s, obs_s, &out_s[0], (uint8_t*)&iout_s[0], &iout_s[1], &out_s[1], &iout_s[2]);
  cx.sync();
  // Epilogue.  Everything it needs is re-derived from the thread index behind an opaque barrier, so that no per-lane
  // address or index stays live (and gets spilled to scratch) across the 20 forward evaluations above.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int slot2 = tid / G, l2 = tid % G;
  int env2 = blockIdx.x * EPB + slot2;
  const bool live2 = env2 < n;
  if (!live2) env2 = n - 1;
  AntScratchT<NB>& s2 = lds[slot2].s;
  const float* obs2 = lds[slot2].io.obs;
  const float* out2 = lds[slot2].io.out;
  const int* iout2 = lds[slot2].io.iout;
  float* rec2 = state + (size_t)env2 * D::REC;
  const int obs_dim = ANT_OBS + (K.observe_blocks ? 3 * NB : 0);
  const uint8_t d = *(const uint8_t*)&iout2[0];
  const int t_new = iout2[2];
  uint32_t episode = (uint32_t)iout2[3];
  if (live2) {
    for (int i = l2; i < obs_dim; i += G) obs[(size_t)env2 * obs_dim + i] = obs2[i];
    if (l2 == 0) {
      reward[env2] = out2[0];
      done[env2] = d;
      if (goal_idx) goal_idx[env2] = iout2[1];
      if (s2.status) atomicOr(&status[env2], s2.status);
    }
    if (info) for (int i = l2; i < 4; i += G) info[(size_t)env2 * 4 + i] = out2[1 + i];
  }
  if (auto_reset && d) {  // masked reset inside the step
    episode += 1;
    uint64_t es = episode_seed(seed, episode);
    // robot coordinates get the reset noise; movable blocks return to their cells (ant.py:84-96)
    for (int i = l2; i < D::NQ; i += G) s2.qpos[i] = i < ANT_NQ ? reset_qpos(K.qpos0[i], es, env0 + (uint64_t)env2, i) : 0.f;
    for (int i = l2; i < D::NV; i += G) { s2.qvel[i] = i < ANT_NV ? reset_qvel(K.reset_kind, D::NQ, es, env0 + (uint64_t)env2, i) : 0.f; s2.warm[i] = 0.f; }
    cx.sync();
    if (l2 == 0) {  // root quaternion normalised in place, as at mz_reset [ASSUME-8]
      float qn = 1.0f / sqrtf(s2.qpos[3] * s2.qpos[3] + s2.qpos[4] * s2.qpos[4] + s2.qpos[5] * s2.qpos[5] + s2.qpos[6] * s2.qpos[6]);
      for (int i = 3; i < 7; i++) s2.qpos[i] *= qn;
    }
  }
  cx.sync();
  if (live2) {
    DevCtx<G, PROF> cx2{l2};
    ant_store<NB>(cx2, s2, rec2);
    if (l2 == 0) { ((int*)rec2)[D::REC_T] = (auto_reset && d) ? 0 : t_new; ((uint32_t*)rec2)[D::REC_T + 1] = episode; }
  }
  if constexpr (PROF) {
    cx.tick(s, 10);
    if (live2 && threadIdx.x == 0 && prof) {
      unsigned long long tot = 0;
      for (int k = 0; k < 13; k++) tot += s.prof[k];
      for (int k = 0; k < 16; k++) if (k != 13 && k != 14) atomicAdd(&prof[k], (unsigned long long)s.prof[k]);
      atomicMax(&prof[13], tot);                  // slowest wave of the accumulation window
      atomicAdd(&prof[14], (tot >> 8) * (tot >> 8));  // sum of squares of the per-wave totals (units of 256 cycles)
    }
  }
}

template <int NB, int G>
__global__ __launch_bounds__(64) void ant_forward_kernel(AntDev K, int n, const float* __restrict__ state,
                                                          const float* __restrict__ actions, float* __restrict__ qacc,
                                                          int* __restrict__ counts) {
  using D = AntDims<NB>;
  constexpr int EPB = 64 / G;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  AntScratchT<NB>* sc = reinterpret_cast<AntScratchT<NB>*>(lds_raw);
  DevCtx<G> cx{(int)threadIdx.x % G};
  const int slot = threadIdx.x / G;
  int env = blockIdx.x * EPB + slot;
  const bool live = env < n;
  if (!live) env = n - 1;
  AntScratchT<NB>& s = sc[slot];
  ant_load<NB>(cx, s, state + (size_t)env * D::REC);
  for (int i = cx.l; i < D::NV; i += G) s.fact[i] = 0.f;
  if (cx.l == 0) s.status = 0;
  cx.sync();
  if (actions)
    for (int u = cx.l; u < ANT_NU; u += G) s.fact[K.act_dof[u]] = K.gear * fminf(fmaxf(actions[(size_t)env * ANT_NU + u], K.ctrl_lo), K.ctrl_hi);
  cx.sync();
  ant_forward<NB>(cx, K, s, true);
  if (live) {
    for (int i = cx.l; i < D::NV; i += G) qacc[(size_t)env * D::NV + i] = s.qacc[i];
    if (cx.l == 0 && counts) { counts[2 * env] = s.ncon; counts[2 * env + 1] = s.iters; }
  }
}

struct AntLayout { int nq, nv, rec, rec_t, obs_dim, nblock3; };  // record layout of the instantiated NB

__global__ void ant_reset_kernel(AntDev K, AntLayout L, int n, float* state, const uint8_t* mask, uint64_t seed, uint64_t env0, float* obs) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  float* rec = state + (size_t)env * L.rec;
  if (!mask || mask[env]) {
    for (int i = 0; i < L.nq; i++) rec[i] = i < ANT_NQ ? reset_qpos(K.qpos0[i], seed, env0 + (uint64_t)env, i) : 0.f;
    {  // set_state -> mj_forward: mj_kinematics normalises the root quaternion in place [ASSUME-8]
      float qn = 1.0f / sqrtf(rec[3] * rec[3] + rec[4] * rec[4] + rec[5] * rec[5] + rec[6] * rec[6]);
      for (int i = 3; i < 7; i++) rec[i] *= qn;
    }
    for (int i = 0; i < L.nv; i++) {
      rec[L.nq + i] = i < ANT_NV ? reset_qvel(K.reset_kind, L.nq, seed, env0 + (uint64_t)env, i) : 0.f;
      rec[L.nq + L.nv + i] = 0.f;
    }
    ((int*)rec)[L.rec_t] = 0;
    ((uint32_t*)rec)[L.rec_t + 1] = 0;
  }
  if (obs) {
    float* o = obs + (size_t)env * L.obs_dim;
    int k = 0;
    for (int i = 0; i < 3; i++) o[k++] = rec[i];
    for (int b = 0; b < L.nblock3 / 3; b++) { o[k++] = K.block_pos0[b][0] + rec[15 + 2 * b]; o[k++] = K.block_pos0[b][1] + rec[16 + 2 * b]; o[k++] = K.block_pos0[b][2]; }
    for (int i = 3; i < ANT_NQ; i++) o[k++] = rec[i];
    for (int i = 0; i < ANT_NV; i++) o[k++] = rec[L.nq + i];
    o[k] = (float)((int*)rec)[L.rec_t] * 0.001f;
  }
}

// row-major API arrays <-> state records
__global__ void ant_set_state_kernel(AntLayout L, int n, float* state, const float* qpos, const float* qvel, const float* warm, const int* t) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int env = idx / L.rec, i = idx % L.rec;
  if (env >= n) return;
  float* rec = state + (size_t)env * L.rec;
  if (i < L.nq) { if (qpos) rec[i] = qpos[(size_t)env * L.nq + i]; }
  else if (i < L.nq + L.nv) { if (qvel) rec[i] = qvel[(size_t)env * L.nv + i - L.nq]; }
  else if (i < L.rec_t) { if (warm) rec[i] = warm[(size_t)env * L.nv + i - L.nq - L.nv]; }
  else if (i == L.rec_t) { if (t) ((int*)rec)[L.rec_t] = t[env]; }
}
__global__ void ant_get_state_kernel(AntLayout L, int n, const float* state, float* qpos, float* qvel, float* warm, int* t) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int env = idx / L.rec, i = idx % L.rec;
  if (env >= n) return;
  const float* rec = state + (size_t)env * L.rec;
  if (i < L.nq) { if (qpos) qpos[(size_t)env * L.nq + i] = rec[i]; }
  else if (i < L.nq + L.nv) { if (qvel) qvel[(size_t)env * L.nv + i - L.nq] = rec[i]; }
  else if (i < L.rec_t) { if (warm) warm[(size_t)env * L.nv + i - L.nq - L.nv] = rec[i]; }
  else if (i == L.rec_t) { if (t) t[env] = ((const int*)rec)[L.rec_t]; }
}

// ------------------------------------------------------------------ Point kernels (SoA: q0 q1 q2 v0 v1 v2 | t | episode)
struct PointState { float* qv; int* t; uint32_t* ep; };

// One MazeEnv.step of the Point (+ NB movable blocks or NS object balls): G lanes per env, PlanarScratch in LDS, SoA state in HBM
// (q_0..q_{NV-1} | v_0..v_{NV-1}, each [n]).
template <int NB, int NS, int G>
__global__ __launch_bounds__(64) void planar_step_kernel(const PointDev* __restrict__ Pp, int n, PointState S,
                                                          const float* __restrict__ actions, float* __restrict__ obs,
                                                          float* __restrict__ reward, uint8_t* __restrict__ done,
                                                          int* __restrict__ goal_idx, float* __restrict__ info,
                                                          int* __restrict__ status, int auto_reset, uint64_t seed, uint64_t env0) {
  using D = PlanarDims<NB, NS>;
  constexpr int NV = D::NV, NOBS = D::NOBS, EPW = 64 / G;
  __shared__ PointDev P;  // segment table + task shared by the block (L2-resident source)
  __shared__ PlanarScratch<NB, NS> scr[EPW];
  __shared__ float obuf[EPW][MZ_MAX_OBS];
  for (int i = threadIdx.x; i < (int)(sizeof(PointDev) / 4); i += blockDim.x) ((uint32_t*)&P)[i] = ((const uint32_t*)Pp)[i];
  __syncthreads();
  DevCtx<G> cx{(int)threadIdx.x % G};
  const int grp = threadIdx.x / G;
  int env = blockIdx.x * EPW + grp;
  const bool live = env < n;
  if (!live) env = n - 1;  // idle groups shadow the last env (no stores) so that every lane reaches the wave-level votes
  PlanarScratch<NB, NS>& s = scr[grp];
  for (int k = cx.l; k < NV; k += G) { s.q[k] = (double)S.qv[(size_t)k * n + env]; s.v[k] = (double)S.qv[(size_t)(NV + k) * n + env]; }
  double a[2] = {(double)actions[(size_t)env * 2], (double)actions[(size_t)env * 2 + 1]};
  const int t_new = S.t[env] + 1;
  cx.sync();
  planar_env_step<NB, NS>(cx, P, s, a);
  float* o = obuf[grp];
  for (int i = cx.l; i < NOBS; i += G) o[i] = planar_obs_elem<NB, NS>(P, s, i, t_new);
  cx.sync();
  float outer; int tm, gi;
  task_eval_dev(P.task, o, &outer, &tm, &gi);  // flags from the fp32 observation that is returned
  const uint8_t d = (uint8_t)((tm ? 1 : 0) | (t_new >= P.task.max_steps ? 2 : 0));
  if (live) {
    for (int i = cx.l; i < NOBS; i += G) obs[(size_t)env * NOBS + i] = o[i];
    if (cx.l == 0) {
      reward[env] = outer;  // Point inner reward is 0.0 (point.py:61)
      done[env] = d;
      if (goal_idx) goal_idx[env] = gi;
      if (info) { info[(size_t)env * 4] = o[0]; info[(size_t)env * 4 + 1] = o[1]; info[(size_t)env * 4 + 2] = 0.f; info[(size_t)env * 4 + 3] = 0.f; }
      int st = s.status;
      bool badv = false;
      for (int k = 0; k < NV; k++) badv = badv || !(fabs(s.q[k]) < 1e10) || !(fabs(s.v[k]) < 1e10);
      if (badv) st |= MZ_STATUS_BAD_STATE;
      if (st) atomicOr(&status[env], st);
    }
    uint32_t ep = S.ep[env];
    const bool rst = auto_reset && d;
    if (rst) ep += 1;
    const uint64_t es = episode_seed(seed, ep);
    for (int k = cx.l; k < NV; k += G) {
      float qk = (float)s.q[k], vk = (float)s.v[k];
      if (rst) {  // point.py:71-81: noise on the robot, blocks back to their spawn state
        qk = k < 3 ? reset_qpos((float)P.qpos0[k], es, env0 + (uint64_t)env, k) : 0.f;
        vk = k < 3 ? reset_qvel(P.reset_kind, NV, es, env0 + (uint64_t)env, k) : 0.f;
      }
      S.qv[(size_t)k * n + env] = qk;
      S.qv[(size_t)(NV + k) * n + env] = vk;
    }
    if (cx.l == 0) { S.t[env] = rst ? 0 : t_new; S.ep[env] = ep; }
  }
}

template <int NB, int NS>
__global__ void point_reset_kernel(const PointDev* Pp, int n, PointState S, const uint8_t* mask, uint64_t seed, uint64_t env0, float* obs) {
  constexpr int NV = 3 + 2 * NB + 3 * NS, NOBS = 7 + 3 * NB + 3 * NS;
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  if (!mask || mask[env]) {
    for (int k = 0; k < NV; k++) {
      S.qv[(size_t)k * n + env] = k < 3 ? reset_qpos((float)Pp->qpos0[k], seed, env0 + (uint64_t)env, k) : 0.f;
      S.qv[(size_t)(NV + k) * n + env] = k < 3 ? reset_qvel(Pp->reset_kind, NV, seed, env0 + (uint64_t)env, k) : 0.f;
    }
    S.t[env] = 0;
    S.ep[env] = 0;
  }
  if (obs) {
    const int nb3 = (Pp->observe_blocks ? 3 * NB : 0) + (Pp->observe_balls ? 3 * NS : 0);
    float* o = obs + (size_t)env * NOBS;
    for (int k = 0; k < 3; k++) { o[k] = S.qv[(size_t)k * n + env]; o[3 + nb3 + k] = S.qv[(size_t)(NV + k) * n + env]; }
    if (NS > 0 && nb3) {
      o[3] = (float)Pp->ball_pos0[0] + S.qv[(size_t)3 * n + env]; o[4] = (float)Pp->ball_pos0[1] + S.qv[(size_t)4 * n + env];
      o[5] = (float)Pp->ball_pos0[2];
    }
    for (int b = 0; b < NB && nb3; b++) {
      o[3 + 3 * b] = (float)Pp->block_pos0[b][0] + S.qv[(size_t)(3 + 2 * b) * n + env];
      o[4 + 3 * b] = (float)Pp->block_pos0[b][1] + S.qv[(size_t)(4 + 2 * b) * n + env];
      o[5 + 3 * b] = (float)Pp->block_pos0[b][2];
    }
    o[6 + nb3] = (float)S.t[env] * 0.001f;
  }
}

template <int KQ>
__global__ void point_set_state_kernel(int n, PointState S, const float* qpos, const float* qvel, const int* t) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  for (int k = 0; k < KQ; k++) {
    if (qpos) S.qv[(size_t)k * n + env] = qpos[(size_t)env * KQ + k];
    if (qvel) S.qv[(size_t)(KQ + k) * n + env] = qvel[(size_t)env * KQ + k];
  }
  if (t) S.t[env] = t[env];
}
template <int KQ>
__global__ void point_get_state_kernel(int n, PointState S, float* qpos, float* qvel, float* warm, int* t) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  for (int k = 0; k < KQ; k++) {
    if (qpos) qpos[(size_t)env * KQ + k] = S.qv[(size_t)k * n + env];
    if (qvel) qvel[(size_t)env * KQ + k] = S.qv[(size_t)(KQ + k) * n + env];
    if (warm) warm[(size_t)env * KQ + k] = 0.f;
  }
  if (t) t[env] = S.t[env];
}

// ------------------------------------------------------------------ Swimmer / Reacher kernels (NL links, NB inert movable blocks;
// SoA: q0..q[NV-1] v0..v[NV-1] | t | episode with NV = NL + 2 + 2 NB)
template <int NL, int NB>
__global__ __launch_bounds__(256) void swimmer_step_kernel(const SwimmerDev* __restrict__ Pp, int n, PointState S,
                                                            const float* __restrict__ actions, float* __restrict__ obs,
                                                            float* __restrict__ reward, uint8_t* __restrict__ done,
                                                            int* __restrict__ goal_idx, float* __restrict__ info,
                                                            int* __restrict__ status, int auto_reset, uint64_t seed, uint64_t env0) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  constexpr int NR = NL + 2, NV = NR + 2 * NB, NH = NL - 1;
  const SwimmerDev& P = *Pp;
  const int nb3 = P.observe_blocks ? 3 * NB : 0, NO = 2 * NR + 1 + nb3;
  double q[NR], v[NR], a[NH], inner, inf4[4];
  for (int k = 0; k < NR; k++) { q[k] = (double)S.qv[(size_t)k * n + env]; v[k] = (double)S.qv[(size_t)(NV + k) * n + env]; }
  for (int k = 0; k < NH; k++) a[k] = (double)actions[(size_t)env * NH + k];
  int t_new;
  int st = swimmer_env_step<NL>(P, q, v, a, S.t[env], &inner, inf4, &t_new);
  // blocks: no contacts (swimmer.xml:3 collision="predefined"), only the medium's drag on a moving box
  float bq[2 * NB + 1], bv[2 * NB + 1];
  for (int b = 0; b < NB; b++) {
    double q2[2] = {(double)S.qv[(size_t)(NR + 2 * b) * n + env], (double)S.qv[(size_t)(NR + 2 * b + 1) * n + env]};
    double v2[2] = {(double)S.qv[(size_t)(NV + NR + 2 * b) * n + env], (double)S.qv[(size_t)(NV + NR + 2 * b + 1) * n + env]};
    if (v2[0] != 0.0 || v2[1] != 0.0) swimmer_block_step(P, q2, v2);
    bq[2 * b] = (float)q2[0]; bq[2 * b + 1] = (float)q2[1]; bv[2 * b] = (float)v2[0]; bv[2 * b + 1] = (float)v2[1];
  }
  float o[2 * NR + 1 + 3 * NB];
  for (int k = 0; k < 3; k++) o[k] = (float)q[k];
  for (int b = 0; b < NB && nb3; b++) {
    o[3 + 3 * b] = (float)(P.block_pos0[b][0] + (double)bq[2 * b]); o[4 + 3 * b] = (float)(P.block_pos0[b][1] + (double)bq[2 * b + 1]);
    o[5 + 3 * b] = (float)P.block_pos0[b][2];
  }
  for (int k = 3; k < NR; k++) o[nb3 + k] = (float)q[k];
  for (int k = 0; k < NR; k++) o[nb3 + NR + k] = (float)v[k];
  o[nb3 + 2 * NR] = (float)t_new * 0.001f;
  float outer; int tm, gi;
  task_eval_dev(P.task, o, &outer, &tm, &gi);
  uint8_t d = (uint8_t)((tm ? 1 : 0) | (t_new >= P.task.max_steps ? 2 : 0));
  for (int k = 0; k < NO; k++) obs[(size_t)env * NO + k] = o[k];
  reward[env] = (float)(P.task.inner_scale * inner) + outer;
  done[env] = d;
  if (goal_idx) goal_idx[env] = gi;
  if (info) for (int k = 0; k < 4; k++) info[(size_t)env * 4 + k] = (float)inf4[k];
  bool badv = false;
  for (int k = 0; k < NR; k++) badv = badv || !(fabs(q[k]) < 1e10) || !(fabs(v[k]) < 1e10);
  if (badv) st |= MZ_STATUS_BAD_STATE;
  if (st) atomicOr(&status[env], st);
  uint32_t ep = S.ep[env];
  const bool rst = auto_reset && d;
  if (rst) { ep += 1; t_new = 0; }
  const uint64_t es = episode_seed(seed, ep);
  for (int k = 0; k < NV; k++) {
    float qk = k < NR ? (float)q[k] : bq[k - NR], vk = k < NR ? (float)v[k] : bv[k - NR];
    if (rst) {
      qk = k < NR ? reset_qpos((float)P.qpos0[k], es, env0 + (uint64_t)env, k) : 0.f;
      vk = k < NR ? reset_qvel(P.reset_kind, NV, es, env0 + (uint64_t)env, k) : 0.f;
    }
    S.qv[(size_t)k * n + env] = qk;
    S.qv[(size_t)(NV + k) * n + env] = vk;
  }
  S.t[env] = t_new;
  S.ep[env] = ep;
}

template <int NL, int NB>
__global__ void swimmer_reset_kernel(const SwimmerDev* Pp, int n, PointState S, const uint8_t* mask, uint64_t seed, uint64_t env0, float* obs) {
  constexpr int NR = NL + 2, NV = NR + 2 * NB;
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  if (!mask || mask[env]) {
    for (int k = 0; k < NV; k++) {
      S.qv[(size_t)k * n + env] = k < NR ? reset_qpos((float)Pp->qpos0[k], seed, env0 + (uint64_t)env, k) : 0.f;
      S.qv[(size_t)(NV + k) * n + env] = k < NR ? reset_qvel(Pp->reset_kind, NV, seed, env0 + (uint64_t)env, k) : 0.f;
    }
    S.t[env] = 0;
    S.ep[env] = 0;
  }
  if (obs) {
    const int nb3 = Pp->observe_blocks ? 3 * NB : 0, NO = 2 * NR + 1 + nb3;
    float* o = obs + (size_t)env * NO;
    for (int k = 0; k < 3; k++) o[k] = S.qv[(size_t)k * n + env];
    for (int b = 0; b < NB && nb3; b++) {
      o[3 + 3 * b] = (float)Pp->block_pos0[b][0] + S.qv[(size_t)(NR + 2 * b) * n + env];
      o[4 + 3 * b] = (float)Pp->block_pos0[b][1] + S.qv[(size_t)(NR + 2 * b + 1) * n + env];
      o[5 + 3 * b] = (float)Pp->block_pos0[b][2];
    }
    for (int k = 3; k < NR; k++) o[nb3 + k] = S.qv[(size_t)k * n + env];
    for (int k = 0; k < NR; k++) o[nb3 + NR + k] = S.qv[(size_t)(NV + k) * n + env];
    o[nb3 + 2 * NR] = (float)S.t[env] * 0.001f;
  }
}

__global__ void fetch_clear_status_kernel(int n, int* status, int* out) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  out[env] = status[env];
  status[env] = 0;
}

// ------------------------------------------------------------------ handle
struct mz_handle {
  mz_model model;
  int n, device, robot;
  AntDev ant;
  AntDev* ant_dev;  // device copy read by the step kernel (refreshed when an option changes it)
  int ant_dirty;
  AntLayout lay;
  PointDev* point_dev;  // device copy
  PointDev point;
  SwimmerDev* swimmer_dev;
  SwimmerDev swimmer;
  float* state;         // ant: [n][48]; point: [6][n]
  int* pt_t;
  uint32_t* pt_ep;
  int* status;
  unsigned long long* prof;  // 16 phase-cycle accumulators (option "profile_phases")
  int auto_reset, lanes, waves_per_block;
  uint64_t seed, env0;  // env0: global slot of local env 0 (sharded runs)
  char err[256];
  // kernel timing ring (option "time_kernels")
  int lanes_set;  // lanes_per_env chosen by the caller (else the per-robot default)
  int ntime, itime;
  hipEvent_t* ev;  // 2 * ntime
  long nsteps;
};

static int set_err(mz_handle* h, int code, const char* what, hipError_t e) {
  if (h) snprintf(h->err, sizeof(h->err), "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
  return code;
}
#define HIPCHK(h, call)                                                      \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) return set_err((h), MZ_ERR_HIP, #call, _e);        \
  } while (0)

template <int NB, int G>
static hipError_t launch_ant_step(mz_handle* h, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, int* gi, float* inf) {
  int wpb = h->waves_per_block;
  int epb = wpb * 64 / G;
  size_t lds = (size_t)epb * sizeof(AntEnvLDS<NB>);
  while (lds > 160 * 1024 && wpb > 1) { wpb /= 2; epb = wpb * 64 / G; lds = (size_t)epb * sizeof(AntEnvLDS<NB>); }
  const dim3 grid((h->n + epb - 1) / epb), block(64 * wpb);
  hipError_t e;
  if (h->prof) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ant_step_kernel<NB, G, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ant_step_kernel<NB, G, true>), grid, block, lds, st, h->ant_dev, h->n, h->state, a, o, r, d, gi, inf, h->status,
                       h->auto_reset, h->seed, h->env0, h->prof);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ant_step_kernel<NB, G, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ant_step_kernel<NB, G, false>), grid, block, lds, st, h->ant_dev, h->n, h->state, a, o, r, d, gi, inf, h->status,
                       h->auto_reset, h->seed, h->env0, (unsigned long long*)nullptr);
  }
  return hipSuccess;
}
template <int NB, int G>
static hipError_t launch_ant_forward(mz_handle* h, hipStream_t st, const float* a, float* qacc, int* counts) {
  constexpr int EPB = 64 / G;
  const size_t lds = (size_t)EPB * sizeof(AntScratchT<NB>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ant_forward_kernel<NB, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ant_forward_kernel<NB, G>), dim3((h->n + EPB - 1) / EPB), dim3(64), lds, st, h->ant, h->n, h->state, a, qacc, counts);
  return hipSuccess;
}
// lane widths: the plain ant is instantiated for 8/16/32/64 lanes per env, mazes with movable blocks for 16/32/64.
// Default: 32 for the plain ant; 64 with blocks (their contact sets keep 64 lanes busy, and the 2048-env batches of those
// configs then fill the chip with two waves per SIMD instead of one).
template <int NB>
static hipError_t dispatch_ant_step(mz_handle* h, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, int* gi, float* inf) {
  if constexpr (NB == 0) {
    if (h->lanes_set && h->lanes == 8) return launch_ant_step<NB, 8>(h, st, a, o, r, d, gi, inf);
  }
  const int lanes = h->lanes_set ? h->lanes : (NB ? 64 : 32);
  if (lanes == 64) return launch_ant_step<NB, 64>(h, st, a, o, r, d, gi, inf);
  if (lanes == 16) return launch_ant_step<NB, 16>(h, st, a, o, r, d, gi, inf);
  return launch_ant_step<NB, 32>(h, st, a, o, r, d, gi, inf);
}
template <int NB>
static hipError_t dispatch_ant_forward(mz_handle* h, hipStream_t st, const float* a, float* qacc, int* counts) {
  if (h->lanes == 16) return launch_ant_forward<NB, 16>(h, st, a, qacc, counts);
  return launch_ant_forward<NB, 32>(h, st, a, qacc, counts);
}

extern "C" {

int mz_abi_version(void) { return MZ_ABI_VERSION; }
uint64_t mz_model_sizeof(void) { return sizeof(mz_model); }

mz_handle* mz_create(const mz_model* model, int32_t num_envs, int32_t device, char* err, int32_t errlen) {
  auto fail = [&](const char* msg) -> mz_handle* {
    if (err && errlen > 0) { strncpy(err, msg, (size_t)errlen - 1); err[errlen - 1] = 0; }
    return nullptr;
  };
  if (!model || num_envs <= 0) return fail("mz_create: bad arguments");
  if (model->abi_version != MZ_ABI_VERSION) return fail("mz_create: mz_model.abi_version mismatch");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("mz_create: no HIP device (the stepper has no CPU path)");
  if (device < 0 || device >= ndev) return fail("mz_create: device index out of range");
  if (hipSetDevice(device) != hipSuccess) return fail("mz_create: hipSetDevice failed");
  mz_handle* h = new (std::nothrow) mz_handle();
  if (!h) return fail("mz_create: out of memory");
  memset(h, 0, sizeof(*h));
  h->model = *model; h->n = num_envs; h->device = device; h->robot = model->robot; h->lanes = 32; h->waves_per_block = 1; h->seed = 0x5EEDULL;
  char msg[200] = {0};
  int rc = MZ_OK;
  if (model->robot == MZ_ROBOT_ANT) rc = ant_dev_from_model(&h->ant, model, msg, sizeof(msg));
  else if (model->robot == MZ_ROBOT_POINT) rc = point_dev_from_model(&h->point, model, msg, sizeof(msg));
  else if (model->robot == MZ_ROBOT_SWIMMER) rc = swimmer_dev_from_model(&h->swimmer, model, msg, sizeof(msg));
  else { rc = MZ_ERR_UNSUPPORTED; snprintf(msg, sizeof(msg), "mz_create: robot kind %d has no device kernel yet", model->robot); }
  if (rc != MZ_OK) { delete h; return fail(msg); }
  hipError_t e = hipSuccess;
  if (h->robot == MZ_ROBOT_ANT) {
    const int nb = h->ant.nblock;
    if (nb > 3) { delete h; return fail("mz_create: more than 3 movable blocks are not instantiated"); }
    h->lay.nq = ANT_NQ + 2 * nb; h->lay.nv = ANT_NV + 2 * nb; h->lay.rec_t = h->lay.nq + 2 * h->lay.nv;
    h->lay.rec = (h->lay.rec_t + 2 + 15) / 16 * 16;
    h->lay.nblock3 = model->observe_blocks ? 3 * nb : 0;
    h->lay.obs_dim = ANT_OBS + h->lay.nblock3;
    if (h->lay.obs_dim != model->obs_dim || h->lay.obs_dim > MZ_MAX_OBS) { delete h; return fail("mz_create: obs_dim mismatch"); }
    e = hipMalloc(&h->ant_dev, sizeof(AntDev));
    h->ant_dirty = 1;
    if (e == hipSuccess) e = hipMalloc(&h->state, (size_t)num_envs * h->lay.rec * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->state, 0, (size_t)num_envs * h->lay.rec * sizeof(float));
  } else {
    const int kq = h->robot == MZ_ROBOT_SWIMMER ? h->swimmer.nlink + 2 + 2 * h->swimmer.nblock : 3 + 2 * h->point.nblock + 3 * h->point.nball;
    e = hipMalloc(&h->state, (size_t)num_envs * 2 * kq * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&h->pt_t, (size_t)num_envs * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->pt_ep, (size_t)num_envs * sizeof(uint32_t));
    if (h->robot == MZ_ROBOT_SWIMMER) {
      if (e == hipSuccess) e = hipMalloc(&h->swimmer_dev, sizeof(SwimmerDev));
      if (e == hipSuccess) e = hipMemcpy(h->swimmer_dev, &h->swimmer, sizeof(SwimmerDev), hipMemcpyHostToDevice);
    } else {
      if (e == hipSuccess) e = hipMalloc(&h->point_dev, sizeof(PointDev));
      if (e == hipSuccess) e = hipMemcpy(h->point_dev, &h->point, sizeof(PointDev), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemset(h->state, 0, (size_t)num_envs * 2 * kq * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->pt_t, 0, (size_t)num_envs * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->pt_ep, 0, (size_t)num_envs * sizeof(uint32_t));
  }
  if (e == hipSuccess) e = hipMalloc(&h->status, (size_t)num_envs * sizeof(int));
  if (e == hipSuccess) e = hipMemset(h->status, 0, (size_t)num_envs * sizeof(int));
  if (e != hipSuccess) {
    snprintf(msg, sizeof(msg), "mz_create: device allocation failed: %s", hipGetErrorString(e));
    mz_destroy(h);
    return fail(msg);
  }
  return h;
}

void mz_destroy(mz_handle* h) {
  if (!h) return;
  if (h->state) (void)hipFree(h->state);
  if (h->pt_t) (void)hipFree(h->pt_t);
  if (h->pt_ep) (void)hipFree(h->pt_ep);
  if (h->point_dev) (void)hipFree(h->point_dev);
  if (h->swimmer_dev) (void)hipFree(h->swimmer_dev);
  if (h->ant_dev) (void)hipFree(h->ant_dev);
  if (h->status) (void)hipFree(h->status);
  if (h->prof) (void)hipFree(h->prof);
  if (h->ev) { for (int i = 0; i < 2 * h->ntime; i++) (void)hipEventDestroy(h->ev[i]); free(h->ev); }
  delete h;
}

const char* mz_last_error(const mz_handle* h) { return h ? h->err : "null handle"; }
int32_t mz_num_envs(const mz_handle* h) { return h->n; }
int32_t mz_obs_dim(const mz_handle* h) { return h->model.obs_dim; }
int32_t mz_nq(const mz_handle* h) { return h->model.nq; }
int32_t mz_nv(const mz_handle* h) { return h->model.nv; }
int32_t mz_nu(const mz_handle* h) { return h->model.nu; }

int32_t mz_set_option(mz_handle* h, const char* key, double value) {
  if (!h || !key) return MZ_ERR_ARG;
  if (!strcmp(key, "auto_reset")) { h->auto_reset = value != 0; return MZ_OK; }
  if (!strcmp(key, "seed")) { h->seed = (uint64_t)value; return MZ_OK; }
  if (!strcmp(key, "env_index_offset")) { h->env0 = (uint64_t)value; return MZ_OK; }
  if (!strcmp(key, "solver_iterations")) { h->ant_dirty = 1; h->ant.max_iter = (int)value; return MZ_OK; }
  if (!strcmp(key, "solver_tolerance")) { h->ant_dirty = 1; h->ant.tol = (float)value; return MZ_OK; }
  if (!strcmp(key, "solver_rtol")) { h->ant_dirty = 1; h->ant.rtol = (float)value; return MZ_OK; }
  if (!strcmp(key, "ls_iterations")) { h->ant_dirty = 1; h->ant.ls_iter = (int)value; return MZ_OK; }
  if (!strcmp(key, "lanes_per_env")) {
    int g = (int)value;
    if (g != 8 && g != 16 && g != 32 && g != 64) return set_err(h, MZ_ERR_ARG, "lanes_per_env must be 8, 16, 32 or 64", hipSuccess);
    h->lanes = g; h->lanes_set = 1;
    return MZ_OK;
  }
  if (!strcmp(key, "profile_phases")) {
    if (value != 0 && !h->prof) { HIPCHK(h, hipMalloc(&h->prof, 16 * sizeof(unsigned long long))); HIPCHK(h, hipMemset(h->prof, 0, 16 * sizeof(unsigned long long))); }
    if (value == 0 && h->prof) { (void)hipFree(h->prof); h->prof = nullptr; }
    return MZ_OK;
  }
  if (!strcmp(key, "waves_per_block")) {
    int w = (int)value;
    if (w != 1 && w != 2 && w != 4) return set_err(h, MZ_ERR_ARG, "waves_per_block must be 1, 2 or 4", hipSuccess);
    h->waves_per_block = w;
    return MZ_OK;
  }
  if (!strcmp(key, "time_kernels")) {
    if (h->ev) { for (int i = 0; i < 2 * h->ntime; i++) (void)hipEventDestroy(h->ev[i]); free(h->ev); h->ev = nullptr; }
    h->ntime = (int)value; h->itime = 0;
    if (h->ntime > 0) {
      h->ev = (hipEvent_t*)calloc((size_t)2 * h->ntime, sizeof(hipEvent_t));
      for (int i = 0; i < 2 * h->ntime; i++) HIPCHK(h, hipEventCreate(&h->ev[i]));
    }
    return MZ_OK;
  }
  return set_err(h, MZ_ERR_ARG, "unknown option", hipSuccess);
}

int32_t mz_reset(mz_handle* h, const uint8_t* mask_dev, uint64_t seed, float* obs_dev, void* stream) {
  if (!h) return MZ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  h->seed = seed;
  int nb = (h->n + 255) / 256;
  if (h->robot == MZ_ROBOT_ANT) hipLaunchKernelGGL(ant_reset_kernel, dim3(nb), dim3(256), 0, st, h->ant, h->lay, h->n, h->state, mask_dev, seed, h->env0, obs_dev);
  else if (h->robot == MZ_ROBOT_SWIMMER) {
    PointState S{h->state, h->pt_t, h->pt_ep};
#define MZ_SW_RESET(NL, NB) hipLaunchKernelGGL((swimmer_reset_kernel<NL, NB>), dim3(nb), dim3(256), 0, st, h->swimmer_dev, h->n, S, mask_dev, seed, h->env0, obs_dev)
    if (h->swimmer.nlink == 3) { if (h->swimmer.nblock) MZ_SW_RESET(3, 1); else MZ_SW_RESET(3, 0); }
    else { if (h->swimmer.nblock) MZ_SW_RESET(2, 1); else MZ_SW_RESET(2, 0); }
#undef MZ_SW_RESET
  } else {
    PointState S{h->state, h->pt_t, h->pt_ep};
    if (h->point.nball) hipLaunchKernelGGL((point_reset_kernel<0, 1>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev);
    else switch (h->point.nblock) {
      case 0: hipLaunchKernelGGL((point_reset_kernel<0, 0>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev); break;
      case 1: hipLaunchKernelGGL((point_reset_kernel<1, 0>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev); break;
      case 2: hipLaunchKernelGGL((point_reset_kernel<2, 0>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev); break;
      default: hipLaunchKernelGGL((point_reset_kernel<3, 0>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev); break;
    }
  }
  HIPCHK(h, hipGetLastError());
  return MZ_OK;
}

int32_t mz_set_state(mz_handle* h, const float* qpos_dev, const float* qvel_dev, const float* warmstart_dev, const int32_t* t_dev,
                     void* stream) {
  if (!h) return MZ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (h->robot == MZ_ROBOT_ANT) {
    int tot = h->n * h->lay.rec;
    hipLaunchKernelGGL(ant_set_state_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, h->lay, h->n, h->state, qpos_dev, qvel_dev, warmstart_dev, t_dev);
  } else {
    PointState S{h->state, h->pt_t, h->pt_ep};
    const int kq = h->robot == MZ_ROBOT_SWIMMER ? h->swimmer.nlink + 2 + 2 * h->swimmer.nblock : 3 + 2 * h->point.nblock + 3 * h->point.nball;
    const dim3 grid((h->n + 255) / 256), blk(256);
    switch (kq) {
      case 3: hipLaunchKernelGGL(point_set_state_kernel<3>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
      case 4: hipLaunchKernelGGL(point_set_state_kernel<4>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
      case 5: hipLaunchKernelGGL(point_set_state_kernel<5>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
      case 6: hipLaunchKernelGGL(point_set_state_kernel<6>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
      case 7: hipLaunchKernelGGL(point_set_state_kernel<7>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
      default: hipLaunchKernelGGL(point_set_state_kernel<9>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
    }
  }
  HIPCHK(h, hipGetLastError());
  return MZ_OK;
}

int32_t mz_get_state(mz_handle* h, float* qpos_dev, float* qvel_dev, float* warmstart_dev, int32_t* t_dev, void* stream) {
  if (!h) return MZ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (h->robot == MZ_ROBOT_ANT) {
    int tot = h->n * h->lay.rec;
    hipLaunchKernelGGL(ant_get_state_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, h->lay, h->n, h->state, qpos_dev, qvel_dev, warmstart_dev, t_dev);
  } else {
    PointState S{h->state, h->pt_t, h->pt_ep};
    const int kq = h->robot == MZ_ROBOT_SWIMMER ? h->swimmer.nlink + 2 + 2 * h->swimmer.nblock : 3 + 2 * h->point.nblock + 3 * h->point.nball;
    const dim3 grid((h->n + 255) / 256), blk(256);
    switch (kq) {
      case 3: hipLaunchKernelGGL(point_get_state_kernel<3>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
      case 4: hipLaunchKernelGGL(point_get_state_kernel<4>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
      case 5: hipLaunchKernelGGL(point_get_state_kernel<5>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
      case 6: hipLaunchKernelGGL(point_get_state_kernel<6>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
      case 7: hipLaunchKernelGGL(point_get_state_kernel<7>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
      default: hipLaunchKernelGGL(point_get_state_kernel<9>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
    }
  }
  HIPCHK(h, hipGetLastError());
  return MZ_OK;
}

int32_t mz_step(mz_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, int32_t* goal_idx_dev,
                float* info_dev, void* stream) {
  if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev) return h ? set_err(h, MZ_ERR_ARG, "mz_step: null array", hipSuccess) : MZ_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int slot = -1;
  if (h->ntime > 0) { slot = h->itime % h->ntime; HIPCHK(h, hipEventRecord(h->ev[2 * slot], st)); }
  if (h->robot == MZ_ROBOT_ANT) {
    if (h->ant_dirty) { HIPCHK(h, hipMemcpyAsync(h->ant_dev, &h->ant, sizeof(AntDev), hipMemcpyHostToDevice, st)); h->ant_dirty = 0; }
    hipError_t le;
    switch (h->ant.nblock) {
      case 0: le = dispatch_ant_step<0>(h, st, actions_dev, obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev); break;
      case 1: le = dispatch_ant_step<1>(h, st, actions_dev, obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev); break;
      case 2: le = dispatch_ant_step<2>(h, st, actions_dev, obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev); break;
      default: le = dispatch_ant_step<3>(h, st, actions_dev, obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev); break;
    }
    HIPCHK(h, le);
  } else if (h->robot == MZ_ROBOT_SWIMMER) {
    PointState S{h->state, h->pt_t, h->pt_ep};
#define MZ_SW_STEP(NL, NB)                                                                                                          \
  hipLaunchKernelGGL((swimmer_step_kernel<NL, NB>), dim3((h->n + 255) / 256), dim3(256), 0, st, h->swimmer_dev, h->n, S, actions_dev, obs_dev, \
                     reward_dev, done_dev, goal_idx_dev, info_dev, h->status, h->auto_reset, h->seed, h->env0)
    if (h->swimmer.nlink == 3) { if (h->swimmer.nblock) MZ_SW_STEP(3, 1); else MZ_SW_STEP(3, 0); }
    else { if (h->swimmer.nblock) MZ_SW_STEP(2, 1); else MZ_SW_STEP(2, 0); }
#undef MZ_SW_STEP
  } else {
    PointState S{h->state, h->pt_t, h->pt_ep};
    // lanes per env: 16 for the bare robot (18 collision enumerators), 32 / 64 with blocks (bigger contact sets in LDS)
#define MZ_PLANAR_LAUNCH(NB, NS, G)                                                                                                    \
  hipLaunchKernelGGL((planar_step_kernel<NB, NS, G>), dim3((h->n + 64 / G - 1) / (64 / G)), dim3(64), 0, st, h->point_dev, h->n, S, actions_dev, \
                     obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev, h->status, h->auto_reset, h->seed, h->env0)
    if (h->point.nball) MZ_PLANAR_LAUNCH(0, 1, 32);
    else switch (h->point.nblock) {
      case 0:
        if (h->lanes_set && h->lanes == 8) MZ_PLANAR_LAUNCH(0, 0, 8);
        else if (h->lanes_set && h->lanes == 32) MZ_PLANAR_LAUNCH(0, 0, 32);
        else MZ_PLANAR_LAUNCH(0, 0, 16);
        break;
      case 1: MZ_PLANAR_LAUNCH(1, 0, 32); break;
      case 2: MZ_PLANAR_LAUNCH(2, 0, 64); break;
      default: MZ_PLANAR_LAUNCH(3, 0, 64); break;
    }
#undef MZ_PLANAR_LAUNCH
  }
  HIPCHK(h, hipGetLastError());
  if (slot >= 0) { HIPCHK(h, hipEventRecord(h->ev[2 * slot + 1], st)); h->itime++; }
  h->nsteps++;
  return MZ_OK;
}

int32_t mz_get_status(mz_handle* h, int32_t* status_dev, void* stream) {
  if (!h || !status_dev) return MZ_ERR_ARG;
  hipLaunchKernelGGL(fetch_clear_status_kernel, dim3((h->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->n, h->status, status_dev);
  HIPCHK(h, hipGetLastError());
  return MZ_OK;
}

int32_t mz_debug_forward(mz_handle* h, const float* actions_dev, float* qacc_dev, int32_t* counts_dev, void* stream) {
  if (!h || !qacc_dev) return MZ_ERR_ARG;
  if (h->robot != MZ_ROBOT_ANT) return set_err(h, MZ_ERR_UNSUPPORTED, "mz_debug_forward: ant only", hipSuccess);
  hipStream_t st = (hipStream_t)stream;
  hipError_t le;
  switch (h->ant.nblock) {
    case 0: le = dispatch_ant_forward<0>(h, st, actions_dev, qacc_dev, counts_dev); break;
    case 1: le = dispatch_ant_forward<1>(h, st, actions_dev, qacc_dev, counts_dev); break;
    case 2: le = dispatch_ant_forward<2>(h, st, actions_dev, qacc_dev, counts_dev); break;
    default: le = dispatch_ant_forward<3>(h, st, actions_dev, qacc_dev, counts_dev); break;
  }
  HIPCHK(h, le);
  HIPCHK(h, hipGetLastError());
  return MZ_OK;
}

// Phase timers of the PROF kernel build: copies the 16 accumulators to host memory and clears them.
int32_t mz_read_phase_cycles(mz_handle* h, uint64_t* out16_host) {
  if (!h || !out16_host || !h->prof) return MZ_ERR_ARG;
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(out16_host, h->prof, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemset(h->prof, 0, 16 * sizeof(unsigned long long)));
  return MZ_OK;
}

// Average duration (ms) of the step kernel over the recorded ring (after synchronising on the last event).
double mz_last_kernel_ms(const mz_handle* h) {
  if (!h || h->ntime <= 0 || h->itime <= 0) return -1.0;
  int cnt = h->itime < h->ntime ? h->itime : h->ntime;
  double tot = 0.0;
  for (int i = 0; i < cnt; i++) {
    if (hipEventSynchronize(h->ev[2 * i + 1]) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]) != hipSuccess) return -1.0;
    tot += ms;
  }
  return tot / cnt;
}

}  // extern "C"
