// ant_kernels.hip — HIP kernels (gfx950 / CDNA4) of the Ant path and their launchers.
//
//   ant_step_kernel<NB,G>  one MazeEnv.step for the Ant (+ NB movable blocks): G lanes per environment, 64/G
//                          environments per one-wavefront workgroup, the env's whole working set (7.7 KB for the plain
//                          ant, AntScratchT<NB>) resident in LDS across the 20 forward-dynamics evaluations of the step;
//                          HBM is touched once per step (192-B state record in, record + obs / reward / done out).
//   reset / state copy / debug kernels.
//
// HBM layout: state[N][REC] fp32 record = qpos[nq] | qvel[nv] | qacc_warmstart[nv] | t | episode | pad
//   (REC = 48 words for the plain ant, 64 with one movable block: AntDims<NB>::REC).  A lane group reads 48 consecutive
//   words: coalesced for lane-group-per-env kernels (SoA would scatter a group's loads over 48 cache lines).
//
// Built with the relaxed floating-point flags of csrc/Makefile (reciprocal division, approximate transcendentals): the
// physics is a 1e-5 tolerance quantity.  The task predicates inside (task_eval_dev) are written so that those flags
// cannot change a flag: fp64, no contraction, squared thresholds instead of sqrt.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ant_dyn.h"
#include "mz_device.h"
#include "ant_newton_rows.h"
#include "ant_forward_rows.h"
#include "mz_internal.h"

// ------------------------------------------------------------------ Ant kernels
template <int NB, int G, bool P>
__device__ __forceinline__ void ant_load(const DevCtx<G, P>& cx, AntScratchCoreT<NB>& s, const float* rec) {
  using D = AntDims<NB>;
  for (int i = cx.l; i < D::REC_T; i += G) {
    float v = rec[i];
    if (i < D::NQ) s.qpos[i] = v;
    else if (i < D::NQ + D::NV) s.qvel[i - D::NQ] = v;
    else s.warm[i - D::NQ - D::NV] = v;
  }
}
template <int NB, int G, bool P>
__device__ __forceinline__ void ant_store(const DevCtx<G, P>& cx, const AntScratchCoreT<NB>& s, float* rec) {
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
// LDS bytes of one env: scratch block + I/O staging.  The quad forward pass (plain ant / one two-slide block at >= 16 lanes per env, not
// instrumented) touches only part 1 of the scratch block (AntScratchT::slim_bytes): 4.7 + 0.3 KB per plain-ant env, 32 envs per CU.
template <int NB, int G, bool PROF>
constexpr bool ant_core_only() { return NB <= 1 && G >= 16 && !PROF; }
// the scratch type a step kernel instantiation allocates and hands down: the core where nothing beneath it may reach further
template <int NB, int G, bool PROF>
using AntScratchOf = std::conditional_t<ant_core_only<NB, G, PROF>(), AntScratchCoreT<NB>, AntScratchT<NB>>;
template <int NB, int G, bool PROF>
constexpr size_t ant_scratch_bytes() { return sizeof(AntScratchOf<NB, G, PROF>); }
template <int NB, int G, bool PROF>
constexpr size_t ant_env_lds_bytes() { return ant_scratch_bytes<NB, G, PROF>() + (sizeof(AntIO) + 15) / 16 * 16; }

// Register budget per lane-group width: the batch is fixed (4096 envs/GPU), so the wave count is 64*N/G and
// the kernel must fit  N*G/64 / 1024 SIMDs  waves per SIMD to be resident in one round.
template <int G>
constexpr int ant_waves_per_simd() { return G >= 64 ? 4 : (G == 32 ? 2 : 1); }
// WPS = 2: the second instantiation of the 16-lane plain ant (round 5), held to 256 registers IN ALL so that two of its waves share a
// SIMD.  The one-wave kernel uses 256 + 38 accumulation registers: beyond 4096 envs (more waves than the 1024 SIMDs) its launch ran in
// rounds — 0.282 ms at 4096 envs, 0.522 ms at 8192.  Forced into 256 registers the compiler spills 36 of them, all per-step lane
// constants re-read once per forward evaluation (no scratch access inside a Newton iteration): 4 % slower with one wave per SIMD
// (13.7 against 14.2 M env-steps/s at 4096 envs), but two co-resident waves fill each other's issue gaps — 21.0 against 15.5 M at 8192
// envs, 23.4 against 16.4 M at 16384.  launch_ant_step picks by the wave count of the launch (option "waves_per_simd" overrides).
template <int NB, int G, int WPS>
constexpr int ant_occupancy() { return WPS ? WPS : (NB ? (G >= 64 ? 2 : 1) : ant_waves_per_simd<G>()); }

template <int NB, int G, bool PROF, int WPS = 0>
__global__ __launch_bounds__(256, (ant_occupancy<NB, G, WPS>())) void ant_step_kernel(const AntDev* __restrict__ Kp, int n, float* __restrict__ state, const float* __restrict__ actions,
                                                       float* __restrict__ obs, float* __restrict__ reward,
                                                       uint8_t* __restrict__ done, int* __restrict__ goal_idx,
                                                       float* __restrict__ info, int* __restrict__ status, int auto_reset,
                                                       uint64_t seed, uint64_t env0, unsigned long long* __restrict__ prof,
                                                       float* __restrict__ final_obs, int ostride, float* __restrict__ record) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  using D = AntDims<NB>;
#ifdef MZ_EXP_STAMPS  // experiment build (tools/exp_launch_stamps.py): when does each wave of the PRODUCT kernel start and end?
  const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
  const AntDev& K = *Kp;  // model constants: scalar loads from a device-resident block (L2 / scalar-cache hits)
  constexpr size_t SB = ant_scratch_bytes<NB, G, PROF>(), EB = ant_env_lds_bytes<NB, G, PROF>();  // scratch block | I/O staging, per env
  const int EPB = blockDim.x / G;  // envs per workgroup (blockDim.x = 64 * waves per workgroup)
  DevCtx<G, PROF> cx{(int)threadIdx.x % G};
  cx.mfma = WPS == 2;  // two waves per SIMD: the Hessian fold goes to the matrix cores (ant_newton_rows.h)
#ifndef MZ_EXP_LCRELOAD
  if constexpr (NB <= 1 && G >= 16 && WPS != 2) ant_lane_consts(K, cx);  // (WPS = 2: per evaluation, ant_forward_rows)
#endif  // per-lane constants of the quad layout (ant_forward_rows.h), once per step
  const int slot = threadIdx.x / G;
  // XCD-aware block of envs (mz_device.h xcd_block; round 4): the workgroups of one XCD take a contiguous range of envs, so the
  // cache lines two neighbouring workgroups share — obs rows are 120 B, records 192 B, reward / done a few bytes per workgroup —
  // are merged in ONE L2 instead of being fetched for ownership and written back by several (PMC, AntPush at two envs per
  // workgroup: FETCH + WRITE 1.61 x the algorithmic bytes with the identity map)
  const int wg = xcd_block(blockIdx.x, gridDim.x);
  int env = wg * EPB + slot;
  const bool live = env < n;
  if (!live) env = n - 1;  // surplus groups shadow the last env (no stores)
  using S = AntScratchOf<NB, G, PROF>;
  S& s = *reinterpret_cast<S*>(lds_raw + (size_t)slot * EB);
  AntIO& io = *reinterpret_cast<AntIO*>(lds_raw + (size_t)slot * EB + SB);
  float* act_s = io.act;
  float* obs_s = io.obs;
  float* out_s = io.out;
  int* iout_s = io.iout;
  float* rec = state + (size_t)env * D::REC;
  ant_load<NB>(cx, s, rec);
  for (int i = cx.l; i < ANT_NU; i += G) act_s[i] = actions[(size_t)env * ANT_NU + i];
  if (cx.l == 0) { iout_s[1] = env; iout_s[2] = ((const int*)rec)[D::REC_T]; iout_s[3] = ((const int*)rec)[D::REC_T + 1]; }  // env slot (per-env goals: ant_env_step), t, episode: parked in LDS for the step
  if constexpr (PROF) { if (cx.l == 0) { for (int k = 0; k < 16; k++) s.prof[k] = 0; s.prof_t0 = __builtin_amdgcn_s_memtime(); } }
#ifdef MZ_EXP_STAMPS
  if (cx.l == 0) { s.red[2] = 0.f; s.red[3] = 0.f; s.bkey[0] = 0; s.bkey[1] = 0; s.bkey[2] = 0; s.bkey[3] = 0; }
#endif
  cx.sync();
  ant_env_step<NB>(cx, K, s, act_s, obs_s, &out_s[0], (uint8_t*)&iout_s[0], &iout_s[1], &out_s[1], &iout_s[2]);
  cx.sync();
  // Epilogue.  Everything it needs is re-derived from the thread index behind an opaque barrier, so that no per-lane
  // address or index stays live (and gets spilled to scratch) across the 20 forward evaluations above.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int slot2 = tid / G, l2 = tid % G;
  int env2 = xcd_block(blockIdx.x, gridDim.x) * EPB + slot2;
  const bool live2 = env2 < n;
  if (!live2) env2 = n - 1;
  S& s2 = *reinterpret_cast<S*>(lds_raw + (size_t)slot2 * EB);
  const AntIO& io2 = *reinterpret_cast<const AntIO*>(lds_raw + (size_t)slot2 * EB + SB);
  const float* obs2 = io2.obs;
  const float* out2 = io2.out;
  const int* iout2 = io2.iout;
  float* rec2 = state + (size_t)env2 * D::REC;
  const int obs_dim = ANT_OBS + ant_obs_extra<NB>(K);
  const uint8_t d = *(const uint8_t*)&iout2[0];
  const int t_new = iout2[2];
  uint32_t episode = (uint32_t)iout2[3];
  // Auto-reset follows the vector-env convention: an env that finished returns its reward / done of the
  // terminal step, the FIRST observation of the new episode in `obs`, and the terminal observation in `final_obs` (when bound).
  const bool rst = auto_reset && d;
  if (live2) {
    // rows are `ostride` floats apart: obs_dim, or obs_dim + MZ_VIEW_DIM with the time entry behind the view (mz_device.h obs_slot)
    float* orow = ((rst && final_obs) ? final_obs : obs) + (size_t)env2 * ostride;
    if (!rst || final_obs) {
      for (int i = l2; i < obs_dim; i += G) orow[obs_slot(i, obs_dim, ostride)] = obs2[i];
      if (ostride != obs_dim) for (int i = l2; i < 2 * D::NBLK; i += G) orow[obs_dim - 1 + i] = ant_block_coord<NB>(K, s2, i >> 1, i & 1);
    }
    // packed record [obs | reward | done] for the sharded all-gather (mz_bind_record): written here, by the kernel that
    // produced it, instead of by three copy launches in front of the collective
    if (record) {
      float* rrow = record + (size_t)env2 * (obs_dim + 2);
      if (!rst) for (int i = l2; i < obs_dim; i += G) rrow[i] = obs2[i];
      if (l2 == 0) { rrow[obs_dim] = out2[0]; rrow[obs_dim + 1] = (float)d; }
    }
    if (l2 == 0) {
      reward[env2] = out2[0];
      done[env2] = d;
      if (goal_idx) goal_idx[env2] = iout2[1];
      if (s2.status) atomicOr(&status[env2], s2.status);
    }
    if (info) for (int i = l2; i < 4; i += G) info[(size_t)env2 * 4 + i] = out2[1 + i];
  }
  if (rst) {  // masked reset inside the step
    episode += 1;
    uint64_t es = episode_seed(seed, episode);
    // robot coordinates get the reset noise; movable blocks return to their cells (ant.py:84-96)
    // (an object ball returns to its spawn pose: its qpos entries are a pose, not displacements)
    for (int i = l2; i < D::NQ; i += G) s2.qpos[i] = i < ANT_NQ ? reset_qpos(K.qpos0[i], es, env0 + (uint64_t)env2, i) : (D::BALL ? K.qpos0[i < ANT_NQ + 7 ? i : 0] : 0.f);
    for (int i = l2; i < D::NV; i += G) { s2.qvel[i] = i < ANT_NV ? reset_qvel(K.reset_kind, D::NQ, es, env0 + (uint64_t)env2, i) : 0.f; s2.warm[i] = 0.f; }
    cx.sync();
    if (l2 == 0) {  // root quaternion normalised in place, as at mz_reset [ASSUME-8]
      float qn = 1.0f / sqrtf(s2.qpos[3] * s2.qpos[3] + s2.qpos[4] * s2.qpos[4] + s2.qpos[5] * s2.qpos[5] + s2.qpos[6] * s2.qpos[6]);
      for (int i = 3; i < 7; i++) s2.qpos[i] *= qn;
    }
    cx.sync();
    if (live2) {
      float* orow = obs + (size_t)env2 * ostride;
      for (int i = l2; i < obs_dim; i += G) orow[obs_slot(i, obs_dim, ostride)] = ant_obs_elem<NB>(K, s2, i, 0);
      if (ostride != obs_dim) for (int i = l2; i < 2 * D::NBLK; i += G) orow[obs_dim - 1 + i] = ant_block_coord<NB>(K, s2, i >> 1, i & 1);
      if (record) { float* rrow = record + (size_t)env2 * (obs_dim + 2); for (int i = l2; i < obs_dim; i += G) rrow[i] = ant_obs_elem<NB>(K, s2, i, 0); }
    }
  }
  cx.sync();
  if (live2) {
    DevCtx<G, PROF> cx2{l2};
    ant_store<NB>(cx2, s2, rec2);
    if (l2 == 0) { ((int*)rec2)[D::REC_T] = rst ? 0 : t_new; ((uint32_t*)rec2)[D::REC_T + 1] = episode; }
  }
  if constexpr (PROF) {
    cx.tick(s, 10);
    if (live2 && threadIdx.x == 0 && prof) {
      unsigned long long tot = 0;
      for (int k = 0; k < 13; k++) tot += s.prof[k];
      // Round 6: NO atomics on the 16 launch-wide accumulators any more (1024 waves x 16 same-address atomics, issued by the waves
      // that finish first, queued in the L2 while the slowest waves were still running: their memory operations — constant
      // reloads, timer reads — waited behind that queue, and the instrumented build showed a 30-55 k-cycle surplus in whichever
      // timer slot held them: the "rk4 slot of the slowest waves" of VERDICT r05, an artefact of the instrument).  Every
      // workgroup writes its own slots; mz_read_phase_cycles reduces them on the host.
      // per-workgroup totals (mz_read_wave_cycles): cycles in the low 40 bits, Newton iterations of the wave above them
      prof[16 + wg] += tot + ((unsigned long long)s.prof[15] << 40);  // (indexed by env block, not by workgroup id)
      for (int k = 0; k < 16; k++) prof[16 + gridDim.x * 0 + n + (size_t)wg * 16 + k] += s.prof[k];  // per-workgroup phases (mz_read_wave_phase_cycles)
    }
  }
#ifdef MZ_EXP_STAMPS
  if constexpr (!PROF) {
    if (prof && threadIdx.x == 0) {
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* w = prof + 16 + n + (size_t)blockIdx.x * 16;
      w[0] = stamp0; w[1] = __builtin_amdgcn_s_memrealtime(); w[2] = hwid; w[3] = xcc;
      float mx = 0.f;
      int npass = 0, nover = 0, pmax = 0, ncyc = 0, nscan = 0;
      for (int q = 0; q < 64 / G; q++) {
        const S& sq = *reinterpret_cast<const S*>(lds_raw + (size_t)q * EB);
        mx = fmaxf(mx, sq.red[3]);
        npass += sq.bkey[0]; nover += sq.bkey[1]; pmax = sq.bkey[0] > pmax ? sq.bkey[0] : pmax; ncyc += sq.bkey[2]; nscan += sq.bkey[3];
      }
      w[9] = (unsigned long long)mz_exp_general_runs;  // (device-wide running total)
      w[11] = (unsigned long long)nscan;  // ... of which the candidate scan
      w[10] = (unsigned long long)ncyc;  // shader cycles between the cell early-out and the end of the narrow phase, summed over the step
      w[6] = (unsigned long long)npass; w[7] = (unsigned long long)nover; w[8] = (unsigned long long)pmax;
      w[4] = (unsigned long long)s.red[2]; w[5] = (unsigned long long)mx;  // lock-step Newton iterations of the wave; most contact-evaluations among its envs
    }
  }
#endif
}

#ifdef MZ_ISA_ONLY
#ifndef MZ_ISA_PROF
#define MZ_ISA_PROF false
#endif
// developer aid (tools/isa_one.sh): compile ONE instantiation of the step kernel to look at its ISA / run the DPP hazard check
// in seconds instead of building all of them:  hipcc ... -DMZ_ISA_ONLY -DMZ_ISA_NB=0 -DMZ_ISA_G=16 -S ant_kernels.hip
#ifndef MZ_ISA_WPS
#define MZ_ISA_WPS 0
#endif
template __global__ void ant_step_kernel<MZ_ISA_NB, MZ_ISA_G, MZ_ISA_PROF, MZ_ISA_WPS>(const AntDev*, int, float*, const float*, float*, float*, uint8_t*, int*, float*, int*, int,
                                                                     uint64_t, uint64_t, unsigned long long*, float*, int, float*);
#else
template <int NB, int G>
__global__ __launch_bounds__(64) void ant_forward_kernel(AntDev K, int n, const float* __restrict__ state,
                                                          const float* __restrict__ actions, float* __restrict__ qacc,
                                                          int* __restrict__ counts) {
  using D = AntDims<NB>;
  constexpr int EPB = 64 / G;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  AntScratchT<NB>* sc = reinterpret_cast<AntScratchT<NB>*>(lds_raw);
  DevCtx<G> cx{(int)threadIdx.x % G};
  if constexpr (NB <= 1 && G >= 16) ant_lane_consts(K, cx);
  const int slot = threadIdx.x / G;
  int env = blockIdx.x * EPB + slot;
  const bool live = env < n;
  if (!live) env = n - 1;
  AntScratchT<NB>& s = sc[slot];
  ant_load<NB>(cx, s, state + (size_t)env * D::REC);
  for (int i = cx.l; i < D::NV; i += G) s.fact[i] = 0.f;
  ant_fill_tables<NB>(cx, K, s);
  if (cx.l == 0) s.status = 0;
  cx.sync();
  if (actions)
    for (int u = cx.l; u < ANT_NU; u += G) s.fact[K.act_dof[u]] = K.gear * fminf(fmaxf(actions[(size_t)env * ANT_NU + u], K.ctrl_lo), K.ctrl_hi);
  cx.sync();
  ant_forward<NB>(cx, K, s, true);
  if (live) {
    for (int i = cx.l; i < D::NV; i += G) qacc[(size_t)env * D::NV + i] = s.qacc[i];
    if (cx.l == 0 && counts) { counts[2 * env] = (G >= 16 && (NB <= 1 || AntDims<NB>::NBLK > 0)) ? s.ncon_true : s.ncon; /* paths that merge a block's contact points report MuJoCo's count */ counts[2 * env + 1] = s.iters; }
  }
}

__global__ void ant_reset_kernel(AntDev K, AntLayout L, int n, float* state, const uint8_t* mask, uint64_t seed, uint64_t env0, float* obs) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  float* rec = state + (size_t)env * L.rec;
  if (!mask || mask[env]) {
    for (int i = 0; i < L.nq; i++) rec[i] = i < ANT_NQ ? reset_qpos(K.qpos0[i], seed, env0 + (uint64_t)env, i) : (K.nball ? K.qpos0[i < ANT_NQ + 7 ? i : 0] : 0.f);
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
    float* o = obs + (size_t)env * L.ostride;
    int k = 0;
    auto block_coord = [&](int b, int c) {
      float v = K.block_pos0[b][c];
      for (int a = 0; a < K.block_nax; a++) v += K.block_axis[a] == c ? rec[15 + K.block_nax * b + a] : 0.f;
      return v;
    };
    for (int i = 0; i < 3; i++) o[k++] = rec[i];
    if (K.nball && K.observe_balls) for (int c = 0; c < 3; c++) o[k++] = rec[ANT_NQ + c];
    for (int b = 0; b < L.nblock3 / 3 && !K.nball; b++)
      for (int c = 0; c < 3; c++) o[k++] = block_coord(b, c);
    for (int i = 3; i < ANT_NQ; i++) o[k++] = rec[i];
    for (int i = 0; i < ANT_NV; i++) o[k++] = rec[L.nq + i];
    if (L.ostride != L.obs_dim)  // top-down view: block x, y parked for mzk_view_fill, time entry behind the view
      for (int b = 0; b < K.nblock; b++) { o[k + 2 * b] = block_coord(b, 0); o[k + 2 * b + 1] = block_coord(b, 1); }
    o[L.ostride - 1] = (float)((int*)rec)[L.rec_t] * 0.001f;
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

// the two-waves-per-SIMD instantiation (WPS = 2) exists for the 16-lane plain ant; taken when the launch has more waves than the
// device has SIMDs (4 per compute unit), or as option "waves_per_simd" says (1 / 2; 0 = by the wave count)
static int device_simds(const mz_handle* h) { return h->simds; }  // 4 per compute unit, read once in mz_create
template <int NB, int G>
static bool ant_two_waves(const mz_handle* h, int waves) {
  if (!(NB == 0 && G == 16)) return false;
  if (h->waves_per_simd) return h->waves_per_simd == 2;
  return waves > device_simds(h);
}
template <int NB, int G>
static hipError_t launch_ant_step(mz_handle* h, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, int* gi, float* inf) {
  int wpb = h->waves_per_block;
  int epb = wpb * 64 / G;
  const size_t per_env = h->prof ? ant_env_lds_bytes<NB, G, true>() : ant_env_lds_bytes<NB, G, false>();
  size_t lds = (size_t)epb * per_env;
  while (lds > 160 * 1024 && wpb > 1) { wpb /= 2; epb = wpb * 64 / G; lds = (size_t)epb * per_env; }
  const dim3 grid((h->n + epb - 1) / epb), block(64 * wpb);
  // the dynamic-LDS attribute is per kernel function: set once per (instantiation, size), not on every step
  static size_t lds_set[2][32] = {};  // [instrumented build][device]
  const int dv = h->device & 31;
  float* rec = h->view.on ? nullptr : h->record;  // with a top-down view the record is packed after mzk_view_fill (mazestep.hip)
  hipError_t e;
#ifdef MZ_EXP_STAMPS
  if (h->prof && !(NB == 0 && G == 16)) return hipErrorNotSupported;
  if (h->prof) {  // the non-instrumented kernel, stamping into the profile buffer
    if (lds_set[0][dv] != lds) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ant_step_kernel<NB, G, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      lds_set[0][dv] = lds;
    }
    hipLaunchKernelGGL((ant_step_kernel<NB, G, false>), grid, block, lds, st, h->ant_dev, h->n, h->state, a, o, r, d, gi, inf, h->status,
                       h->auto_reset, h->seed, h->env0, h->prof, h->final_obs, h->lay.ostride, rec);
    return hipSuccess;
  }
#endif
  if (h->prof) {
    if (lds_set[1][dv] != lds) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ant_step_kernel<NB, G, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      lds_set[1][dv] = lds;
    }
    hipLaunchKernelGGL((ant_step_kernel<NB, G, true>), grid, block, lds, st, h->ant_dev, h->n, h->state, a, o, r, d, gi, inf, h->status,
                       h->auto_reset, h->seed, h->env0, h->prof, h->final_obs, h->lay.ostride, rec);
  } else if (ant_two_waves<NB, G>(h, (int)grid.x * wpb)) {
    if constexpr (NB == 0 && G == 16) {
      static size_t lds_set2[32] = {};
      if (lds_set2[dv] != lds) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ant_step_kernel<NB, G, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set2[dv] = lds;
      }
      hipLaunchKernelGGL((ant_step_kernel<NB, G, false, 2>), grid, block, lds, st, h->ant_dev, h->n, h->state, a, o, r, d, gi, inf, h->status,
                         h->auto_reset, h->seed, h->env0, (unsigned long long*)nullptr, h->final_obs, h->lay.ostride, rec);
    }
  } else {
    if (lds_set[0][dv] != lds) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ant_step_kernel<NB, G, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      lds_set[0][dv] = lds;
    }
    hipLaunchKernelGGL((ant_step_kernel<NB, G, false>), grid, block, lds, st, h->ant_dev, h->n, h->state, a, o, r, d, gi, inf, h->status,
                       h->auto_reset, h->seed, h->env0, (unsigned long long*)nullptr, h->final_obs, h->lay.ostride, rec);
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
// Default: 16 for the plain ant — one DPP row per env, four envs per wavefront, one wavefront per SIMD at the 4096-env
// batch of the metric: since the row solver removed most LDS waits the kernel is bound by vector-instruction count, and
// at 16 lanes every phase with <= 16 items (dofs, bodies, geoms, the whole Newton iteration) costs half the instructions
// per env that it costs at 32 (measured: 8.9 vs 8.1 M env-steps/s; round 1, wait-bound, had it the other way round);
// 64 with blocks (their contact sets keep 64 lanes busy, and the 2048-env batches of those configs then fill the chip with
// two waves per SIMD instead of one).
template <int NB>
static int ant_lanes(const mz_handle* h);
template <int NB>
static hipError_t dispatch_ant_step(mz_handle* h, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, int* gi, float* inf) {
  if constexpr (NB == 0) {
    if (h->lanes_set && h->lanes == 8) return launch_ant_step<NB, 8>(h, st, a, o, r, d, gi, inf);
  }
  const int lanes = ant_lanes<NB>(h);
  if (lanes == 64) return launch_ant_step<NB, 64>(h, st, a, o, r, d, gi, inf);
  if (lanes == 16) return launch_ant_step<NB, 16>(h, st, a, o, r, d, gi, inf);
  return launch_ant_step<NB, 32>(h, st, a, o, r, d, gi, inf);
}
// lanes per env of the handle: ONE rule for the step and for mz_debug_forward, so that the diagnostic evaluates the very
// instantiation that steps (8 lanes exist for the plain ant's step only; the forward diagnostic then uses 16)
template <int NB>
static int ant_lanes(const mz_handle* h) {
  // one movable block: the row solver (16 dofs in one DPP row, the block's own contacts spread over the group) at 32 lanes per env
  // — 2048 envs then put exactly one wave on every SIMD; more blocks / the ball: the lane-group solver at 64
  // (the plain ant: 16 at every batch size since the solver's DPP operands were fused, round 3 — 8192 / 16384 / 32768 envs run
  // 12.8 / 13.4 / 13.8 M env-steps/s at 16 lanes against 11.6 / 12.4 / 13.1 M at 32)
  // (round 4, tried: 16 lanes for the one-block ant from 4096 envs on — no gain at 4096 (four envs per wave need 47 KB of LDS: three
  // waves per CU, two rounds), 6.2 against 5.0 M at 8192; and one env in 4096 then sits 2.5e-5 off the oracle: not taken)
  // Round 5: 16 lanes for the one-block ant as soon as 32 lanes would mean more waves than SIMDs (beyond 2048 envs on a 256-CU
  // device): with 32 contact slots four 16-lane waves fit a CU's LDS (AntDims::NC), and the iterative refinement of the accepted
  // Newton step (ant_newton_rows.h) closed the parity gap noted above.  AntPush 4096 / 8192 envs: 5.4 / 5.8 -> 8.6 / 9.4 M
  // env-steps/s, AntFall 4.5 / 4.8 -> 7.3 / 8.0 M (profiles/r05/push_nc.txt); 2048 envs (BASELINE config 5) stay at 32 lanes.
  if (h->lanes_set) return h->lanes;
  if (NB == 1) return h->n / 2 > device_simds(h) ? 16 : 32;
  return NB ? 64 : 16;
}
template <int NB>
static hipError_t dispatch_ant_forward(mz_handle* h, hipStream_t st, const float* a, float* qacc, int* counts) {
  const int lanes = ant_lanes<NB>(h);
  if (lanes == 64) return launch_ant_forward<NB, 64>(h, st, a, qacc, counts);
  if (lanes == 32) return launch_ant_forward<NB, 32>(h, st, a, qacc, counts);
  return launch_ant_forward<NB, 16>(h, st, a, qacc, counts);
}

// MazeTask.reward / termination on rows of observations (parity tests: the instance of task_eval_dev that is inlined into
// the Ant step kernel of THIS translation unit, i.e. compiled with its relaxed floating-point flags)
__global__ void ant_task_eval_kernel(const AntDev* __restrict__ Kp, int n, int nenv, int obs_dim, const float* __restrict__ obs,
                                     float* __restrict__ reward, uint8_t* __restrict__ done, int* __restrict__ goal_idx) {
  int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float o6[6];
  for (int k = 0; k < 6; k++) o6[k] = obs[(size_t)row * obs_dim + k];
  float r; int tm, gi;
  task_eval_dev(Kp->task, o6, &r, &tm, &gi, row < nenv ? row : -1);  // per-env goals (mz_bind_env_goals): row r is env r
  reward[row] = r;
  done[row] = (uint8_t)(tm ? 1 : 0);
  if (goal_idx) goal_idx[row] = gi;
}

// ------------------------------------------------------------------ entry points of this translation unit (mz_internal.h)
// AntDims configuration of the handle's model: 0-3 two-slide blocks, 4 one three-slide block, 5 one free-joint object ball
static int ant_config(const mz_handle* h) { return h->ant.nball ? 5 : ((h->ant.nblock == 1 && h->ant.block_nax == 3) ? 4 : h->ant.nblock); }
static hipError_t ant_sync_constants(mz_handle* h, hipStream_t st) {
  if (!h->ant_dirty) return hipSuccess;
  hipError_t e = hipMemcpyAsync(h->ant_dev, &h->ant, sizeof(AntDev), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) h->ant_dirty = 0;
  return e;
}

hipError_t mzk_ant_step(mz_handle* h, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, int* gi, float* inf) {
  hipError_t e = ant_sync_constants(h, st);
  if (e != hipSuccess) return e;
#if defined(MZ_DEV_NB0)  // developer build (make dev: libmazestep_dev.so, a minute instead of six): the plain ant's instantiations only
  if (ant_config(h) != 0) return hipErrorNotSupported;
  return dispatch_ant_step<0>(h, st, a, o, r, d, gi, inf);
#elif defined(MZ_DEV_NB01)  // (make dev1: plus the ant with one two-slide block)
  if (ant_config(h) > 1) return hipErrorNotSupported;
  return ant_config(h) ? dispatch_ant_step<1>(h, st, a, o, r, d, gi, inf) : dispatch_ant_step<0>(h, st, a, o, r, d, gi, inf);
#else
  switch (ant_config(h)) {  // configuration of movable bodies (AntDims)
    case 0: return dispatch_ant_step<0>(h, st, a, o, r, d, gi, inf);
    case 1: return dispatch_ant_step<1>(h, st, a, o, r, d, gi, inf);
    case 2: return dispatch_ant_step<2>(h, st, a, o, r, d, gi, inf);
    case 4: return dispatch_ant_step<4>(h, st, a, o, r, d, gi, inf);
    case 5: return dispatch_ant_step<5>(h, st, a, o, r, d, gi, inf);
    default: return dispatch_ant_step<3>(h, st, a, o, r, d, gi, inf);
  }
#endif
}

hipError_t mzk_ant_forward(mz_handle* h, hipStream_t st, const float* a, float* qacc, int* counts) {
#if defined(MZ_DEV_NB0)
  if (ant_config(h) != 0) return hipErrorNotSupported;
  return dispatch_ant_forward<0>(h, st, a, qacc, counts);
#elif defined(MZ_DEV_NB01)
  if (ant_config(h) > 1) return hipErrorNotSupported;
  return ant_config(h) ? dispatch_ant_forward<1>(h, st, a, qacc, counts) : dispatch_ant_forward<0>(h, st, a, qacc, counts);
#else
  switch (ant_config(h)) {
    case 0: return dispatch_ant_forward<0>(h, st, a, qacc, counts);
    case 1: return dispatch_ant_forward<1>(h, st, a, qacc, counts);
    case 2: return dispatch_ant_forward<2>(h, st, a, qacc, counts);
    case 4: return dispatch_ant_forward<4>(h, st, a, qacc, counts);
    case 5: return dispatch_ant_forward<5>(h, st, a, qacc, counts);
    default: return dispatch_ant_forward<3>(h, st, a, qacc, counts);
  }
#endif
}

void mzk_ant_shape(const mz_handle* h, int* lanes, int* waves_per_simd) {
  int l = 16;
  switch (ant_config(h)) {
    case 0: l = (h->lanes_set && h->lanes == 8) ? 8 : ant_lanes<0>(h); break;
    case 1: l = ant_lanes<1>(h); break;
    default: l = ant_lanes<2>(h); break;  // (one rule for every configuration with more movable bodies)
  }
  *lanes = l;
  const int wpb = h->waves_per_block, epb = wpb * 64 / 16;  // launch_ant_step's grid: workgroups of wpb waves, four 16-lane envs per wave
  *waves_per_simd = (ant_config(h) == 0 && l == 16 && ant_two_waves<0, 16>(h, (h->n + epb - 1) / epb * wpb)) ? 2 : 1;
}

hipError_t mzk_ant_reset(mz_handle* h, hipStream_t st, const uint8_t* mask, uint64_t seed, float* obs) {
  hipLaunchKernelGGL(ant_reset_kernel, dim3((h->n + 255) / 256), dim3(256), 0, st, h->ant, h->lay, h->n, h->state, mask, seed, h->env0, obs);
  return hipGetLastError();
}

hipError_t mzk_ant_set_state(mz_handle* h, hipStream_t st, const float* qpos, const float* qvel, const float* warm, const int* t) {
  int tot = h->n * h->lay.rec;
  hipLaunchKernelGGL(ant_set_state_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, h->lay, h->n, h->state, qpos, qvel, warm, t);
  return hipGetLastError();
}

hipError_t mzk_ant_get_state(mz_handle* h, hipStream_t st, float* qpos, float* qvel, float* warm, int* t) {
  int tot = h->n * h->lay.rec;
  hipLaunchKernelGGL(ant_get_state_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, h->lay, h->n, h->state, qpos, qvel, warm, t);
  return hipGetLastError();
}

hipError_t mzk_ant_task_eval(mz_handle* h, hipStream_t st, int n, const float* obs, float* reward, uint8_t* done, int* goal_idx) {
  hipError_t e = ant_sync_constants(h, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ant_task_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h->ant_dev, n, h->n, h->lay.ostride, obs, reward, done, goal_idx);
  return hipGetLastError();
}
#endif  // MZ_ISA_ONLY
