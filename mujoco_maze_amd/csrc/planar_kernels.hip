// planar_kernels.hip — HIP kernels (gfx950 / CDNA4) of the Point, Swimmer and Reacher paths and their launchers.
//
//   planar_step_kernel<NB,NS,G>  one MazeEnv.step for the Point (+ NB movable blocks or NS object balls): lane group
//                                per env, PlanarScratch in LDS, fp64.
//   swimmer_step_kernel<NL,NB,G> one MazeEnv.step for the Swimmer (NL = 3) / Reacher (NL = 2): G = 4 lanes per env (lane b = link b), fp64.
//   reset / state copy kernels; debug kernels for the parity tests (task predicates, the Point's wall detector).
//
// HBM layout: SoA  q_0..q_{NV-1} | v_0..v_{NV-1}, each [N] fp32; t[N], episode[N] i32.  API arrays are row-major [N, k].
//
// This translation unit is built without -fapprox-func / -fno-signed-zeros (csrc/Makefile); the dynamics may use
// reciprocal division, but the Point's manual wall detector (point_dyn.h: point_detect / point_bounce) and the task
// predicates reproduce the reference's float64 decisions bit for bit — inside those functions contraction, reciprocal
// division and reassociation are switched off (#pragma clang fp ...), and sqrt is correctly rounded.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ant_dyn.h"
#include "point_dyn.h"
#include "planar_dyn.h"
#include "swimmer_dyn.h"
#include "mz_device.h"
#include "mz_internal.h"

// ------------------------------------------------------------------ state in HBM
// Chains (Swimmer / Reacher: 64 envs per workgroup): SoA qv[2 NV][n] + t[n] + episode[n] — a wave's loads of one coordinate are one line.
// Point (`rec` > 0; round 4): env-major records qv[n][rec] = q[NV] | v[NV] | t | episode (| pad to a multiple of 4 floats).  A Point
// workgroup is ONE wave = 2 envs (32 lanes each): in SoA it touched 8 bytes of eight different 128-byte lines per array pass, and
// every such partial store left the L2 as a transaction of its own (PMC, round 4: WRITE_SIZE 1.9 MB per launch against 0.3 MB of
// algorithmic writes, FETCH_SIZE 1.1 MB against 0.2); its two records are 64 contiguous bytes.
struct PointState { float* qv; int* t; uint32_t* ep; int rec; };
// The bare Point's record (no movable bodies) carries the solver's warm start behind t | episode: qacc - qacc_smooth of the step's last
// forward evaluation, the first guess of the next step's first solve (round 6; reset and set_state clear it)
constexpr int PT_WARM_OFF = 8;
__device__ __forceinline__ float& st_q(const PointState& S, int n, int nv, int k, int env) { return S.rec ? S.qv[(size_t)env * S.rec + k] : S.qv[(size_t)k * n + env]; }
__device__ __forceinline__ float& st_v(const PointState& S, int n, int nv, int k, int env) { return S.rec ? S.qv[(size_t)env * S.rec + nv + k] : S.qv[(size_t)(nv + k) * n + env]; }
__device__ __forceinline__ int& st_t(const PointState& S, int nv, int env) { return S.rec ? reinterpret_cast<int*>(S.qv)[(size_t)env * S.rec + 2 * nv] : S.t[env]; }
__device__ __forceinline__ uint32_t& st_ep(const PointState& S, int nv, int env) { return S.rec ? reinterpret_cast<uint32_t*>(S.qv)[(size_t)env * S.rec + 2 * nv + 1] : S.ep[env]; }

// One MazeEnv.step of the Point (+ NB movable blocks or NS object balls): G lanes per env, PlanarScratch in LDS, SoA state in HBM
// (q_0..q_{NV-1} | v_0..v_{NV-1}, each [n]).
// The bare Point at 32 lanes per env runs two waves per SIMD (4096 envs = 2048 waves on 1024 SIMDs): its register budget is pinned
// to 256 so that a few registers more do not silently halve the occupancy and send half the waves into a second round.
template <int NB, int NS, int G>
__device__ __forceinline__ void planar_step_body(const PointDev& P, PlanarScratch<NB, NS>& s, float* o, const DevCtx<G>& cx, int n, const PointState& S,
                                                 int env, bool live, const float* __restrict__ actions, float* __restrict__ obs,
                                                 float* __restrict__ reward, uint8_t* __restrict__ done, int* __restrict__ goal_idx,
                                                 float* __restrict__ info, int* __restrict__ status, int auto_reset, uint64_t seed, uint64_t env0,
                                                 float* __restrict__ final_obs, int ostride) {
  using D = PlanarDims<NB, NS>;
  constexpr int NV = D::NV, NOBS = D::NOBS;
  for (int k = cx.l; k < NV; k += G) { s.q[k] = (double)st_q(S, n, NV, k, env); s.v[k] = (double)st_v(S, n, NV, k, env); }
  if constexpr (NB == 0 && NS == 0) { for (int k = cx.l; k < 3; k += G) s.wds[k] = (double)S.qv[(size_t)env * S.rec + PT_WARM_OFF + k]; }  // the bare Point's warm start
  double a[2] = {(double)actions[(size_t)env * 2], (double)actions[(size_t)env * 2 + 1]};
  cx.sync();
#ifdef MZ_EXP_PROF
  if constexpr (NB == 0 && NS == 0) { if (cx.l == 0) { for (int k = 0; k < 12; k++) s.prof[k] = 0; s.prof_t0 = __builtin_amdgcn_s_memtime(); } }
#endif
  planar_env_step<NB, NS>(cx, P, s, a);
#ifdef MZ_EXP_PROF
  if constexpr (NB == 0 && NS == 0) {
    if (blockIdx.x % 97 == 5 && threadIdx.x == 0)
      printf("PROF %d setup %llu enum %llu prefix %llu fill %llu newton %llu pre %llu rk4 %llu detect %llu fwdtail %llu\n", (int)blockIdx.x, s.prof[0], s.prof[1], s.prof[2],
             s.prof[3], s.prof[4], s.prof[5], s.prof[6], s.prof[7], s.prof[8]);
  }
#endif
  const int t_new = st_t(S, NV, env) + 1;  // (read here, behind the step: one register less carried through it)
  for (int i = cx.l; i < NOBS; i += G) o[i] = planar_obs_elem<NB, NS>(P, s, i, t_new);
  cx.sync();
  float outer; int tm, gi;
  task_eval_dev(P.task, o, &outer, &tm, &gi, env);  // flags from the fp32 observation that is returned
  const uint8_t d = (uint8_t)((tm ? 1 : 0) | (t_new >= P.task.max_steps ? 2 : 0));
  const bool rst = auto_reset && d;  // vector-env convention: obs <- first observation of the new episode, terminal one -> final_obs
  if (live) {
    // rows are `ostride` floats apart: NOBS, or NOBS + MZ_VIEW_DIM with the time entry behind the view (mz_device.h obs_slot)
    float* orow = ((rst && final_obs) ? final_obs : obs) + (size_t)env * ostride;
    if (!rst || final_obs) {
      for (int i = cx.l; i < NOBS; i += G) orow[obs_slot(i, NOBS, ostride)] = o[i];
      if (ostride != NOBS) for (int i = cx.l; i < 2 * NB; i += G) orow[NOBS - 1 + i] = planar_block_coord<NB, NS>(P, s, i >> 1, i & 1);
    }
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
  }
  uint32_t ep = st_ep(S, NV, env);
  if (rst) {  // point.py:71-81: noise on the robot, blocks / ball back to their spawn state
    ep += 1;
    const uint64_t es = episode_seed(seed, ep);
    cx.sync();
    for (int k = cx.l; k < NV; k += G) {
      s.q[k] = k < 3 ? (double)reset_qpos((float)P.qpos0[k], es, env0 + (uint64_t)env, k) : 0.0;
      s.v[k] = k < 3 ? (double)reset_qvel(P.reset_kind, NV, es, env0 + (uint64_t)env, k) : 0.0;
    }
    cx.sync();
    if (live) {
      float* orow = obs + (size_t)env * ostride;
      for (int i = cx.l; i < NOBS; i += G) orow[obs_slot(i, NOBS, ostride)] = planar_obs_elem<NB, NS>(P, s, i, 0);
      if (ostride != NOBS) for (int i = cx.l; i < 2 * NB; i += G) orow[NOBS - 1 + i] = planar_block_coord<NB, NS>(P, s, i >> 1, i & 1);
    }
  }
  if (live) {
    for (int k = cx.l; k < NV; k += G) {
      st_q(S, n, NV, k, env) = (float)s.q[k];
      st_v(S, n, NV, k, env) = (float)s.v[k];
    }
    if (cx.l == 0) { st_t(S, NV, env) = rst ? 0 : t_new; st_ep(S, NV, env) = ep; }
    if constexpr (NB == 0 && NS == 0) { for (int k = cx.l; k < 3; k += G) S.qv[(size_t)env * S.rec + PT_WARM_OFF + k] = rst ? 0.f : (float)s.wds[k]; }
  }
}

template <int NB, int NS, int G>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((NB == 0 && NS == 0 && G == 32) ? 2 : 1))) void planar_step_kernel(const PointDev* __restrict__ Pp, int n, PointState S,
                                                          const float* __restrict__ actions, float* __restrict__ obs,
                                                          float* __restrict__ reward, uint8_t* __restrict__ done,
                                                          int* __restrict__ goal_idx, float* __restrict__ info,
                                                          int* __restrict__ status, int auto_reset, uint64_t seed, uint64_t env0,
                                                          float* __restrict__ final_obs, int ostride) {
  constexpr int EPW = 64 / G;
  __shared__ PointDev P;  // segment table + task shared by the block (L2-resident source)
  __shared__ PlanarScratch<NB, NS> scr[EPW];
  __shared__ float obuf[EPW][MZ_MAX_OBS];
  for (int i = threadIdx.x; i < (int)(sizeof(PointDev) / 4); i += blockDim.x) ((uint32_t*)&P)[i] = ((const uint32_t*)Pp)[i];
  __syncthreads();
  DevCtx<G> cx{(int)threadIdx.x % G};
  const int grp = threadIdx.x / G;
  int env = xcd_block(blockIdx.x, gridDim.x) * EPW + grp;
  const bool live = env < n;
  if (!live) env = n - 1;  // idle groups shadow the last env (no stores) so that every lane reaches the wave-level votes
  planar_step_body<NB, NS, G>(P, scr[grp], obuf[grp], cx, n, S, env, live, actions, obs, reward, done, goal_idx, info, status, auto_reset, seed, env0, final_obs,
                              ostride);
}

template <int NB, int NS>
__global__ void point_reset_kernel(const PointDev* Pp, int n, PointState S, const uint8_t* mask, uint64_t seed, uint64_t env0, float* obs,
                                   int ostride) {
  constexpr int NV = 3 + 2 * NB + 3 * NS, NOBS = 7 + 3 * NB + 3 * NS;
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  if (!mask || mask[env]) {
    for (int k = 0; k < NV; k++) {
      st_q(S, n, NV, k, env) = k < 3 ? reset_qpos((float)Pp->qpos0[k], seed, env0 + (uint64_t)env, k) : 0.f;
      st_v(S, n, NV, k, env) = k < 3 ? reset_qvel(Pp->reset_kind, NV, seed, env0 + (uint64_t)env, k) : 0.f;
    }
    st_t(S, NV, env) = 0;
    st_ep(S, NV, env) = 0;
    if (NB == 0 && NS == 0) for (int k = 0; k < 3; k++) S.qv[(size_t)env * S.rec + PT_WARM_OFF + k] = 0.f;
  }
  if (obs) {
    const int nb3 = (Pp->observe_blocks ? 3 * NB : 0) + (Pp->observe_balls ? 3 * NS : 0);
    float* o = obs + (size_t)env * ostride;
    for (int k = 0; k < 3; k++) { o[k] = st_q(S, n, NV, k, env); o[3 + nb3 + k] = st_v(S, n, NV, k, env); }
    if (NS > 0 && nb3) {
      o[3] = (float)Pp->ball_pos0[0] + st_q(S, n, NV, 3, env); o[4] = (float)Pp->ball_pos0[1] + st_q(S, n, NV, 4, env);
      o[5] = (float)Pp->ball_pos0[2];
    }
    for (int b = 0; b < NB && nb3; b++) {
      float p3[3] = {(float)Pp->block_pos0[b][0], (float)Pp->block_pos0[b][1], (float)Pp->block_pos0[b][2]};
      p3[Pp->block_axis[0]] += st_q(S, n, NV, 3 + 2 * b, env);
      p3[Pp->block_axis[1]] += st_q(S, n, NV, 4 + 2 * b, env);
      o[3 + 3 * b] = p3[0]; o[4 + 3 * b] = p3[1]; o[5 + 3 * b] = p3[2];
    }
    if (ostride != NOBS)  // top-down view: block x, y parked for mzk_view_fill, time entry behind the view
      for (int b = 0; b < NB; b++) {
        float p3[3] = {(float)Pp->block_pos0[b][0], (float)Pp->block_pos0[b][1], (float)Pp->block_pos0[b][2]};
        p3[Pp->block_axis[0]] += st_q(S, n, NV, 3 + 2 * b, env);
        p3[Pp->block_axis[1]] += st_q(S, n, NV, 4 + 2 * b, env);
        o[NOBS - 1 + 2 * b] = p3[0]; o[NOBS + 2 * b] = p3[1];
      }
    o[ostride - 1] = (float)st_t(S, NV, env) * 0.001f;
  }
}

template <int KQ>
__global__ void point_set_state_kernel(int n, PointState S, const float* qpos, const float* qvel, const int* t) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  for (int k = 0; k < KQ; k++) {
    if (qpos) st_q(S, n, KQ, k, env) = qpos[(size_t)env * KQ + k];
    if (qvel) st_v(S, n, KQ, k, env) = qvel[(size_t)env * KQ + k];
  }
  if (t) st_t(S, KQ, env) = t[env];
  if (KQ == 3 && S.rec > PT_WARM_OFF && (qpos || qvel)) for (int k = 0; k < 3; k++) S.qv[(size_t)env * S.rec + PT_WARM_OFF + k] = 0.f;  // a new state: no guess
}
template <int KQ>
__global__ void point_get_state_kernel(int n, PointState S, float* qpos, float* qvel, float* warm, int* t) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  for (int k = 0; k < KQ; k++) {
    if (qpos) qpos[(size_t)env * KQ + k] = st_q(S, n, KQ, k, env);
    if (qvel) qvel[(size_t)env * KQ + k] = st_v(S, n, KQ, k, env);
    if (warm) warm[(size_t)env * KQ + k] = (KQ == 3 && S.rec > PT_WARM_OFF) ? S.qv[(size_t)env * S.rec + PT_WARM_OFF + k] : 0.f;
  }
  if (t) t[env] = st_t(S, KQ, env);
}

// ------------------------------------------------------------------ Swimmer / Reacher kernels (NL links, one movable block with BD
// slide dofs: 0 none, 2 or 3; SoA: q0..q[NV-1] v0..v[NV-1] | t | episode with NV = NL + 2 + BD)
//
// Observation / reset layout: swimmer_dyn.h (swimmer_obs_row).  Reset (swimmer.py:56-69): U(-0.1, 0.1) noise on ALL nq
// coordinates and ALL nv velocities, the block's included.
// one observation row o[NO] (swimmer_obs_row) into a row of `ostride` floats: NO, or NO + MZ_VIEW_DIM with the time entry behind
// the view and the movable block's x, y parked for mzk_view_fill (mz_device.h obs_slot)
template <int NL, int BD>
__device__ __forceinline__ void swimmer_store_row(const SwimmerDev& P, const float* qf, const float* o, int NO, int ostride, float* row) {
  for (int k = 0; k < NO; k++) row[obs_slot(k, NO, ostride)] = o[k];
  if constexpr (BD > 0) {
    if (ostride != NO) {
      double p[3] = {P.block_pos0[0][0], P.block_pos0[0][1], P.block_pos0[0][2]};
      for (int a = 0; a < BD; a++) p[P.bd_axis[a]] += (double)qf[NL + 2 + a];
      row[NO - 1] = (float)p[0];
      row[NO] = (float)p[1];
    }
  }
}

// lane-group context of the chain kernels (swimmer_dyn.h): G = 4 lanes per env for chains of up to four links, 8 beyond
template <int G>
struct SwimmerCtx {
  static constexpr int nlanes = G;
  int l;
  __device__ __forceinline__ int lane0() const { return l; }
  __device__ __forceinline__ double gsum(double x) const { return DevCtx<G>{l}.gsum(x); }
  // value of lane k of this lane's group (ds_bpermute on the two halves of the double; a handful per forward evaluation)
  __device__ __forceinline__ double from_lane(double x, int k) const { return __shfl(x, (int)(threadIdx.x & 63u) - l + k, 64); }
#ifdef MZ_EXP_SWPROF
  mutable unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
  __device__ __forceinline__ void stamp(int k) const { __builtin_amdgcn_sched_barrier(0); unsigned long long t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); prof[k] += t - t0; t0 = t; }
#else
  __device__ __forceinline__ void stamp(int) const {}
#endif
};

template <int NL, int NB, int G>
__global__ __launch_bounds__(256) void swimmer_step_kernel(const SwimmerDev* __restrict__ Pp, int n, PointState S,
                                                            const float* __restrict__ actions, float* __restrict__ obs,
                                                            float* __restrict__ reward, uint8_t* __restrict__ done,
                                                            int* __restrict__ goal_idx, float* __restrict__ info,
                                                            int* __restrict__ status, int auto_reset, uint64_t seed, uint64_t env0,
                                                            float* __restrict__ final_obs, int ostride) {
  // G adjacent lanes advance one env (lane b owns link b of the chain); everything outside the forward dynamics' per-link part
  // is computed redundantly by the G lanes, the group's first lane stores
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int env = gid / G;
  const SwimmerCtx<G> cx{(int)(threadIdx.x % G)};
#ifdef MZ_EXP_SWPROF
  cx.t0 = __builtin_amdgcn_s_memtime();
#endif
  const bool live = env < n && cx.l == 0;
  if (env >= n) env = n - 1;  // surplus groups shadow the last env (no stores): every lane reaches the group sums
  constexpr int NR = NL + 2, NV = NR + NB, NH = NL - 1;  // NB: slide dofs of the movable block
  const SwimmerDev& P = *Pp;
  const int nb3 = (NB && P.observe_blocks) ? 3 : 0, NO = 2 * NV + 1 + nb3;
  float qf[NV], vf[NV], af[NH > 0 ? NH : 1], o[2 * NV + 4];
  double inner, inf4[4];
  for (int k = 0; k < NV; k++) { qf[k] = S.qv[(size_t)k * n + env]; vf[k] = S.qv[(size_t)(NV + k) * n + env]; }
  for (int k = 0; k < NH; k++) af[k] = actions[(size_t)env * NH + k];
  int t_new;
  int st = swimmer_maze_step<NL, NB>(P, qf, vf, af, S.t[env], o, &inner, inf4, &t_new, cx);
#ifdef MZ_EXP_SWPROF
  cx.stamp(6);
  if (blockIdx.x == 3 && threadIdx.x == 0)
    printf("PROF %d sincos %llu links %llu gsum %llu solve %llu limits %llu rk4 %llu rest %llu\n", (int)blockIdx.x, cx.prof[0], cx.prof[1], cx.prof[2], cx.prof[3], cx.prof[4],
           cx.prof[5], cx.prof[6]);
#endif
  if (!live) return;
  float outer; int tm, gi;
  task_eval_dev(P.task, o, &outer, &tm, &gi, env);
  uint8_t d = (uint8_t)((tm ? 1 : 0) | (t_new >= P.task.max_steps ? 2 : 0));
  const bool rst = auto_reset && d;
  if (!rst || final_obs) swimmer_store_row<NL, NB>(P, qf, o, NO, ostride, ((rst && final_obs) ? final_obs : obs) + (size_t)env * ostride);
  reward[env] = (float)(P.task.inner_scale * inner) + outer;
  done[env] = d;
  if (goal_idx) goal_idx[env] = gi;
  if (info) for (int k = 0; k < 4; k++) info[(size_t)env * 4 + k] = (float)inf4[k];
  if (st) atomicOr(&status[env], st);
  uint32_t ep = S.ep[env];
  if (rst) {
    ep += 1; t_new = 0;
    const uint64_t es = episode_seed(seed, ep);
    for (int k = 0; k < NV; k++) {
      qf[k] = reset_qpos(k < NR ? (float)P.qpos0[k] : 0.f, es, env0 + (uint64_t)env, k);
      vf[k] = reset_qvel(P.reset_kind, NV, es, env0 + (uint64_t)env, k);
    }
    swimmer_obs_row<NL, NB>(P, qf, vf, 0, o);
    swimmer_store_row<NL, NB>(P, qf, o, NO, ostride, obs + (size_t)env * ostride);
  }
  for (int k = 0; k < NV; k++) {
    S.qv[(size_t)k * n + env] = qf[k];
    S.qv[(size_t)(NV + k) * n + env] = vf[k];
  }
  S.t[env] = t_new;
  S.ep[env] = ep;
}

template <int NL, int NB>
__global__ void swimmer_reset_kernel(const SwimmerDev* Pp, int n, PointState S, const uint8_t* mask, uint64_t seed, uint64_t env0, float* obs,
                                     int ostride) {
  constexpr int NR = NL + 2, NV = NR + NB;
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  if (!mask || mask[env]) {
    for (int k = 0; k < NV; k++) {
      S.qv[(size_t)k * n + env] = reset_qpos(k < NR ? (float)Pp->qpos0[k] : 0.f, seed, env0 + (uint64_t)env, k);
      S.qv[(size_t)(NV + k) * n + env] = reset_qvel(Pp->reset_kind, NV, seed, env0 + (uint64_t)env, k);
    }
    S.t[env] = 0;
    S.ep[env] = 0;
  }
  if (obs) {
    const int nb3 = (NB && Pp->observe_blocks) ? 3 : 0, NO = 2 * NV + 1 + nb3;
    float qf[NV], vf[NV], o[2 * NV + 4];
    for (int k = 0; k < NV; k++) { qf[k] = S.qv[(size_t)k * n + env]; vf[k] = S.qv[(size_t)(NV + k) * n + env]; }
    swimmer_obs_row<NL, NB>(*Pp, qf, vf, S.t[env], o);
    swimmer_store_row<NL, NB>(*Pp, qf, o, NO, ostride, obs + (size_t)env * ostride);
  }
}

// ------------------------------------------------------------------ top-down view (every robot; mz_view.h)
// One thread per env: fills the MZ_VIEW_DIM view entries of the env's observation row — and of its final_obs row when the env
// just finished under auto-reset — from the row's own torso position and the block positions the step / reset kernel parked.
__global__ void view_fill_kernel(ViewDev V, int n, int ostride, int view_off, float* __restrict__ obs, float* __restrict__ final_obs,
                                 const uint8_t* __restrict__ done) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  mzv_fill_row(V, obs + (size_t)env * ostride, view_off);
  if (final_obs && done && done[env]) mzv_fill_row(V, final_obs + (size_t)env * ostride, view_off);
}

#ifdef MZ_ISA_POINT  // developer aid (tools/isa_point.sh): the bare Point's kernel only, for a look at its ISA / register budget
template __global__ void planar_step_kernel<0, 0, 32>(const PointDev*, int, PointState, const float*, float*, float*, uint8_t*, int*, float*, int*, int, uint64_t,
                                                        uint64_t, float*, int);
#else
hipError_t mzk_view_fill(mz_handle* h, hipStream_t st, float* obs, float* final_obs, const uint8_t* done) {
  if (!h->view.on) return hipSuccess;
  hipLaunchKernelGGL(view_fill_kernel, dim3((h->n + 63) / 64), dim3(64), 0, st, h->view, h->n, h->model.obs_dim, h->base_obs - 1, obs, final_obs, done);
  return hipGetLastError();
}

// ------------------------------------------------------------------ parity-test kernels
// MazeTask.reward / termination on rows of observations: the task_eval_dev instance of this translation unit
__global__ void planar_task_eval_kernel(const TaskDev* __restrict__ Tp, int n, int nenv, int obs_dim, const float* __restrict__ obs,
                                        float* __restrict__ reward, uint8_t* __restrict__ done, int* __restrict__ goal_idx) {
  int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float o6[6];
  for (int k = 0; k < 6; k++) o6[k] = obs[(size_t)row * obs_dim + k];
  float r; int tm, gi;
  task_eval_dev(*Tp, o6, &r, &tm, &gi, row < nenv ? row : -1);  // per-env goals (mz_bind_env_goals): row r is env r
  reward[row] = r;
  done[row] = (uint8_t)(tm ? 1 : 0);
  if (goal_idx) goal_idx[row] = gi;
}

// CollisionDetector.detect + the bounce / give-up rule of MazeEnv.step on rows of float64 (old_xy, new_xy) moves: the
// point_detect / point_bounce instances the Point step kernel runs.  hit: 0 none, 1 bounced, 2 gave up, -1 collinear.
__global__ void point_detect_kernel(const PointDev* __restrict__ Pp, int n, const double* __restrict__ old_xy,
                                    const double* __restrict__ new_xy, int* __restrict__ hit, double* __restrict__ point,
                                    double* __restrict__ final_xy) {
  __shared__ PointDev P;
  for (int i = threadIdx.x; i < (int)(sizeof(PointDev) / 4); i += blockDim.x) ((uint32_t*)&P)[i] = ((const uint32_t*)Pp)[i];
  __syncthreads();
  int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double o[2] = {old_xy[2 * row], old_xy[2 * row + 1]}, nw[2] = {new_xy[2 * row], new_xy[2 * row + 1]}, fin[2], pt[2];
  int r = point_bounce(P, o, nw, fin, pt);
  hit[row] = r;
  if (point) { point[2 * row] = pt[0]; point[2 * row + 1] = pt[1]; }
  final_xy[2 * row] = fin[0]; final_xy[2 * row + 1] = fin[1];
}

// ------------------------------------------------------------------ entry points of this translation unit (mz_internal.h)
// floats per env of the Point's env-major state record (q | v | t | episode, padded to 16 bytes); 0 = SoA (the chains)
int mzk_planar_record_width(const mz_handle* h) {
  if (h->robot == MZ_ROBOT_POINT && h->point.nblock == 0 && h->point.nball == 0) return PT_WARM_OFF + 4;  // q[3] v[3] t episode | warm start[3] | pad: 48 B
  return h->robot == MZ_ROBOT_POINT ? (2 * mzk_planar_state_width(h) + 2 + 3) / 4 * 4 : 0;
}
int mzk_planar_state_width(const mz_handle* h) {
  return h->robot == MZ_ROBOT_SWIMMER ? h->swimmer.nlink + 2 + (h->swimmer.nblock ? h->swimmer.nbdof : 0) : 3 + 2 * h->point.nblock + 3 * h->point.nball;
}

// lanes per env of the next step launch (one rule for the launch and for mz_get_info)
int mzk_planar_lanes(const mz_handle* h) {
  if (h->robot == MZ_ROBOT_SWIMMER) return h->swimmer.nlink <= 4 ? 4 : 8;
  if (h->point.nball || h->point.nblock == 1) return 32;
  if (h->point.nblock >= 2) return 64;
  // the bare Point: 16 lanes per env while that still gives every wave a SIMD of its own (up to 4096 envs on a 256-CU device), else 32
  const int lanes = h->lanes_set ? h->lanes : ((h->n + 3) / 4 <= h->simds ? 16 : 32);
  return (lanes == 8 || lanes == 16) ? lanes : 32;
}

hipError_t mzk_planar_step(mz_handle* h, hipStream_t st, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
                           int* goal_idx_dev, float* info_dev) {
  PointState S{h->state, h->pt_t, h->pt_ep, h->pt_rec};
  if (h->robot == MZ_ROBOT_SWIMMER) {
#define MZ_SW_STEP(NL, NB)                                                                                                          \
  hipLaunchKernelGGL((swimmer_step_kernel<NL, NB, (NL <= 4 ? 4 : 8)>), dim3((unsigned)(((size_t)h->n * (NL <= 4 ? 4 : 8) + sw_bd - 1) / sw_bd)), dim3(sw_bd), 0, st, \
                     h->swimmer_dev, h->n, S, actions_dev, obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev, h->status, h->auto_reset, h->seed,    \
                     h->env0, h->final_obs, h->model.obs_dim)
    const int bd = h->swimmer.nblock ? h->swimmer.nbdof : 0;
    const unsigned sw_bd = 64u * (unsigned)(h->wpb_set ? h->waves_per_block : 4);  // workgroup size (option "waves_per_block")
    if (h->swimmer.nlink == 3) { if (bd == 3) MZ_SW_STEP(3, 3); else if (bd == 2) MZ_SW_STEP(3, 2); else MZ_SW_STEP(3, 0); }
    else if (h->swimmer.nlink == 2) { if (bd == 3) MZ_SW_STEP(2, 3); else if (bd == 2) MZ_SW_STEP(2, 2); else MZ_SW_STEP(2, 0); }
    else if (h->swimmer.nlink == 4) MZ_SW_STEP(4, 0);  // longer chains (user MJCF): no movable blocks
    else if (h->swimmer.nlink == 5) MZ_SW_STEP(5, 0);
    else MZ_SW_STEP(6, 0);
#undef MZ_SW_STEP
    return hipGetLastError();
  }
  // lanes per env: 32 for the bare robot — its 18 collision enumerators then share one round, and 4096 envs make two waves per
  // SIMD: a wave runs as long as the slowest of its envs (only envs at a wall enumerate, fill and iterate), so two envs per wave
  // and a second wave to fill the gaps beat four envs per wave (measured, round 3: 34.8 vs 31.6 M env-steps/s on PointUMaze) —
  // 32 / 64 with blocks (bigger contact sets in LDS)
#define MZ_PLANAR_LAUNCH(NB, NS, G)                                                                                                    \
  hipLaunchKernelGGL((planar_step_kernel<NB, NS, G>), dim3((h->n + 64 / G - 1) / (64 / G)), dim3(64), 0, st, h->point_dev, h->n, S, actions_dev, \
                     obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev, h->status, h->auto_reset, h->seed, h->env0, h->final_obs, h->model.obs_dim)
  if (h->point.nball) MZ_PLANAR_LAUNCH(0, 1, 32);
  else switch (h->point.nblock) {
    case 0: {
      // Round 5: 16 lanes per env by default while that still gives every wave a SIMD of its own (up to 4096 envs on a 256-CU device).
      // Round 3 measured the opposite (34.8 against 31.6 M at 32 lanes: two envs per wave and a second wave to fill the gaps); with
      // the step in registers (point_bare.h) and the unit-step solver the waves are short and even enough that four envs per wave on
      // a SIMD of their own win: PointUMaze 4096 envs 74.3 -> 79.6 M env-steps/s, Point4Rooms 79.6 -> 82.6 M; beyond (8192 envs:
      // 108.0 at 32 lanes against 106.3) it stays at 32 (profiles/r05/point_knobs.txt).
      const int lanes = mzk_planar_lanes(h);
      if (lanes == 8) MZ_PLANAR_LAUNCH(0, 0, 8);
      else if (lanes == 16) MZ_PLANAR_LAUNCH(0, 0, 16);
      else MZ_PLANAR_LAUNCH(0, 0, 32);
      break;
    }
    case 1: MZ_PLANAR_LAUNCH(1, 0, 32); break;
    case 2: MZ_PLANAR_LAUNCH(2, 0, 64); break;
    default: MZ_PLANAR_LAUNCH(3, 0, 64); break;
  }
#undef MZ_PLANAR_LAUNCH
  return hipGetLastError();
}

hipError_t mzk_planar_reset(mz_handle* h, hipStream_t st, const uint8_t* mask_dev, uint64_t seed, float* obs_dev) {
  PointState S{h->state, h->pt_t, h->pt_ep, h->pt_rec};
  const int nb = (h->n + 255) / 256;
  if (h->robot == MZ_ROBOT_SWIMMER) {
#define MZ_SW_RESET(NL, NB) hipLaunchKernelGGL((swimmer_reset_kernel<NL, NB>), dim3(nb), dim3(256), 0, st, h->swimmer_dev, h->n, S, mask_dev, seed, h->env0, obs_dev, h->model.obs_dim)
    const int bd = h->swimmer.nblock ? h->swimmer.nbdof : 0;
    if (h->swimmer.nlink == 3) { if (bd == 3) MZ_SW_RESET(3, 3); else if (bd == 2) MZ_SW_RESET(3, 2); else MZ_SW_RESET(3, 0); }
    else if (h->swimmer.nlink == 2) { if (bd == 3) MZ_SW_RESET(2, 3); else if (bd == 2) MZ_SW_RESET(2, 2); else MZ_SW_RESET(2, 0); }
    else if (h->swimmer.nlink == 4) MZ_SW_RESET(4, 0);
    else if (h->swimmer.nlink == 5) MZ_SW_RESET(5, 0);
    else MZ_SW_RESET(6, 0);
#undef MZ_SW_RESET
    return hipGetLastError();
  }
  if (h->point.nball) hipLaunchKernelGGL((point_reset_kernel<0, 1>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev, h->model.obs_dim);
  else switch (h->point.nblock) {
    case 0: hipLaunchKernelGGL((point_reset_kernel<0, 0>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev, h->model.obs_dim); break;
    case 1: hipLaunchKernelGGL((point_reset_kernel<1, 0>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev, h->model.obs_dim); break;
    case 2: hipLaunchKernelGGL((point_reset_kernel<2, 0>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev, h->model.obs_dim); break;
    default: hipLaunchKernelGGL((point_reset_kernel<3, 0>), dim3(nb), dim3(256), 0, st, h->point_dev, h->n, S, mask_dev, seed, h->env0, obs_dev, h->model.obs_dim); break;
  }
  return hipGetLastError();
}

hipError_t mzk_planar_set_state(mz_handle* h, hipStream_t st, const float* qpos_dev, const float* qvel_dev, const int* t_dev) {
  PointState S{h->state, h->pt_t, h->pt_ep, h->pt_rec};
  const dim3 grid((h->n + 255) / 256), blk(256);
  switch (mzk_planar_state_width(h)) {
    case 3: hipLaunchKernelGGL(point_set_state_kernel<3>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
    case 4: hipLaunchKernelGGL(point_set_state_kernel<4>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
    case 5: hipLaunchKernelGGL(point_set_state_kernel<5>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
    case 6: hipLaunchKernelGGL(point_set_state_kernel<6>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
    case 7: hipLaunchKernelGGL(point_set_state_kernel<7>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
    case 8: hipLaunchKernelGGL(point_set_state_kernel<8>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
    default: hipLaunchKernelGGL(point_set_state_kernel<9>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, t_dev); break;
  }
  return hipGetLastError();
}

hipError_t mzk_planar_get_state(mz_handle* h, hipStream_t st, float* qpos_dev, float* qvel_dev, float* warmstart_dev, int* t_dev) {
  PointState S{h->state, h->pt_t, h->pt_ep, h->pt_rec};
  const dim3 grid((h->n + 255) / 256), blk(256);
  switch (mzk_planar_state_width(h)) {
    case 3: hipLaunchKernelGGL(point_get_state_kernel<3>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
    case 4: hipLaunchKernelGGL(point_get_state_kernel<4>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
    case 5: hipLaunchKernelGGL(point_get_state_kernel<5>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
    case 6: hipLaunchKernelGGL(point_get_state_kernel<6>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
    case 7: hipLaunchKernelGGL(point_get_state_kernel<7>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
    case 8: hipLaunchKernelGGL(point_get_state_kernel<8>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
    default: hipLaunchKernelGGL(point_get_state_kernel<9>, grid, blk, 0, st, h->n, S, qpos_dev, qvel_dev, warmstart_dev, t_dev); break;
  }
  return hipGetLastError();
}

hipError_t mzk_planar_task_eval(mz_handle* h, hipStream_t st, int n, const float* obs, float* reward, uint8_t* done, int* goal_idx) {
  const TaskDev* Tp = h->robot == MZ_ROBOT_SWIMMER ? &h->swimmer_dev->task : &h->point_dev->task;
  hipLaunchKernelGGL(planar_task_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, st, Tp, n, h->n, h->model.obs_dim, obs, reward, done, goal_idx);
  return hipGetLastError();
}

hipError_t mzk_point_detect(mz_handle* h, hipStream_t st, int n, const double* old_xy, const double* new_xy, int* hit, double* point,
                            double* final_xy) {
  hipLaunchKernelGGL(point_detect_kernel, dim3((n + 63) / 64), dim3(64), 0, st, h->point_dev, n, old_xy, new_xy, hit, point, final_xy);
  return hipGetLastError();
}
#endif  // MZ_ISA_POINT
