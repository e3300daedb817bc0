// generic_kernels.hip — HIP kernels (gfx950 / CDNA4) of the GENERAL ENGINE (csrc/generic_dyn.h): any compiled mz_model none of the
// specialised kernels steps — a user's AgentModel of any tree topology, mazes with SPIN plates (a box on a ball joint), more than
// three movable blocks — and any model at all when the caller asks for it (mz_model.engine = 1: the cross-check of the
// specialised kernels).
//
//   generic_step_kernel   one MazeEnv.step per env: ONE WAVEFRONT (64 lanes) per environment, the env's working set
//                         (GenScratch, 77 KB, float64) in LDS for the frame_skip x 4 forward evaluations of the step — two
//                         envs share a CU's 160 KB; the compiled mz_model itself is the constant block (global memory).
//                         The tree passes run level by level, the factorisations row-parallel (generic_dyn.h).
//   reset / state copy kernels.
//
// HBM layout: state[N][REC] fp32 record = qpos[nq] | qvel[nv] | qacc_warmstart[nv] | t | episode (REC = nq + 2 nv + 2).
// Built with the strict floating-point flags of csrc/Makefile (no fast-math): the path computes in float64 and is compared
// with the float64 oracle at 1e-6; the Point's manual wall bounce (point_dyn.h) is bit-exact under these flags.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "generic_dyn.h"
#include "mz_device.h"
#include "mz_internal.h"

struct GenIO { float act[MZ_MAX_ACT], obs[MZ_MAX_OBS + MZ_VIEW_DIM], out[8]; int iout[4]; };
// LDS of one env: I/O staging | scratch block, the latter cut behind the Jacobian rows its model needs (gen_scratch_bytes).  Two envs
// (workgroups of one wavefront) share a CU while that stays within half of the 160 KB: models of up to 22 dofs.
constexpr size_t GEN_IO_BYTES = (sizeof(GenIO) + 15) / 16 * 16;
static size_t gen_env_lds_bytes(int nv) { return GEN_IO_BYTES + gen_scratch_bytes(nv); }

__global__ __launch_bounds__(64) void generic_step_kernel(const GenDev* __restrict__ Kp, int n, float* __restrict__ state, const float* __restrict__ actions,
                                                           float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
                                                           int* __restrict__ goal_idx, float* __restrict__ info, int* __restrict__ status, int auto_reset,
                                                           uint64_t seed, uint64_t env0, float* __restrict__ final_obs, float* __restrict__ record) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  struct { GenIO& io; } L{*reinterpret_cast<GenIO*>(lds_raw)};
  const GenDev& K = *Kp;
  const mz_model& m = K.m;
  GenScratch& s = *reinterpret_cast<GenScratch*>(lds_raw + GEN_IO_BYTES);
  DevCtx<64> cx{(int)threadIdx.x};
  const int env = blockIdx.x, nq = m.nq, nv = m.nv, nu = m.nu, rec_t = nq + 2 * nv, REC = rec_t + 2, obs_dim = m.obs_dim;
  float* rec = state + (size_t)env * REC;
  for (int i = cx.l; i < rec_t; i += 64) {
    const double v = (double)rec[i];
    if (i < nq) s.qpos[i] = v; else if (i < nq + nv) s.qvel[i - nq] = v; else s.warm[i - nq - nv] = v;
  }
  for (int i = cx.l; i < nu; i += 64) L.io.act[i] = actions[(size_t)env * nu + i];
  if (cx.l == 0) { L.io.iout[2] = ((const int*)rec)[rec_t]; L.io.iout[3] = ((const int*)rec)[rec_t + 1]; }
#ifdef MZ_EXP_GENPROF
  if (cx.l == 0) { for (int k = 0; k < 20; k++) s.prof[k] = 0; s.prof_t0 = __builtin_amdgcn_s_memtime(); }
#endif
  cx.sync();
  gen_env_step(cx, K, s, L.io.act, L.io.obs, &L.io.out[0], (uint8_t*)&L.io.iout[0], &L.io.iout[1], &L.io.out[1], &L.io.iout[2], env);
  cx.sync();
#ifdef MZ_EXP_GENPROF
  if (cx.l == 0 && blockIdx.x % 97 == 5) {
    unsigned long long tot = 0;
    for (int k = 0; k < 16; k++) tot += s.prof[k];
    printf("GENPROF %d total %llu kin %llu items %llu collide %llu crb+rne %llu mass/compact/limits %llu force/rows %llu qas %llu | warmcost %llu Mx/cu %llu grad %llu H %llu chol %llu ls %llu cost %llu tail %llu | rk4/io %llu iters %d ncon %d\n",
           (int)blockIdx.x, tot, s.prof[0], s.prof[1], s.prof[2], s.prof[3], s.prof[4], s.prof[5], s.prof[6], s.prof[7], s.prof[8], s.prof[9], s.prof[10], s.prof[11], s.prof[12],
           s.prof[13], s.prof[14], s.prof[15], s.iters, s.ncon);
  }
#endif
  const uint8_t d = *(const uint8_t*)&L.io.iout[0];
  const int t_new = L.io.iout[2];
  uint32_t episode = (uint32_t)L.io.iout[3];
  const bool rst = auto_reset && d;  // vector-env convention: obs <- first observation of the new episode, terminal one -> final_obs
  float* orow = ((rst && final_obs) ? final_obs : obs) + (size_t)env * obs_dim;
  if (!rst || final_obs) for (int i = cx.l; i < obs_dim; i += 64) orow[i] = L.io.obs[i];
  float* rrow = record ? record + (size_t)env * (obs_dim + 2) : nullptr;
  if (rrow && !rst) for (int i = cx.l; i < obs_dim; i += 64) rrow[i] = L.io.obs[i];
  if (cx.l == 0) {
    reward[env] = L.io.out[0];
    done[env] = d;
    if (goal_idx) goal_idx[env] = L.io.iout[1];
    if (s.status) atomicOr(&status[env], s.status);
    if (rrow) { rrow[obs_dim] = L.io.out[0]; rrow[obs_dim + 1] = (float)d; }
  }
  if (info) for (int i = cx.l; i < 4; i += 64) info[(size_t)env * 4 + i] = L.io.out[1 + i];
  if (rst) {  // masked reset inside the step: noise on the robot's coordinates (ant.py:84-96 / swimmer.py:56-69 pattern)
    episode += 1;
    const uint64_t es = episode_seed(seed, episode);
    for (int i = cx.l; i < nq; i += 64) s.qpos[i] = i < m.nq_robot ? (double)reset_qpos((float)m.qpos0[i], es, env0 + (uint64_t)env, i) : m.qpos0[i];
    for (int i = cx.l; i < nv; i += 64) { s.qvel[i] = i < m.nv_robot ? (double)reset_qvel(m.reset_qvel_kind, nq, es, env0 + (uint64_t)env, i) : 0.0; s.warm[i] = 0.0; }
    cx.sync();
    if (cx.l == 0 && m.njnt > 0 && m.jnt_type[0] == MZ_JNT_FREE) gd_quat_norm(s.qpos + m.jnt_qposadr[0] + 3);  // [ASSUME-8]
    cx.sync();
    for (int i = cx.l; i < obs_dim - (m.top_down_view ? MZ_VIEW_DIM : 0); i += 64) gen_store_obs(K, s.qpos, s.qvel, 0, L.io.obs, i);
    cx.sync();
    for (int i = cx.l; i < obs_dim; i += 64) { const float v = L.io.obs[i]; obs[(size_t)env * obs_dim + i] = v; if (rrow) rrow[i] = v; }
  }
  cx.sync();
  for (int i = cx.l; i < rec_t; i += 64) rec[i] = (float)(i < nq ? s.qpos[i] : (i < nq + nv ? s.qvel[i - nq] : s.warm[i - nq - nv]));
  if (cx.l == 0) { ((int*)rec)[rec_t] = rst ? 0 : t_new; ((uint32_t*)rec)[rec_t + 1] = episode; }
}

__global__ void generic_reset_kernel(const GenDev* __restrict__ Kp, int n, float* state, const uint8_t* mask, uint64_t seed, uint64_t env0, float* obs) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  const mz_model& m = Kp->m;
  const int nq = m.nq, nv = m.nv, rec_t = nq + 2 * nv, REC = rec_t + 2;
  float* rec = state + (size_t)env * REC;
  if (!mask || mask[env]) {
    for (int i = 0; i < nq; i++) rec[i] = i < m.nq_robot ? reset_qpos((float)m.qpos0[i], seed, env0 + (uint64_t)env, i) : (float)m.qpos0[i];
    if (m.njnt > 0 && m.jnt_type[0] == MZ_JNT_FREE) {  // set_state -> mj_forward normalises the root quaternion in place [ASSUME-8]
      float* q = rec + m.jnt_qposadr[0] + 3;
      const float qn = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int k = 0; k < 4; k++) q[k] *= qn;
    }
    for (int i = 0; i < nv; i++) { rec[nq + i] = i < m.nv_robot ? reset_qvel(m.reset_qvel_kind, nq, seed, env0 + (uint64_t)env, i) : 0.f; rec[nq + nv + i] = 0.f; }
    ((int*)rec)[rec_t] = 0;
    ((uint32_t*)rec)[rec_t + 1] = 0;
  }
  if (obs) {
    float* o = obs + (size_t)env * m.obs_dim;
    for (int i = 0; i < m.obs_dim - (m.top_down_view ? MZ_VIEW_DIM : 0); i++) gen_store_obs(*Kp, rec, rec + nq, ((int*)rec)[rec_t], o, i);
  }
}

__global__ void generic_set_state_kernel(int nq, int nv, int n, float* state, const float* qpos, const float* qvel, const float* warm, const int* t) {
  const int REC = nq + 2 * nv + 2, idx = blockIdx.x * blockDim.x + threadIdx.x, env = idx / REC, i = idx % REC;
  if (env >= n) return;
  float* rec = state + (size_t)env * REC;
  if (i < nq) { if (qpos) rec[i] = qpos[(size_t)env * nq + i]; }
  else if (i < nq + nv) { if (qvel) rec[i] = qvel[(size_t)env * nv + i - nq]; }
  else if (i < nq + 2 * nv) { if (warm) rec[i] = warm[(size_t)env * nv + i - nq - nv]; }
  else if (i == nq + 2 * nv) { if (t) ((int*)rec)[i] = t[env]; }
}
__global__ void generic_get_state_kernel(int nq, int nv, int n, const float* state, float* qpos, float* qvel, float* warm, int* t) {
  const int REC = nq + 2 * nv + 2, idx = blockIdx.x * blockDim.x + threadIdx.x, env = idx / REC, i = idx % REC;
  if (env >= n) return;
  const float* rec = state + (size_t)env * REC;
  if (i < nq) { if (qpos) qpos[(size_t)env * nq + i] = rec[i]; }
  else if (i < nq + nv) { if (qvel) qvel[(size_t)env * nv + i - nq] = rec[i]; }
  else if (i < nq + 2 * nv) { if (warm) warm[(size_t)env * nv + i - nq - nv] = rec[i]; }
  else if (i == nq + 2 * nv) { if (t) t[env] = ((const int*)rec)[i]; }
}

__global__ void generic_task_eval_kernel(const GenDev* __restrict__ Kp, int n, int nenv, int obs_dim, const float* __restrict__ obs, float* __restrict__ reward,
                                         uint8_t* __restrict__ done, int* __restrict__ goal_idx) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  float o6[6];
  for (int k = 0; k < 6; k++) o6[k] = k < obs_dim ? obs[(size_t)row * obs_dim + k] : 0.f;
  float r; int tm, gi;
  task_eval_dev(Kp->task, o6, &r, &tm, &gi, row < nenv ? row : -1);  // per-env goals (mz_bind_env_goals): row r is env r
  reward[row] = r; done[row] = (uint8_t)(tm ? 1 : 0);
  if (goal_idx) goal_idx[row] = gi;
}

// ------------------------------------------------------------------ entry points of this translation unit (mz_internal.h)
int mzk_generic_needed(const mz_model* m) { return gen_model_needs_general_engine(m); }
int mzk_generic_create(mz_handle* h, char* err, int errlen) {
  GenDev* g = (GenDev*)malloc(sizeof(GenDev));
  if (!g) return MZ_ERR_HIP;
  int rc = gen_dev_from_model(g, &h->model, err, errlen);
  if (rc == MZ_OK) {
    if (hipMalloc(&h->gen_dev, sizeof(GenDev)) != hipSuccess || hipMemcpy(h->gen_dev, g, sizeof(GenDev), hipMemcpyHostToDevice) != hipSuccess)
      rc = gen_fail(err, errlen, "general engine: device allocation failed") == MZ_ERR_UNSUPPORTED ? MZ_ERR_HIP : MZ_ERR_HIP;
  }
  free(g);
  return rc;
}
hipError_t mzk_generic_set_task(mz_handle* h, const TaskDev* task) {
  return hipMemcpy(reinterpret_cast<char*>(h->gen_dev) + offsetof(GenDev, task), task, sizeof(TaskDev), hipMemcpyHostToDevice);
}
void mzk_generic_destroy(mz_handle* h) { if (h->gen_dev) { (void)hipFree(h->gen_dev); h->gen_dev = nullptr; } }
hipError_t mzk_generic_step(mz_handle* h, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, int* gi, float* inf) {
  static int lds_set[32] = {};
  const int lds = (int)gen_env_lds_bytes(h->model.nv), lds_max = (int)gen_env_lds_bytes(GN_NV), dv = h->device & 31;
  if (lds_set[dv] != lds_max) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e != hipSuccess) return e;
    lds_set[dv] = lds_max;
  }
  hipLaunchKernelGGL(generic_step_kernel, dim3(h->n), dim3(64), lds, st, h->gen_dev, h->n, h->state, a, o, r, d, gi, inf, h->status, h->auto_reset, h->seed,
                     h->env0, h->final_obs, h->record);
  return hipGetLastError();
}
hipError_t mzk_generic_reset(mz_handle* h, hipStream_t st, const uint8_t* mask, uint64_t seed, float* obs) {
  hipLaunchKernelGGL(generic_reset_kernel, dim3((h->n + 255) / 256), dim3(256), 0, st, h->gen_dev, h->n, h->state, mask, seed, h->env0, obs);
  return hipGetLastError();
}
hipError_t mzk_generic_set_state(mz_handle* h, hipStream_t st, const float* qpos, const float* qvel, const float* warm, const int* t) {
  const int tot = h->n * (h->model.nq + 2 * h->model.nv + 2);
  hipLaunchKernelGGL(generic_set_state_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, h->model.nq, h->model.nv, h->n, h->state, qpos, qvel, warm, t);
  return hipGetLastError();
}
hipError_t mzk_generic_get_state(mz_handle* h, hipStream_t st, float* qpos, float* qvel, float* warm, int* t) {
  const int tot = h->n * (h->model.nq + 2 * h->model.nv + 2);
  hipLaunchKernelGGL(generic_get_state_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, h->model.nq, h->model.nv, h->n, h->state, qpos, qvel, warm, t);
  return hipGetLastError();
}
hipError_t mzk_generic_task_eval(mz_handle* h, hipStream_t st, int n, const float* obs, float* reward, uint8_t* done, int* goal_idx) {
  hipLaunchKernelGGL(generic_task_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h->gen_dev, n, h->n, h->model.obs_dim, obs, reward, done, goal_idx);
  return hipGetLastError();
}
