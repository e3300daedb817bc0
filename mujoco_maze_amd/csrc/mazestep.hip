// mazestep.hip — the C-ABI of include/mazestep.h: handle life cycle, options, argument checks and the dispatch to the
// kernels of ant_kernels.hip (Ant) and planar_kernels.hip (Point / Swimmer / Reacher).  See mz_internal.h for how the
// library is split into translation units and why they are built with different floating-point flags.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "mz_internal.h"

__global__ void fetch_clear_status_kernel(int n, int* status, int* out) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  out[env] = status[env];
  status[env] = 0;
}

// packed record [obs (obs_dim) | reward | done] of one step: the generic path (Point / Swimmer / Reacher, and any robot with a
// top-down view, whose rows are completed by mzk_view_fill after the step kernel).  The Ant step kernel writes the record
// itself (ant_kernels.hip epilogue).
__global__ void pack_record_kernel(int n, int obs_dim, const float* __restrict__ obs, const float* __restrict__ reward,
                                   const uint8_t* __restrict__ done, float* __restrict__ record) {
  const int w = obs_dim + 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n * w) return;
  const int env = (int)(idx / w), i = (int)(idx - (size_t)env * w);
  record[idx] = i < obs_dim ? obs[(size_t)env * obs_dim + i] : (i == obs_dim ? reward[env] : (float)done[env]);
}

static int set_err(mz_handle* h, int code, const char* what, hipError_t e) {
  if (h) snprintf(h->err, sizeof(h->err), "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
  return code;
}
#define HIPCHK(h, call)                                                      \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) return set_err((h), MZ_ERR_HIP, #call, _e);        \
  } while (0)

// Every entry point runs with the handle's device current and restores the caller's device on return, so that a
// handle on GPU 1 in a process whose current device is GPU 0 (or a C caller passing stream = NULL) allocates, sets
// kernel attributes and launches on the right device — and torch's notion of the current device is left alone.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

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
  DeviceScope scope(device);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device) return fail("mz_create: hipSetDevice failed");
  mz_handle* h = new (std::nothrow) mz_handle();
  if (!h) return fail("mz_create: out of memory");
  memset(h, 0, sizeof(*h));
  h->model = *model; h->n = num_envs; h->device = device; h->robot = model->robot; h->lanes = 32; h->waves_per_block = 1; h->seed = 0x5EEDULL;
  {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cu <= 0) cu = 256;
    h->simds = 4 * cu;
  }
  // the general engine (generic_dyn.h) steps what no specialised kernel does — user robots, SPIN plates, more than three blocks —
  // and whatever the caller asks it to (mz_model.engine = 1); h->robot is the dispatch key, h->model.robot stays the family
  if (mzk_generic_needed(model)) h->robot = MZ_ROBOT_GENERIC;
  char msg[200] = {0};
  int rc = MZ_OK;
  if (h->robot == MZ_ROBOT_GENERIC) rc = MZ_OK;  // checked by mzk_generic_create below (it needs the device)
  else if (model->robot == MZ_ROBOT_ANT) rc = ant_dev_from_model(&h->ant, model, msg, sizeof(msg));
  else if (model->robot == MZ_ROBOT_POINT) rc = point_dev_from_model(&h->point, model, msg, sizeof(msg));
  else if (model->robot == MZ_ROBOT_SWIMMER) rc = swimmer_dev_from_model(&h->swimmer, model, msg, sizeof(msg));
  else { rc = MZ_ERR_UNSUPPORTED; snprintf(msg, sizeof(msg), "mz_create: robot kind %d has no device kernel yet", model->robot); }
  if (rc != MZ_OK) { delete h; return fail(msg); }
  hipError_t e = hipSuccess;
  view_dev_from_model(model, &h->view);
  const int vdim = model->top_down_view ? MZ_VIEW_DIM : 0;
  if (model->top_down_view && model->nblock > 4) { delete h; return fail("mz_create: top-down view with more than 4 movable blocks"); }
  if (h->robot == MZ_ROBOT_GENERIC) {
    rc = mzk_generic_create(h, msg, sizeof(msg));
    if (rc != MZ_OK) { mz_destroy(h); return fail(msg); }
    h->base_obs = model->obs_dim - vdim;
    const size_t rec = (size_t)model->nq + 2 * model->nv + 2;
    e = hipMalloc(&h->state, (size_t)num_envs * rec * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->state, 0, (size_t)num_envs * rec * sizeof(float));
  } else if (h->robot == MZ_ROBOT_ANT) {
    const int nb = h->ant.nblock;
    if (nb > 3) { delete h; return fail("mz_create: more than 3 movable blocks are not instantiated"); }
    h->lay.nq = ANT_NQ + h->ant.block_nax * nb + 7 * h->ant.nball; h->lay.nv = ANT_NV + h->ant.block_nax * nb + 6 * h->ant.nball;
    h->lay.rec_t = h->lay.nq + 2 * h->lay.nv;
    h->lay.rec = (h->lay.rec_t + 2 + 15) / 16 * 16;
    h->lay.nblock3 = model->observe_blocks ? 3 * nb : 0;
    h->lay.obs_dim = ANT_OBS + h->lay.nblock3 + ((h->ant.nball && model->observe_balls) ? 3 : 0);
    h->base_obs = h->lay.obs_dim;
    h->lay.ostride = h->lay.obs_dim + vdim;
    if (h->base_obs + vdim != model->obs_dim || h->base_obs > MZ_MAX_OBS) { delete h; return fail("mz_create: obs_dim mismatch"); }
    e = hipMalloc(&h->ant_dev, sizeof(AntDev));
    h->ant_dirty = 1;
    if (e == hipSuccess) e = hipMalloc(&h->state, (size_t)num_envs * h->lay.rec * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->state, 0, (size_t)num_envs * h->lay.rec * sizeof(float));
  } else {
    const int kq = mzk_planar_state_width(h);
    const int nb3 = h->robot == MZ_ROBOT_SWIMMER ? ((h->swimmer.observe_blocks && h->swimmer.nblock) ? 3 : 0)
                                                 : (h->point.observe_blocks ? 3 * h->point.nblock : 0) + (h->point.observe_balls ? 3 * h->point.nball : 0);
    const int want = h->robot == MZ_ROBOT_SWIMMER ? 2 * kq + 1 + nb3 : 7 + nb3;
    h->base_obs = want;
    if (want + vdim != model->obs_dim || want > MZ_MAX_OBS) { delete h; return fail("mz_create: obs_dim mismatch"); }
    h->pt_rec = mzk_planar_record_width(h);
    const size_t state_floats = (size_t)num_envs * (h->pt_rec ? h->pt_rec : 2 * kq);
    e = hipMalloc(&h->state, state_floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&h->pt_t, (size_t)num_envs * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->pt_ep, (size_t)num_envs * sizeof(uint32_t));
    if (h->robot == MZ_ROBOT_SWIMMER) {
      if (e == hipSuccess) e = hipMalloc(&h->swimmer_dev, sizeof(SwimmerDev));
      if (e == hipSuccess) e = hipMemcpy(h->swimmer_dev, &h->swimmer, sizeof(SwimmerDev), hipMemcpyHostToDevice);
    } else {
      if (e == hipSuccess) e = hipMalloc(&h->point_dev, sizeof(PointDev));
      if (e == hipSuccess) e = hipMemcpy(h->point_dev, &h->point, sizeof(PointDev), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemset(h->state, 0, state_floats * sizeof(float));
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
  DeviceScope scope(h->device);
  if (h->state) (void)hipFree(h->state);
  if (h->pt_t) (void)hipFree(h->pt_t);
  if (h->pt_ep) (void)hipFree(h->pt_ep);
  if (h->point_dev) (void)hipFree(h->point_dev);
  if (h->swimmer_dev) (void)hipFree(h->swimmer_dev);
  if (h->ant_dev) (void)hipFree(h->ant_dev);
  mzk_generic_destroy(h);
  if (h->status) (void)hipFree(h->status);
  if (h->prof) (void)hipFree(h->prof);
  if (h->ev) { for (int i = 0; i < 2 * h->ntime; i++) (void)hipEventDestroy(h->ev[i]); free(h->ev); }
  delete h;
}

const char* mz_last_error(const mz_handle* h) { return h ? h->err : "null handle"; }
int32_t mz_num_envs(const mz_handle* h) { return h ? h->n : MZ_ERR_ARG; }
int32_t mz_obs_dim(const mz_handle* h) { return h ? h->model.obs_dim : MZ_ERR_ARG; }
int32_t mz_nq(const mz_handle* h) { return h ? h->model.nq : MZ_ERR_ARG; }
int32_t mz_nv(const mz_handle* h) { return h ? h->model.nv : MZ_ERR_ARG; }
int32_t mz_nu(const mz_handle* h) { return h ? h->model.nu : MZ_ERR_ARG; }

int32_t mz_set_option(mz_handle* h, const char* key, double value) {
  if (!h || !key) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  if (!strcmp(key, "auto_reset")) { h->auto_reset = value != 0; return MZ_OK; }
  if (!strcmp(key, "seed")) { h->seed = (uint64_t)value; return MZ_OK; }
  if (!strcmp(key, "env_index_offset")) { h->env0 = (uint64_t)value; return MZ_OK; }
  if (h->robot == MZ_ROBOT_POINT && !strcmp(key, "ls_fast_iterations")) {  // the Point's float64 solvers: unit steps before the exact search (0 = search in every iteration)
    if (value < 0 || value > 50) return set_err(h, MZ_ERR_ARG, "ls_fast_iterations must be 0 .. 50", hipSuccess);
    h->point.unit_steps = (int)value;
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(h->point_dev, &h->point, sizeof(PointDev), hipMemcpyHostToDevice));
    return MZ_OK;
  }
  if (h->robot != MZ_ROBOT_ANT && (!strcmp(key, "solver_iterations") || !strcmp(key, "solver_tolerance") || !strcmp(key, "solver_rtol") ||
                                   !strcmp(key, "ls_iterations") || !strcmp(key, "ls_fast_iterations") || !strcmp(key, "ls_fast") || !strcmp(key, "debug_frame_skip") || !strcmp(key, "waves_per_block") || !strcmp(key, "waves_per_simd")))
    return set_err(h, MZ_ERR_UNSUPPORTED, "mz_set_option: this key tunes the Ant kernels only (the other robots' solvers have fixed settings)", hipSuccess);
  if (h->robot == MZ_ROBOT_GENERIC && !strcmp(key, "lanes_per_env"))
    return set_err(h, MZ_ERR_UNSUPPORTED, "mz_set_option: the generic-robot kernel runs one wavefront per env", hipSuccess);
  if (!strcmp(key, "solver_iterations")) { h->ant_dirty = 1; h->ant.max_iter = (int)value; return MZ_OK; }
  if (!strcmp(key, "solver_tolerance")) { h->ant_dirty = 1; h->ant.tol = (float)value; return MZ_OK; }
  if (!strcmp(key, "solver_rtol")) { h->ant_dirty = 1; h->ant.rtol = (float)value; return MZ_OK; }
  if (!strcmp(key, "ls_iterations")) { h->ant_dirty = 1; h->ant.ls_iter = (int)value; return MZ_OK; }
  if (!strcmp(key, "ls_fast_iterations")) { h->ant_dirty = 1; h->ant.ls_fast_iters = (int)value; return MZ_OK; }
  if (!strcmp(key, "ls_fast")) { h->ant_dirty = 1; h->ant.ls_fast = (int)value; return MZ_OK; }
  if (!strcmp(key, "debug_frame_skip")) { h->ant_dirty = 1; h->ant.frame_skip = (int)value; return MZ_OK; }  // diagnostics: mj_steps per env.step (Ant)
  if (!strcmp(key, "lanes_per_env")) {
    int g = (int)value;
    if (g != 8 && g != 16 && g != 32 && g != 64) return set_err(h, MZ_ERR_ARG, "lanes_per_env must be 8, 16, 32 or 64", hipSuccess);
    h->lanes = g; h->lanes_set = 1;
    return MZ_OK;
  }
  if (!strcmp(key, "profile_phases")) {
    if (value != 0 && !h->prof) {  // 16 accumulators + one total per workgroup (at most one workgroup per env)
      HIPCHK(h, hipMalloc(&h->prof, (16 + 17 * (size_t)h->n) * sizeof(unsigned long long)));
      HIPCHK(h, hipMemset(h->prof, 0, (16 + 17 * (size_t)h->n) * sizeof(unsigned long long)));
    }
    if (value == 0 && h->prof) {
      HIPCHK(h, hipDeviceSynchronize());  // an instrumented step kernel may still be adding to the accumulators
      (void)hipFree(h->prof); h->prof = nullptr;
    }
    return MZ_OK;
  }
  if (!strcmp(key, "waves_per_block")) {
    int w = (int)value;
    if (w != 1 && w != 2 && w != 4) return set_err(h, MZ_ERR_ARG, "waves_per_block must be 1, 2 or 4", hipSuccess);
    h->waves_per_block = w; h->wpb_set = 1;
    return MZ_OK;
  }
  if (!strcmp(key, "waves_per_simd")) {
    int w = (int)value;
    if (w < 0 || w > 2) return set_err(h, MZ_ERR_ARG, "waves_per_simd must be 0 (by the launch's wave count), 1 or 2", hipSuccess);
    h->waves_per_simd = w;
    return MZ_OK;
  }
  if (!strcmp(key, "time_kernels")) {
    if (h->ev) { for (int i = 0; i < 2 * h->ntime; i++) (void)hipEventDestroy(h->ev[i]); free(h->ev); h->ev = nullptr; }
    h->ntime = (int)value; h->itime = 0; h->time_phase = 0;
    if (h->ntime > 0) {
      h->ev = (hipEvent_t*)calloc((size_t)2 * h->ntime, sizeof(hipEvent_t));
      // (no system-scope fence: the default event writes back and invalidates the caches at every record — the timestamps are read by this process
      // after a synchronise, nothing on the host polls the memory the kernels write)
      for (int i = 0; i < 2 * h->ntime; i++) HIPCHK(h, hipEventCreateWithFlags(&h->ev[i], hipEventDisableSystemFence));
    }
    return MZ_OK;
  }
  if (!strcmp(key, "time_kernels_stride")) {  // time every k-th launch only: an event pair costs a small kernel 7.6 us of its 45 (profiles/r06/event_overhead.txt)
    if (value < 1) return set_err(h, MZ_ERR_ARG, "time_kernels_stride must be >= 1", hipSuccess);
    h->time_stride = (int)value; h->time_phase = 0;
    return MZ_OK;
  }
  return set_err(h, MZ_ERR_ARG, "unknown option", hipSuccess);
}

// What the next mz_step will launch — the kernel instantiation depends on the robot, the batch size and the device's compute-unit
// count (include/mazestep.h), so a caller that compares runs across batch sizes or devices can record it.
int32_t mz_get_info(const mz_handle* h, const char* key, double* value) {
  if (!h || !key || !value) return MZ_ERR_ARG;
  if (!strcmp(key, "engine")) { *value = h->robot == MZ_ROBOT_GENERIC ? 1.0 : 0.0; return MZ_OK; }
  if (!strcmp(key, "device_simds")) { *value = (double)h->simds; return MZ_OK; }
  if (!strcmp(key, "ls_fast_iterations")) {  // Newton iterations of a solve that take unit steps (-1: this handle's kernels have no such phase)
    *value = h->robot == MZ_ROBOT_ANT ? (double)h->ant.ls_fast_iters : (h->robot == MZ_ROBOT_POINT ? (double)h->point.unit_steps : -1.0);
    return MZ_OK;
  }
  if (!strcmp(key, "lanes_per_env") || !strcmp(key, "waves_per_simd")) {
    int lanes = 64, wps = 1;
    if (h->robot == MZ_ROBOT_ANT) mzk_ant_shape(h, &lanes, &wps);
    else if (h->robot != MZ_ROBOT_GENERIC) lanes = mzk_planar_lanes(h);
    *value = (double)(key[0] == 'l' ? lanes : wps);
    return MZ_OK;
  }
  return MZ_ERR_ARG;
}

int32_t mz_bind_final_obs(mz_handle* h, float* final_obs_dev) {
  if (!h) return MZ_ERR_ARG;
  h->final_obs = final_obs_dev;
  return MZ_OK;
}

int32_t mz_bind_record(mz_handle* h, float* record_dev) {
  if (!h) return MZ_ERR_ARG;
  h->record = record_dev;
  return MZ_OK;
}

}  // extern "C"
// (re)build the task block from h->model (+ the bound per-env goal table) and upload it to the handle's kernels
static int upload_task(mz_handle* h) {
  mz_model* m = &h->model;
  if (h->robot == MZ_ROBOT_ANT) {
    task_dev_from_model(&h->ant.task, m); h->ant.task.env_goals = h->env_goals;
    HIPCHK(h, hipMemcpy(reinterpret_cast<char*>(h->ant_dev) + offsetof(AntDev, task), &h->ant.task, sizeof(TaskDev), hipMemcpyHostToDevice));
  } else if (h->robot == MZ_ROBOT_POINT) {
    task_dev_from_model(&h->point.task, m); h->point.task.env_goals = h->env_goals;
    HIPCHK(h, hipMemcpy(reinterpret_cast<char*>(h->point_dev) + offsetof(PointDev, task), &h->point.task, sizeof(TaskDev), hipMemcpyHostToDevice));
  } else if (h->robot == MZ_ROBOT_SWIMMER) {
    task_dev_from_model(&h->swimmer.task, m); h->swimmer.task.env_goals = h->env_goals;
    HIPCHK(h, hipMemcpy(reinterpret_cast<char*>(h->swimmer_dev) + offsetof(SwimmerDev, task), &h->swimmer.task, sizeof(TaskDev), hipMemcpyHostToDevice));
  } else {
    TaskDev t;
    task_dev_from_model(&t, m); t.env_goals = h->env_goals;
    HIPCHK(h, mzk_generic_set_task(h, &t));
  }
  return MZ_OK;
}
extern "C" {

// Per-env goal positions.  The reference gives every env its own task object and resamples its goals at EVERY episode reset
// (MazeEnv.reset -> MazeTask.sample_goals, maze_env.py:374-376); a batch that shares one goal table cannot.  goal_pos_dev: DEVICE
// pointer, [num_envs][MZ_MAX_GOAL][3] float64, caller-owned and caller-updated (the host mirror rewrites the rows of the envs
// whose episode ended, between two steps); NULL unbinds (back to the shared table of mz_set_goals / the model).  Thresholds, reward
// scales, dims and the goal COUNT stay those of the shared table.  Synchronises on `stream` first.  Returns MZ_OK or an error code.
int32_t mz_bind_env_goals(mz_handle* h, const double* goal_pos_dev, void* stream) {
  if (!h) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
  h->env_goals = goal_pos_dev;
  return upload_task(h);
}

int32_t mz_set_goals(mz_handle* h, int32_t ngoal, const double* pos, const double* threshold, const double* reward_scale, const int32_t* dim,
                     void* stream) {
  if (!h) return MZ_ERR_ARG;
  if (ngoal < 0 || ngoal > MZ_MAX_GOAL || (ngoal > 0 && (!pos || !threshold || !reward_scale || !dim)))
    return set_err(h, MZ_ERR_ARG, "mz_set_goals: 0 <= ngoal <= MZ_MAX_GOAL rows of pos / threshold / reward_scale / dim", hipSuccess);
  for (int g = 0; g < ngoal; g++)
    if ((dim[g] != 2 && dim[g] != 3) || !(threshold[g] >= 0.0)) return set_err(h, MZ_ERR_ARG, "mz_set_goals: dim must be 2 or 3, threshold >= 0", hipSuccess);
  DeviceScope scope(h->device);
  mz_model* m = &h->model;
  m->ngoal = ngoal;
  for (int g = 0; g < ngoal; g++) {
    for (int k = 0; k < 3; k++) m->goal_pos[g][k] = pos[3 * g + k];
    m->goal_threshold[g] = threshold[g]; m->goal_reward_scale[g] = reward_scale[g]; m->goal_dim[g] = dim[g];
  }
  HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));  // steps already queued keep the goals they were launched with
  return upload_task(h);
}

int32_t mz_reset(mz_handle* h, const uint8_t* mask_dev, uint64_t seed, float* obs_dev, void* stream) {
  if (!h) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  hipStream_t st = (hipStream_t)stream;
  h->seed = seed;
  if (h->robot == MZ_ROBOT_ANT) HIPCHK(h, mzk_ant_reset(h, st, mask_dev, seed, obs_dev));
  else if (h->robot == MZ_ROBOT_GENERIC) HIPCHK(h, mzk_generic_reset(h, st, mask_dev, seed, obs_dev));
  else HIPCHK(h, mzk_planar_reset(h, st, mask_dev, seed, obs_dev));
  if (h->view.on && obs_dev) HIPCHK(h, mzk_view_fill(h, st, obs_dev, NULL, NULL));
  return MZ_OK;
}

int32_t mz_set_state(mz_handle* h, const float* qpos_dev, const float* qvel_dev, const float* warmstart_dev, const int32_t* t_dev,
                     void* stream) {
  if (!h) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  hipStream_t st = (hipStream_t)stream;
  if (h->robot == MZ_ROBOT_ANT) HIPCHK(h, mzk_ant_set_state(h, st, qpos_dev, qvel_dev, warmstart_dev, t_dev));
  else if (h->robot == MZ_ROBOT_GENERIC) HIPCHK(h, mzk_generic_set_state(h, st, qpos_dev, qvel_dev, warmstart_dev, t_dev));
  else HIPCHK(h, mzk_planar_set_state(h, st, qpos_dev, qvel_dev, t_dev));
  return MZ_OK;
}

int32_t mz_get_state(mz_handle* h, float* qpos_dev, float* qvel_dev, float* warmstart_dev, int32_t* t_dev, void* stream) {
  if (!h) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  hipStream_t st = (hipStream_t)stream;
  if (h->robot == MZ_ROBOT_ANT) HIPCHK(h, mzk_ant_get_state(h, st, qpos_dev, qvel_dev, warmstart_dev, t_dev));
  else if (h->robot == MZ_ROBOT_GENERIC) HIPCHK(h, mzk_generic_get_state(h, st, qpos_dev, qvel_dev, warmstart_dev, t_dev));
  else HIPCHK(h, mzk_planar_get_state(h, st, qpos_dev, qvel_dev, warmstart_dev, t_dev));
  return MZ_OK;
}

int32_t mz_step(mz_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, int32_t* goal_idx_dev,
                float* info_dev, void* stream) {
  if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev) return h ? set_err(h, MZ_ERR_ARG, "mz_step: null array", hipSuccess) : MZ_ERR_ARG;
  DeviceScope scope(h->device);
  hipStream_t st = (hipStream_t)stream;
  int slot = -1;
  if (h->ntime > 0 && (h->time_phase++ % (h->time_stride > 0 ? h->time_stride : 1)) == 0) { slot = h->itime % h->ntime; HIPCHK(h, hipEventRecord(h->ev[2 * slot], st)); }
  if (h->robot == MZ_ROBOT_ANT) HIPCHK(h, mzk_ant_step(h, st, actions_dev, obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev));
  else if (h->robot == MZ_ROBOT_GENERIC) HIPCHK(h, mzk_generic_step(h, st, actions_dev, obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev));
  else HIPCHK(h, mzk_planar_step(h, st, actions_dev, obs_dev, reward_dev, done_dev, goal_idx_dev, info_dev));
  if (h->view.on) HIPCHK(h, mzk_view_fill(h, st, obs_dev, h->auto_reset ? h->final_obs : NULL, done_dev));
  if (h->record && ((h->robot != MZ_ROBOT_ANT && h->robot != MZ_ROBOT_GENERIC) || h->view.on)) {
    const size_t tot = (size_t)h->n * (h->model.obs_dim + 2);
    hipLaunchKernelGGL(pack_record_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->n, h->model.obs_dim, obs_dev, reward_dev, done_dev, h->record);
  }
  HIPCHK(h, hipGetLastError());
  if (slot >= 0) { HIPCHK(h, hipEventRecord(h->ev[2 * slot + 1], st)); h->itime++; }
  h->nsteps++;
  return MZ_OK;
}

int32_t mz_get_status(mz_handle* h, int32_t* status_dev, void* stream) {
  if (!h || !status_dev) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  hipLaunchKernelGGL(fetch_clear_status_kernel, dim3((h->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->n, h->status, status_dev);
  HIPCHK(h, hipGetLastError());
  return MZ_OK;
}

int32_t mz_debug_forward(mz_handle* h, const float* actions_dev, float* qacc_dev, int32_t* counts_dev, void* stream) {
  if (!h || !qacc_dev) return MZ_ERR_ARG;
  if (h->robot != MZ_ROBOT_ANT) return set_err(h, MZ_ERR_UNSUPPORTED, "mz_debug_forward: ant only", hipSuccess);
  DeviceScope scope(h->device);
  HIPCHK(h, mzk_ant_forward(h, (hipStream_t)stream, actions_dev, qacc_dev, counts_dev));
  HIPCHK(h, hipGetLastError());
  return MZ_OK;
}

int32_t mz_debug_task_eval(mz_handle* h, int32_t n_rows, const float* obs_dev, float* reward_dev, uint8_t* done_dev, int32_t* goal_idx_dev,
                           void* stream) {
  if (!h || n_rows <= 0 || !obs_dev || !reward_dev || !done_dev) return h ? set_err(h, MZ_ERR_ARG, "mz_debug_task_eval: bad arguments", hipSuccess) : MZ_ERR_ARG;
  DeviceScope scope(h->device);
  if (h->robot == MZ_ROBOT_ANT) HIPCHK(h, mzk_ant_task_eval(h, (hipStream_t)stream, n_rows, obs_dev, reward_dev, done_dev, goal_idx_dev));
  else if (h->robot == MZ_ROBOT_GENERIC) HIPCHK(h, mzk_generic_task_eval(h, (hipStream_t)stream, n_rows, obs_dev, reward_dev, done_dev, goal_idx_dev));
  else HIPCHK(h, mzk_planar_task_eval(h, (hipStream_t)stream, n_rows, obs_dev, reward_dev, done_dev, goal_idx_dev));
  return MZ_OK;
}

int32_t mz_debug_detect(mz_handle* h, int32_t n_rows, const double* old_xy_dev, const double* new_xy_dev, int32_t* hit_dev, double* point_dev,
                        double* final_xy_dev, void* stream) {
  if (!h || n_rows <= 0 || !old_xy_dev || !new_xy_dev || !hit_dev || !final_xy_dev)
    return h ? set_err(h, MZ_ERR_ARG, "mz_debug_detect: bad arguments", hipSuccess) : MZ_ERR_ARG;
  if (h->robot != MZ_ROBOT_POINT || h->point.nseg <= 0) return set_err(h, MZ_ERR_UNSUPPORTED, "mz_debug_detect: the manual wall detector exists for the Point only", hipSuccess);
  DeviceScope scope(h->device);
  HIPCHK(h, mzk_point_detect(h, (hipStream_t)stream, n_rows, old_xy_dev, new_xy_dev, hit_dev, point_dev, final_xy_dev));
  return MZ_OK;
}

// Phase timers of the PROF kernel build: copies the 16 accumulators to host memory and clears them.
int32_t mz_read_phase_cycles(mz_handle* h, uint64_t* out16_host) {
  if (!h || !out16_host || !h->prof) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  HIPCHK(h, hipDeviceSynchronize());
  // the step kernels write per-workgroup slots only (no same-address atomics: they disturbed what they measured, ant_kernels.hip);
  // the 16 launch-wide accumulators are reduced here from the per-workgroup phase slots — accumulated since the last
  // mz_read_wave_phase_cycles, which hands those slots out and clears them; this call clears nothing
  const size_t nw = (size_t)h->n;
  unsigned long long* buf = (unsigned long long*)malloc(17 * nw * sizeof(unsigned long long));
  if (!buf) return set_err(h, MZ_ERR_HIP, "mz_read_phase_cycles: out of memory", hipSuccess);
  hipError_t e = hipMemcpy(buf, h->prof + 16, 17 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(buf); return set_err(h, MZ_ERR_HIP, "mz_read_phase_cycles", e); }
  for (int k = 0; k < 16; k++) out16_host[k] = 0;
  for (size_t w = 0; w < nw; w++) {
    const unsigned long long* ph = buf + nw + 16 * w;
    unsigned long long tot = 0;
    for (int k = 0; k < 13; k++) tot += ph[k];
    if (!tot) continue;  // (slots of workgroups that do not exist at this lane width stay empty)
    for (int k = 0; k < 16; k++) if (k != 13 && k != 14) out16_host[k] += ph[k];
    if (tot > out16_host[13]) out16_host[13] = tot;   // slowest workgroup of the window (per-workgroup sums over the window's steps)
    out16_host[14] += (tot >> 8) * (tot >> 8);         // sum of squares of the per-workgroup totals (units of 256 cycles)
  }
  free(buf);
  return MZ_OK;
}

// Per-workgroup totals accumulated by the PROF kernel build since the last call (then cleared): n_host entries, each
// cycles (low 40 bits) + the Newton iterations the workgroup ran (bits 40 and up).
int32_t mz_read_wave_cycles(mz_handle* h, uint64_t* out_host, int32_t n_host) {
  if (!h || !out_host || !h->prof || n_host <= 0 || n_host > h->n) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(out_host, h->prof + 16, (size_t)n_host * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemset(h->prof + 16, 0, (size_t)h->n * sizeof(unsigned long long)));
  return MZ_OK;
}

// Per-workgroup, per-phase cycles of the PROF kernel build accumulated since the last call (then cleared): [n_host][16].
int32_t mz_read_wave_phase_cycles(mz_handle* h, uint64_t* out_host, int32_t n_host) {
  if (!h || !out_host || !h->prof || n_host <= 0 || n_host > h->n) return MZ_ERR_ARG;
  DeviceScope scope(h->device);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(out_host, h->prof + 16 + h->n, (size_t)n_host * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemset(h->prof + 16 + h->n, 0, (size_t)h->n * 16 * sizeof(unsigned long long)));
  return MZ_OK;
}

// Average duration (ms) of the step kernel over the recorded ring (after synchronising on the last event).
double mz_last_kernel_ms(const mz_handle* h) {
  if (!h || h->ntime <= 0 || h->itime <= 0) return -1.0;
  DeviceScope scope(h->device);
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
