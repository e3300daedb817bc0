// ant_emu.cpp — CPU emulation of the Ant kernel's lane-group code (TEST INFRASTRUCTURE).
//
// Compiles mujoco_maze_amd/csrc/ant_dyn.h — the exact source the HIP kernel
// instantiates — with the one-lane HostCtx, so that the kernel's *logic and fp32
// numerics* can be compared with the float64 oracle on a machine without a GPU.
// It is not a fallback: nothing under mujoco_maze_amd/ loads this library and the
// product path raises if the HIP library / a GPU is missing.
#include <stdlib.h>

#include "../../mujoco_maze_amd/csrc/ant_dyn.h"

extern "C" {

int emu_ant_sizeof_scratch(void) { return (int)sizeof(AntScratch); }

// One MazeEnv.step for n envs (row-major arrays as in the C-ABI's get/set_state).
int emu_ant_env_step(const mz_model* m, int n, float* qpos, float* qvel, float* warm, int32_t* t, const float* actions,
                     float* obs, float* reward, uint8_t* done, int32_t* goal_idx, float* info, int32_t* status,
                     int32_t* iters, int max_iter, float tol, float rtol) {
  AntDev K;
  char err[128];
  int rc = ant_dev_from_model(&K, m, err, sizeof(err));
  if (rc != MZ_OK) return rc;
  if (max_iter > 0) K.max_iter = max_iter;
  if (tol > 0) K.tol = tol;
  if (rtol >= 0) K.rtol = rtol;
  HostCtx cx;
  AntScratch* s = (AntScratch*)calloc(1, sizeof(AntScratch));
  for (int e = 0; e < n; e++) {
    for (int k = 0; k < ANT_NQ; k++) s->qpos[k] = qpos[e * ANT_NQ + k];
    for (int k = 0; k < ANT_NV; k++) { s->qvel[k] = qvel[e * ANT_NV + k]; s->warm[k] = warm[e * ANT_NV + k]; }
    int gi = -1, tout = 0;
    ant_env_step(cx, K, *s, actions + e * ANT_NU, t[e], obs + e * ANT_OBS, reward + e, done + e, &gi, info ? info + 4 * e : nullptr, &tout);
    for (int k = 0; k < ANT_NQ; k++) qpos[e * ANT_NQ + k] = s->qpos[k];
    for (int k = 0; k < ANT_NV; k++) { qvel[e * ANT_NV + k] = s->qvel[k]; warm[e * ANT_NV + k] = s->warm[k]; }
    t[e] = tout;
    if (goal_idx) goal_idx[e] = gi;
    if (status) status[e] = s->status;
    if (iters) iters[e] = s->iters;
  }
  free(s);
  return MZ_OK;
}

// One forward-dynamics evaluation per env: qacc [n,14], counts [n,2] = (ncon, newton iterations),
// optional dense mass matrix [n,14,14], bias [n,14], qacc_smooth [n,14].
int emu_ant_forward(const mz_model* m, int n, const float* qpos, const float* qvel, const float* warm, const float* actions,
                    float* qacc, int32_t* counts, float* Mout, float* bias, float* qas, int max_iter, float tol, float rtol) {
  AntDev K;
  char err[128];
  int rc = ant_dev_from_model(&K, m, err, sizeof(err));
  if (rc != MZ_OK) return rc;
  if (max_iter > 0) K.max_iter = max_iter;
  if (tol > 0) K.tol = tol;
  if (rtol >= 0) K.rtol = rtol;
  HostCtx cx;
  AntScratch* s = (AntScratch*)calloc(1, sizeof(AntScratch));
  for (int e = 0; e < n; e++) {
    for (int k = 0; k < ANT_NQ; k++) s->qpos[k] = qpos[e * ANT_NQ + k];
    for (int k = 0; k < ANT_NV; k++) { s->qvel[k] = qvel[e * ANT_NV + k]; s->warm[k] = warm ? warm[e * ANT_NV + k] : 0.f; s->fact[k] = 0.f; }
    if (actions)
      for (int u = 0; u < ANT_NU; u++) s->fact[K.act_dof[u]] = K.gear * fminf(fmaxf(actions[e * ANT_NU + u], K.ctrl_lo), K.ctrl_hi);
    s->status = 0;
    ant_forward(cx, K, *s, true);
    for (int k = 0; k < ANT_NV; k++) qacc[e * ANT_NV + k] = s->qacc[k];
    if (counts) { counts[2 * e] = s->ncon; counts[2 * e + 1] = s->iters; }
    if (bias) for (int k = 0; k < ANT_NV; k++) bias[e * ANT_NV + k] = s->bias[k];
    if (qas) for (int k = 0; k < ANT_NV; k++) qas[e * ANT_NV + k] = s->qas[k];
    if (Mout) {
      float* M = Mout + (size_t)e * ANT_NV * ANT_NV;
      float x[ANT_NV];
      for (int j = 0; j < ANT_NV; j++) {
        for (int k = 0; k < ANT_NV; k++) x[k] = k == j ? 1.f : 0.f;
        for (int i = 0; i < ANT_NV; i++) M[i * ANT_NV + j] = arrow_row_mul(s->M, x, i);
      }
    }
  }
  free(s);
  return MZ_OK;
}
}
