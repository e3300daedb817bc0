// mz_internal.h — what the three translation units of libmazestep.so share on the HOST side:
//   mazestep.hip        the C-ABI of include/mazestep.h (handle life cycle, options, argument checks, dispatch)
//   ant_kernels.hip     Ant kernels (fp32, built with the relaxed floating-point flags of csrc/Makefile)
//   planar_kernels.hip  Point / Swimmer / Reacher kernels (fp64, built with strict IEEE flags: the Point's manual wall
//                       detector and every task predicate must reproduce the reference's float64 decisions bit for bit)
// Kernels never call across translation units, so no relocatable device code is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ant_model.h"
#include "mz_view.h"
#include "point_dyn.h"
#include "swimmer_dyn.h"

struct GenDev;  // generic_dyn.h (the generic-robot path keeps its float64 constant block out of the other translation units)

struct AntLayout { int nq, nv, rec, rec_t, obs_dim, nblock3, ostride; };  // record layout of the instantiated block count; obs_dim: without the
                                                                          // top-down view, ostride: floats between observation rows

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
  GenDev* gen_dev;      // generic robot (MZ_ROBOT_GENERIC): device copy of the constant block (mz_model + pair tables)
  float* state;         // ant: [n][REC]; point / swimmer: SoA [2 NV][n]
  int* pt_t;
  uint32_t* pt_ep;
  int pt_rec;           // Point: floats per env-major state record (planar_kernels.hip PointState); 0: SoA
  int* status;
  ViewDev view;         // MazeTask.TOP_DOWN_VIEW: the maze bitmasks the view kernel reads (passed by value)
  int base_obs;         // observation width without the view; rows are model.obs_dim = base_obs (+ MZ_VIEW_DIM) floats apart
  float* final_obs;     // caller's buffer for terminal observations under auto-reset (mz_bind_final_obs), or NULL
  float* record;        // caller's [n, obs_dim + 2] buffer for the packed record obs | reward | done (mz_bind_record), or NULL
  unsigned long long* prof;  // 16 phase-cycle accumulators (option "profile_phases")
  int auto_reset, lanes, waves_per_block, wpb_set;
  int waves_per_simd;  // option "waves_per_simd": 0 = chosen by the launch's wave count, 1 / 2 = the plain ant's one- / two-wave kernel (ant_kernels.hip)
  uint64_t seed, env0;  // env0: global slot of local env 0 (sharded runs)
  char err[256];
  int lanes_set;  // lanes_per_env chosen by the caller (else the per-robot default)
  const double* env_goals;  // caller's per-env goal positions (mz_bind_env_goals), or NULL: carried into every TaskDev copy
  int simds;      // 4 x the device's compute units, read once in mz_create: the batch-size rules of the launch shapes compare wave counts with it
  // kernel timing ring (option "time_kernels")
  int ntime, itime;
  int time_stride, time_phase;  // option "time_kernels_stride": every k-th launch carries the event pair
  hipEvent_t* ev;  // 2 * ntime
  long nsteps;
};

// ---- ant_kernels.hip
hipError_t mzk_ant_step(mz_handle* h, hipStream_t st, const float* actions, float* obs, float* reward, uint8_t* done, int* goal_idx, float* info);
hipError_t mzk_ant_forward(mz_handle* h, hipStream_t st, const float* actions, float* qacc, int* counts);
hipError_t mzk_ant_reset(mz_handle* h, hipStream_t st, const uint8_t* mask, uint64_t seed, float* obs);
hipError_t mzk_ant_set_state(mz_handle* h, hipStream_t st, const float* qpos, const float* qvel, const float* warm, const int* t);
hipError_t mzk_ant_get_state(mz_handle* h, hipStream_t st, float* qpos, float* qvel, float* warm, int* t);
hipError_t mzk_ant_task_eval(mz_handle* h, hipStream_t st, int n, const float* obs, float* reward, uint8_t* done, int* goal_idx);

// launch shape the next mz_step takes (mz_get_info): lanes per env, and for the plain ant at 16 lanes which instantiation (1 / 2 waves per SIMD)
void mzk_ant_shape(const mz_handle* h, int* lanes, int* waves_per_simd);
int mzk_planar_lanes(const mz_handle* h);

// ---- planar_kernels.hip (Point, Swimmer, Reacher)
hipError_t mzk_planar_step(mz_handle* h, hipStream_t st, const float* actions, float* obs, float* reward, uint8_t* done, int* goal_idx, float* info);
hipError_t mzk_planar_reset(mz_handle* h, hipStream_t st, const uint8_t* mask, uint64_t seed, float* obs);
hipError_t mzk_planar_set_state(mz_handle* h, hipStream_t st, const float* qpos, const float* qvel, const int* t);
hipError_t mzk_planar_get_state(mz_handle* h, hipStream_t st, float* qpos, float* qvel, float* warm, int* t);
hipError_t mzk_planar_task_eval(mz_handle* h, hipStream_t st, int n, const float* obs, float* reward, uint8_t* done, int* goal_idx);
hipError_t mzk_point_detect(mz_handle* h, hipStream_t st, int n, const double* old_xy, const double* new_xy, int* hit, double* point,
                            double* final_xy);
// MazeEnv.get_top_down_view into the rows the step / reset kernel just wrote (no-op unless the task has TOP_DOWN_VIEW):
// every row of obs, and the rows of final_obs of envs that finished (done != NULL: the step under auto-reset)
hipError_t mzk_view_fill(mz_handle* h, hipStream_t st, float* obs, float* final_obs, const uint8_t* done);
int mzk_planar_state_width(const mz_handle* h);  // coordinates per env of the state (NV)
int mzk_planar_record_width(const mz_handle* h); // Point: floats per env-major record; 0 for the chains (SoA)

// ---- generic_kernels.hip (a user robot of any tree topology, csrc/generic_dyn.h)
int mzk_generic_needed(const mz_model* m);  // no specialised kernel steps this model, or the caller asked for the general engine (mz_model.engine)
int mzk_generic_create(mz_handle* h, char* err, int errlen);  // builds + uploads the constant block; MZ_OK or MZ_ERR_*
void mzk_generic_destroy(mz_handle* h);
hipError_t mzk_generic_step(mz_handle* h, hipStream_t st, const float* actions, float* obs, float* reward, uint8_t* done, int* goal_idx, float* info);
hipError_t mzk_generic_reset(mz_handle* h, hipStream_t st, const uint8_t* mask, uint64_t seed, float* obs);
hipError_t mzk_generic_set_state(mz_handle* h, hipStream_t st, const float* qpos, const float* qvel, const float* warm, const int* t);
hipError_t mzk_generic_get_state(mz_handle* h, hipStream_t st, float* qpos, float* qvel, float* warm, int* t);
hipError_t mzk_generic_set_task(mz_handle* h, const TaskDev* task);  // replaces the task block of the device constants (mz_set_goals)
hipError_t mzk_generic_task_eval(mz_handle* h, hipStream_t st, int n, const float* obs, float* reward, uint8_t* done, int* goal_idx);
