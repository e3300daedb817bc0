/* mzo.h — CPU oracle (test infrastructure). See mzo_physics.c / mzo_env.c. */
#ifndef MZO_H
#define MZO_H
#include "../include/mazestep.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MZO_MAX_CON 96
#define MZO_MAX_EFC (8 + 2 * MZ_MAX_JNT + 4 * MZO_MAX_CON)
#define MZO_STATUS_UNSUPPORTED_PAIR 256

typedef struct {
  double dist, pos[3], frame[9], includemargin, mu, solref[2], solimp[5];
  int dim, body1, body2, geom1, geom2, efc_address;
} mzo_contact;

typedef struct mzo_data {
  /* state */
  double qpos[MZ_MAX_Q], qvel[MZ_MAX_DOF], qacc_warmstart[MZ_MAX_DOF], time;
  /* position-dependent */
  double xpos[MZ_MAX_BODY][3], xquat[MZ_MAX_BODY][4], xmat[MZ_MAX_BODY][9], xipos[MZ_MAX_BODY][3];
  double xanchor[MZ_MAX_JNT][3], xaxis[MZ_MAX_JNT][3];
  double geom_xpos[MZ_MAX_GEOM][3], geom_xmat[MZ_MAX_GEOM][9];
  double refpoint[3], S[MZ_MAX_DOF][6], cinert[MZ_MAX_BODY][10], cvel[MZ_MAX_BODY][6];
  double M[MZ_MAX_DOF][MZ_MAX_DOF];
  /* forces / accelerations */
  double qfrc_bias[MZ_MAX_DOF], qfrc_passive[MZ_MAX_DOF], qfrc_actuator[MZ_MAX_DOF], qfrc_smooth[MZ_MAX_DOF];
  double qacc_smooth[MZ_MAX_DOF], qacc[MZ_MAX_DOF];
  /* contacts and constraint rows */
  int ncon, nefc, solver_iter, status;
  mzo_contact con[MZO_MAX_CON];
  double efc_J[MZO_MAX_EFC][MZ_MAX_DOF];
  double efc_pos[MZO_MAX_EFC], efc_margin[MZO_MAX_EFC], efc_R[MZO_MAX_EFC], efc_D[MZO_MAX_EFC], efc_K[MZO_MAX_EFC],
      efc_B[MZO_MAX_EFC], efc_imp[MZO_MAX_EFC], efc_aref[MZO_MAX_EFC], efc_force[MZO_MAX_EFC];
  /* solver knobs (0 -> defaults 1e-10 / 100) */
  double solver_tol;
  int solver_maxiter;
} mzo_data;

uint64_t mzo_model_sizeof(void);
uint64_t mzo_data_sizeof(void);
void mzo_data_init(const mz_model* m, mzo_data* d);
void mzo_forward(const mz_model* m, mzo_data* d, const double* ctrl);
void mzo_mj_step(const mz_model* m, mzo_data* d, const double* ctrl);
double mzo_energy(const mz_model* m, mzo_data* d);
void mzo_fluid_passive(const mz_model* m, mzo_data* d);

/* --- maze-level restatement (mzo_env.c) ------------------------------------- */
/* CollisionDetector.detect (maze_env_utils.py:186-206). Returns 1 on hit, -1 on the
 * reference's latent division by zero (collinear), else 0. */
int mzo_detect(const mz_model* m, const double* old_xy, const double* new_xy, double* point, double* reflection);
int mzo_bounce(const mz_model* m, const double* old_xy, const double* new_xy, double* final_xy);
/* MazeTask.reward / termination on an observation (maze_task.py:43-47,77-81,110-111,403-407) */
void mzo_task_eval(const mz_model* m, const double* obs, double* reward, int* done, int* goal_idx);

/* One MazeEnv.step (maze_env.py:448-481) for one env, state in/out. */
typedef struct {
  double qpos[MZ_MAX_Q], qvel[MZ_MAX_DOF], warmstart[MZ_MAX_DOF];
  int32_t t, status;
} mzo_env_state;
void mzo_env_step(const mz_model* m, mzo_env_state* s, const double* action, double* obs, double* reward, uint8_t* done,
                  int32_t* goal_idx, double* info4, double solver_tol);
void mzo_env_obs(const mz_model* m, const mzo_env_state* s, double* obs);
/* MazeEnv.get_top_down_view (maze_env.py:262-349): view[MZ_VIEW_DIM] from the torso position and the movable blocks' xy */
void mzo_top_down_view(const mz_model* m, double robot_x, double robot_y, int nblock, const double* block_xy, double* view);
/* reset distribution (ant.py:84-96, point.py:71-81) with the library's counter-based RNG */
void mzo_env_reset(const mz_model* m, mzo_env_state* s, uint64_t seed, uint64_t env_index);

/* batch drivers (row-major arrays, OpenMP over envs when nthreads > 1) */
void mzo_batch_step(const mz_model* m, int n, double* qpos, double* qvel, double* warm, int32_t* t, const double* actions,
                    double* obs, double* reward, uint8_t* done, int32_t* goal_idx, double* info, int32_t* status,
                    int nthreads, double solver_tol);
void mzo_batch_reset(const mz_model* m, int n, const uint8_t* mask, uint64_t seed, uint64_t env0, double* qpos, double* qvel,
                     double* warm, int32_t* t, double* obs);
/* diagnostics: one forward evaluation per env */
void mzo_batch_forward(const mz_model* m, int n, const double* qpos, const double* qvel, const double* warm,
                       const double* actions, double* qacc, int32_t* counts, double* Mout, double* bias);

void mzo_forward_report(const mz_model* m, const double* qpos, const double* qvel, const double* warm, const double* ctrl,
                        double* report, double* qacc_out);
int mzo_probe_pair(int kind, const double* pos1, const double* mat1, const double* size1, const double* pos2, const double* mat2,
                   const double* size2, double margin, int max_con, double* out);
int mzo_contacts(const mz_model* m, const double* qpos, int max_con, double* out);
void mzo_raw_steps(const mz_model* m, double* qpos, double* qvel, const double* ctrl, int nsteps, double* energy_out);

/* RNG shared by oracle and kernels: 64-bit counter hash -> uniform [0,1) */
uint32_t mzo_rng_u32(uint64_t seed, uint64_t env, uint32_t counter);
#ifdef __cplusplus
}
#endif
#endif
