/* mazestep.h — C-ABI of the MI355X batched maze-environment stepper.
 *
 * This is the drop-in boundary for the hot path of kngwyu/mujoco-maze:
 *
 *   reference call (Python, one env)                       replaced by (batched, device)
 *   -----------------------------------------------------  ------------------------------
 *   MazeEnv.__init__       mujoco_maze/maze_env.py:28-233   mz_create   (compiled mz_model in)
 *   MazeEnv.reset          mujoco_maze/maze_env.py:371-382  mz_reset
 *     AntEnv.reset_model   mujoco_maze/ant.py:84-96
 *     PointEnv.reset_model mujoco_maze/point.py:71-81
 *   MazeEnv.step           mujoco_maze/maze_env.py:448-481  mz_step
 *     AntEnv.step          mujoco_maze/ant.py:61-73           (ctrl write + mj_step x frame_skip,
 *     PointEnv.step        mujoco_maze/point.py:44-61          forward reward, ctrl cost)
 *     CollisionDetector.detect  maze_env_utils.py:186-206      (Point wall bounce)
 *     MazeEnv._get_obs     mujoco_maze/maze_env.py:351-369     (obs assembly)
 *     MazeTask.reward / termination  maze_task.py:77-81,110-111,403-407
 *   MujocoEnv.set_state (gym) via set_xy  ant.py:105-108    mz_set_state / mz_get_state
 *   mujoco-py exceptions / warnings                         integer status + mz_last_error,
 *                                                           per-env status words (mz_get_status)
 *
 * All array arguments named *_dev are DEVICE pointers (HBM) owned by the caller
 * (e.g. torch tensors' data_ptr()); the library never frees caller memory.  The
 * handle owns the persistent per-env state.  Calls are asynchronous and ordered
 * on the hipStream_t passed as `stream` (void* here so that this header needs
 * no HIP include; NULL = the default stream).  One host thread per handle.
 *
 * No torch / C++ types appear in any signature.
 */
#ifndef MAZESTEP_H
#define MAZESTEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZ_ABI_VERSION 8

#define MZ_MAX_BODY 24
#define MZ_MAX_JNT 24
#define MZ_MAX_DOF 24
#define MZ_MAX_Q 28
#define MZ_MAX_GEOM 24
#define MZ_MAX_ACT 8
#define MZ_MAX_GRID 12
#define MZ_MAX_SEG 96
#define MZ_MAX_GOAL 8
#define MZ_MAX_OBS 48        /* observation without the top-down view */
#define MZ_VIEW_DIM 75       /* MazeEnv.get_top_down_view: 5 x 5 cells x (walls, chasms, movable blocks), maze_env.py:95 */

/* robot kinds */
#define MZ_ROBOT_POINT 0
#define MZ_ROBOT_ANT 1
#define MZ_ROBOT_SWIMMER 2 /* planar link chains: Swimmer (3 links) and Reacher (2 links; reacher.xml is <mujoco model="swimmer">) */
#define MZ_ROBOT_GENERIC 3 /* a user's AgentModel of any tree topology (agent_model.py:12-41, README.md:127): stepped by walking the compiled tree */

/* joint types (MuJoCo numbering) */
#define MZ_JNT_FREE 0
#define MZ_JNT_BALL 1
#define MZ_JNT_SLIDE 2
#define MZ_JNT_HINGE 3

/* geom types (MuJoCo numbering; pair order is by type) */
#define MZ_GEOM_PLANE 0
#define MZ_GEOM_SPHERE 2
#define MZ_GEOM_CAPSULE 3
#define MZ_GEOM_BOX 6

/* maze cell codes in mz_model.grid (MazeCell, maze_env_utils.py:19-33) */
#define MZ_CELL_EMPTY 0
#define MZ_CELL_BLOCK 1
#define MZ_CELL_CHASM 2
#define MZ_CELL_ROBOT 255

/* reward kinds (mujoco_maze_amd/maze_task.py) */
#define MZ_REWARD_ZERO 0
#define MZ_REWARD_FIRST_MATCH 1
#define MZ_REWARD_NEG_DIST 2
#define MZ_SLOT_AGENT 0
#define MZ_SLOT_OBJECT 1

/* per-env status bits (mz_get_status) */
#define MZ_STATUS_OK 0
#define MZ_STATUS_BAD_STATE 1      /* NaN / |x| > 1e10 in qpos/qvel/qacc (MuJoCo mj_check*) */
#define MZ_STATUS_CONTACT_OVERFLOW 2 /* more simultaneous contacts (or, generic robots, joint-limit rows) than the kernel's buffers */
#define MZ_STATUS_SOLVER_MAXITER 4 /* constraint solver hit its iteration cap */

/* error codes */
#define MZ_OK 0
#define MZ_ERR_ARG (-1)
#define MZ_ERR_HIP (-2)
#define MZ_ERR_UNSUPPORTED (-3)

/* The compiled model: what MazeEnv.__init__ + the MuJoCo model compiler produce
 * in the reference (maze_env.py:97-218 writes MJCF; MuJoCo compiles it).  Built
 * on the host by mujoco_maze_amd/model.py.  Plain doubles/ints; the library
 * converts to fp32 device constants. */
typedef struct mz_model {
  int32_t abi_version;
  int32_t robot; /* MZ_ROBOT_* */
  int32_t nbody, njnt, nq, nv, ngeom, nu;
  int32_t nq_robot, nv_robot; /* leading coordinates the robot's own _get_obs reports and reset_model re-randomises: ant 15/14
                               * (ant.py:75-96), point 3/3 (point.py:63-81), swimmer / reacher ALL nq / nv, movable blocks
                               * included (swimmer.py:50-69) */
  int32_t frame_skip;
  int32_t integrator_rk4; /* 1: RK4 (every reference asset); 0: MuJoCo's default Euler — semi-implicit, implicit in the joint damping
                           * (mj_EulerSkip) — for user robots, stepped by the general engine */
  int32_t collision_predefined; /* swimmer: no dynamic contact pairs */
  int32_t manual_collision;     /* Point: CollisionDetector bounce */
  int32_t max_episode_steps;    /* gym TimeLimit (mujoco_maze/__init__.py:31) */
  int32_t obs_dim;
  int32_t reset_qvel_kind; /* 0 normal*0.1, 1 U[0,1)*0.1, 2 U(-.1,.1) */
  int32_t engine; /* 0: the robot family's specialised kernels where they can step this model, the general engine (csrc/generic_dyn.h:
                   * walks the compiled tree; any joints / geoms / movable bodies this struct can describe) where they cannot — SPIN
                   * plates, user robots, more than three blocks; 1: the general engine whatever the robot (ABI 8) */
  double timestep;
  double gravity[3];
  double density, viscosity;
  double meaninertia;

  /* bodies (0 = world) */
  int32_t body_parent[MZ_MAX_BODY];
  int32_t body_jntadr[MZ_MAX_BODY];
  int32_t body_jntnum[MZ_MAX_BODY];
  int32_t body_dofadr[MZ_MAX_BODY];
  int32_t body_dofnum[MZ_MAX_BODY];
  double body_pos[MZ_MAX_BODY][3];
  double body_quat[MZ_MAX_BODY][4];
  double body_ipos[MZ_MAX_BODY][3];
  double body_inertia[MZ_MAX_BODY][6]; /* xx yy zz xy xz yz about COM, body frame */
  /* MuJoCo's own form of the same tensor: principal moments and the inertial frame's orientation in the body frame
   * (mjModel.body_inertia / body_iquat; ximat = xmat * R(iquat)).  A body with ONE geom takes that geom's frame and moments
   * (a capsule's fromto frame, axial moment third); several geoms: eigen-decomposition of the summed tensor, moments in
   * decreasing order.  The inertia-box fluid model works in this frame. */
  double body_iquat[MZ_MAX_BODY][4];
  double body_pinertia[MZ_MAX_BODY][3];
  double body_mass[MZ_MAX_BODY];
  double body_invweight0[MZ_MAX_BODY][2];

  /* joints */
  int32_t jnt_type[MZ_MAX_JNT];
  int32_t jnt_qposadr[MZ_MAX_JNT];
  int32_t jnt_dofadr[MZ_MAX_JNT];
  int32_t jnt_bodyid[MZ_MAX_JNT];
  int32_t jnt_limited[MZ_MAX_JNT];
  double jnt_pos[MZ_MAX_JNT][3];
  double jnt_axis[MZ_MAX_JNT][3];
  double jnt_range[MZ_MAX_JNT][2];
  double jnt_margin[MZ_MAX_JNT];
  double jnt_solref[MZ_MAX_JNT][2];
  double jnt_solimp[MZ_MAX_JNT][5];
  double jnt_stiffness[MZ_MAX_JNT]; /* hinge / slide spring (MJCF joint stiffness): passive force -k (q - springref); a model with any goes to the general engine */
  double jnt_springref[MZ_MAX_JNT]; /* rest position of that spring in joint coordinates (radians / metres; MJCF springref) */

  /* dofs */
  int32_t dof_bodyid[MZ_MAX_DOF];
  int32_t dof_jntid[MZ_MAX_DOF];
  double dof_armature[MZ_MAX_DOF];
  double dof_damping[MZ_MAX_DOF];
  double dof_invweight0[MZ_MAX_DOF];
  double qpos0[MZ_MAX_Q];

  /* geoms (0 = floor plane; maze wall boxes are implicit in `grid`) */
  int32_t geom_type[MZ_MAX_GEOM];
  int32_t geom_bodyid[MZ_MAX_GEOM];
  int32_t geom_contype[MZ_MAX_GEOM];
  int32_t geom_conaffinity[MZ_MAX_GEOM];
  int32_t geom_condim[MZ_MAX_GEOM];
  double geom_pos[MZ_MAX_GEOM][3];
  double geom_quat[MZ_MAX_GEOM][4];
  double geom_size[MZ_MAX_GEOM][3];
  double geom_friction[MZ_MAX_GEOM][3];
  double geom_solref[MZ_MAX_GEOM][2];
  double geom_solimp[MZ_MAX_GEOM][5];
  double geom_margin[MZ_MAX_GEOM];
  double geom_gap[MZ_MAX_GEOM];
  double geom_rbound[MZ_MAX_GEOM];

  /* actuators: motor on one dof */
  int32_t act_dofid[MZ_MAX_ACT];
  int32_t act_ctrllimited[MZ_MAX_ACT];
  double act_gear[MZ_MAX_ACT];
  double act_ctrlrange[MZ_MAX_ACT][2];
  double act_gainprm[MZ_MAX_ACT];    /* actuator force = gainprm * ctrl + biasprm[0] + biasprm[1] * length + biasprm[2] * velocity, with length =  */
  double act_biasprm[MZ_MAX_ACT][3]; /* gear * q, velocity = gear * qvel of its joint; on the dof: gear * force.  motor: 1 | 0 0 0; <position kp>:
                                      * kp | 0 -kp 0; <velocity kv>: kv | 0 0 -kv.  Anything but a motor is stepped by the general engine */

  /* maze world (maze_env.py:116-152): cell (i,j) centre = (j*s - torso_x, i*s - torso_y) */
  int32_t grid_rows, grid_cols;
  uint8_t grid[MZ_MAX_GRID][MZ_MAX_GRID];
  double maze_scale, torso_x, torso_y;
  double wall_half_xy, wall_half_z, wall_center_z; /* box half sizes / centre height */
  int32_t wall_contype, wall_conaffinity, wall_condim;
  int32_t step_kind; /* the robot's own step (AgentModel.step): 0 = the robot family's; 1 = motors (ant.py:61-73, swimmer.py:37-48: clamped
                      * motors, forward reward, control cost); 2 = the Point's (point.py:44-61: heading / position moved by the action,
                      * velocity clip, no control).  Read by the general engine; a user robot picks one (ABI 8) */
  double wall_friction[3];
  double wall_solref[2];
  double wall_solimp[5];
  double wall_margin, wall_gap;

  /* Point manual collision (maze_env_utils.py:151-184; maze_env.py:36,457-464) */
  int32_t nseg, pad2;
  double seg[MZ_MAX_SEG][4]; /* x1 y1 x2 y2 */
  double restitution;
  double velocity_limit; /* point.py:33,56 */

  /* task (maze_task.py) */
  int32_t ngoal;
  int32_t reward_kind, reward_slot, reward_binary, term_slot;
  int32_t goal_dim[MZ_MAX_GOAL];
  double goal_pos[MZ_MAX_GOAL][3];
  double goal_threshold[MZ_MAX_GOAL];
  double goal_reward_scale[MZ_MAX_GOAL];
  double penalty, task_scale, inner_reward_scaling;
  double forward_reward_weight, ctrl_cost_weight; /* ant.py:47-53 */

  /* movable XY blocks (maze_env.py:563-660: body + slide-x + slide-y + box geom appended after the
   * robot); observed as body xpos at obs[3:3+3*nblock] when observe_blocks (maze_env.py:364-368) */
  int32_t nblock, observe_blocks;
  int32_t block_bodyid[4];
  int32_t block_geomid[4];

  /* object balls (Billiard; maze_env.py:167-191, 489-536: body at z = 0 with slide-x, slide-y and a z hinge, sphere geom
   * of radius r at height r); observed as body xpos BEFORE the blocks when observe_balls (maze_env.py:360-363) */
  int32_t nball, observe_balls;
  int32_t ball_bodyid[4];
  int32_t ball_geomid[4];

  /* elevated mazes (Fall / MultiFall, maze_env.py:102-107,124-137): under every cell that is not a CHASM stands a platform
   * box of the wall's footprint from z = 0 to z = height_offset (centre wall_half_z, half height wall_half_z); the walls
   * stand on top of it (wall_center_z = wall_half_z + height_offset) and the robot's torso starts 0.75 above it.
   * Movable blocks of such mazes slide along z as well (limited joints; block_bodyid / the joint arrays describe them). */
  int32_t elevated;
  /* MazeTask.TOP_DOWN_VIEW (maze_env.py:54,351-369): the robot-centred 5 x 5 x 3 occupancy view of get_top_down_view
   * (maze_env.py:262-349), flattened row-major, sits between the robot's observation and the trailing t * 0.001; obs_dim
   * counts its MZ_VIEW_DIM entries */
  int32_t top_down_view;
  double height_offset;
} mz_model;

typedef struct mz_handle mz_handle;

int mz_abi_version(void);
uint64_t mz_model_sizeof(void); /* host-side struct layout check for FFI users */

/* Create a batch of `num_envs` environments on HIP device `device`.
 * Returns NULL on failure with a message in err (if non-NULL). */
mz_handle* mz_create(const mz_model* model, int32_t num_envs, int32_t device, char* err, int32_t errlen);
void mz_destroy(mz_handle* h);
const char* mz_last_error(const mz_handle* h);

/* accessors: the value, or MZ_ERR_ARG for a NULL handle */
int32_t mz_num_envs(const mz_handle* h);
int32_t mz_obs_dim(const mz_handle* h);
int32_t mz_nq(const mz_handle* h);
int32_t mz_nv(const mz_handle* h);
int32_t mz_nu(const mz_handle* h);

/* Tunables: "auto_reset" (0/1), "seed", "env_index_offset" (global slot of local env 0 in a
 * sharded run: keys the reset RNG), "solver_iterations", "solver_tolerance", "solver_rtol",
 * "ls_iterations" (Ant solver: evaluations of the exact line search), "ls_fast_iterations" / "ls_fast" (the first n Newton
 * iterations of a solve search the line with at most that many evaluations; defaults 5 / 0: unit steps; "ls_fast_iterations" also
 * sets the Point's float64 solvers; 0 = MuJoCo's search in every iteration, monotone on any maze at ~7 % of the Ant kernel's time:
 * what the host mirror sets for custom tasks, whose stiffness nobody has soaked — the registered mazes are), "lanes_per_env" (8/16/32/64 lanes of a wavefront per environment; defaults: plain Ant 16,
 * Ant with one movable block 32 — 16 for batches beyond 2048 envs, where 32 would mean more waves than the device has SIMDs —, with more blocks / a three-slide block / the ball 64; Point 16 while that leaves every wave a SIMD of its own (up to 4096 envs), else 32; with one block or the ball 32, with two or three blocks 64;
 * Swimmer / Reacher 4, fixed), "waves_per_block" (1/2/4, Ant), "waves_per_simd" (plain Ant at 16 lanes: 1 = the one-wave
 * kernel, 2 = the kernel held to 256 registers so that two waves share a SIMD, 0 = default: by the wave count of the
 * launch — more waves than SIMDs takes the second), "profile_phases" (0/1: instrumented Ant kernel, see
 * mz_read_phase_cycles), "time_kernels" (n > 0: ring of n HIP event pairs around the step kernel, see
 * mz_last_kernel_ms), "time_kernels_stride" (k >= 1, default 1: only every k-th mz_step carries the pair — a pair costs the queue
 * ~5 us per launch, a tenth of a Point or Swimmer step; bench.py samples every 8th). Returns MZ_OK, MZ_ERR_ARG, or MZ_ERR_UNSUPPORTED for an Ant-only key on another robot. */
int32_t mz_set_option(mz_handle* h, const char* key, double value);

/* What the next mz_step launches (ABI 8).  The kernel instantiation a handle steps with is chosen from the robot, its maze's
 * movable bodies, the BATCH SIZE and the device's compute-unit count (lanes per env, one or two waves per SIMD: see
 * "lanes_per_env" / "waves_per_simd" above), and the instantiations agree to fp32 round-off, not bit for bit: the same seed and
 * states give round-off-different trajectories at different num_envs, on shards of unequal size or on devices with another
 * CU count.  Keys: "engine" (0 = the robot family's specialised kernel, 1 = the general engine of mz_model.engine),
 * "lanes_per_env", "waves_per_simd", "device_simds" (4 x compute units), "ls_fast_iterations".  Returns MZ_OK or MZ_ERR_ARG. */
int32_t mz_get_info(const mz_handle* h, const char* key, double* value);

/* Auto-reset observation convention.  With option "auto_reset" = 1 an env whose step ended its episode (done != 0) is
 * re-seeded inside the same kernel; mz_step then returns, for that env, the reward / done / goal index of the terminal step
 * and in obs_dev the FIRST observation of the new episode (t = 0) — what a policy must act on next.  The terminal
 * observation (the one the reference's MazeEnv.step returns, maze_env.py:474-481) is written to row i of the buffer bound
 * here ([N, obs_dim] fp32, device, caller-owned; rows of envs that did not finish are left untouched); NULL unbinds.
 * Without auto-reset obs_dev always holds the step's own observation and this buffer is never written. */
int32_t mz_bind_final_obs(mz_handle* h, float* final_obs_dev);

/* Packed per-env record for sharded runs (the one collective of the path is the all-gather of this tensor).
 * When bound ([N, obs_dim + 2] fp32, device, caller-owned; NULL unbinds), every mz_step also writes row i =
 * obs_dev row i (obs_dim floats, i.e. under auto-reset the first observation of the new episode) | reward | done as float,
 * from inside the step kernel (Ant) or with one extra pack launch (Point / Swimmer / Reacher, top-down-view tasks), so the
 * caller can hand the buffer to the collective without copy launches of its own.  No reference counterpart (the
 * reference steps one env per process: mujoco_maze/maze_env.py:448-481 returns the three values separately). */
int32_t mz_bind_record(mz_handle* h, float* record_dev);

/* Goal resampling (MazeEnv.reset calls MazeTask.sample_goals() and, when it returns True, moves the goal markers:
 * mujoco_maze/maze_env.py:374-376; MazeTask.sample_goals: maze_task.py:66-67).  Replaces the goal table the task predicate of
 * every later mz_step / mz_debug_task_eval evaluates: ngoal <= MZ_MAX_GOAL rows of HOST arrays — pos [ngoal][3] (z ignored where
 * dim = 2), threshold [ngoal], reward_scale [ngoal], dim [ngoal] (2 or 3) — float64, the reference's own values
 * (maze_task.py:26-47).  Work already queued on `stream` is finished first (the call synchronises on it), so a step launched
 * before the call is judged on the old goals and every step after it on the new ones.  Reward kind, slots, PENALTY stay
 * those of the model.  Returns MZ_OK or MZ_ERR_ARG. */
int32_t mz_set_goals(mz_handle* h, int32_t ngoal, const double* pos, const double* threshold, const double* reward_scale,
                     const int32_t* dim, void* stream);

/* Per-env goal positions (ABI 8).  The reference gives every env its own task object and resamples its goals at EVERY episode reset
 * (maze_env.py:374-376); a batch that shares one goal table can only do so at a full reset.  goal_pos_dev: DEVICE pointer,
 * [num_envs][MZ_MAX_GOAL][3] float64, caller-owned and caller-updated between steps (the host mirror rewrites the rows of the envs
 * whose episode just ended); NULL unbinds.  Thresholds, reward scales, dims and the goal count stay those of the shared table
 * (mz_set_goals / the model).  Synchronises on `stream` first.  Returns MZ_OK, MZ_ERR_ARG or MZ_ERR_HIP. */
int32_t mz_bind_env_goals(mz_handle* h, const double* goal_pos_dev, void* stream);

/* reset(): envs with mask_dev[i] != 0 (all when NULL) get t = 0 and a fresh state
 * from the reference's reset distribution (counter-based RNG keyed by seed and
 * env slot; streams differ from numpy's — distributional parity only).
 * obs_dev [N, obs_dim] may be NULL. */
int32_t mz_reset(mz_handle* h, const uint8_t* mask_dev, uint64_t seed, float* obs_dev, void* stream);

/* State injection / read-back (row-major [N, nq], [N, nv], [N, nv], [N]).
 * Any pointer may be NULL to skip that field.  warmstart = the constraint solver's starting guess carried between steps
 * (MuJoCo's qacc_warmstart): the Ant and the general engine keep the last evaluation's qacc; the bare Point keeps its constraint part
 * qacc - qacc_smooth (read back by mz_get_state; mz_set_state with a new qpos / qvel clears it — a guess only, results agree to the
 * solver tolerance whatever it holds); the Swimmer / Reacher and the Point with movable bodies keep none (reads 0, writes ignored). */
int32_t mz_set_state(mz_handle* h, const float* qpos_dev, const float* qvel_dev, const float* warmstart_dev,
                     const int32_t* t_dev, void* stream);
int32_t mz_get_state(mz_handle* h, float* qpos_dev, float* qvel_dev, float* warmstart_dev, int32_t* t_dev,
                     void* stream);

/* One MazeEnv.step for every env.
 *  actions_dev [N, nu] fp32 row-major
 *  obs_dev     [N, obs_dim]   the step's observation (under auto-reset: see mz_bind_final_obs)
 *  reward_dev  [N]            inner_reward_scaling * inner + task reward
 *  done_dev    [N] u8         bit0 = task termination, bit1 = TimeLimit truncation
 *  goal_idx_dev[N] i32        first matching goal (list order) or -1 (nullable): the goal whose reward_scale the reward is
 *                             (maze_task.py:403-407); for zero / distance rewards the first goal that terminates
 *  info_dev    [N, 4]         position x, y, reward_forward, reward_ctrl (nullable) */
int32_t mz_step(mz_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
                int32_t* goal_idx_dev, float* info_dev, void* stream);

/* Per-env status words accumulated since the last call (then cleared). [N] i32. */
int32_t mz_get_status(mz_handle* h, int32_t* status_dev, void* stream);

/* Diagnostics for parity tests: one forward-dynamics evaluation at the current
 * state with ctrl = actions; writes qacc [N, nv] and ncon/nefc [N, 2]. */
int32_t mz_debug_forward(mz_handle* h, const float* actions_dev, float* qacc_dev, int32_t* counts_dev, void* stream);

/* Parity-test entries for the two places where the reference's float64 DECISIONS must be reproduced bit for bit.
 * mz_debug_task_eval: MazeTask.reward / termination / first matching goal (maze_task.py:43-47,77-81,110-111,403-407) on
 *   n_rows observation rows [n_rows, obs_dim] fp32 — the very task_eval_dev instance the handle's step kernel runs:
 *   reward_dev[n_rows] = task reward only (no inner reward), done_dev[n_rows] u8 = termination, goal_idx_dev (nullable).
 *   With per-env goals bound (mz_bind_env_goals) row r < num_envs is judged with env r's goals, later rows with the shared table.
 * mz_debug_detect: CollisionDetector.detect + the bounce / give-up rule of MazeEnv.step (maze_env_utils.py:96-123,186-206;
 *   maze_env.py:457-464) on n_rows float64 moves old_xy -> new_xy ([n_rows, 2] each): hit_dev[n_rows] = 0 no wall hit,
 *   1 bounced, 2 gave up (position restored), -1 collinear move (the reference raises ZeroDivisionError);
 *   point_dev[n_rows, 2] (nullable) = first collision point, final_xy_dev[n_rows, 2] = position after the rule.  Point only. */
int32_t mz_debug_task_eval(mz_handle* h, int32_t n_rows, const float* obs_dev, float* reward_dev, uint8_t* done_dev,
                           int32_t* goal_idx_dev, void* stream);
int32_t mz_debug_detect(mz_handle* h, int32_t n_rows, const double* old_xy_dev, const double* new_xy_dev, int32_t* hit_dev,
                        double* point_dev, double* final_xy_dev, void* stream);

/* Kernel-internal phase timers (option "profile_phases" = 1 selects the instrumented kernel build):
 * 16 accumulators over the window since the last mz_read_wave_phase_cycles (which clears the slots this call sums; this call clears
 * nothing) — slots 0..12 shader cycles per phase (ids: tools/phase_profile.py), 13 the most loaded workgroup's total over the window,
 * 14 the sum of squared per-workgroup window totals (units of 256 cycles), 15 Newton iterations — taken from the first env group of
 * every workgroup and reduced on the host from the per-workgroup slots (the kernels issue no same-address atomics: they disturbed
 * the waves still running); out16_host is HOST memory. */
int32_t mz_read_phase_cycles(mz_handle* h, uint64_t* out16_host);
/* Same instrumented build: total shader cycles of every workgroup (one wavefront = 64 / lanes_per_env envs) accumulated
 * since the last call, n_host <= num_envs entries into HOST memory; cleared on read.  For load-balance analysis. */
int32_t mz_read_wave_cycles(mz_handle* h, uint64_t* out_host, int32_t n_host);
/* Same build: the 16 phase slots of every workgroup ([n_host][16] into HOST memory, accumulated since the last call, cleared on
 * read) — which phase makes the slowest waves slow (Ant kernels). */
int32_t mz_read_wave_phase_cycles(mz_handle* h, uint64_t* out_host, int32_t n_host);

/* Average duration (ms) of the step kernel over the launches recorded since "time_kernels" was set (HIP events on
 * the stream each mz_step was given; synchronises on them); returns < 0 if not available. */
double mz_last_kernel_ms(const mz_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* MAZESTEP_H */
