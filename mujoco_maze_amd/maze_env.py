"""Batched maze environment on the MI355X and the reference-shaped single env.

`VecMazeEnv` is the drop-in for the reference's hot path (`MazeEnv.step/reset`,
mujoco_maze/maze_env.py:371-382,448-481) over `num_envs` environments in
lock-step: Python -> ctypes -> C-ABI (include/mazestep.h) -> HIP kernels.  Tensors
stay on the GPU (torch is only the owner of device memory and streams).

`MazeEnv` keeps the reference constructor signature (maze_env.py:28-44) and the
`reset() -> (obs, info)` / `step(a) -> (obs, reward, done, info)` shapes for one
environment (numpy in / numpy out, float64 like the reference), implemented as a
`VecMazeEnv` of size 1.
"""
import ctypes as C
from typing import Optional, Tuple, Type

import numpy as np

from mujoco_maze_amd import _capi
from mujoco_maze_amd.agent_model import AgentModel
from mujoco_maze_amd.maze_task import MazeTask
from mujoco_maze_amd.model import CompiledModel, compile_model


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency)."""

    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32

    def sample(self, rng: Optional[np.random.Generator] = None):
        rng = rng or np.random.default_rng()
        return rng.uniform(self.low, self.high).astype(np.float32)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class WrappedRobot:
    """`MazeEnv.wrapped_env` of the reference (maze_env.py:218): the robot object user code reaches into for
    `get_xy()` / `set_xy(xy)` / `get_ori()` (agent_model.py:35-41, ant.py:98-111, point.py:83-92, swimmer.py:70-76).  The
    simulators themselves live in the HIP kernels, so this object is a view on the device batch through
    mz_get_state / mz_set_state; class-level knobs of the robot class (FILE, ORI_IND, RADIUS, ...) read through.

    On a `VecMazeEnv` the methods are batched (device tensors [N, 2] / [N]); on the single-env `MazeEnv` they take and return
    float64 numpy values like the reference."""

    def __init__(self, vec: "VecMazeEnv", model_cls, single: bool = False) -> None:
        self._vec, self._cls, self._single = vec, model_cls, single

    def __getattr__(self, name):  # class attributes of the robot class (ORI_IND, RADIUS, MANUAL_COLLISION, ...)
        # (only reached for names the instance does not have: private names and a half-built instance — copy / pickle probe
        # __setstate__ & co. before __init__ has run — must not recurse through the missing `_cls`)
        if name.startswith("_") or "_cls" not in self.__dict__:
            raise AttributeError(name)
        return getattr(self._cls, name)

    def get_xy(self):
        """qpos[:2] (ant.py:110-111, point.py:83-84, swimmer.py:75-76): a fresh copy."""
        xy = self._vec.get_state()[0][:, :2].clone()
        return xy[0].double().cpu().numpy() if self._single else xy

    def set_xy(self, xy) -> None:
        """qpos[:2] = xy, everything else (qvel, qacc_warmstart, t) untouched: ant.py:105-108, point.py:86-89 call
        `set_state(qpos, qvel)` with the current qvel; the warm start is not part of mujoco-py's state and survives."""
        torch, vec = self._vec._torch, self._vec
        qpos = vec.get_state()[0]
        new = torch.as_tensor(np.asarray(xy, dtype=np.float32) if not torch.is_tensor(xy) else xy, device=vec.device).to(torch.float32)
        if self._single:
            new = new.reshape(1, 2)
        if tuple(new.shape) != (vec.num_envs, 2):
            raise ValueError(f"xy must have shape {(vec.num_envs, 2)}, got {tuple(new.shape)}")
        qpos[:, :2] = new
        vec.set_state(qpos=qpos)

    def get_ori(self):
        """Heading: the Point's qpos[ORI_IND] (point.py:91-92); the Ant's torso x axis rotated by its quaternion
        qpos[ORI_IND:ORI_IND+4] and projected on the plane, arctan2(y, x) (ant.py:98-103, q_mult / q_inv :26-35)."""
        ind = getattr(self._cls, "ORI_IND", None)
        if ind is None:
            raise AttributeError(f"{self._cls.__name__} has no ORI_IND: the reference defines get_ori for the Point and the Ant only")
        qpos = self._vec.get_state()[0].double()
        torch = self._vec._torch
        if getattr(self._cls, "ROBOT", None) == "ant":
            w, x, y, z = (qpos[:, ind + k] for k in range(4))
            # rot * (0, 1, 0, 0) * conj(rot), components 1..2, written out (ant.py:26-35 multiply without normalising)
            ox = w * w + x * x - y * y - z * z
            oy = 2.0 * (x * y + w * z)
            ori = torch.atan2(oy, ox)
        elif getattr(self._cls, "ROBOT", None) == "point":
            ori = qpos[:, ind].clone()
        else:
            # a user robot: qpos[ORI_IND] is a yaw angle only if its class says so (a free-joint robot's entry there is a
            # quaternion component) — it supplies the rule as a static `ori_from_qpos(qpos [N, nq]) -> [N]`
            fn = getattr(self._cls, "ori_from_qpos", None)
            if fn is None:
                raise AttributeError(f"{self._cls.__name__}: get_ori is defined for the Point (qpos[ORI_IND]) and the Ant (torso quaternion); "
                                     "a user robot provides `ori_from_qpos(qpos)`")
            ori = fn(qpos)
        return float(ori[0].item()) if self._single else ori


def _is_registered(model_cls, task_cls, scale) -> bool:
    """Is (robot class, task class, maze scale) one of the 145 registered combinations (mujoco_maze/__init__.py:22-78)?"""
    import mujoco_maze_amd as mm

    for spec in mm.REGISTRY.values():
        kw = spec.kwargs
        if kw["model_cls"] is model_cls and kw["maze_task"] is task_cls and float(kw["maze_size_scaling"]) == float(scale):
            return True
    return False


class VecMazeEnv:
    def __init__(self, model_cls: Type[AgentModel], maze_task: Type[MazeTask] = MazeTask, num_envs: int = 1,
                 maze_height: float = 0.5, maze_size_scaling: float = 4.0, inner_reward_scaling: float = 1.0,
                 restitution_coef: float = 0.8, task_kwargs: Optional[dict] = None, device=None,
                 max_episode_steps: int = 1000, auto_reset: bool = False, seed: int = 0, include_position: bool = True,
                 **kwargs) -> None:
        import torch  # device memory + streams only

        self._torch = torch
        self.num_envs = int(num_envs)
        self._task = maze_task(maze_size_scaling, **(task_kwargs or {}))
        robot = getattr(model_cls, "ROBOT", None)
        if robot not in ("ant", "point", "swimmer", "reacher", "generic"):
            raise NotImplementedError(f"robot {model_cls.__name__}: set ROBOT to one of the built-in families or to \"generic\" (any tree "
                                      "topology, stepped from its MJCF: AgentModel.FILE)")
        if robot == "generic":  # the user's own robot (agent_model.py:12-41, README.md:127): its MJCF, frame_skip, reset distribution
            kwargs.setdefault("robot_xml", getattr(model_cls, "FILE", None))
            # round 6: STEP = "motors" (ant.py:61-73, the default) or "point" (point.py:44-61: the action moves heading and position,
            # velocities clipped to VELOCITY_LIMITS); MANUAL_COLLISION / RADIUS as for the built-in Point
            gen_kw = dict(frame_skip=int(getattr(model_cls, "FRAME_SKIP", 1)), reset_qvel=getattr(model_cls, "RESET_QVEL", "normal"),
                          step=getattr(model_cls, "STEP", "motors"), velocity_limit=getattr(model_cls, "VELOCITY_LIMITS", None))
        else:
            gen_kw = {}
        # engine="general": step this env on the general engine (csrc/generic_dyn.h) whatever the robot — the second, independent
        # device implementation the specialised kernels are cross-checked with; "auto" (default): only where they cannot step it
        gen_kw["engine"] = kwargs.pop("engine", "auto")
        has_xml = kwargs.get("robot_xml") is not None
        self.model: CompiledModel = compile_model(
            robot, self._task, maze_size_scaling, inner_reward_scaling=inner_reward_scaling,
            restitution_coef=restitution_coef, maze_height=maze_height, max_episode_steps=max_episode_steps,
            forward_reward_weight=kwargs.pop("forward_reward_weight", 1.0), ctrl_cost_weight=kwargs.pop("ctrl_cost_weight", 1e-4),
            manual_collision=getattr(model_cls, "MANUAL_COLLISION", False), radius=getattr(model_cls, "RADIUS", None),
            robot_xml=kwargs.pop("robot_xml", None), **gen_kw)
        self.wrapped_cls = model_cls
        self.wrapped_env = WrappedRobot(self, model_cls)
        from mujoco_maze_amd.model import device_unsupported_reason

        why = device_unsupported_reason(self.model)
        if why:
            raise NotImplementedError(why)
        if not torch.cuda.is_available():
            raise _capi.MazeStepError("no GPU visible: mujoco_maze_amd steps environments on an MI355X only (no CPU fallback)")
        self._lib = _capi.load()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        err = C.create_string_buffer(256)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._h = self._lib.mz_create(C.byref(self.model.c), self.num_envs, idx, err, 256)
        if not self._h:
            raise _capi.MazeStepError(f"mz_create failed: {err.value.decode()}")
        m = self.model.c
        self.nq, self.nv, self.nu, self.obs_dim = m.nq, m.nv, m.nu, m.obs_dim
        # The Ant's and the Point's Newton solvers take unit steps in the first iterations of a solve (ant_newton_rows.h).  The registered
        # mazes are soaked with them (1e8 env-steps, no solve at the iteration cap); a CUSTOM task, maze or robot variant is not, so
        # there every iteration searches the line exactly (ls_fast_iterations = 0: MuJoCo's monotone iteration on any maze, ~7 % of the
        # Ant kernel's time; a guarded unit step was measured and costs the same: profiles/r06/unit_guard_ab.txt).
        # set_option("ls_fast_iterations", 5) brings the unit steps back.  The general engine and the chain kernels search the line in
        # every iteration / solve in closed form.
        from mujoco_maze_amd.model import needs_general_engine

        self.custom_task = not _is_registered(model_cls, maze_task, maze_size_scaling) or has_xml
        if self.custom_task and robot in ("ant", "point") and not needs_general_engine(self.model):
            _capi.check(self._lib, self._h, self._lib.mz_set_option(self._h, b"ls_fast_iterations", 0.0), "mz_set_option(ls_fast_iterations)")
        n, dev = self.num_envs, self.device
        self._obs = torch.empty((n, self.obs_dim), dtype=torch.float32, device=dev)
        self._reward = torch.empty(n, dtype=torch.float32, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._goal = torch.empty(n, dtype=torch.int32, device=dev)
        self._info = torch.empty((n, 4), dtype=torch.float32, device=dev)
        self._seed = int(seed)
        self._host_rewards = not self.model.device_rewards
        self._final_obs = None
        self._host_mask = None
        self._record = None
        self._resampled_goals = False
        self._warned_goal_resampling = False
        self._env_goals = None   # float64 [N, MZ_MAX_GOAL, 3] once the task's sample_goals() said the goals move (per-env goal positions)
        self._goal_pool = None
        self._goal_gen = None
        self.set_auto_reset(auto_reset)
        lo = np.array([m.act_ctrlrange[a][0] for a in range(m.nu)], dtype=np.float32)
        hi = np.array([m.act_ctrlrange[a][1] for a in range(m.nu)], dtype=np.float32)
        self.action_space = Box(lo, hi)
        high = np.full(self.obs_dim, np.inf, dtype=np.float32)
        low = -high
        low[0], high[0], low[1], high[1] = self.model.world.xy_limits()
        self.observation_space = Box(low, high)

    # -- plumbing ----------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def set_option(self, key: str, value: float) -> None:
        if key == "auto_reset":
            return self.set_auto_reset(value != 0)
        _capi.check(self._lib, self._h, self._lib.mz_set_option(self._h, key.encode(), float(value)), f"mz_set_option({key})")

    def launch_info(self) -> dict:
        """What the next step launches (mz_get_info): the engine (0 = the robot family's specialised kernel, 1 = the general engine),
        lanes per env, waves per SIMD, the device's SIMD count.  The instantiation depends on the batch size and the device; its
        results agree across instantiations to fp32 round-off, not bit for bit (include/mazestep.h)."""
        out = {}
        for key in ("engine", "lanes_per_env", "waves_per_simd", "device_simds", "ls_fast_iterations"):
            v = C.c_double(0.0)
            _capi.check(self._lib, self._h, self._lib.mz_get_info(self._h, key.encode(), C.byref(v)), f"mz_get_info({key})")
            out[key] = int(v.value)
        return out

    def set_auto_reset(self, on: bool) -> None:
        """Auto-reset follows the vector-env convention: an env whose step ended its episode returns the reward / done of
        that terminal step, the FIRST observation of the next episode in `obs`, and its terminal observation in
        `info["final_observation"]` (rows where done != 0).  Tasks whose reward()/termination() are Python overrides are
        judged on the host, so for them the device's own verdict must not reset anything: the kernel's auto-reset stays
        off and the host issues the masked reset after it has evaluated the task."""
        self._auto_reset = bool(on)
        if self._auto_reset and self._final_obs is None:
            self._final_obs = self._torch.zeros((self.num_envs, self.obs_dim), dtype=self._torch.float32, device=self.device)
        device_side = self._auto_reset and not self._host_rewards
        _capi.check(self._lib, self._h, self._lib.mz_bind_final_obs(self._h, _ptr(self._final_obs) if device_side else None), "mz_bind_final_obs")
        _capi.check(self._lib, self._h, self._lib.mz_set_option(self._h, b"auto_reset", 1.0 if device_side else 0.0), "mz_set_option(auto_reset)")

    def bind_record(self, record) -> None:
        """Bind (or, with None, unbind) a float32 [N, obs_dim + 2] device tensor that every step fills with the packed record
        obs | reward | done — the send buffer of the sharded run's all-gather (mz_bind_record); the caller keeps it alive."""
        if record is not None:
            if self._host_rewards:
                raise ValueError("this task's reward()/termination() are Python overrides judged on the host after the kernel: the "
                                 "kernel-written record would carry the device's built-in verdict.  Pack obs / reward / done after "
                                 "step() instead (sharding.RecordGatherer.start(obs, reward, done))")
            if tuple(record.shape) != (self.num_envs, self.obs_dim + 2) or record.dtype != self._torch.float32 or not record.is_contiguous() \
                    or record.device != self.device:
                raise ValueError(f"record must be a contiguous float32 tensor of shape {(self.num_envs, self.obs_dim + 2)} on {self.device}")
        self._record = record
        _capi.check(self._lib, self._h, self._lib.mz_bind_record(self._h, _ptr(record)), "mz_bind_record")

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._torch.cuda.synchronize(self.device)
            self._lib.mz_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- API ---------------------------------------------------------------
    def get_ori(self):
        return self.wrapped_env.get_ori()  # maze_env.py:231-232

    def set_goals(self, goals=None) -> None:
        """Upload the task's goal list (default: `self._task.goals`) to the device: position, threshold, reward scale and
        dimension of every MazeGoal replace the table the step kernel's predicate reads (mz_set_goals).  The counterpart of
        the reference's `set_marker()` after `sample_goals()` (maze_env.py:374-376, 384-387)."""
        goals = self._task.goals if goals is None else goals
        n = len(goals)
        if n > 8:
            raise ValueError("too many goals (MZ_MAX_GOAL = 8)")
        pos = np.zeros((max(n, 1), 3), np.float64)
        for i, g in enumerate(goals):
            pos[i, : g.dim] = np.asarray(g.pos, np.float64)[: g.dim]
        thr = np.array([g.threshold for g in goals] or [0.0], np.float64)
        rs = np.array([g.reward_scale for g in goals] or [0.0], np.float64)
        dim = np.array([g.dim for g in goals] or [2], np.int32)
        rc = self._lib.mz_set_goals(self._h, n, pos.ctypes.data_as(C.c_void_p), thr.ctypes.data_as(C.c_void_p),
                                    rs.ctypes.data_as(C.c_void_p), dim.ctypes.data_as(C.c_void_p), self._stream())
        _capi.check(self._lib, self._h, rc, "mz_set_goals")
        # the host copy of the model follows (tools that judge with `env.model` must see the new goals; model.py fills the same fields)
        m = self.model.c
        m.ngoal = n
        for i in range(n):
            for k in range(3):
                m.goal_pos[i][k] = float(pos[i, k])
            m.goal_threshold[i], m.goal_reward_scale[i], m.goal_dim[i] = float(thr[i]), float(rs[i]), int(dim[i])

    @property
    def env_goals(self):
        """float64 [N, MZ_MAX_GOAL, 3] device tensor of per-env goal positions (None while the batch shares one goal table): the
        tensor the step kernel's goal predicate reads (mz_bind_env_goals).  Writing into it moves the goals of single envs."""
        return self._env_goals

    def _draw_goal_pool(self) -> None:
        """Per-env goals (maze_env.py:374-376: the reference calls `sample_goals()` at every episode reset of every env, each env its own
        task object).  Here one task object serves the batch, so a full reset() draws N goal tables from it — env i starts with draw i —
        and keeps the N draws as the pool from which the episodes that end later (device auto-reset, masked reset) take their next
        goals, picked by a device-side RNG: no host round trip in the step loop.  Positions only; number, thresholds, reward scales
        and dimensions of the goals are those of the shared table."""
        torch, n = self._torch, self.num_envs
        first = [np.asarray(g.pos, np.float64).copy() for g in self._task.goals]
        rows = np.zeros((n, 8, 3), np.float64)
        for r in range(n):
            if r and not self._task.sample_goals():
                raise ValueError(f"{type(self._task).__name__}.sample_goals() returned True, then False: it must keep resampling")
            if len(self._task.goals) != len(first) or len(first) > 8:
                raise ValueError("sample_goals() must keep the number of goals (and at most MZ_MAX_GOAL = 8 of them)")
            for i, g in enumerate(self._task.goals):
                rows[r, i, : g.dim] = np.asarray(g.pos, np.float64)[: g.dim]
        for g, p0 in zip(self._task.goals, first):  # the task object (and the uploaded shared table) keep the FIRST draw
            g.pos = p0
        self._goal_pool = torch.as_tensor(rows, device=self.device)
        if self._env_goals is None:
            self._env_goals = self._goal_pool.clone()
            self._goal_gen = torch.Generator(device=self.device)
            rc = self._lib.mz_bind_env_goals(self._h, _ptr(self._env_goals), self._stream())
            _capi.check(self._lib, self._h, rc, "mz_bind_env_goals")
        else:
            self._env_goals.copy_(self._goal_pool)
        self._goal_gen.manual_seed(self._seed)

    def _resample_env_goals(self, which) -> None:
        """Envs in `which` (bool [N], device) take a random member of the pool as their goals; the others keep theirs."""
        torch = self._torch
        idx = torch.randint(0, self.num_envs, (self.num_envs,), device=self.device, generator=self._goal_gen)
        torch.where(which[:, None, None], self._goal_pool[idx], self._env_goals, out=self._env_goals)

    def _sync_goals_across_ranks(self) -> None:
        """Sharded runs (sharding.py): every rank called sample_goals() on its own unsynchronised RNG — rank 0's goals win, so that
        the batch really has ONE goal table (positions only: thresholds / reward scales are class constants of the task)."""
        # Only where the caller SAID the env is one shard of a node-wide batch (sharding.ShardedVecMazeEnv sets goal_sync_group):
        # a collective hidden behind dist.is_initialized() would deadlock a trainer whose ranks reset at different times
        group = getattr(self, "goal_sync_group", None)
        if group is None:
            return
        import torch.distributed as dist

        grp = None if group == "world" else group
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(grp) > 1):
            return
        torch = self._torch
        buf = torch.zeros((8, 3), dtype=torch.float64, device=self.device if dist.get_backend(grp) == "nccl" else "cpu")
        for i, g in enumerate(self._task.goals[:8]):
            buf[i, : g.dim] = torch.as_tensor(np.asarray(g.pos, np.float64)[: g.dim])
        dist.broadcast(buf, src=dist.get_global_rank(grp, 0) if grp is not None else 0, group=grp)
        host = buf.cpu().numpy()
        for i, g in enumerate(self._task.goals[:8]):
            g.pos = host[i, : g.dim].copy()

    def reset(self, mask=None, seed: Optional[int] = None):
        """Reset all (or the masked) envs; returns the observation tensor [N, obs_dim] on the GPU.

        A full reset (mask None) first asks the task for new goals — `MazeTask.sample_goals()`, maze_env.py:374-376 — and when it
        says they changed the batch moves to PER-ENV goal positions (`env_goals`, _draw_goal_pool): every env its own draw, and
        every later episode start (masked reset, device auto-reset) a new one for that env, as the reference's one-task-per-env
        does.  Tasks whose goals never move keep the one shared table."""
        if seed is not None:
            self._seed = int(seed)
        if mask is None and self._task.sample_goals():
            self._resampled_goals = True
            self._sync_goals_across_ranks()
            self.set_goals()  # the shared table follows the draw (thresholds, reward scales, dims; the host copy of the model)
            if not self._host_rewards:
                self._draw_goal_pool()  # every env its own draw of sample_goals(); the step kernel reads the per-env table from here on
            elif self._auto_reset and not self._warned_goal_resampling:
                import warnings

                self._warned_goal_resampling = True
                warnings.warn(f"{type(self._task).__name__}.sample_goals() returned True, auto_reset is on and the task's reward()/termination() are "
                              "Python overrides judged on the host from ONE task object: the reference resamples goals at EVERY episode reset "
                              "(maze_env.py:374-376), here they change at full reset() calls only")
        elif mask is not None and self._env_goals is not None:
            mkb = self._torch.as_tensor(mask, device=self.device).to(self._torch.bool)
            self._resample_env_goals(mkb)  # the masked envs start a new episode: new goals for them (maze_env.py:374-376), the others keep theirs
        mk = None
        if mask is not None:
            mk = self._torch.as_tensor(mask, device=self.device).to(self._torch.uint8).contiguous()
        rc = self._lib.mz_reset(self._h, _ptr(mk), C.c_uint64(self._seed), _ptr(self._obs), self._stream())
        _capi.check(self._lib, self._h, rc, "mz_reset")
        self._seed += 1
        return self._obs

    def step(self, actions):
        """actions: float32 tensor [N, nu] on the same GPU.  Returns (obs, reward, done, info) tensors;
        `done` is uint8 with bit0 = task termination, bit1 = TimeLimit truncation."""
        torch = self._torch
        a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions, dtype=np.float32), device=self.device)
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(a.shape) != (self.num_envs, self.nu):
            raise ValueError(f"actions must have shape {(self.num_envs, self.nu)}, got {tuple(a.shape)}")
        rc = self._lib.mz_step(self._h, _ptr(a), _ptr(self._obs), _ptr(self._reward), _ptr(self._done), _ptr(self._goal),
                               _ptr(self._info), self._stream())
        _capi.check(self._lib, self._h, rc, "mz_step")
        if self._host_rewards:
            self._apply_host_task()
        elif self._env_goals is not None and self._auto_reset:
            self._resample_env_goals(self._done != 0)  # the kernel judged this step with the old goals and restarted these envs
        info = {"position": self._info[:, :2], "reward_forward": self._info[:, 2], "reward_ctrl": self._info[:, 3],
                "goal_index": self._goal}
        if self._auto_reset:
            info["final_observation"] = self._final_obs  # valid in the rows where done != 0
        return self._obs, self._reward, self._done, info

    def _apply_host_task(self):
        """User-defined Python reward()/termination(): evaluated on the host from the obs batch (one device-to-host copy
        per step; the kernel supplied the inner reward terms, the task part is computed here).  With
        auto-reset the host then resets exactly the envs ITS verdict finished (the kernel's auto-reset is off for these
        tasks, see set_auto_reset)."""
        torch = self._torch
        m = self.model.c
        # one device-to-host copy: obs | info | done packed into one tensor
        packed = torch.cat([self._obs, self._info, self._done.to(torch.float32).unsqueeze(1)], dim=1).double().cpu().numpy()
        obs, inf, dev_done = packed[:, : self.obs_dim], packed[:, self.obs_dim: self.obs_dim + 4], packed[:, -1].astype(np.uint8)
        inner = (m.forward_reward_weight * inf[:, 2] + inf[:, 3]) * m.inner_reward_scaling  # ant.py:68, swimmer.py:43
        if m.robot == 0:
            inner[:] = 0.0  # point.py:61
        rew = np.array([self._task.reward(o) for o in obs]) + inner
        term = np.array([bool(self._task.termination(o)) for o in obs])
        done = (term.astype(np.uint8) | (dev_done & 2)).astype(np.uint8)
        self._reward.copy_(torch.as_tensor(rew, dtype=torch.float32))
        self._done.copy_(torch.as_tensor(done))
        # the device's first-match goal index belongs to the built-in task descriptor, not to the Python override that just
        # judged the step: it is not defined for host-judged tasks
        self._goal.fill_(-1)
        if self._auto_reset and done.any():
            if self._host_mask is None:  # persistent: the reset kernel may still read it after this call returns
                self._host_mask = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
            torch.ne(self._done, 0, out=self._host_mask.view(torch.bool))
            self._final_obs[self._host_mask.view(torch.bool)] = self._obs[self._host_mask.view(torch.bool)]
            rc = self._lib.mz_reset(self._h, _ptr(self._host_mask), C.c_uint64(self._seed), _ptr(self._obs), self._stream())
            _capi.check(self._lib, self._h, rc, "mz_reset (host-judged auto-reset)")
            self._seed += 1

    def get_state(self):
        torch, n = self._torch, self.num_envs
        qpos = torch.empty((n, self.nq), dtype=torch.float32, device=self.device)
        qvel = torch.empty((n, self.nv), dtype=torch.float32, device=self.device)
        warm = torch.empty((n, self.nv), dtype=torch.float32, device=self.device)
        t = torch.empty(n, dtype=torch.int32, device=self.device)
        _capi.check(self._lib, self._h, self._lib.mz_get_state(self._h, _ptr(qpos), _ptr(qvel), _ptr(warm), _ptr(t), self._stream()), "mz_get_state")
        return qpos, qvel, warm, t

    def set_state(self, qpos=None, qvel=None, warmstart=None, t=None):
        torch = self._torch

        def prep(x, dt, shape):
            if x is None:
                return None
            x = torch.as_tensor(x, device=self.device).to(dt).contiguous()
            if tuple(x.shape) != shape:
                raise ValueError(f"expected shape {shape}, got {tuple(x.shape)}")
            return x

        n = self.num_envs
        qpos = prep(qpos, torch.float32, (n, self.nq)); qvel = prep(qvel, torch.float32, (n, self.nv))
        warmstart = prep(warmstart, torch.float32, (n, self.nv)); t = prep(t, torch.int32, (n,))
        rc = self._lib.mz_set_state(self._h, _ptr(qpos), _ptr(qvel), _ptr(warmstart), _ptr(t), self._stream())
        _capi.check(self._lib, self._h, rc, "mz_set_state")
        torch.cuda.current_stream(self.device).synchronize()  # inputs may be temporaries

    def status(self):
        out = self._torch.empty(self.num_envs, dtype=self._torch.int32, device=self.device)
        _capi.check(self._lib, self._h, self._lib.mz_get_status(self._h, _ptr(out), self._stream()), "mz_get_status")
        return out

    def debug_forward(self, actions=None):
        torch, n = self._torch, self.num_envs
        qacc = torch.empty((n, self.nv), dtype=torch.float32, device=self.device)
        counts = torch.empty((n, 2), dtype=torch.int32, device=self.device)
        a = None if actions is None else torch.as_tensor(actions, device=self.device).to(torch.float32).contiguous()
        _capi.check(self._lib, self._h, self._lib.mz_debug_forward(self._h, _ptr(a), _ptr(qacc), _ptr(counts), self._stream()), "mz_debug_forward")
        return qacc, counts

    def debug_task_eval(self, obs):
        """Task reward / termination / first matching goal for rows of observations [n, obs_dim] (parity tests)."""
        torch = self._torch
        o = torch.as_tensor(obs, device=self.device).to(torch.float32).contiguous()
        n = o.shape[0]
        rew = torch.empty(n, dtype=torch.float32, device=self.device)
        done = torch.empty(n, dtype=torch.uint8, device=self.device)
        gi = torch.empty(n, dtype=torch.int32, device=self.device)
        _capi.check(self._lib, self._h, self._lib.mz_debug_task_eval(self._h, n, _ptr(o), _ptr(rew), _ptr(done), _ptr(gi), self._stream()), "mz_debug_task_eval")
        torch.cuda.current_stream(self.device).synchronize()
        return rew, done, gi

    def debug_detect(self, old_xy, new_xy):
        """The Point's manual wall rule on float64 moves [n, 2]: (hit, point, final_xy); hit 0 none, 1 bounce, 2 give-up, -1 collinear."""
        torch = self._torch
        o = torch.as_tensor(np.asarray(old_xy, np.float64), device=self.device).contiguous()
        w = torch.as_tensor(np.asarray(new_xy, np.float64), device=self.device).contiguous()
        n = o.shape[0]
        hit = torch.empty(n, dtype=torch.int32, device=self.device)
        pt = torch.zeros((n, 2), dtype=torch.float64, device=self.device)
        fin = torch.empty((n, 2), dtype=torch.float64, device=self.device)
        _capi.check(self._lib, self._h, self._lib.mz_debug_detect(self._h, n, _ptr(o), _ptr(w), _ptr(hit), _ptr(pt), _ptr(fin), self._stream()), "mz_debug_detect")
        torch.cuda.current_stream(self.device).synchronize()
        return hit, pt, fin

    def phase_cycles(self):
        """Phase timers of the instrumented kernel (set_option('profile_phases', 1) first)."""
        out = (C.c_uint64 * 16)()
        _capi.check(self._lib, self._h, self._lib.mz_read_phase_cycles(self._h, out), "mz_read_phase_cycles")
        return list(out)

    def wave_cycles(self, n: int):
        """Per-workgroup cycle totals of the instrumented kernel since the last call (numpy uint64 [n])."""
        out = np.zeros(n, np.uint64)
        _capi.check(self._lib, self._h, self._lib.mz_read_wave_cycles(self._h, out.ctypes.data_as(C.c_void_p), n), "mz_read_wave_cycles")
        self.last_wave_newton_iters = out >> np.uint64(40)  # Newton iterations the wave ran (packed above the cycle count)
        return out & np.uint64((1 << 40) - 1)

    def wave_phase_cycles(self, n: int):
        """Per-workgroup phase cycles of the instrumented kernel since the last call (numpy uint64 [n, 16])."""
        out = np.zeros((n, 16), np.uint64)
        _capi.check(self._lib, self._h, self._lib.mz_read_wave_phase_cycles(self._h, out.ctypes.data_as(C.c_void_p), n), "mz_read_wave_phase_cycles")
        return out

    def kernel_ms(self) -> float:
        return float(self._lib.mz_last_kernel_ms(self._h))

    # -- render / export bridge (render.py; reference: MazeEnv.render, maze_env.py:389-420) ----------------------------
    def render(self, mode: str = "rgb_array", env_index: int = 0, image_shape: Tuple[int, int] = (600, 480)):
        """Top view of env `env_index` as an uint8 [H, W, 3] array: its state is pulled from the device and rasterised on the
        host (render.render_top_down).  There is no GL context or websocket server next to a device batch: every mode
        returns the array."""
        from mujoco_maze_amd import render as R

        qpos = self.get_state()[0][env_index].double().cpu().numpy()
        return R.render_top_down(self.model, qpos, image_shape)

    def state_for_viewer(self, env_index: int = 0) -> dict:
        """MJCF of the model + the env's qpos / qvel (plain lists) for replay in an external MuJoCo viewer."""
        from mujoco_maze_amd import render as R

        qpos, qvel, _, _ = self.get_state()
        return R.state_for_viewer(self.model, qpos[env_index].double().cpu().numpy(), qvel[env_index].double().cpu().numpy())

    @property
    def has_extended_obs(self) -> bool:
        return bool(self._task.TOP_DOWN_VIEW or self._task.OBSERVE_BLOCKS or self._task.OBSERVE_BALLS)


class MazeEnv:
    """Single environment with the reference's constructor and step/reset shapes."""

    def __init__(self, model_cls: Type[AgentModel], maze_task: Type[MazeTask] = MazeTask, include_position: bool = True,
                 maze_height: float = 0.5, maze_size_scaling: float = 4.0, inner_reward_scaling: float = 1.0,
                 restitution_coef: float = 0.8, task_kwargs: Optional[dict] = None, websock_port: Optional[int] = None,
                 camera_move_x=None, camera_move_y=None, camera_zoom=None, image_shape: Tuple[int, int] = (600, 480),
                 **kwargs) -> None:
        self.vec = VecMazeEnv(model_cls, maze_task, num_envs=1, maze_height=maze_height, maze_size_scaling=maze_size_scaling,
                              inner_reward_scaling=inner_reward_scaling, restitution_coef=restitution_coef,
                              task_kwargs=task_kwargs, **kwargs)
        self._task = self.vec._task
        self.wrapped_env = WrappedRobot(self.vec, model_cls, single=True)
        self.t = 0
        self.action_space = self.vec.action_space
        self.observation_space = self.vec.observation_space
        self._max_steps = self.vec.model.c.max_episode_steps
        self._image_shape = image_shape

    def render(self, mode="human", **kwargs):
        """The reference hands back MuJoCo's camera image (or pushes it to its websocket viewer); here: the host-side top
        view of the device state (render.py), as an uint8 [H, W, 3] array in every mode."""
        return self.vec.render(mode, 0, self._image_shape)

    @property
    def unwrapped(self):
        return self

    @property
    def has_extended_obs(self) -> bool:
        return self.vec.has_extended_obs

    @property
    def _observe_balls(self) -> bool:
        return self._task.OBSERVE_BALLS

    def get_ori(self) -> float:
        return self.wrapped_env.get_ori()  # maze_env.py:231-232

    def reset(self, *, return_info: bool = True, **kwargs):
        """`(obs, info)` like the reference's MazeEnv.reset (maze_env.py:371-382).  `return_info=False` hands back the bare
        observation — the old-gym convention the reference's own tests are written against (`env.reset().shape`,
        tests/test_envs.py:13,27), for callers that have not moved to the tuple."""
        self.t = 0
        obs = self.vec.reset(seed=kwargs.get("seed"))[0].double().cpu().numpy()
        return (obs, {}) if return_info else obs

    def step(self, action):
        self.t += 1
        v = self.vec
        torch = v._torch
        if getattr(self, "_act_dev", None) is None:  # pinned staging for the action: an asynchronous upload, no allocation per step
            self._act_host = torch.empty((1, v.nu), dtype=torch.float32, pin_memory=True)
            self._act_dev = torch.empty((1, v.nu), dtype=torch.float32, device=v.device)
        self._act_host.numpy()[0, :] = np.asarray(action, dtype=np.float32).reshape(-1)
        self._act_dev.copy_(self._act_host, non_blocking=True)
        obs, reward, done, info = v.step(self._act_dev)
        # ONE device-to-host copy per step (obs | reward | done | info packed on the device): six separate reads were six synchronisations
        # (round 6: 158 -> 60 us per PointUMaze step)
        host = torch.cat([obs[0], reward, done.to(torch.float32), v._info[0]]).cpu().numpy().astype(np.float64)
        k = v.obs_dim
        d = int(host[k + 1])
        out_info = {"position": host[k + 2: k + 4].copy(), "reward_forward": float(host[k + 4]), "reward_ctrl": float(host[k + 5])}
        if d & 2:
            out_info["TimeLimit.truncated"] = not (d & 1)
        return host[:k].copy(), float(host[k]), bool(d), out_info

    def close(self) -> None:
        self.vec.close()
