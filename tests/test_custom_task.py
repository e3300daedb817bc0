"""User-defined tasks (the reference's README "How to customize" flow: subclass MazeTask, give it goals, a maze and a
reward): known reward kinds run inside the kernel, anything else is evaluated on the host from the returned observations."""
import numpy as np
import pytest

import mujoco_maze_amd as mm
from mujoco_maze_amd import maze_task as T
from mujoco_maze_amd import model
from mujoco_maze_amd.maze_env_utils import MazeCell
from mujoco_maze_amd.maze_task import MazeGoal, MazeTask


class GoalRewardCross(MazeTask):
    """A user maze with a hand-written reward: +1 at the goal, a small living cost that grows with the distance otherwise."""

    REWARD_THRESHOLD = 0.9
    PENALTY = -0.0001

    def __init__(self, scale: float) -> None:
        super().__init__(scale)
        self.goals = [MazeGoal(np.array([2.0 * scale, 0.0]))]

    def reward(self, obs: np.ndarray) -> float:
        if self.termination(obs):
            return 1.0
        return self.PENALTY * (1.0 + float(np.hypot(*(obs[:2] - self.goals[0].pos))))

    @staticmethod
    def create_maze():
        E, B, R = MazeCell.EMPTY, MazeCell.BLOCK, MazeCell.ROBOT
        return [
            [B, B, B, B, B, B, B],
            [B, B, B, E, B, B, B],
            [B, R, E, E, E, E, B],
            [B, B, B, E, B, B, B],
            [B, B, B, B, B, B, B],
        ]


class InheritedRewardCross(GoalRewardCross):
    """Same maze, but the stock goal reward of the reference (`GoalRewardUMaze.reward`, maze_task.py:110-111)."""

    reward = T.GoalRewardUMaze.reward


def test_reward_classification():
    # a stock reward function on a user maze runs inside the kernel; a hand-written one does not
    assert T.device_reward_descriptor(InheritedRewardCross(4.0)) is not None
    assert T.device_reward_descriptor(GoalRewardCross(4.0)) is None
    cm = model.compile_model("point", GoalRewardCross(4.0), 4.0)
    assert not cm.device_rewards and cm.c.ngoal == 1 and (cm.world.rows, cm.world.cols) == (5, 7)
    assert model.compile_model("point", InheritedRewardCross(4.0), 4.0).device_rewards


@pytest.mark.gpu
@pytest.mark.parametrize("model_cls", [mm.PointEnv, mm.AntEnv])
def test_user_task_with_python_reward_on_the_device(model_cls):
    import torch

    from mujoco_maze_amd.maze_env import VecMazeEnv

    n, scale = 64, 4.0
    env = VecMazeEnv(model_cls, GoalRewardCross, maze_size_scaling=scale, num_envs=n, inner_reward_scaling=0.0)
    task = GoalRewardCross(scale)
    env.reset(seed=3)
    # put a quarter of the envs next to the goal so that termination fires
    qpos, qvel, warm, t = env.get_state()
    qpos[: n // 4, 0] = 2.0 * scale - 0.1
    env.set_state(qpos=qpos)
    rng = np.random.default_rng(0)
    lo, hi = env.action_space.low, env.action_space.high
    obs, rew, done, info = env.step(torch.as_tensor(0.1 * rng.uniform(lo, hi, (n, env.nu)).astype(np.float32), device=env.device))
    o = obs.double().cpu().numpy()
    want_r = np.array([task.reward(x) for x in o])
    want_d = np.array([task.termination(x) for x in o])
    assert np.allclose(rew.cpu().numpy(), want_r, atol=1e-6)
    assert np.array_equal((done.cpu().numpy() & 1).astype(bool), want_d) and want_d[: n // 4].all() and not want_d[n // 4:].any()
    env.close()


class FarGoalCross(GoalRewardCross):
    """A user task whose termination() is a Python override that DISAGREES with the stock goal-neighbour rule the kernel
    knows (it ends the episode three times further out), so a device-side verdict would reset the wrong envs."""

    def termination(self, obs: np.ndarray) -> bool:
        return bool(np.linalg.norm(obs[:2] - self.goals[0].pos) <= 1.8)


@pytest.mark.gpu
@pytest.mark.parametrize("model_cls", [mm.PointEnv, mm.AntEnv])
def test_host_judged_task_with_auto_reset(model_cls):
    """for host-evaluated tasks the envs that reset must be exactly the ones the user's termination() ended —
    the kernel's own auto-reset is off and the host issues the masked reset; obs holds the new episode's first observation,
    info["final_observation"] the terminal one; the inner reward uses forward_reward_weight."""
    import torch

    from mujoco_maze_amd.maze_env import VecMazeEnv

    n, scale = 64, 4.0
    env = VecMazeEnv(model_cls, FarGoalCross, maze_size_scaling=scale, num_envs=n, auto_reset=True, forward_reward_weight=2.5,
                     inner_reward_scaling=1.0)
    task = FarGoalCross(scale)
    assert env._host_rewards and env._auto_reset
    env.reset(seed=3)
    qpos, qvel, warm, t = env.get_state()
    qpos[: n // 4, 0] = 2.0 * scale - 1.2   # inside the user's radius 1.8, outside the stock threshold 0.6
    qpos[n // 4: n // 2, 0] = 2.0 * scale - 0.1  # inside both
    env.set_state(qpos=qpos)
    act = torch.zeros((n, env.nu), device=env.device)
    obs, rew, done, info = env.step(act)
    d = done.cpu().numpy()
    fin = info["final_observation"].double().cpu().numpy()
    want = np.array([task.termination(x) for x in fin[: n // 2]])
    assert want.all() and np.all(d[: n // 2] & 1) and not np.any(d[n // 2:])
    t_after = env.get_state()[3].cpu().numpy()
    assert np.all(t_after[: n // 2] == 0) and np.all(t_after[n // 2:] == 1)  # exactly the user's verdict was reset
    o = obs.cpu().numpy()
    assert np.all(o[: n // 2, -1] == 0.0) and np.all(np.abs(o[: n // 2, 0]) <= 0.1 + 1e-6)  # first obs of the new episode
    assert np.allclose(o[n // 2:, -1], 0.001)
    # reward of the terminal step: user's reward on the terminal obs + inner reward with the forward weight (ant.py:68)
    inner = 0.0 if model_cls is mm.PointEnv else (2.5 * info["reward_forward"] + info["reward_ctrl"]).double().cpu().numpy()
    last = np.where((d != 0)[:, None], fin, obs.double().cpu().numpy())
    want_r = np.array([task.reward(x) for x in last]) + inner
    assert np.allclose(rew.cpu().numpy(), want_r, atol=1e-5)
    env.close()


class MovingGoalCross(InheritedRewardCross):
    """A task that resamples its goal on every reset (MazeTask.sample_goals, maze_task.py:66-67; MazeEnv.reset acts on its
    return value, maze_env.py:374-376): the goal alternates between the east arm and the north arm of the cross."""

    def __init__(self, scale: float) -> None:
        super().__init__(scale)
        self._k = 0

    def sample_goals(self) -> bool:
        self._k += 1
        east, north = np.array([2.0 * self.scale, 0.0]), np.array([2.0 * self.scale, -1.0 * self.scale])
        self.goals = [MazeGoal(east if self._k % 2 else north, threshold=0.8)]
        return True


@pytest.mark.gpu
@pytest.mark.parametrize("model_cls", [mm.PointEnv, mm.AntEnv, mm.SwimmerEnv])
def test_resampled_goals_reach_the_device(model_cls):
    """reset() asks the task for new goals (maze_env.py:374-376).  The reference has one task object per env, so every env gets its
    OWN draw: the batch moves to per-env goal positions (mz_bind_env_goals) and termination / reward of the kernel's predicate
    follow each env's goal, threshold included; the shared table (mz_set_goals) keeps the first draw."""
    import torch

    from mujoco_maze_amd.maze_env import VecMazeEnv

    n, scale = 32, 4.0
    env = VecMazeEnv(model_cls, MovingGoalCross, maze_size_scaling=scale, num_envs=n, inner_reward_scaling=0.0)
    assert not env._host_rewards and env.env_goals is None  # the stock reward: judged inside the kernel; one table until goals move
    zero = torch.zeros((n, env.nu), device=env.device)
    east, north = np.array([8.0, 0.0]), np.array([8.0, -4.0])
    for episode in range(3):
        env.reset(seed=episode)
        eg = env.env_goals.cpu().numpy()
        assert eg.shape == (n, 8, 3) and not eg[:, 1:].any() and not eg[:, 0, 2].any()
        eg = eg[:, 0, :2]
        is_east = np.all(eg == east, axis=1)
        assert np.all(is_east | np.all(eg == north, axis=1)) and 0 < is_east.sum() < n  # every env one of the task's draws, both occur
        goal = env._task.goals[0]
        assert np.allclose(goal.pos, eg[0])  # the task object keeps the FIRST draw ...
        other = np.where(is_east[:, None], north, east)
        qpos = env.get_state()[0]
        q = qpos.cpu().numpy()
        q[: n // 4, 0], q[: n // 4, 1] = eg[: n // 4, 0] - 0.7, eg[: n // 4, 1]                # inside 0.8, outside the default 0.6
        q[n // 4: n // 2, 0], q[n // 4: n // 2, 1] = other[n // 4: n // 2, 0] - 0.1, other[n // 4: n // 2, 1]  # at the goal of OTHER envs
        env.set_state(qpos=torch.as_tensor(q, device=env.device))
        obs, rew, done, info = env.step(zero)
        o = obs.double().cpu().numpy()
        want = np.linalg.norm(o[:, :2] - eg, axis=1) <= goal.threshold
        got = (done.cpu().numpy() & 1).astype(bool)
        assert np.array_equal(got, want), episode
        assert want[: n // 4].all() and not want[n // 4:].any()
        assert np.allclose(rew.cpu().numpy(), np.where(want, 1.0, env._task.PENALTY), atol=1e-7)
        assert np.array_equal(info["goal_index"].cpu().numpy(), np.where(want, 0, -1))
        # the parity-test entry judges row r with env r's goals
        r2, d2, g2 = env.debug_task_eval(obs)
        assert np.array_equal(d2.cpu().numpy().astype(bool), want)
        # ... and so do the shared table and the host copy of the model (what the oracle / parity tools judge env 0 with)
        assert env.model.c.ngoal == 1 and np.allclose([env.model.c.goal_pos[0][k] for k in range(2)], goal.pos)
        assert env.model.c.goal_threshold[0] == goal.threshold
    # writing into the tensor moves the goal of a single env
    env.reset(seed=9)
    env.env_goals[5, 0, :2] = torch.as_tensor([0.3, 0.0], dtype=torch.float64)  # next to the start cell
    obs, rew, done, info = env.step(zero)
    d = done.cpu().numpy() & 1
    assert d[5] == 1 and d.sum() == 1
    env.close()


@pytest.mark.gpu
def test_per_env_goals_on_the_general_engine():
    """The same per-env goal table through the general engine (a user robot: tests/user_robots.py): every env judged with its own goal."""
    import torch

    from mujoco_maze_amd.maze_env import VecMazeEnv
    from tests import user_robots

    BipedAnt, _ = user_robots.robot_classes()
    n = 16
    env = VecMazeEnv(BipedAnt, MovingGoalCross, maze_size_scaling=4.0, num_envs=n, inner_reward_scaling=0.0)
    assert env.launch_info()["engine"] == 1 and not env._host_rewards
    env.reset(seed=0)
    eg = env.env_goals.cpu().numpy()[:, 0, :2]
    assert len({tuple(g) for g in eg}) == 2
    q = env.get_state()[0].cpu().numpy()
    near = np.arange(n) % 4 == 0
    q[near, 0], q[near, 1] = eg[near, 0] - 0.5, eg[near, 1]
    env.set_state(qpos=torch.as_tensor(q, device=env.device))
    obs, rew, done, info = env.step(torch.zeros((n, env.nu), device=env.device))
    o = obs.double().cpu().numpy()
    want = np.linalg.norm(o[:, :2] - eg, axis=1) <= 0.8
    assert want[near].all() and not want[~near].any()
    assert np.array_equal((done.cpu().numpy() & 1).astype(bool), want)
    assert np.array_equal(info["goal_index"].cpu().numpy(), np.where(want, 0, -1))
    env.close()


class RandomGoalCross(InheritedRewardCross):
    """Goal drawn uniformly along the east arm at every episode start (a continuous distribution: no two draws coincide)."""

    def __init__(self, scale: float) -> None:
        super().__init__(scale)
        self._rng = np.random.default_rng(5)

    def sample_goals(self) -> bool:
        self.goals = [MazeGoal(np.array([self._rng.uniform(1.0, 2.0) * self.scale, 0.0]), threshold=0.8)]
        return True


@pytest.mark.gpu
@pytest.mark.parametrize("model_cls", [mm.PointEnv, mm.AntEnv])
def test_goals_are_resampled_per_env_at_every_episode_start(model_cls):
    """maze_env.py:374-376 under the device auto-reset and under a masked reset: exactly the envs whose episode ended get new goals
    (members of the pool drawn at the last full reset), the others keep theirs; the terminal step itself was judged with the OLD goal.
    No host round trip: the rows are rewritten by device-side ops after the step kernel."""
    import warnings

    import torch

    from mujoco_maze_amd.maze_env import VecMazeEnv

    n = 64
    env = VecMazeEnv(model_cls, RandomGoalCross, maze_size_scaling=4.0, num_envs=n, auto_reset=True, inner_reward_scaling=0.0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # no "goals change at full reset() only" warning any more
        env.reset(seed=1)
    pool = env._goal_pool.cpu().numpy()
    g0 = env.env_goals.cpu().numpy().copy()
    assert np.array_equal(g0, pool) and len(np.unique(g0[:, 0, 0])) == n  # env i starts with draw i
    q = env.get_state()[0].cpu().numpy()
    at_goal = np.arange(n) % 4 == 0
    q[at_goal, 0], q[at_goal, 1] = g0[at_goal, 0, 0] - 0.2, g0[at_goal, 0, 1]
    env.set_state(qpos=torch.as_tensor(q, device=env.device))
    zero = torch.zeros((n, env.nu), device=env.device)
    obs, rew, done, info = env.step(zero)
    d = (done.cpu().numpy() & 1).astype(bool)
    assert np.array_equal(d, at_goal) and np.allclose(rew.cpu().numpy()[at_goal], 1.0)
    g1 = env.env_goals.cpu().numpy()
    assert np.array_equal(g1[~at_goal], g0[~at_goal])
    changed = np.any(g1 != g0, axis=(1, 2))
    assert not changed[~at_goal].any() and changed[at_goal].sum() >= at_goal.sum() - 2  # a pool member picked at random (may pick its own)
    assert all(np.any(np.all(pool == row, axis=(1, 2))) for row in g1)
    # the restarted envs are judged with their NEW goals from the next step on: put them at the old goal -> not done (unless the new one is near)
    q = env.get_state()[0].cpu().numpy()
    q[at_goal, 0], q[at_goal, 1] = g0[at_goal, 0, 0] - 0.2, g0[at_goal, 0, 1]
    env.set_state(qpos=torch.as_tensor(q, device=env.device))
    obs, rew, done, info = env.step(zero)
    fin = np.where((done.cpu().numpy() != 0)[:, None], info["final_observation"].double().cpu().numpy(), obs.double().cpu().numpy())
    want = np.linalg.norm(fin[:, :2] - g1[:, 0, :2], axis=1) <= 0.8
    assert np.array_equal((done.cpu().numpy() & 1).astype(bool), want) and want.sum() < at_goal.sum()
    # masked reset: the same rule
    g2 = env.env_goals.cpu().numpy().copy()
    mask = np.zeros(n, np.uint8)
    mask[1::8] = 1
    env.reset(mask=torch.as_tensor(mask, device=env.device))
    g3 = env.env_goals.cpu().numpy()
    assert np.array_equal(g3[mask == 0], g2[mask == 0]) and np.any(g3[mask == 1] != g2[mask == 1])
    # a full reset draws a fresh pool
    env.reset(seed=2)
    assert not np.array_equal(env._goal_pool.cpu().numpy(), pool) and np.array_equal(env.env_goals.cpu().numpy(), env._goal_pool.cpu().numpy())
    env.close()


class MovingGoalPythonReward(MovingGoalCross):
    def reward(self, obs):
        return 1.0 if self.termination(obs) else -0.5


@pytest.mark.gpu
def test_goal_resampling_of_host_judged_tasks_under_auto_reset_warns_once():
    """Python reward()/termination() overrides are judged on the host from the ONE task object: its goals can only change at a full
    reset() (the reference resamples at every episode reset, maze_env.py:374-376).  Such a task under auto_reset says so (once)."""
    import warnings

    from mujoco_maze_amd.maze_env import VecMazeEnv

    env = VecMazeEnv(mm.PointEnv, MovingGoalPythonReward, maze_size_scaling=4.0, num_envs=4, auto_reset=True)
    assert env._host_rewards
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env.reset(seed=0)
        env.reset(seed=1)
    assert sum("sample_goals" in str(x.message) for x in w) == 1 and env.env_goals is None
    env.close()


def test_wrapped_robot_is_safe_to_copy_and_pickle():
    """WrappedRobot forwards unknown attributes to the robot class; copy / pickle probe a bare instance (no `_cls` yet) for
    __setstate__ / __deepcopy__ and must get AttributeError, not a recursion."""
    import copy
    import pickle

    from mujoco_maze_amd.maze_env import WrappedRobot

    w = WrappedRobot(None, mm.AntEnv)
    assert w.ORI_IND == mm.AntEnv.ORI_IND
    w2 = copy.copy(w)
    assert w2._cls is mm.AntEnv
    w3 = pickle.loads(pickle.dumps(w))
    assert w3._cls is mm.AntEnv and w3.ORI_IND == mm.AntEnv.ORI_IND
    with pytest.raises(AttributeError):
        w._no_such_private
    bare = WrappedRobot.__new__(WrappedRobot)
    with pytest.raises(AttributeError):
        bare.anything


@pytest.mark.gpu
def test_set_goals_argument_checks_and_goal_free_tasks():
    from mujoco_maze_amd import _capi
    from mujoco_maze_amd.maze_env import VecMazeEnv

    env = VecMazeEnv(mm.PointEnv, InheritedRewardCross, maze_size_scaling=4.0, num_envs=8)
    with pytest.raises(ValueError):
        env.set_goals([MazeGoal(np.zeros(2))] * 9)
    env.set_goals([])  # a task without goals never terminates (Corridor-v2: maze_task.py:483-503)
    import torch
    env.reset(seed=0)
    obs, rew, done, info = env.step(torch.zeros((8, 2), device=env.device))
    assert not done.cpu().numpy().any()
    env.set_goals([MazeGoal(np.array([0.0, 0.0]), threshold=5.0), MazeGoal(np.array([0.0, 0.0, 0.0]), reward_scale=0.5)])
    obs, rew, done, info = env.step(torch.zeros((8, 2), device=env.device))
    assert (done.cpu().numpy() & 1).all() and (info["goal_index"].cpu().numpy() == 0).all()
    env.close()


@pytest.mark.gpu
def test_record_binding_is_refused_for_host_judged_tasks():
    import torch

    from mujoco_maze_amd.maze_env import VecMazeEnv

    env = VecMazeEnv(mm.PointEnv, GoalRewardCross, maze_size_scaling=4.0, num_envs=8)
    assert env._host_rewards
    with pytest.raises(ValueError, match="host"):
        env.bind_record(torch.zeros((8, env.obs_dim + 2), device=env.device))
    env.bind_record(None)
    env.close()
