"""mujoco_maze_amd — MI355X-native batched maze-environment stepper.

Drop-in for the hot path of kngwyu/mujoco-maze (`MazeEnv.step/reset`,
`AgentModel.step`, `MazeTask.reward/termination`): same env ids, same
`MazeTask` / `MazeGoal` / `MazeCell` / `TaskRegistry` / `AgentModel` plugin
surface, but `step()` advances thousands of environments in lock-step inside
hand-written HIP kernels (csrc/) behind a C-ABI (include/mazestep.h).
"""
from mujoco_maze_amd.agent_model import AgentModel, AntEnv, PointEnv, ReacherEnv, SwimmerEnv
from mujoco_maze_amd.maze_env_utils import MazeCell
from mujoco_maze_amd.maze_task import MazeGoal, MazeTask, Scaling, TaskRegistry
from mujoco_maze_amd.registration import REGISTRY, EnvSpec, register

__version__ = "0.1.0"


def make(id: str, num_envs: int = 1, **overrides):
    """`gym.make` equivalent.  `num_envs == 1` returns the reference-shaped single
    environment (`MazeEnv`); larger batches return `VecMazeEnv`."""
    from mujoco_maze_amd.maze_env import MazeEnv, VecMazeEnv

    if id not in REGISTRY:
        raise KeyError(f"No registered env with id: {id}")
    spec = REGISTRY[id]
    kwargs = dict(spec.kwargs)
    kwargs.update(overrides)
    kwargs.setdefault("max_episode_steps", spec.max_episode_steps)
    if num_envs == 1 and not kwargs.pop("force_vec", False):
        return MazeEnv(**kwargs)
    return VecMazeEnv(num_envs=num_envs, **kwargs)
