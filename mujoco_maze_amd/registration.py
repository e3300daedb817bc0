"""Environment ids — mirror of the reference's registration loop
(`mujoco_maze/__init__.py:17-78`): `"{Point,Ant,Reacher,Swimmer}{Maze}-v{i}"`
for every task in `TaskRegistry`, `max_episode_steps=1000`,
`reward_threshold=task.REWARD_THRESHOLD`.  gym is absent on the build and GPU
boxes, so the ids live in an internal registry; when gym *is* importable the
same ids are also registered there with `mujoco_maze_amd.maze_env:MazeEnv` as
entry point (single-env façade, num_envs=1).
"""
from dataclasses import dataclass, field
from typing import Dict

from mujoco_maze_amd.agent_model import AntEnv, PointEnv, ReacherEnv, SwimmerEnv
from mujoco_maze_amd.maze_task import TaskRegistry


@dataclass
class EnvSpec:
    id: str
    kwargs: dict = field(default_factory=dict)
    max_episode_steps: int = 1000
    reward_threshold: float = 0.0
    entry_point: str = "mujoco_maze_amd.maze_env:MazeEnv"


REGISTRY: Dict[str, EnvSpec] = {}


def register(id: str, kwargs: dict, max_episode_steps: int, reward_threshold: float) -> None:
    REGISTRY[id] = EnvSpec(id, dict(kwargs), max_episode_steps, reward_threshold)
    try:  # optional: also expose through gym when it exists
        import gym  # type: ignore

        gym.envs.register(id=id, entry_point="mujoco_maze_amd.maze_env:MazeEnv", kwargs=dict(kwargs),
                          max_episode_steps=max_episode_steps, reward_threshold=reward_threshold)
    except Exception:
        pass


def _register_all() -> None:
    for maze_id in TaskRegistry.keys():
        for i, task_cls in enumerate(TaskRegistry.tasks(maze_id)):
            scaling = task_cls.MAZE_SIZE_SCALING
            plan = [("Point", PointEnv, scaling.point), ("Ant", AntEnv, scaling.ant),
                    ("Reacher", ReacherEnv, scaling.swimmer), ("Swimmer", SwimmerEnv, scaling.swimmer)]
            for prefix, model_cls, scale in plan:
                if scale is None:
                    continue
                register(f"{prefix}{maze_id}-v{i}",
                         dict(model_cls=model_cls, maze_task=task_cls, maze_size_scaling=scale,
                              inner_reward_scaling=task_cls.INNER_REWARD_SCALING),
                         1000, task_cls.REWARD_THRESHOLD)


_register_all()
