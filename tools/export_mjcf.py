#!/usr/bin/env python3
"""Write the MJCF of the model this repository steps for a registered env id (VERDICT r01 #2b).

    python tools/export_mjcf.py AntPush-v0 > antpush.xml

The XML is assembled from the SAME data `compile_model` turns into device constants — `robots.RobotSpec` (pinned against
the reference's assets by tests/golden/robots.json), the maze boxes of `MazeWorld` (pinned by tests/golden/worlds.json),
the movable-block / object-ball bodies compile_model appends, the goal sites — in the element order of the reference's
generator (mujoco_maze/maze_env.py:97-218: the asset's floor and robot first, then per cell the wall box or the movable
body, then the goal sites).  Anyone with MuJoCo can load it and compare against this repository's oracle and kernels:
tests/test_mujoco_crosscheck.py does exactly that when `import mujoco` works (it does not in the build image, where
the physics of the oracle is therefore PARITY UNPINNED — DESIGN.md section 5).
"""
import os
import sys
import xml.etree.ElementTree as ET

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def compile_for(env_id, **overrides):
    import mujoco_maze_amd as mm
    from mujoco_maze_amd import model

    spec = mm.REGISTRY[env_id]
    kw = dict(spec.kwargs)
    kw.update(overrides)
    scale = kw["maze_size_scaling"]
    cls = kw["model_cls"]
    return model.compile_model(cls.ROBOT, kw["maze_task"](scale, **(kw.get("task_kwargs") or {})), scale,
                               inner_reward_scaling=kw.get("inner_reward_scaling"), restitution_coef=kw.get("restitution_coef", 0.8),
                               maze_height=kw.get("maze_height", 0.5), max_episode_steps=spec.max_episode_steps)


def export_mjcf(env_id: str, cm=None, **overrides) -> str:
    from mujoco_maze_amd import mjcf

    return mjcf.world_to_mjcf(cm or compile_for(env_id, **overrides), name=env_id)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit("usage: export_mjcf.py <env id> [output.xml]")
    text = export_mjcf(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)
