#!/usr/bin/env python3
"""Round-6 fixtures, generated from the reference in the build container (needs /root/reference; never runs on the GPU box).

  spin_worlds.json   the MJCF world the reference's MazeEnv.__init__ generates (maze_env.py:97-218) for mazes with SPIN plates —
                     a thin box on two slides and a ball joint (maze_env.py:119-120,575,649-660; MazeCell.SPIN,
                     maze_env_utils.py:33,74-75):
                       SpinUMaze/{ant,point}   GoalRewardUMaze with PUT_SPIN_NEAR_AGENT = True (maze_task.py:67): the plate is dropped
                                               onto the robot's own cell, a quarter cell along x; `self.blocks` stays False, so the
                                               default geoms keep their solimp
                       SpinCellMaze/{ant,point} a custom maze with an explicit SPIN cell next to an XY_BLOCK (`self.blocks` True:
                                               solimp .995), OBSERVE_BLOCKS on
                     plus the observation the reference's MazeEnv._get_obs assembles at the spawn state (robot fake: zeros).
Data only: positions, sizes, masses, joint lists, names, observation sizes.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the gym / mujoco stubs and imports the reference)

from mujoco_maze import maze_env, maze_task  # noqa: E402


class _SpinUMaze(maze_task.GoalRewardUMaze):
    PUT_SPIN_NEAR_AGENT = True
    OBSERVE_BLOCKS = True


class _SpinCellMaze(maze_task.GoalRewardPush):
    OBSERVE_BLOCKS = True

    @staticmethod
    def create_maze():
        E, B, R, S, M = (maze_task.MazeCell.EMPTY, maze_task.MazeCell.BLOCK, maze_task.MazeCell.ROBOT, maze_task.MazeCell.SPIN,
                         maze_task.MazeCell.XY_BLOCK)
        return [[B, B, B, B, B],
                [B, E, S, E, B],
                [B, R, E, M, B],
                [B, E, E, E, B],
                [B, B, B, B, B]]


def main():
    out = {}
    for task, tname, scales in ((_SpinUMaze, "SpinUMaze", dict(ant=8.0, point=4.0)), (_SpinCellMaze, "SpinCellMaze", dict(ant=4.0, point=4.0))):
        for fake, tag in ((G._FakeAnt, "ant"), (G._FakePoint, "point")):
            env = maze_env.MazeEnv(model_cls=fake, maze_task=task, maze_size_scaling=scales[tag])
            real_build = G.build_env
            G.build_env = lambda _id, env=env: env
            try:
                w = G.dump_world("custom")
            finally:
                G.build_env = real_build
            out[f"{tname}/{tag}"] = dict(w, grid=G.grid_text(env._maze_structure), scale=scales[tag],
                                         put_spin_near_agent=bool(env._put_spin_near_agent),
                                         obs0=[float(v) for v in env._get_obs()])
    with open(os.path.join(HERE, "spin_worlds.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    for n, v in out.items():
        print(n, v["blocks"], v["default_geom_solimp"], v["obs_dim"], [(b["name"], b["pos"], b["geom"]["size"], [j["type"] for j in b["joints"]]) for b in v["movable"]])


if __name__ == "__main__":
    main()
