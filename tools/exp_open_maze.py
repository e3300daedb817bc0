#!/usr/bin/env python3
"""GPU box helper (experiment): the plain ant's step kernel in a maze WITHOUT reachable walls (a 9 x 9 room at the ant's scale: no ant
gets near a wall within the run) against AntUMaze-v0 — how much of a launch is the wall narrow phase of the ~1 % of the waves that
run it?  Same kernel, same code; only the data differ.
    python tools/exp_open_maze.py [envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from mujoco_maze_amd import maze_task as T
from mujoco_maze_amd.maze_env import VecMazeEnv
from mujoco_maze_amd.maze_env_utils import MazeCell
from mujoco_maze_amd.maze_task import MazeGoal


class OpenRoom(T.MazeTask):
    REWARD_THRESHOLD = 0.9
    PENALTY = -0.0001
    reward = T.GoalRewardUMaze.reward

    def __init__(self, scale):
        super().__init__(scale)
        self.goals = [MazeGoal(np.array([3.0 * scale, 3.0 * scale]))]

    @staticmethod
    def create_maze():
        E, B, R = MazeCell.EMPTY, MazeCell.BLOCK, MazeCell.ROBOT
        rows = [[B] * 11]
        for i in range(9):
            rows.append([B] + [R if (i == 4 and j == 4) else E for j in range(9)] + [B])
        rows.append([B] * 11)
        return rows


n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def run(env, label):
    env.reset(seed=1)
    g = torch.Generator(device=env.device).manual_seed(0)
    acts = [(torch.rand((n, env.nu), device=env.device, generator=g) * 60 - 30) for _ in range(16)]
    for i in range(200): env.step(acts[i % 16])
    env.set_option("time_kernels", 500)
    for i in range(500): env.step(acts[i % 16])
    torch.cuda.synchronize()
    ms = env.kernel_ms()
    print(f"{label:32s} kernel {ms:.4f} ms   ({n / ms / 1e3:.2f} M env-steps/s by kernel time)   flagged {int((env.status() != 0).sum())}")


run(VecMazeEnv(mm.AntEnv, OpenRoom, num_envs=n, auto_reset=True, maze_size_scaling=8.0), "open 9 x 9 room (no walls near)")
run(mm.make("AntUMaze-v0", num_envs=n, auto_reset=True, force_vec=True), "AntUMaze-v0")
