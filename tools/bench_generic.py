#!/usr/bin/env python3
"""GPU box helper: throughput of the generic tree kernel on the two user robots of tests/user_robots.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mujoco_maze_amd import maze_task as T
from mujoco_maze_amd.maze_env import VecMazeEnv
from tests import user_robots
B, Y = user_robots.robot_classes()
for cls, amp in ((B, 20.0), (Y, 1.0)):
    n = 4096
    env = VecMazeEnv(cls, T.DistRewardUMaze, num_envs=n, maze_size_scaling=4.0, auto_reset=True)
    env.reset(seed=1)
    a = (torch.rand((n, env.nu), device=env.device) * 2 - 1) * amp
    for _ in range(20): env.step(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): env.step(a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(cls.__name__, f"{n * 50 / dt / 1e6:.3f} M env-steps/s  {dt / 50 * 1e3:.3f} ms/step  flagged {int(((env.status() & 3) != 0).sum())}")
    env.close()
