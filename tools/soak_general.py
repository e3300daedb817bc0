"""Soak of the general engine: long auto-reset rollouts under random actions, status bits accumulated (NaN / contact overflow / solver cap).
    python tools/soak_general.py [steps] [envs]   ->  profiles/r06/soak_general.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mujoco_maze_amd as mm  # noqa: E402
from mujoco_maze_amd import maze_task as T  # noqa: E402
from mujoco_maze_amd.maze_env import VecMazeEnv  # noqa: E402
from tests import user_robots  # noqa: E402
from tests.test_general_engine import SPIN_TASKS  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
BipedAnt, YSwimmer = user_robots.robot_classes()
cases = [("SpinUMaze / Ant", lambda: VecMazeEnv(mm.AntEnv, SPIN_TASKS["SpinUMaze"], num_envs=n, maze_size_scaling=4.0)),
         ("SpinCellMaze / Ant", lambda: VecMazeEnv(mm.AntEnv, SPIN_TASKS["SpinCellMaze"], num_envs=n, maze_size_scaling=4.0)),
         ("SpinUMaze / Point", lambda: VecMazeEnv(mm.PointEnv, SPIN_TASKS["SpinUMaze"], num_envs=n, maze_size_scaling=4.0)),
         ("BipedAnt / DistRewardPush", lambda: VecMazeEnv(BipedAnt, T.DistRewardPush, num_envs=n, maze_size_scaling=4.0)),
         ("BipedAnt / GoalRewardUMaze", lambda: VecMazeEnv(BipedAnt, T.GoalRewardUMaze, num_envs=n, maze_size_scaling=4.0)),
         ("Pincer (self-colliding) / DistRewardUMaze", lambda: VecMazeEnv(user_robots.pincer_class(), T.DistRewardUMaze, num_envs=n, maze_size_scaling=4.0)),
         ("YSwimmer / DistRewardUMaze", lambda: VecMazeEnv(YSwimmer, T.DistRewardUMaze, num_envs=n, maze_size_scaling=4.0)),
         ("AntPushMaze-v0 engine=general", lambda: mm.make("AntPushMaze-v0", num_envs=n, force_vec=True, engine="general")),
         ("AntMultiFall-v0 engine=general", lambda: mm.make("AntMultiFall-v0", num_envs=n, force_vec=True, engine="general")),
         ("PointBilliard-v0 engine=general", lambda: mm.make("PointBilliard-v0", num_envs=n, force_vec=True, engine="general"))]
for name, make in cases:
    env = make()
    env.set_auto_reset(True)
    env.reset(seed=3)
    lo = torch.as_tensor(env.action_space.low, device=env.device); hi = torch.as_tensor(env.action_space.high, device=env.device)
    g = torch.Generator(device=env.device).manual_seed(1)
    acc = torch.zeros(n, dtype=torch.int32, device=env.device)
    dones = 0
    t0 = time.perf_counter()
    for k in range(steps):
        obs, rew, done, info = env.step(lo + (hi - lo) * torch.rand((n, env.nu), device=env.device, generator=g))
        if k % 100 == 99:
            acc |= env.status()
            dones += int((done != 0).sum())
            assert torch.isfinite(obs).all(), (name, k)
    acc |= env.status()
    a = acc.cpu().numpy()
    print("%-44s n=%d steps=%d: %5.1f s  envs ever flagged: nan %d overflow %d maxiter %d; dones sampled %d" % (
        name, n, steps, time.perf_counter() - t0, int(((a & 1) != 0).sum()), int(((a & 2) != 0).sum()), int(((a & 4) != 0).sum()), dones), flush=True)
    env.close()
