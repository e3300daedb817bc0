"""Throughput of the general engine (csrc/generic_dyn.h: float64, one wavefront per env, any compiled mz_model) next to the specialised
kernels — the price of generality, measured: SPIN-plate mazes and user robots (the only engine that steps them), and registered ids
forced onto it with engine="general".  Whole-call wall clock over `steps` steps under auto-reset, random actions, inputs resident.
    python tools/general_engine_bench.py [envs] [steps]  ->  table on stdout (profiles/r06/general_engine.txt)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mujoco_maze_amd as mm  # noqa: E402
from mujoco_maze_amd import maze_task as T  # noqa: E402
from mujoco_maze_amd.maze_env import VecMazeEnv  # noqa: E402


def timed(env, steps, seed=0):
    n, nu = env.num_envs, env.nu
    lo = torch.as_tensor(env.action_space.low, device=env.device)
    hi = torch.as_tensor(env.action_space.high, device=env.device)
    g = torch.Generator(device=env.device).manual_seed(seed)
    acts = [lo + (hi - lo) * torch.rand((n, nu), device=env.device, generator=g) for _ in range(16)]
    env.set_auto_reset(True)
    env.reset(seed=seed)
    for k in range(20):
        env.step(acts[k % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = int((env.status() & 7).ne(0).sum())
    return n * steps / dt, dt / steps * 1e3, bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    from tests import user_robots
    from tests.test_general_engine import SPIN_TASKS

    BipedAnt, _ = user_robots.robot_classes()
    rows = []

    def run(label, make):
        env = make()
        info = env.launch_info()
        v, ms, bad = timed(env, steps)
        rows.append((label, info["engine"], v, ms, bad))
        env.close()

    for tname in sorted(SPIN_TASKS):
        run(f"{tname} / Ant (SPIN plate: ball joint + box of general orientation)", lambda: VecMazeEnv(mm.AntEnv, SPIN_TASKS[tname], num_envs=n, maze_size_scaling=4.0))
        run(f"{tname} / Point", lambda: VecMazeEnv(mm.PointEnv, SPIN_TASKS[tname], num_envs=n, maze_size_scaling=4.0))
    run("user robot BipedAnt in DistRewardPush (movable block)", lambda: VecMazeEnv(BipedAnt, T.DistRewardPush, num_envs=n, maze_size_scaling=4.0))
    run("user robot BipedAnt in GoalRewardUMaze", lambda: VecMazeEnv(BipedAnt, T.GoalRewardUMaze, num_envs=n, maze_size_scaling=4.0))
    for env_id in ("PointUMaze-v0", "AntUMaze-v0", "AntPush-v0", "AntFall-v0", "SwimmerUMaze-v0"):
        run(f"{env_id} engine=general", lambda: mm.make(env_id, num_envs=n, force_vec=True, engine="general"))
        run(f"{env_id} specialised kernel", lambda: mm.make(env_id, num_envs=n, force_vec=True))
    print(f"general engine vs specialised kernels, {n} envs, {steps} timed steps (auto-reset, random actions; whole-call wall clock)")
    print("%-78s %7s %14s %10s %8s" % ("workload", "engine", "env-steps/s", "ms/step", "flagged"))
    for label, eng, v, ms, bad in rows:
        print("%-78s %7s %12.3f M %10.4f %8d" % (label, "general" if eng else "special", v / 1e6, ms, bad))


if __name__ == "__main__":
    main()
