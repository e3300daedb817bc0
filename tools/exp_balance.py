#!/usr/bin/env python3
"""GPU box helper (experiment): what would DEALING the expensive envs out — one per wave — buy?  A launch lasts as long as its slowest
wave (tools/exp_launch_stamps.py), and the slow waves are those that happen to hold two or three ants leaning on a wall.  The same batch
of states is stepped in two arrangements — as it is, and permuted so that the envs nearest to a wall are spread over the waves (four
consecutive env slots = one wave of the 16-lane kernel) — and the kernel time of that one step is compared.  The physics is the same set
of env-steps either way.
    python tools/exp_balance.py [envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = mm.make("AntUMaze-v0", num_envs=n, auto_reset=True, force_vec=True)
env.reset(seed=20260928)
g = torch.Generator(device=env.device).manual_seed(1234)
pool = [torch.rand((n, 8), device=env.device, generator=g) * 60 - 30 for _ in range(32)]
for i in range(250): env.step(pool[i % 32])
c = env.model.c
grid = np.array([[c.grid[i][j] for j in range(c.grid_cols)] for i in range(c.grid_rows)])
bi, bj = np.nonzero(grid == 1)
bx, by = bj * c.maze_scale - c.torso_x, bi * c.maze_scale - c.torso_y


def wall_gap(qpos):
    dx = np.maximum(np.abs(qpos[:, None, 0] - bx[None]) - c.wall_half_xy, 0.0)
    dy = np.maximum(np.abs(qpos[:, None, 1] - by[None]) - c.wall_half_xy, 0.0)
    return np.sqrt(dx * dx + dy * dy).min(1)


def timed_step(st, act):
    env.set_state(*st)
    env.set_option("time_kernels", 1)
    env.step(act)
    torch.cuda.synchronize()
    return env.kernel_ms()


nw = n // 4
res = []
for trial in range(24):
    for i in range(8): env.step(pool[(trial * 8 + i) % 32])
    st = [x.cpu().numpy().copy() for x in env.get_state()]
    act = pool[trial % 32]
    gap = wall_gap(st[0])
    order = np.argsort(gap, kind="stable")            # nearest to a wall first
    perm = np.empty(n, np.int64)                      # perm[slot] = env that sits in that slot
    k = np.arange(n)
    perm[(k % nw) * 4 + (k // nw)] = order            # k-th nearest -> wave k mod nw, lane group k div nw
    rnd = np.random.default_rng(trial).permutation(n)
    t_a = timed_step(st, act)
    t_b = timed_step([x[perm] for x in st], act[torch.as_tensor(perm, device=act.device)])
    t_r = timed_step([x[rnd] for x in st], act[torch.as_tensor(rnd, device=act.device)])
    t_a2 = timed_step(st, act)
    res.append((t_a, t_b, t_r, t_a2, int((gap < 0.9).sum())))
    env.set_state(*st)
r = np.array(res)
print(f"AntUMaze-v0 {n} envs, {len(res)} batches of states, kernel ms of ONE step: as it is {r[:, 0].mean():.4f} (again {r[:, 3].mean():.4f}) | envs nearest to a wall dealt one per wave {r[:, 1].mean():.4f} | random permutation {r[:, 2].mean():.4f};  envs within 0.9 of a wall: mean {r[:, 4].mean():.0f}")
print("per batch (as is, dealt, random):", " ".join(f"{a:.4f}/{b:.4f}/{c_:.4f}" for a, b, c_, _, _ in res[:12]))
