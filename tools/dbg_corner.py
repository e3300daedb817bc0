#!/usr/bin/env python3
"""GPU box helper: the corner-contact states of tests/test_gpu_parity.py::test_ant_corner_contacts_overflow_the_staging stepped by the
8-lane (lane-group formulation) and the 16-lane (quad forward + row solver) instantiations, against the oracle and its perturbed runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib
from tests.test_gpu_parity import _rollout_states, _f32
oracle = oracle_lib.load()
n = 256
res = {}
for lanes in (8, 16):
    env = mm.make("AntUMaze-v0", num_envs=n)
    env.set_option("lanes_per_env", lanes)
    cm = env.model
    st = _rollout_states(oracle, cm, n, 8, {40})[40]
    rng = np.random.default_rng(9)
    st["qpos"][:, 0] = 19.0 + rng.uniform(0.0, 0.7, n); st["qpos"][:, 1] = -3.0 - rng.uniform(0.0, 0.7, n)
    st["qvel"][:, 0] = 1.0; st["qvel"][:, 1] = -1.0
    st = _f32(st)
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    env.set_state(st["qpos"], st["qvel"], st["warm"], st["t"])
    env.step(torch.as_tensor(act, device=env.device))
    res[lanes] = [x.cpu().numpy() for x in env.get_state()][:2]
    env.close()
start = {k: v.copy() for k, v in st.items()}
ref = {k: v.copy() for k, v in st.items()}
oracle.step(cm, ref, act.astype(np.float64), nthreads=8)
e8 = np.abs(res[8][1] - ref["qvel"]).max(1); e16 = np.abs(res[16][1] - ref["qvel"]).max(1); d = np.abs(res[8][1] - res[16][1]).max(1)
bad = np.where((e8 > 1e-5) | (e16 > 1e-5))[0]
fwd = oracle.forward(cm, start["qpos"], start["qvel"], act.astype(np.float64), start["warm"])["counts"][:, 0]
print("outliers:", len(bad))
prng = np.random.default_rng(1)
for e in bad:
    sp = []
    for _ in range(12):
        p = {k: v[e:e+1].copy() for k, v in start.items()}
        p["qpos"] = p["qpos"] + prng.uniform(-1e-6, 1e-6, p["qpos"].shape) * np.maximum(1.0, np.abs(p["qpos"]))
        oracle.step(cm, p, act[e:e+1].astype(np.float64))
        sp.append(np.abs(p["qvel"][0] - ref["qvel"][e]).max())
    print(f"env {e:3d} ncon {fwd[e]:2d}: |8-lane - oracle| {e8[e]:.2e}  |16-lane - oracle| {e16[e]:.2e}  |8 - 16| {d[e]:.2e}   oracle's own spread under 1e-6 perturbation: median {np.median(sp):.2e} max {max(sp):.2e}")
