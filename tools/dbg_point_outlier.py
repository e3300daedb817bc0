#!/usr/bin/env python3
"""GPU box helper: outliers of the PointPush-family parity test — which coordinate, how far, what the oracle does nearby."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib
from tests.test_gpu_parity import _f32
env_id = sys.argv[1] if len(sys.argv) > 1 else "PointPush-v0"
oracle = oracle_lib.load()
n = 1024
env = mm.make(env_id, num_envs=n)
cm = env.model
st, _ = oracle.reset(cm, n, 1)
rng = np.random.default_rng(0)
np.set_printoptions(precision=9, linewidth=200, suppress=False)
for k in range(81):
    act = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.25, 0.25, n)], 1).astype(np.float32)
    if k % 20 == 0:
        s64 = _f32(st); start = _f32(st)
        env.set_state(s64["qpos"], s64["qvel"], None, s64["t"])
        env.step(torch.as_tensor(act, device=env.device))
        qpos, qvel, _, t = [x.cpu().numpy() for x in env.get_state()]
        oracle.step(cm, s64, act.astype(np.float64), nthreads=8)
        e_all = np.maximum(np.abs(qpos - s64["qpos"]).max(1), np.abs(qvel - s64["qvel"]).max(1))
        bad = np.where(e_all > 2e-6 + 1e-5 * 10)[0]
        print(f"step {k}: max err {e_all.max():.2e}, envs beyond 1e-4: {(e_all > 1e-4).sum()}, beyond 1e-5: {(e_all > 1e-5).sum()}, beyond 3e-6: {(e_all > 3e-6).sum()}")
        for e in np.argsort(-e_all)[:4]:
            f = oracle.forward(cm, start["qpos"][e], start["qvel"][e])
            print(f"  env {e}: err {e_all[e]:.2e}  ncon/nefc at start {f['counts'][0]}")
            print("     start qpos", start["qpos"][e], "qvel", start["qvel"][e], "act", act[e])
            print("     dev  qpos", qpos[e], "qvel", qvel[e])
            print("     orac qpos", s64["qpos"][e], "qvel", s64["qvel"][e])
            sp = []
            prng = np.random.default_rng(3)
            for _ in range(8):
                p = {kk: v[e:e+1].copy() for kk, v in start.items()}
                p["qpos"] = p["qpos"] + prng.uniform(-1e-6, 1e-6, p["qpos"].shape)
                oracle.step(cm, p, act[e:e+1].astype(np.float64))
                sp.append(max(np.abs(p["qvel"][0] - s64["qvel"][e]).max(), np.abs(p["qpos"][0] - s64["qpos"][e]).max()))
            print(f"     oracle spread under 1e-6 perturbation: {np.array(sp)}")
    oracle.step(cm, st, act.astype(np.float64), nthreads=8)
