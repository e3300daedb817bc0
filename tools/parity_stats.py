#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""GPU box helper: worst / quantile single-step errors vs the oracle (AntUMaze-v0, rollout checkpoints)."""
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib
oracle = oracle_lib.load()
n = 1024
env = mm.make("AntUMaze-v0", num_envs=n)
cm = env.model
rng = np.random.default_rng(11)
st, _ = oracle.reset(cm, n, 11)
worst_v = worst_q = 0.0; q99 = []
for k in range(101):
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    if k in (0, 1, 10, 50, 100):
        s = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
        env.set_state(s["qpos"], s["qvel"], s["warm"], s["t"])
        env.step(torch.as_tensor(act, device=env.device))
        qpos, qvel, _, _ = [x.cpu().numpy() for x in env.get_state()]
        oracle.step(cm, s, act.astype(np.float64), nthreads=8)
        ev = np.abs(qvel - s["qvel"]).max(1); eq = np.abs(qpos - s["qpos"]).max(1)
        worst_v = max(worst_v, ev.max()); worst_q = max(worst_q, eq.max()); q99.append(np.quantile(ev, 0.99))
    oracle.step(cm, st, act.astype(np.float64), nthreads=8)
print(f"worst |dqvel| {worst_v:.2e}  worst |dqpos| {worst_q:.2e}  99% quantile of |dqvel| per checkpoint {['%.1e' % x for x in q99]}")
