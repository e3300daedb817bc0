#!/usr/bin/env python3
"""GPU box helper: where does AntPush's 1e-5 tail come from?  Single-step error quantiles under solver settings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib
oracle = oracle_lib.load()
env_id, n, checks = "AntPush-v0", 2048, (0, 10, 50, 100)
for opts in ({}, {"solver_rtol": 1e-7}, {"solver_rtol": 0.0, "solver_tolerance": 1e-12}, {"lanes_per_env": 64}):
    env = mm.make(env_id, num_envs=n, force_vec=True)
    for k, v in opts.items(): env.set_option(k, v)
    cm = env.model
    rng = np.random.default_rng(11)
    st, _ = oracle.reset(cm, n, 11)
    errs, which = [], []
    for k in range(max(checks) + 1):
        act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
        if k in checks:
            s = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
            env.set_state(s["qpos"], s["qvel"], s["warm"], s["t"])
            env.step(torch.as_tensor(act, device=env.device))
            qpos, qvel, _, _ = [x.cpu().numpy() for x in env.get_state()]
            oracle.step(cm, s, act.astype(np.float64), nthreads=16)
            ev = np.abs(qvel - s["qvel"]) / (1 + np.abs(s["qvel"]))
            errs.append(ev.max(1)); which.append(ev.argmax(1))
        oracle.step(cm, st, act.astype(np.float64), nthreads=16)
    e = np.concatenate(errs); w = np.concatenate(which)
    big = e > 5e-6
    print(f"{opts}: median {np.median(e):.1e} 99% {np.quantile(e, .99):.1e} 99.9% {np.quantile(e, .999):.1e} max {e[e < 1e-3].max():.1e}; envs > 1e-5: {(e > 1e-5).sum()}; dof of the largest error among envs > 5e-6: {np.bincount(w[big], minlength=16)}")
    env.close()
# the worst envs of the default settings: what do they touch?
env = mm.make(env_id, num_envs=n, force_vec=True)
cm = env.model
rng = np.random.default_rng(11)
st, _ = oracle.reset(cm, n, 11)
np.set_printoptions(precision=3, linewidth=220, suppress=True)
for k in range(max(checks) + 1):
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    if k in checks:
        s = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
        s0 = {kk: v.copy() for kk, v in s.items()}
        env.set_state(s["qpos"], s["qvel"], s["warm"], s["t"])
        env.step(torch.as_tensor(act, device=env.device))
        qpos, qvel, _, _ = [x.cpu().numpy() for x in env.get_state()]
        oracle.step(cm, s, act.astype(np.float64), nthreads=16)
        ev = np.abs(qvel - s["qvel"]) / (1 + np.abs(s["qvel"]))
        for e in np.where(ev.max(1) > 8e-6)[0][:6]:
            c = oracle.contacts(cm, s0["qpos"][e])
            print(f"step {k} env {e}: err {ev[e].max():.1e} at dof {ev[e].argmax()}; torso xyz {s0['qpos'][e, :3]} block {s0['qpos'][e, 15:]}; |qvel| max {np.abs(s0['qvel'][e]).max():.1f}; contacts (geom1, geom2, dist): {[(int(r[7]), int(r[8]), round(r[0], 4)) for r in c]}")
            print(f"      err per dof {ev[e] * 1e6} (1e-6)")
    oracle.step(cm, st, act.astype(np.float64), nthreads=16)
