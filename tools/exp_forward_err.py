#!/usr/bin/env python3
"""GPU box helper: error of ONE forward-dynamics evaluation (device qacc against the float64 oracle's) per dof, by number of robot
contacts, for AntUMaze-v0 (solimp .8) and AntPush-v0 (solimp .995): where does the stiff mazes' wider error tail enter?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib
oracle = oracle_lib.load()
np.set_printoptions(precision=1, linewidth=220, suppress=True)
n = 4096
for env_id in ("AntUMaze-v0", "AntPush-v0"):
    env = mm.make(env_id, num_envs=n, force_vec=True)
    cm = env.model
    rng = np.random.default_rng(11)
    st, _ = oracle.reset(cm, n, 11)
    for k in range(31):
        act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
        if k in (10, 30):
            s = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
            env.set_state(s["qpos"], s["qvel"], s["warm"], s["t"])
            qacc, counts = env.debug_forward(torch.as_tensor(act, device=env.device))
            ref = oracle.forward(cm, s["qpos"], s["qvel"], act.astype(np.float64), s["warm"])
            qa, rq = qacc.cpu().numpy()[:, :14].astype(np.float64), ref["qacc"][:, :14]
            nb = 4 if env_id == "AntPush-v0" else 0  # the block's own floor contacts
            nc = ref["counts"][:, 0] - nb
            same = counts.cpu().numpy()[:, 0] == ref["counts"][:, 0]
            err = np.abs(qa - rq)
            print(f"{env_id} step {k}: contact counts agree on {same.mean():.3f}")
            for c in range(0, 7):
                m = (nc == c) & same
                if m.sum() < 20: continue
                print(f"   {c} robot contacts ({m.sum():4d} envs): |dqacc| median per dof {np.median(err[m], axis=0) * 1e3} 99 % {np.quantile(err[m], .99, axis=0) * 1e3}  (1e-3; |qacc| median {np.median(np.abs(rq[m]), axis=0)})")
        oracle.step(cm, st, act.astype(np.float64), nthreads=16)
    env.close()
