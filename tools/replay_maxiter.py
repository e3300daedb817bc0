#!/usr/bin/env python3
"""GPU box helper (debug): for states captured by tools/catch_nan.py with status bit 4 (gpurun_in/nan_cases.npy), find the
mj_step at which the Newton solver reaches its cap and look at how the iterate moves with the cap (stationary or cycling)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
env_id = sys.argv[1] if len(sys.argv) > 1 else "AntPush-v0"
np.set_printoptions(precision=7, suppress=True, linewidth=220)
cases = np.load("gpurun_in/nan_cases.npy", allow_pickle=True)
for c in cases:
    act = None
    for f in range(1, 6):
        env = mm.make(env_id, num_envs=1, force_vec=True)
        if os.environ.get("MZ_SOAK_RTOL"): env.set_option("solver_rtol", float(os.environ["MZ_SOAK_RTOL"]))
        env.set_option("debug_frame_skip", f)
        act = torch.as_tensor(c["act"][None], device=env.device)
        env.set_state(c["qpos"][None], c["qvel"][None], c["warm"][None], np.array([c["t"]], np.int32))
        env.step(act)
        bit = int(env.status()[0]) & 4
        if bit: break
        state = [x.cpu().numpy() for x in env.get_state()]
        env.close()
    print("case step", c["step"], "env", c["env"], "first frame with MAXITER:", f if bit else None)
    if not bit: continue
    # the state at the start of that frame; stage 0 evaluation with growing caps
    env = mm.make(env_id, num_envs=1, force_vec=True)
    if os.environ.get("MZ_SOAK_RTOL"): env.set_option("solver_rtol", float(os.environ["MZ_SOAK_RTOL"]))
    q0, v0, w0 = (c["qpos"][None], c["qvel"][None], c["warm"][None]) if f == 1 else state[:3]
    prev = None
    for cap in (5, 10, 20, 30, 40, 41, 42, 43, 44, 50, 100):
        env.set_option("solver_iterations", cap)
        env.set_state(q0, v0, w0, np.array([c["t"]], np.int32))
        qacc, counts = env.debug_forward(act)
        qa = qacc[0].cpu().numpy()
        print("   cap", cap, "ncon/iters", counts[0].tolist(), "max|qacc|", np.abs(qa).max(), "moved since previous cap", None if prev is None else float(np.abs(qa - prev).max()))
        prev = qa
    env.close()
