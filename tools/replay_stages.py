#!/usr/bin/env python3
"""GPU box helper (debug): for plain-ant states captured with status bit 4 (gpurun_in/nan_cases.npy), rebuild the four RK4 stage
states of the first mj_step on the host (stage k + 1 from the device's own qacc of stage k, as ant_mj_step does) and run each
through mz_debug_forward with growing iteration caps: which evaluation reaches the cap, and does the iterate still move?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
env_id = sys.argv[1] if len(sys.argv) > 1 else "AntUMaze-v0"
np.set_printoptions(precision=7, suppress=True, linewidth=220)
H = 0.02


def integrate(q0, vel, h):
    q = q0.copy()
    q[:3] = q0[:3] + h * vel[:3]
    quat = q0[3:7] / np.linalg.norm(q0[3:7]); w = vel[3:6]; nrm = np.linalg.norm(w)
    if nrm > 1e-15:
        ang = 0.5 * h * nrm; r = np.concatenate([[np.cos(ang)], w * np.sin(ang) / nrm])
        o = np.array([quat[0]*r[0]-quat[1]*r[1]-quat[2]*r[2]-quat[3]*r[3], quat[0]*r[1]+quat[1]*r[0]+quat[2]*r[3]-quat[3]*r[2],
                      quat[0]*r[2]-quat[1]*r[3]+quat[2]*r[0]+quat[3]*r[1], quat[0]*r[3]+quat[1]*r[2]-quat[2]*r[1]+quat[3]*r[0]])
        quat = o / np.linalg.norm(o)
    q[3:7] = quat
    q[7:] = q0[7:] + h * vel[6:]
    return q


cases = np.load("gpurun_in/nan_cases.npy", allow_pickle=True)
for c in cases:
    env = mm.make(env_id, num_envs=1, force_vec=True)
    act = torch.as_tensor(c["act"][None], device=env.device)
    q0, v0, warm = c["qpos"].astype(np.float64), c["qvel"].astype(np.float64), c["warm"].astype(np.float64)
    q, v = q0.copy(), v0.copy()
    for st in range(4):
        res = {}
        for cap in (10, 50, 51, 52, 100, 200):
            env.set_option("solver_iterations", cap)
            env.set_state(q[None].astype(np.float32), v[None].astype(np.float32), warm[None].astype(np.float32), np.array([c["t"]], np.int32))
            qacc, counts = env.debug_forward(act)
            res[cap] = (qacc[0].cpu().numpy().astype(np.float64), counts[0].tolist())
        qa = res[200][0]
        print(f"case step {c['step']} env {c['env']} stage {st}: ncon/iters at caps", {k: r[1] for k, r in res.items()},
              " |qacc| max", float(np.abs(qa).max()), " moved 50->51", float(np.abs(res[51][0] - res[50][0]).max()), " 51->52", float(np.abs(res[52][0] - res[51][0]).max()),
              " 100->200", float(np.abs(res[200][0] - res[100][0]).max()))
        aw = 1.0 if st == 2 else 0.5
        q = integrate(q0, aw * v, H); v = v0 + H * aw * qa; warm = qa
    env.close()
