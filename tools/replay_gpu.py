#!/usr/bin/env python3
"""GPU box helper (debug): find the RK4 stage of a captured state (gpurun_in/nan_cases.npy: copy tools/catch_nan.py's gpurun_out/nan_cases.npy there — gpurun_out/ does not travel to the box) at which the device goes non-finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import emu_lib, oracle_lib
env_id = sys.argv[1] if len(sys.argv) > 1 else "AntPushMaze-v0"
c = np.load("gpurun_in/nan_cases.npy", allow_pickle=True)[0]
np.set_printoptions(precision=6, suppress=True, linewidth=220)
env = mm.make(env_id, num_envs=1, force_vec=True)
cm = env.model
act = torch.as_tensor(c["act"][None], device=env.device)
state = (c["qpos"][None].copy(), c["qvel"][None].copy(), c["warm"][None].copy())
bad_frame = None
for f in range(1, 6):
    env.set_option("debug_frame_skip", f)
    env.set_state(state[0], state[1], state[2], np.array([c["t"]], np.int32))
    env.step(act)
    x = env.get_state()[0][0, 0].item()
    print("frames", f, "x'", x)
    if not np.isfinite(x):
        bad_frame = f
        break
env.set_option("debug_frame_skip", bad_frame - 1)
if bad_frame > 1:
    env.set_state(state[0], state[1], state[2], np.array([c["t"]], np.int32)); env.step(act)
    q0, v0, w0, _ = [x.cpu().numpy().astype(np.float64) for x in env.get_state()]
else:
    q0, v0, w0 = [s.astype(np.float64) for s in state]
print("state at the start of the bad frame: qpos", q0[0], "qvel", v0[0])
np.save("gpurun_out/bad_frame_state.npy", dict(qpos=q0[0], qvel=v0[0], warm=w0[0], act=c["act"]), allow_pickle=True)
h = 0.02
def quat_mul(a, b):
    return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2], a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])
def integrate(q, v, hh):
    out = q.copy(); out[:3] += hh * v[:3]
    w = v[3:6]; n = np.linalg.norm(w); qq = q[3:7] / np.linalg.norm(q[3:7])
    if n > 1e-15:
        ang = 0.5 * hh * n; qq = quat_mul(qq, np.concatenate([[np.cos(ang)], np.sin(ang) * w / n])); qq /= np.linalg.norm(qq)
    out[3:7] = qq; out[7:] += hh * v[6:]
    return out
qs, vs = q0[0].copy(), v0[0].copy()
for stage in range(4):
    env.set_state(qs[None], vs[None], w0, np.array([c["t"]], np.int32))
    qacc, counts = env.debug_forward(act)
    qa = qacc[0].cpu().numpy().astype(np.float64)
    ef = emu_lib.forward(cm, qs[None], vs[None], c["act"][None], w0)
    if stage == 0: print("  dev qacc", qa, "\n  emu qacc", ef["qacc"][0])
    print("stage", stage, "device ncon/iters", counts[0].tolist(), "finite", bool(np.isfinite(qa).all()), "| emu ncon", ef["counts"][0].tolist(), "max |dev - emu| qacc", float(np.abs(qa - ef["qacc"][0]).max()))
    if not np.isfinite(qa).all():
        np.save("gpurun_out/nan_stage_state.npy", dict(qpos=qs, qvel=vs, warm=w0[0], act=c["act"]), allow_pickle=True)
        print("  qpos", qs, "\n  qvel", vs, "\n  device qacc", qa, "\n  emu qacc", ef["qacc"][0])
        break
    aw = 1.0 if stage == 2 else 0.5
    qs, vs = integrate(q0[0], aw * vs, h), v0[0] + h * aw * qa
