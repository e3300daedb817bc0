#!/usr/bin/env python3
"""GPU box helper (debug): step the states of gpurun_in/nan_cases.npy once with the trace experiment library (csrc/exp_TRACE.so,
built -DMZ_EXP_TRACE: the row solver prints its iterates from the 7th iteration on)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import numpy as np, torch, mujoco_maze_amd as mm
cases = np.load('gpurun_in/nan_cases.npy', allow_pickle=True)
for c in cases:
    env = mm.make('%s', num_envs=1, force_vec=True)
    env.set_option('debug_frame_skip', 1)
    env.set_state(c['qpos'][None], c['qvel'][None], c['warm'][None], np.array([c['t']], np.int32))
    env.step(torch.as_tensor(c['act'][None], device=env.device))
    torch.cuda.synchronize()
    print('status', int(env.status()[0]))
""" % (sys.argv[1] if len(sys.argv) > 1 else "AntUMaze-v0")
env = dict(os.environ, MZ_DEBUG="1", MZ_LIBMAZESTEP_EXPERIMENT=os.path.join(root, "mujoco_maze_amd/csrc/exp_TRACE.so"))
out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True)
lines = [l for l in out.stdout.splitlines() if l.startswith(("TRACE", "status"))]
print("\n".join(lines))
print("...", len(lines), "lines")
