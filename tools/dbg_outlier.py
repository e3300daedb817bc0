import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
np.set_printoptions(linewidth=220, precision=6, suppress=True)
import mujoco_maze_amd as mm
n = 4096
A = mm.make("AntUMaze-v0", num_envs=n, force_vec=True)            # rows solver (32 lanes)
B = mm.make("AntUMaze-v0", num_envs=n, force_vec=True); B.set_option("lanes_per_env", 8)   # generic solver
A.reset(seed=3)
g = torch.Generator(device=A.device).manual_seed(0)
tot = 0; bad_total = 0
for k in range(200):
    act = torch.rand((n, 8), device=A.device, generator=g) * 60 - 30
    qp, qv, wm, t = A.get_state()
    B.set_state(qp, qv, wm, t)
    _, counts = A.debug_forward(act)
    A.step(act); B.step(act)
    va, vb = A.get_state()[1], B.get_state()[1]
    err = (va - vb).abs().max(dim=1).values
    bad = (err > 1e-4).nonzero().flatten()
    tot += n; bad_total += len(bad)
    if len(bad) and k > 20:
        c = counts.cpu().numpy()
        print("step", k, "mismatches", len(bad), "errs", err[bad].cpu().numpy()[:6], "ncon/iters first eval", c[bad.cpu().numpy()][:6].tolist())
print("total env-steps", tot, "rows-vs-generic mismatches > 1e-4:", bad_total)
