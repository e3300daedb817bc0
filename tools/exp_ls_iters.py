#!/usr/bin/env python3
"""GPU box helper (experiment): what does the Ant solver's exact line search buy?  The same rollout under `ls_iterations` = 12 (product), 4, 3,
2, 1, 0 (0: every Newton step is a unit step): kernel time per step, Newton iterations per step (instrumented pass), status bits, and the drift of
the observations from the product setting after a short rollout (a converged solve does not depend on how it got there).
    python tools/exp_ls_iters.py [env id [envs]]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
import mujoco_maze_amd as mm
env_id = sys.argv[1] if len(sys.argv) > 1 else "AntUMaze-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ref = None
# (fast iterations k, evaluations in them): the first k Newton iterations of an evaluation search the line with at most that many evaluations
SETTINGS = [(0, 0), (1, 0), (2, 0), (3, 0), (50, 0), (1, 1), (2, 1), (3, 1), (50, 1)]
if os.environ.get("MZ_LS_SETTINGS"):  # "3:0,4:0,6:0"
    SETTINGS = [tuple(int(x) for x in kv.split(":")) for kv in os.environ["MZ_LS_SETTINGS"].split(",")]
for kf, ls in SETTINGS:
    env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
    env.set_option("ls_fast_iterations", kf); env.set_option("ls_fast", ls)
    env.reset(seed=1)
    g = torch.Generator(device=env.device).manual_seed(0)
    acts = [(torch.rand((n, env.nu), device=env.device, generator=g) * 2 - 1) * 30 for _ in range(16)]
    for i in range(40): out = env.step(acts[i % 16])
    obs40 = out[0].float().cpu().numpy().copy()
    for i in range(160): env.step(acts[i % 16])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(300): env.step(acts[i % 16])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    st = env.status().cpu().numpy()
    env.set_option("profile_phases", 1)
    env.step(acts[0]); env.phase_cycles(); env.wave_cycles(n)
    its = []
    for k in range(8):
        env.step(acts[(k + 1) % 16]); env.wave_cycles(n); its.append(env.last_wave_newton_iters.astype(np.int64).copy()); env.phase_cycles()
    its = np.concatenate(its)
    if ref is None: ref = obs40
    d = np.abs(obs40 - ref)
    print(f"{env_id} {n} envs  fast iterations {kf:2d} with {ls} evaluations: {ms:.4f} ms/step (host loop)  iterations per wave-step mean {its.mean():.1f} q99 {np.quantile(its, .99):.0f} max {its.max()}"
          f"  status bits set {np.bitwise_or.reduce(st):#x} in {int((st != 0).sum())} envs  |obs - obs(product)| after 40 steps: q99.9 {np.quantile(d, .999):.2e} max {d.max():.2e}", flush=True)
