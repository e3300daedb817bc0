#!/usr/bin/env python3
"""GPU box helper: which waves are slow?  Per-workgroup cycle totals of ONE step (instrumented kernel) against per-env
properties of the state the step started from (contacts, Newton iterations, near-wall flag, auto-reset)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
env_id = sys.argv[2] if len(sys.argv) > 2 else "AntUMaze-v0"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 16   # the default lane-group width of the plain ant at this batch size
EPW = 64 // lanes                                      # envs per wavefront (= per workgroup)
NW = n // EPW
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
if len(sys.argv) > 1: env.set_option("lanes_per_env", lanes)
env.reset(seed=1)
g = torch.Generator(device=env.device).manual_seed(0)
acts = [(torch.rand((n, 8), device=env.device, generator=g) * 60 - 30) for _ in range(16)]
for i in range(150): env.step(acts[i % 16])
env.set_option("profile_phases", 1)
env.step(acts[0]); env.phase_cycles(); env.wave_cycles(NW)
boxes = np.array(env.model.world.wall_boxes())  # x, y, z, hx, hy, hz
def near_wall(xy, reach=1.25):
    dx = np.maximum(np.abs(xy[:, None, 0] - boxes[None, :, 0]) - boxes[None, :, 3], 0.0)
    dy = np.maximum(np.abs(xy[:, None, 1] - boxes[None, :, 1]) - boxes[None, :, 4], 0.0)
    return ((dx * dx + dy * dy) < reach * reach).any(1)
rows = []
for k in range(8):
    nw = near_wall(env.get_state()[0][:, :2].cpu().numpy()).reshape(-1, EPW)
    qacc, counts = env.debug_forward(acts[(k + 1) % 16])   # ncon / iters of the first forward evaluation of the coming step
    c = counts.cpu().numpy()
    obs, rew, done, info = env.step(acts[(k + 1) % 16])
    cyc = env.wave_cycles(NW).astype(np.float64)
    wit = env.last_wave_newton_iters.astype(np.float64)
    env.phase_cycles()
    ncon = c[:, 0].reshape(-1, EPW); it = c[:, 1].reshape(-1, EPW)
    d = done.cpu().numpy().reshape(-1, EPW)
    rows.append(np.stack([cyc, ncon.max(1), ncon.sum(1), it.max(1), (d != 0).any(1), wit, nw.sum(1)], 1))
R = np.concatenate(rows)
cyc = R[:, 0]
print(f"waves {len(cyc)}: mean {cyc.mean():.0f} std {cyc.std():.0f} min {cyc.min():.0f} q50 {np.median(cyc):.0f} q99 {np.quantile(cyc,0.99):.0f} max {cyc.max():.0f}")
for name, col in (("max ncon of the wave", 1), ("sum ncon", 2), ("max first-eval iters", 3), ("auto-reset in wave", 4)):
    print(f"  corr(cycles, {name}) = {np.corrcoef(cyc, R[:, col])[0,1]:.3f}")
print(f"  Newton iterations per wave-step: mean {R[:,5].mean():.1f} std {R[:,5].std():.1f} min {R[:,5].min():.0f} max {R[:,5].max():.0f};  "
      f"corr(cycles, iterations of the wave in this step) = {np.corrcoef(cyc, R[:,5])[0,1]:.3f}")
A = np.stack([R[:, 5], np.ones(len(cyc))], 1)
coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
res = cyc - A @ coef
print(f"  cycles ~ {coef[0]:.0f} * iterations + {coef[1]:.0f};  residual std {res.std():.0f} (of {cyc.std():.0f})")
for v in range(0, EPW + 1):
    m = R[:, 6] == v
    if m.sum() > 5: print(f"  envs of the wave within the wall broad phase == {v}: {m.sum():5d} waves, mean cycles {cyc[m].mean():.0f}  std {cyc[m].std():.0f}")
A = np.stack([R[:, 5], R[:, 6], R[:, 2], np.ones(len(cyc))], 1)
coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
print(f"  cycles ~ {coef[0]:.0f} * iterations + {coef[1]:.0f} * near-wall envs + {coef[2]:.0f} * contacts + {coef[3]:.0f};  residual std {(cyc - A @ coef).std():.0f}")
for v in range(0, 9):
    m = R[:, 1] == v
    if m.sum() > 20: print(f"  max ncon == {v}: {m.sum():5d} waves, mean cycles {cyc[m].mean():.0f}  std {cyc[m].std():.0f}")
for v in range(0, 8):
    m = R[:, 3] == v
    if m.sum() > 20: print(f"  first-eval iters == {v}: {m.sum():5d} waves, mean cycles {cyc[m].mean():.0f}  std {cyc[m].std():.0f}")
# position in the launch: do late-numbered workgroups take longer (second wave of a SIMD)?
W = rows[-1][:, 0]
q = len(W) // 8
print("  mean cycles by workgroup-index octile:", [f"{W[i*q:(i+1)*q].mean():.0f}" for i in range(8)])
# is the spread the work or the hardware?  the same states and actions stepped twice: per-wave cycles of run 1 against run 2
qpos, qvel, warm, t = [x.clone() for x in env.get_state()]
env.set_auto_reset(False)
env.step(acts[3]); c1 = env.wave_cycles(NW).astype(np.float64); env.phase_cycles()
env.set_state(qpos, qvel, warm, t)
env.step(acts[3]); c2 = env.wave_cycles(NW).astype(np.float64); env.phase_cycles()
print(f"  identical step repeated: corr(run 1, run 2) = {np.corrcoef(c1, c2)[0,1]:.3f}; std of the difference {np.std(c1 - c2):.0f} (std of one run {c1.std():.0f})")
