#!/usr/bin/env python3
"""GPU box helper: WHICH PHASE makes the slowest waves slow?  Per-workgroup phase cycles of single steps (instrumented Ant kernel):
mean breakdown of all waves against that of the slowest 1 % and of the slowest wave of each step (the one a launch waits for).
    python tools/tail_phases.py [lanes [env id [envs]]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
env_id = sys.argv[2] if len(sys.argv) > 2 else "AntUMaze-v0"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
NW = n // (64 // lanes)
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
env.set_option("lanes_per_env", lanes)
env.reset(seed=1)
g = torch.Generator(device=env.device).manual_seed(0)
acts = [(torch.rand((n, env.nu), device=env.device, generator=g) * 60 - 30) for _ in range(16)]
for i in range(150): env.step(acts[i % 16])
env.set_option("profile_phases", 1)
env.step(acts[0]); env.phase_cycles(); env.wave_cycles(NW); env.wave_phase_cycles(NW)
names = ["K kinematics", "V bias", "C enum", "I+M", "solve:setup", "solve:grad+H", "solve:elim", "solve:vote/ls/update", "solve:tail", "rk4", "io", "C records", "(prologue)"]
allp, slow, nears, saved, subs = [], [], [], [], []
boxes = np.array(env.model.world.wall_boxes())  # x, y, z, hx, hy, hz
def near_wall(xy, reach):
    dx = np.maximum(np.abs(xy[:, None, 0] - boxes[None, :, 0]) - boxes[None, :, 3], 0.0)
    dy = np.maximum(np.abs(xy[:, None, 1] - boxes[None, :, 1]) - boxes[None, :, 4], 0.0)
    return ((dx * dx + dy * dy) < reach * reach).any(1)
for k in range(12):
    st0 = [x.cpu().numpy().copy() for x in env.get_state()]
    xy = st0[0][:, :2]
    nears.append(np.stack([near_wall(xy, 1.2).reshape(NW, -1).sum(1), near_wall(xy, 0.7).reshape(NW, -1).sum(1)], 1))
    env.step(acts[(k + 1) % 16])
    raw = env.wave_phase_cycles(NW).astype(np.float64)
    ph = raw[:, :13]
    subs.append(raw[:, 13:15])  # sub-timers of experiment builds (-DMZ_EXP_RK4TICK: rk4 arithmetic + stores | the hand-off behind them); 0 otherwise
    it = env.wave_cycles(NW); its = env.last_wave_newton_iters.astype(np.float64)
    env.phase_cycles()
    tot = ph.sum(1)
    allp.append(np.concatenate([ph, its[:, None]], 1))
    slow.append(np.concatenate([ph[np.argmax(tot)], [its[np.argmax(tot)]]]))
    wv = int(np.argmax(ph[:, 2])); epw = n // NW   # the wave with the most expensive contact enumeration: its envs' states, for offline analysis
    saved.append(dict(wave=wv, enum=ph[wv, 2], qpos=st0[0][wv * epw:(wv + 1) * epw], qvel=st0[1][wv * epw:(wv + 1) * epw], act=acts[(k + 1) % 16][wv * epw:(wv + 1) * epw].cpu().numpy()))
A = np.concatenate(allp); S = np.stack(slow)
tot = A[:, :13].sum(1)
top = A[tot >= np.quantile(tot, 0.99)]
print(f"{env_id} n={n} lanes={lanes}: waves {len(A)}  mean {tot.mean():.0f}  q50 {np.median(tot):.0f}  q90 {np.quantile(tot,0.9):.0f}  q99 {np.quantile(tot,0.99):.0f}  max {tot.max():.0f}; slowest of each step: mean {S[:, :13].sum(1).mean():.0f}")
print(f"{'phase':24s} {'all waves':>10s} {'slowest 1%':>11s} {'slowest/step':>13s}   (cycles per wave-step)")
for i, nm in enumerate(names):
    print(f"{nm:24s} {A[:, i].mean():10.0f} {top[:, i].mean():11.0f} {S[:, i].mean():13.0f}")
print(f"{'Newton iterations':24s} {A[:, 13].mean():10.1f} {top[:, 13].mean():11.1f} {S[:, 13].mean():13.1f}")
X = np.concatenate(subs)
if X.any():  # experiment build: slots 13 / 14 split the "rk4" slot (they are NOT part of the totals above)
    sel = tot >= np.quantile(tot, 0.99)
    for i, nm in enumerate(("sub 13: rk4 arithmetic+stores", "sub 14: hand-off behind it")):
        print(f"{nm:30s} {X[:, i].mean():10.0f} {X[sel, i].mean():11.0f}")
N = np.concatenate(nears)
for col, nm in ((0, "torso within 1.2 of a wall"), (1, "torso within 0.7 of a wall")):
    print(f"envs of the wave with the {nm}:")
    for v in range(0, 5):
        m = N[:, col] == v
        if m.sum() > 3: print(f"   {v}: {m.sum():5d} waves  total {tot[m].mean():8.0f}  C enum {A[m, 2].mean():7.0f}  C rows {A[m, 11].mean():7.0f}  grad+H {A[m, 5].mean():7.0f}  vote/ls {A[m, 7].mean():7.0f}  iters {A[m, 13].mean():.1f}")
os.makedirs("gpurun_out/dev_ant", exist_ok=True)
np.savez("gpurun_out/dev_ant/slow_enum_states.npz", enum=np.array([d["enum"] for d in saved]), qpos=np.stack([d["qpos"] for d in saved]), qvel=np.stack([d["qvel"] for d in saved]),
         act=np.stack([d["act"] for d in saved]))
