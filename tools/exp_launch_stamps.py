#!/usr/bin/env python3
"""GPU box helper (experiment library built with -DMZ_EXP_STAMPS: make -C mujoco_maze_amd/csrc dev DEVFLAGS=-DMZ_EXP_STAMPS): every wave
of the PRODUCT plain-ant kernel stamps its start and its end (s_memrealtime, 100 MHz) — what are the last 100 k cycles of a launch?
Start spread, duration spread, and for the wave that ends last: how late it started, how long it ran, where it ran."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = mm.make("AntUMaze-v0", num_envs=n, auto_reset=True, force_vec=True)
env.reset(seed=20260928)
g = torch.Generator(device=env.device).manual_seed(1234)
pool = [torch.rand((n, 8), device=env.device, generator=g) * 60 - 30 for _ in range(32)]
for i in range(200): env.step(pool[i % 32])
env.set_option("profile_phases", 1)
nw = (n + 3) // 4
rows = []
feats = []
gen_total, run_total = [], []
cyc_rows = []
for i in range(200):
    env.step(pool[i % 32])
    w = env.wave_phase_cycles(nw).astype(np.int64)
    st, en, hw, xcc = w[:, 0], w[:, 1], w[:, 2], w[:, 3]
    its, ncs = w[:, 4].astype(float), w[:, 5].astype(float)
    dur = (en - st).astype(float)
    gen_total.append(int(w[:, 9].max())); run_total.append(int(w[:, 6].sum()))
    cyc_rows.append(np.stack([w[:, 10].astype(float), w[:, 6].astype(float), w[:, 11].astype(float)], 1))
    feats.append(np.stack([dur, its, ncs, w[:, 6].astype(float), w[:, 7].astype(float), w[:, 8].astype(float)], 1))
    t0 = st.min()
    last = int(np.argmax(en))
    rows.append(dict(span=(en.max() - t0), start_spread=(st.max() - t0), dur_mean=(en - st).mean(), dur_max=(en - st).max(), last_start=(st[last] - t0), last_dur=(en[last] - st[last]),
                     last_is_longest=int(np.argmax(en - st) == last), n_after_90=int((en - t0 > 0.9 * (en.max() - t0)).sum()), xcc_last=int(xcc[last] & 15),
                     start_by_xcc=[int(np.median(st[(xcc & 15) == x] - t0)) if ((xcc & 15) == x).any() else -1 for x in range(8)]))
tick = 10.0  # ns per s_memrealtime tick (100 MHz)
f = lambda k: np.array([r[k] for r in rows], float)
print(f"AntUMaze-v0, {n} envs, {nw} waves, 200 launches; times in microseconds (s_memrealtime)")
print(f"  launch span (first start -> last end): mean {f('span').mean() * tick / 1e3:.1f}  | wave duration: mean {f('dur_mean').mean() * tick / 1e3:.1f}, max {f('dur_max').mean() * tick / 1e3:.1f}")
print(f"  start spread (last start - first start): mean {f('start_spread').mean() * tick / 1e3:.1f}, max {f('start_spread').max() * tick / 1e3:.1f}")
print(f"  the wave that ends last: started {f('last_start').mean() * tick / 1e3:.1f} after the first, ran {f('last_dur').mean() * tick / 1e3:.1f}; it is the longest-running wave in {100 * f('last_is_longest').mean():.0f} % of the launches")
print(f"  waves still running in the last 10 % of the span: mean {f('n_after_90').mean():.1f} of {nw}")
print(f"  median start offset by XCC id (ticks of 10 ns): {np.median(np.array([r['start_by_xcc'] for r in rows]), axis=0)}")
F = np.concatenate(feats)
dur, its, ncs, npass, nover, pmax = F[:, 0] * tick / 1e3, F[:, 1], F[:, 2], F[:, 3], F[:, 4], F[:, 5]
A = np.stack([np.ones_like(its), its, ncs, npass, nover], 1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
res = dur - A @ coef
print(f"  per wave over all launches: duration = {coef[0]:.1f} + {coef[1]:.3f} x (lock-step Newton iterations: mean {its.mean():.1f}, 99 % {np.quantile(its, .99):.0f}) + {coef[2]:.3f} x (contact-evaluations of its busiest env: mean {ncs.mean():.1f}, 99 % {np.quantile(ncs, .99):.0f}) + {coef[3]:.3f} x (wall narrow-phase runs of its envs' geoms: mean {npass.mean():.1f}, 99 % {np.quantile(npass, .99):.0f}) + {coef[4]:.2f} x (evaluations on the staging-overflow fall-back: mean {nover.mean():.3f}) us; residual std {res.std():.1f}; corr(dur, iters) {np.corrcoef(dur, its)[0, 1]:.2f}, corr(dur, contacts) {np.corrcoef(dur, ncs)[0, 1]:.2f}")
per = len(feats[0])
slow = np.array([fe[np.argmax(fe[:, 0])] for fe in feats])
print(f"  the slowest wave of a launch: {slow[:, 0].mean() * tick / 1e3:.1f} us, {slow[:, 1].mean():.1f} iterations (all waves: {its.mean():.1f}), {slow[:, 2].mean():.1f} contact-evaluations (all: {ncs.mean():.1f}), {slow[:, 3].mean():.1f} narrow-phase runs (all: {npass.mean():.1f}), {slow[:, 4].mean():.2f} fall-back evaluations (all: {nover.mean():.3f})")
for lo, hi in ((0, 1), (1, 20), (20, 60), (60, 150), (150, 100000)):
    m = (npass >= lo) & (npass < hi)
    if m.sum(): print(f"     waves with {lo:3d}-{hi:6d} wall narrow-phase runs per step: {100 * m.mean():5.1f} %  duration mean {dur[m].mean():.1f} us  max {dur[m].max():.1f}  iterations {its[m].mean():.1f}")
for lo, hi in ((0, 45), (45, 60), (60, 80), (80, 120), (120, 1000)):
    m = (ncs >= lo) & (ncs < hi)
    if m.sum(): print(f"     waves whose busiest env has {lo:3d}-{hi:3d} contact-evaluations per step: {100 * m.mean():5.1f} %  duration mean {dur[m].mean():.1f} us  iterations {its[m].mean():.1f}")
print(f"  capsule / sphere tests against wall boxes over the 200 launches: {sum(run_total)}, of which capsule tests outside the face case: {gen_total[-1] - gen_total[0] + 0} (device-wide counter, first to last launch)")

cr = np.concatenate(cyc_rows)
m = cr[:, 1] > 0
if m.any():
    print(f"  shader cycles between the cell early-out and the end of the narrow phase (waves with runs): {cr[m, 0].sum() / cr[m, 1].sum():.0f} per run "
          f"(= {cr[m, 0].sum() / cr[m, 1].sum() / 2400:.2f} us at 2.4 GHz); waves without a run but past the early-out: mean {cr[~m, 0].mean():.0f} cycles per step; of the region, the candidate scan: {100 * cr[m, 2].sum() / cr[m, 0].sum():.0f} % in waves with runs")
