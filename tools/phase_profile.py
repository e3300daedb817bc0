#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""GPU box helper: per-phase shader-cycle breakdown of ant_step_kernel (instrumented build)."""
import sys, json
import torch
import mujoco_maze_amd as mm
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
env_id = sys.argv[2] if len(sys.argv) > 2 else "AntUMaze-v0"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
env.set_option("lanes_per_env", lanes)
env.reset(seed=1)
g = torch.Generator(device=env.device).manual_seed(0)
acts = [(torch.rand((n, 8), device=env.device, generator=g) * 60 - 30) for _ in range(16)]
for i in range(100): env.step(acts[i % 16])
env.set_option("profile_phases", 1)
wgs0 = n // (64 // lanes)
env.step(acts[0]); env.wave_cycles(wgs0); env.wave_phase_cycles(wgs0)  # (these two reads clear the per-workgroup slots phase_cycles sums)
steps = 50
for i in range(steps): env.step(acts[i % 16])
cyc = env.phase_cycles()
# plain ant (row solver): slot 4 = limit rows + initial guess, 5 = curvature publish + gradient / Hessian rows, 6 = row elimination,
# 7 = J search, vote, line search, update; 11 = contact rows; 12 = row of M + qacc_smooth by the row elimination
names = ["P0 kinematics+broadphase", "P1 inertia|contact count", "P2 crb|bias legs|contact fill", "P3 hub M|bias dofs", "solve:limits+start", "solve:grad+H rows", "solve:elimination", "solve:vote/linesearch/update", "solve:tail",
         "rk4/integrate", "io+epilogue", "P4 contact rows", "P5 row of M -> qacc_smooth"]
if env.model.c.nblock == 0 and env.model.c.nball == 0 and lanes >= 16:
    # the plain ant since round 4 (ant_forward_rows.h: everything before the solver in the registers of the leg quads); the compiler
    # moves arithmetic across the timer reads, so neighbouring pre-solver slots blur into each other by a few hundred cycles
    names[:4] = ["K kinematics (quads)", "V bias forces (RNE)", "C contact enumeration", "I+M inertias, mass matrix rows"]
    names[11] = "contact records (+ fall-back)"
    names[12] = "(solver prologue)"
wgs = n // (64 // lanes)
tot = sum(cyc[:13])
print(f"{env_id} n={n} lanes={lanes}  total cycles/step/wave = {tot/steps/wgs:.0f}   newton iters per forward eval (mean over group 0 envs) = {cyc[15]/steps/wgs/20:.2f}")
for k, nm in enumerate(names):
    print(f"  {nm:36s} {cyc[k]/steps/wgs:10.0f} cycles/step  {100*cyc[k]/tot:5.1f} %")
import math
# since round 6 slots 13 / 14 are reduced on the host from per-workgroup sums over the WINDOW (no same-address atomics in the kernel): the
# most loaded workgroup's window total and the sum of squares of the window totals — per-STEP spreads: tools/tail_phases.py
mean = tot / steps / wgs
sq = cyc[14] * 65536.0 / wgs / (steps * steps)
std = math.sqrt(max(0.0, sq - mean * mean))
print(f"  per-wave total per step, averaged over the {steps}-step window: mean {mean:.0f}  std across waves {std:.0f}  most loaded wave {cyc[13]/steps:.0f}  "
      f"(x {cyc[13]/steps/mean:.2f}); the per-step slowest wave — what a launch waits for: tools/tail_phases.py")
