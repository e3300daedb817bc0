#!/usr/bin/env python3
"""GPU box helper (experiment): WHICH envs of AntPush need many Newton iterations, and what do their iterations look like?
Part 1 (product library, 64 lanes per env = one env per wave, instrumented kernel): per-env iteration counts of single steps; the
states in front of the steps with the most iterations (and two ordinary ones) are saved.  Part 2 (trace library, MZ_EXP_TRACE:
csrc/libmazestep_dev.so built `make dev1 DEVFLAGS=-DMZ_EXP_TRACE`): those states stepped alone, one line per Newton iteration.
    python tools/exp_push_iters.py [env id [envs]]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
env_id = sys.argv[1] if len(sys.argv) > 1 else "AntPush-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
if os.environ.get("MZ_PART") != "2":
    import numpy as np, torch
    import mujoco_maze_amd as mm
    env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
    env.set_option("lanes_per_env", 64)
    env.reset(seed=1)
    g = torch.Generator(device=env.device).manual_seed(0)
    acts = [(torch.rand((n, env.nu), device=env.device, generator=g) * 60 - 30) for _ in range(16)]
    for i in range(150): env.step(acts[i % 16])
    env.set_option("profile_phases", 1)
    env.step(acts[0]); env.phase_cycles(); env.wave_cycles(n)
    cases, hist = [], []
    for k in range(8):
        st = [x.cpu().numpy().copy() for x in env.get_state()]
        a = acts[(k + 1) % 16]
        env.step(a)
        env.wave_cycles(n); its = env.last_wave_newton_iters.astype(np.int64); env.phase_cycles()
        hist.append(its)
        order = np.argsort(its)
        for e in list(order[-2:]) + [order[n // 2]]:
            cases.append(dict(step=k, env=int(e), iters=int(its[e]), qpos=st[0][e], qvel=st[1][e], warm=st[2][e], t=int(st[3][e]), act=a[e].cpu().numpy()))
    its = np.concatenate(hist)
    print(f"{env_id}: iterations per env-step: mean {its.mean():.1f}  q50 {np.quantile(its, .5):.0f}  q90 {np.quantile(its, .9):.0f}  q99 {np.quantile(its, .99):.0f}  max {its.max()}")
    print("histogram (iterations // 10):", np.bincount(its // 10))
    # does an env's count persist from step to step?
    h = np.stack(hist).astype(np.float64)
    print("corr(iters[t], iters[t+1]) per env: %.3f" % np.corrcoef(h[:-1].ravel(), h[1:].ravel())[0, 1])
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(root, "gpurun_out/push_iter_cases.npy"), np.array(cases, dtype=object), allow_pickle=True)
    sub = subprocess.run([sys.executable, __file__, env_id], env=dict(os.environ, MZ_PART="2", MZ_DEBUG="1", MZ_LIBMAZESTEP_EXPERIMENT=os.path.join(root, "mujoco_maze_amd/csrc/libmazestep_dev.so")),
                         capture_output=True, text=True, cwd=root)
    print(sub.stdout[-60000:])
    print(sub.stderr[-2000:])
else:
    import numpy as np, torch
    import mujoco_maze_amd as mm
    cases = np.load(os.path.join(root, "gpurun_out/push_iter_cases.npy"), allow_pickle=True)
    for c in cases:
        if c["step"] > 2: continue
        env = mm.make(env_id, num_envs=1, force_vec=True)
        env.set_option("lanes_per_env", 32)
        env.set_state(c["qpos"][None], c["qvel"][None], c["warm"][None], np.array([c["t"]], np.int32))
        print(f"CASE step {c['step']} env {c['env']} iters {c['iters']}  block q {c['qpos'][15:17]} v {c['qvel'][14:16]}  torso {c['qpos'][:3]}", flush=True)
        env.step(torch.as_tensor(c["act"][None], device=env.device))
        torch.cuda.synchronize()
        sys.stdout.flush()
        print("status", int(env.status()[0]), flush=True)
