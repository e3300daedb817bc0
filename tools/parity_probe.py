#!/usr/bin/env python3
"""GPU box helper: the numbers behind the thresholds of tests/test_gpu_parity.py (run, read, then set the tests to <= 2x)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib
from tests.test_gpu_parity import _rollout_states, _f32, _close

oracle = oracle_lib.load()


def sensitive(cm, start, act, ref_qvel, e, atol, rng):
    spread = 0.0
    for _ in range(12):
        p = {k: v[e:e + 1].copy() for k, v in start.items()}
        p["qpos"] = p["qpos"] + rng.uniform(-2e-7, 2e-7, p["qpos"].shape) * np.maximum(1.0, np.abs(p["qpos"]))
        p["qvel"] = p["qvel"] + rng.uniform(-2e-7, 2e-7, p["qvel"].shape) * np.maximum(1.0, np.abs(p["qvel"]))
        oracle.step(cm, p, act[e:e + 1].astype(np.float64))
        spread = max(spread, np.abs(p["qvel"] - ref_qvel[e]).max())
    return spread


def report(name, env, cm, st, act, atol=1e-5):
    start = {k: v.copy() for k, v in st.items()}
    env.set_state(st["qpos"], st["qvel"], st["warm"], st["t"])
    obs, rew, done, info = env.step(torch.as_tensor(act, device=env.device))
    qpos, qvel, warm, t = [x.cpu().numpy() for x in env.get_state()]
    ref = oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    ok = np.all(_close(qpos, st["qpos"], atol=atol), axis=1) & np.all(_close(qvel, st["qvel"], atol=atol), axis=1)
    bad = np.where(~ok)[0]
    rng = np.random.default_rng(1)
    smooth = [int(e) for e in bad if sensitive(cm, start, act, st["qvel"], e, atol, rng) <= atol]
    per = np.maximum((np.abs(qvel - st["qvel"]) / (1 + np.abs(st["qvel"]))).max(1), (np.abs(qpos - st["qpos"]) / (1 + np.abs(st["qpos"]))).max(1))
    print(f"{name}: n={len(ok)} outliers={len(bad)} ({100*len(bad)/len(ok):.3f} %) of which oracle-smooth={len(smooth)} "
          f"median={np.median(per):.2e} q99={np.quantile(per,0.99):.2e} max_ok={per[ok].max():.2e} "
          f"done_mismatch={(done.cpu().numpy()!=ref['done']).sum()} obs_err_ok={np.abs(obs.cpu().numpy()-ref['obs'])[ok].max():.2e} "
          f"rew_err_ok={np.abs(rew.cpu().numpy()-ref['reward'])[ok].max():.2e}", flush=True)
    if smooth:
        print("   smooth outliers:", smooth[:10], [f"{per[e]:.2e}" for e in smooth[:10]])


# 1. debug_forward
n = 256
env = mm.make("AntUMaze-v0", num_envs=n); cm = env.model
st = _rollout_states(oracle, cm, n, 3, {30})[30]
act = np.random.default_rng(1).uniform(-30, 30, (n, 8)).astype(np.float32)
env.set_state(st["qpos"], st["qvel"], st["warm"], st["t"])
qacc, counts = env.debug_forward(act)
ref = oracle.forward(cm, st["qpos"], st["qvel"], act.astype(np.float64), st["warm"])
err = np.abs(qacc.cpu().numpy() - ref["qacc"])
print("debug_forward: max abs err", err.max(), "max |qacc|", np.abs(ref["qacc"]).max(), "max rel-ish", (err / (1 + np.abs(ref["qacc"]))).max(), flush=True)
env.close()

# 2. lane widths
n = 128
for g in (8, 16, 32, 64):
    env = mm.make("AntUMaze-v0", num_envs=n); env.set_option("lanes_per_env", g); cm = env.model
    s = _rollout_states(oracle, cm, n, 21, {20})[20]
    act = np.random.default_rng(2).uniform(-30, 30, (n, 8)).astype(np.float32)
    report(f"lanes {g}", env, cm, s, act)
    env.close()

# 3. wall contacts
from tests.test_gpu_parity import _place_ant
n = 64
env = mm.make("AntUMaze-v0", num_envs=n); cm = env.model
st = _rollout_states(oracle, cm, n, 8, {40})[40]
rng = np.random.default_rng(4)
st["qpos"][: n // 2, 0] = 18.9 + rng.uniform(0.0, 0.5, n // 2)
st["qpos"][: n // 2, 1] = rng.uniform(-1, 1, n // 2)
st["qvel"][: n // 2, 0] = 2.0
st["qpos"][n // 2:, 0] = rng.uniform(-0.5, 0.5, n - n // 2)
st["qpos"][n // 2:, 1] = 16.0 + rng.uniform(-0.7, 0.7, n - n // 2)
st = _f32(st)
act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
report("wall contacts", env, cm, st, act)
env.close()

# 4. AntPush
n = 2048
env = mm.make("AntPush-v0", num_envs=n); cm = env.model
st, _ = oracle.reset(cm, n, 5)
rng = np.random.default_rng(0)
st["qpos"][: n // 3, 1] = 3.0 + rng.uniform(0.2, 0.6, n // 3)
st["qvel"][: n // 3, 1] = 1.5
for k in range(12):
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    if k in (2, 11):
        report(f"AntPush k={k}", env, cm, _f32(st), act)
    oracle.step(cm, st, act.astype(np.float64), nthreads=8)
env.close()

# 5. multi-block
for env_id, nblock in (("AntMultiPush-v0", 2), ("AntPushMaze-v0", 3)):
    n = 512
    env = mm.make(env_id, num_envs=n, maze_size_scaling=2.0); cm = env.model
    st, _ = oracle.reset(cm, n, 5)
    rng = np.random.default_rng(0)
    st["qvel"][:, 14:] = rng.uniform(-3, 3, (n, 2 * nblock))
    for k in range(21):
        act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
        if k in (0, 5, 20):
            report(f"{env_id} k={k}", env, cm, _f32(st), act)
        oracle.step(cm, st, act.astype(np.float64), nthreads=8)
    env.close()
