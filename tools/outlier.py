import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
np.set_printoptions(linewidth=220, precision=6, suppress=True)
import mujoco_maze_amd as mm
from tests import oracle_lib, emu_lib
oracle = oracle_lib.load()
n = 1024
env = mm.make("AntUMaze-v0", num_envs=n)
cm = env.model
rng = np.random.default_rng(11)
st, _ = oracle.reset(cm, n, 11)
for k in range(101):
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    if k in (0, 1, 10, 50, 100):
        s = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
        s0 = {kk: v.copy() for kk, v in s.items()}
        env.set_state(s["qpos"], s["qvel"], s["warm"], s["t"])
        env.step(torch.as_tensor(act, device=env.device))
        stt = env.status().cpu().numpy()
        qpos, qvel, _, _ = [x.cpu().numpy() for x in env.get_state()]
        ro = oracle.step(cm, s, act.astype(np.float64), nthreads=8)
        ev = np.abs(qvel - s["qvel"]).max(1)
        bad = np.where(ev > 2e-5)[0]
        print("checkpoint", k, "outliers", bad, ev[bad], "gpu status", stt[bad], "oracle status", ro["status"][bad])
        for e in bad[:2]:
            s32 = emu_lib.f32_state({kk: v[e:e+1].copy() for kk, v in s0.items()})
            re = emu_lib.env_step(cm, s32, act[e:e+1])
            print("   emu vs oracle", np.abs(s32["qvel"] - s["qvel"][e]).max(), "emu vs gpu", np.abs(s32["qvel"] - qvel[e]).max(), "emu status", re["status"], "iters", re["iters"])
            re2 = emu_lib.env_step(cm, emu_lib.f32_state({kk: v[e:e+1].copy() for kk, v in s0.items()}), act[e:e+1], max_iter=40)
            s32b = emu_lib.f32_state({kk: v[e:e+1].copy() for kk, v in s0.items()}); emu_lib.env_step(cm, s32b, act[e:e+1], max_iter=40, tol=1e-9, rtol=0.0)
            print("   emu(max_iter 40, tight) vs oracle", np.abs(s32b["qvel"] - s["qvel"][e]).max())
            print("   z, quat", s0["qpos"][e][2:7], "ncon oracle start", oracle.forward(cm, s0["qpos"][e], s0["qvel"][e], act[e].astype(np.float64), s0["warm"][e])["counts"])
    oracle.step(cm, st, act.astype(np.float64), nthreads=8)
