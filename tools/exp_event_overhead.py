"""What the per-kernel HIP event pairs of the timing ring (option "time_kernels": bench.py's roofline.kernel_ms) cost the wall clock of small kernels.
    python tools/exp_event_overhead.py [env id] [envs] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mujoco_maze_amd as mm  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "PointUMaze-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
env.reset(seed=0)
lo = torch.as_tensor(env.action_space.low, device=env.device); hi = torch.as_tensor(env.action_space.high, device=env.device)
g = torch.Generator(device=env.device).manual_seed(0)
acts = [lo + (hi - lo) * torch.rand((n, env.nu), device=env.device, generator=g) for _ in range(16)]
for t in range(300):
    env.step(acts[t % 16])
for rep in range(3):
    for timed, stride in ((0, 1), (steps, 1), (steps // 8, 8)):
        env.set_option("time_kernels", timed)
        env.set_option("time_kernels_stride", stride)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            env.step(acts[t % 16])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s %d envs  events %-8s  %.3f M env-steps/s  %.4f ms per step%s" % (env_id, n, ("every %d" % stride) if timed else "off", n * steps / dt / 1e6, dt / steps * 1e3,
                                                                                 ("  kernel %.4f ms" % env.kernel_ms()) if timed else ""))
env.close()
