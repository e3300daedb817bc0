"""Experiment: K independent sub-batches of one GPU's envs, each a VecMazeEnv of its own stepped on its own HIP stream.  A launch lasts as long
as its slowest wave; with K launches in flight the SIMDs a finished wave frees are taken by another sub-batch's next step, so the device
works at the MEAN wave duration instead of the maximum.  (Not the BASELINE metric's lock-step batch: what an asynchronous vector env does.)
    python tools/exp_two_streams.py [env id] [total envs] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mujoco_maze_amd as mm  # noqa: E402


def run(env_id, total, k, steps):
    n = total // k
    streams = [torch.cuda.Stream() for _ in range(k)]
    envs, acts = [], []
    for i in range(k):
        with torch.cuda.stream(streams[i]):
            e = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
            e.reset(seed=100 + i)
            lo = torch.as_tensor(e.action_space.low, device=e.device); hi = torch.as_tensor(e.action_space.high, device=e.device)
            g = torch.Generator(device=e.device).manual_seed(i)
            acts.append([lo + (hi - lo) * torch.rand((n, e.nu), device=e.device, generator=g) for _ in range(16)])
            envs.append(e)
    for t in range(200):
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                envs[i].step(acts[i][t % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                envs[i].step(acts[i][t % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = sum(int((e.status() & 7).ne(0).sum()) for e in envs)
    for e in envs:
        e.close()
    return total * steps / dt, bad


if __name__ == "__main__":
    env_id = sys.argv[1] if len(sys.argv) > 1 else "AntUMaze-v0"
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    for k in (1, 2, 4, 8):
        v, bad = run(env_id, total, k, steps)
        print("%s  %d envs as %d sub-batch(es) of %d on %d stream(s): %.3f M env-steps/s  (flagged %d)" % (env_id, total, k, total // k, k, v / 1e6, bad))
