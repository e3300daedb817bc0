"""The N > 1 device path under the driver's eyes: two ranks share the box's ONE GPU (gloo rendezvous —
RCCL wants one device per rank), each steps its shard of Ant4Rooms-v0 (BASELINE configs[3]) with the HIP kernels through
`ShardedVecMazeEnv`, and the all-gathered batch must equal the unsharded device run slot for slot: observation, reward, done
of every env, every step.  What an 8-GPU node changes is the backend string and the device index, nothing else in the path."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from mujoco_maze_amd import sharding
import mujoco_maze_amd as mm

rank, world, n_local, steps = int(sys.argv[1]), 2, 256, 12
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", world_size=world, rank=rank)
env = sharding.ShardedVecMazeEnv("Ant4Rooms-v0", n_local, device=dev, gather=True, always_collective=True, auto_reset=True)
assert (env.lo, env.hi) == (rank * n_local, (rank + 1) * n_local) and env._device_record
env.reset(seed=11)
g = torch.Generator(device="cpu").manual_seed(5)
acts = [(torch.rand((world * n_local, 8), generator=g) * 60 - 30) for _ in range(steps)]   # one table, indexed by GLOBAL slot
# a few envs of each shard next to the goal (24, -24) so that termination + auto-reset happen inside the compared window
qpos = env.env.get_state()[0]
qpos[:8, 0], qpos[:8, 1] = 24.0 - 0.3, -24.0
env.env.set_state(qpos=qpos)
out = []
for k in range(steps):
    obs, rew, done, info = env.step(acts[k][env.lo:env.hi].to(dev))
    got = env.gathered()                      # [world * n_local, obs_dim + 2] in global slot order
    assert torch.equal(got[env.lo:env.hi, :30], obs) and torch.equal(got[env.lo:env.hi, 30], rew)
    out.append(got.cpu().clone())
if rank == 0:
    # the unsharded run on the same GPU: one VecMazeEnv of world * n_local envs, same seed, same action table
    full = mm.make("Ant4Rooms-v0", num_envs=world * n_local, device=dev, auto_reset=True)
    full.reset(seed=11)
    q = full.get_state()[0]
    for r in range(world):
        q[r * n_local: r * n_local + 8, 0], q[r * n_local: r * n_local + 8, 1] = 24.0 - 0.3, -24.0
    full.set_state(qpos=q)
    ndone = 0
    for k in range(steps):
        obs, rew, done, info = full.step(acts[k].to(dev))
        ref = torch.cat([obs, rew[:, None], done.float()[:, None]], dim=1).cpu()
        assert torch.equal(out[k], ref), (k, (out[k] != ref).nonzero()[:5])
        ndone += int((done != 0).sum())
    assert ndone >= 2 * 8, ndone            # the goal-side envs terminated and were re-seeded by their global slot
    full.close()
    print("TWO_RANK_HIP_OK", ndone)
dist.barrier()
env.close()
dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_reproduce_the_unsharded_device_batch(tmp_path):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rank.py"
    script.write_text(RANK % dict(root=ROOT, port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, o[-2000:] + e[-4000:]
    assert "TWO_RANK_HIP_OK" in outs[0][1]
