"""The RCCL code path on real hardware: one rank, backend "nccl" (= RCCL on ROCm), the
record all-gather forced down `all_gather_into_tensor(async_op=True)` and overlapped with the next step's kernel — the
same calls `bench.py --gpus N` and `ShardedVecMazeEnv` make on an 8-GPU node.  Runs in a subprocess so that the process
group does not leak into the other tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from mujoco_maze_amd import sharding

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%(port)d", world_size=1, rank=0, device_id=dev)
assert dist.get_backend() == "nccl"
n = 1024
env = sharding.ShardedVecMazeEnv("Ant4Rooms-v0", n, device=dev, gather=True, always_collective=True, auto_reset=True)  # BASELINE configs[3]'s env
assert env.gatherer._always and (env.lo, env.hi) == (0, n)
env.reset(seed=3)
g = torch.Generator(device=dev).manual_seed(0)
acts = [torch.rand((n, 8), device=dev, generator=g) * 60 - 30 for _ in range(8)]
for k in range(50):
    obs, rew, done, info = env.step(acts[k %% 8])           # launches the step kernel, then the async all-gather of its record
    expect = torch.cat([obs, rew[:, None], done.float()[:, None]], dim=1).clone()
    got = env.gathered()                                    # waits for THIS step's collective
    assert got.shape == (n, env.env.obs_dim + 2)
    assert torch.equal(got, expect), k
    o, r, d = env.gatherer.split()
    assert torch.equal(o, obs) and torch.equal(r, rew) and torch.equal(d, done.float())
assert env._device_record                                   # the step kernel wrote the record the collective sent (mz_bind_record)
# overlap: the gather of step k is still in flight when the kernel of step k + 1 is launched (what bench.py does); the kernel
# writes its record into the OTHER send buffer (RecordGatherer.flip)
gt = env.gatherer
snap = None
for k in range(20):
    env.env.bind_record(gt.flip())
    obs, rew, done, _ = env.env.step(acts[k %% 8])          # kernel k overlaps gather k - 1
    if snap is not None:
        assert torch.equal(gt.wait(), snap), k              # gather k - 1 delivered record k - 1 although kernel k ran meanwhile
    snap = torch.cat([obs, rew[:, None], done.float()[:, None]], dim=1).clone()
    gt.start()                                              # gather k, straight from the kernel-written buffer
assert torch.equal(gt.wait(), snap)
torch.cuda.synchronize()
env.close()
dist.destroy_process_group()
print("RCCL_PATH_OK")
"""


def test_record_allgather_over_rccl_single_rank():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT, port=port)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "RCCL_PATH_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
