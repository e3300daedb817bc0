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


TWO_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from mujoco_maze_amd import sharding
import mujoco_maze_amd as mm

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
single = os.environ.get("MZ_TEST_SINGLE_GPU") == "1"          # rehearsal on a one-GPU box: both ranks on cuda:0, gloo (RCCL wants a device per rank)
torch.cuda.set_device(0 if single else rank)
dev = torch.device("cuda", 0 if single else rank)
if single:
    dist.init_process_group(backend="gloo")
else:
    dist.init_process_group(backend="nccl", device_id=dev)
n_local = 512
env = sharding.ShardedVecMazeEnv("Ant4Rooms-v0", n_local, device=dev, gather=True, auto_reset=True)   # envs PER RANK
assert (env.lo, env.hi) == (rank * n_local, (rank + 1) * n_local) and env._device_record
env.reset(seed=3)
whole = mm.make("Ant4Rooms-v0", num_envs=n_local * world, auto_reset=True, device=dev, force_vec=True) if rank == 0 else None
if whole is not None:
    whole.reset(seed=3)
g = torch.Generator(device="cpu").manual_seed(0)
for k in range(30):
    act = (torch.rand((n_local * world, 8), generator=g) * 60 - 30)   # one table, indexed by GLOBAL slot (same generator on every rank)
    obs, rew, done, info = env.step(act[env.lo:env.hi].to(dev))
    got = env.gathered()                                       # [n_local * world, obs_dim + 2] on every rank, over RCCL / xGMI
    mine = torch.cat([obs, rew[:, None], done.float()[:, None]], dim=1)
    assert torch.equal(got[env.lo:env.hi], mine), k
    if whole is not None:                                      # sharded == unsharded (same kernel instantiation: 512 and 1024 envs)
        o, r, d, _ = whole.step(act.to(dev))
        assert torch.equal(got, torch.cat([o, r[:, None], d.float()[:, None]], dim=1)), k
dist.barrier()
env.close()
dist.destroy_process_group()
if rank == 0:
    print("TWO_RANKS_NCCL_OK")
"""


def test_two_ranks_nccl():
    """the first time TWO GPUs are visible this runs instead of skipping — two processes, one per GPU, backend "nccl"
    (RCCL over xGMI): rank-local records arrive in every rank's gathered tensor, and the sharded run equals the unsharded one bit for
    bit; then `bench.py --gpus 2` itself (real nccl, gather on / off pair in its JSON line).  One-GPU boxes skip (the same code path
    under gloo: tests/test_sharding_gloo.py, tests/test_gpu_two_ranks.py)."""
    import json
    import socket

    import torch

    two = torch.cuda.device_count() >= 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    if not two:
        env["MZ_TEST_SINGLE_GPU"] = "1"  # rehearse the very script on the one GPU (gloo), so that its first real run is not its first run
    script = os.path.join(ROOT, "gpurun_out", "_two_ranks_nccl.py")
    os.makedirs(os.path.dirname(script), exist_ok=True)
    with open(script, "w") as f:
        f.write(TWO_RANK_SCRIPT % dict(root=ROOT))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port), script]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "TWO_RANKS_NCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    if not two:
        pytest.skip("the nccl leg needs two visible GPUs (the two-rank script itself was rehearsed on this GPU over gloo: passed)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "10", "--no-cpu-baseline", "--no-live-pmc",
                          "--sustained", "0"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
