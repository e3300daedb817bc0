"""N > 1 path on CPU: two processes (gloo) each own a shard of the env slots, produce the
per-env records (here with the CPU oracle standing in for the device step — tests may do
that), all-gather them with mujoco_maze_amd.sharding.RecordGatherer, and the gathered batch
must equal the unsharded run slot for slot (reset noise is keyed by the global slot)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(n_local, lo, seed, steps):
    from mujoco_maze_amd import maze_task as T
    from mujoco_maze_amd import model
    from tests import oracle_lib

    oracle = oracle_lib.load()
    cm = model.compile_model("ant", T.DistRewardUMaze(8.0), 8.0)
    st, _ = oracle.reset(cm, n_local, seed, env0=lo)
    out = None
    for k in range(steps):
        rng = np.random.default_rng(1000 + k)  # same action table on every rank, indexed by global slot
        act = rng.uniform(-30, 30, (64, 8))[lo:lo + n_local]
        out = oracle.step(cm, st, act)
    return out


def _worker(rank, world, port, n_local, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mujoco_maze_amd import sharding

    lo, hi = sharding.shard_range(rank, world, n_local)
    out = _records(n_local, lo, seed=77, steps=2)
    g = sharding.RecordGatherer(n_local, 30, torch.device("cpu"))
    g.start(torch.from_numpy(out["obs"]).float(), torch.from_numpy(out["reward"]).float(), torch.from_numpy(out["done"]).float())
    gathered = g.wait().clone()
    obs, rew, done = g.split(gathered)
    assert obs.shape == (n_local * world, 30)
    if rank == 0:
        q.put(gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_reproduce_the_unsharded_batch():
    world, n_local = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_local, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = _records(n_local * world, 0, seed=77, steps=2)
    ref = np.concatenate([full["obs"], full["reward"][:, None], full["done"][:, None].astype(np.float64)], 1).astype(np.float32)
    assert np.array_equal(gathered, ref)


def test_shard_ranges_partition_the_slots():
    from mujoco_maze_amd import sharding

    seen = []
    for r in range(8):
        lo, hi = sharding.shard_range(r, 8, 4096)
        seen += list(range(lo, hi, 1024))
        assert hi - lo == 4096
    assert seen == list(range(0, 8 * 4096, 1024))
    with pytest.raises(ValueError):
        sharding.shard_range(8, 8, 4096)
    assert sharding.record_width(30) == 32
