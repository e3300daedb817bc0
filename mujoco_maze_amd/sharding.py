"""Env sharding across the GPUs of one node.

Environments are independent, so rank r simply owns the global env slots
`[r * n_local, (r + 1) * n_local)`; nothing is exchanged inside the physics.  The
one collective is the all-gather of the packed per-env record
`[obs (obs_dim) | reward | done]` for consumers that want the concatenated batch
(`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box; the same code
runs on "gloo" CPU tensors in the world_size-2 test).

Reset noise is keyed by the *global* env slot (`env_index_offset` option of the
C-ABI), so a sharded run reproduces the unsharded one slot for slot.
"""
from typing import Optional, Tuple


def shard_range(rank: int, world_size: int, n_local: int) -> Tuple[int, int]:
    """Global env slots owned by `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return rank * n_local, (rank + 1) * n_local


def record_width(obs_dim: int) -> int:
    return obs_dim + 2


def pack_record(obs, reward, done, out):
    """out[:, :obs_dim] = obs; out[:, obs_dim] = reward; out[:, obs_dim + 1] = done (as float)."""
    d = obs.shape[1]
    out[:, :d].copy_(obs)
    out[:, d].copy_(reward)
    out[:, d + 1].copy_(done)
    return out


class RecordGatherer:
    """Owns the send/receive buffers and issues one all-gather per batch step.

    Two send buffers alternate (`flip()`): while the all-gather of step k reads one of them on RCCL's stream, the step kernel
    of step k + 1 — which writes its packed record itself (mz_bind_record) — fills the other, so the collective still hides
    behind the next step's physics."""

    def __init__(self, n_local: int, obs_dim: int, device, group=None, always_collective: bool = False):
        """`always_collective`: issue the all-gather even in a one-rank group (where it degenerates to a copy), so that the
        collective path itself — RCCL on the GPU box — is what runs; used by the single-GPU RCCL test."""
        import torch
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_local, self.width = n_local, record_width(obs_dim)
        self._bufs = [torch.empty((n_local, self.width), dtype=torch.float32, device=device) for _ in range(2)]
        self._cur = 0
        self._buf_work = [None, None]  # the gather that last read each send buffer
        self.gathered = torch.empty((n_local * self.world, self.width), dtype=torch.float32, device=device)
        self._work = None
        self._always = bool(always_collective) and dist.is_initialized()

    @property
    def packed(self):
        """The current send buffer."""
        return self._bufs[self._cur]

    def flip(self):
        """Switch to the other send buffer and return it, after the gather that last read it has finished (stream-ordered:
        the wait blocks the current stream, not the host)."""
        self._cur ^= 1
        w = self._buf_work[self._cur]
        if w is not None:
            w.wait()
            self._buf_work[self._cur] = None
        return self._bufs[self._cur]

    def start(self, obs=None, reward=None, done=None):
        """Launch the all-gather of `packed` (asynchronous); call wait() before reading `gathered`.  With arguments the record
        is packed here first (three copies: CPU tensors of the gloo test, host-judged tasks); without, `packed` is the buffer the
        step kernel itself filled (VecMazeEnv.bind_record -> mz_bind_record): kernel -> collective, no launch in between."""
        if obs is not None:
            pack_record(obs, reward, done, self.packed)
        if self.world == 1 and not self._always:
            self.gathered.copy_(self.packed)
            self._work = None
        else:
            self._work = self._dist.all_gather_into_tensor(self.gathered, self.packed, group=self.group, async_op=True)
        self._buf_work[self._cur] = self._work
        return self

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self.gathered

    def split(self, gathered=None):
        """(obs, reward, done) views of the gathered batch in global env-slot order."""
        g = self.gathered if gathered is None else gathered
        d = self.width - 2
        return g[:, :d], g[:, d], g[:, d + 1]


class ShardedVecMazeEnv:
    """One rank's shard of a node-wide batch: a local `VecMazeEnv` + the record all-gather."""

    def __init__(self, env_id: str, envs_per_rank: int, device=None, gather: bool = True, group=None, always_collective: bool = False,
                 **kwargs):
        import torch
        import torch.distributed as dist

        import mujoco_maze_amd as mm

        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.env = mm.make(env_id, num_envs=envs_per_rank, device=device, force_vec=True, **kwargs)
        self.lo, self.hi = shard_range(self.rank, self.world, envs_per_rank)
        self.env.set_option("env_index_offset", float(self.lo))
        # a task that resamples its goals on reset (MazeTask.sample_goals): rank 0's goals are broadcast over THIS group, and the
        # ranks must therefore call reset() together (maze_env.py _sync_goals_across_ranks)
        self.env.goal_sync_group = group if group is not None else "world"
        self.gatherer: Optional[RecordGatherer] = RecordGatherer(envs_per_rank, self.env.obs_dim, self.env.device, group, always_collective) if gather else None
        # the step kernel writes the packed record straight into the gatherer's send buffer — unless the task is judged by
        # Python overrides on the host, whose reward / done only exist after the kernel
        self._device_record = self.gatherer is not None and not self.env._host_rewards
        self._torch = torch

    def reset(self, seed: int = 0):
        return self.env.reset(seed=seed)  # same seed on every rank: streams differ through the global slot index

    def step(self, actions):
        if self._device_record:
            self.env.bind_record(self.gatherer.flip())  # the buffer the previous gather is NOT reading
        obs, rew, done, info = self.env.step(actions)
        if self.gatherer is not None:
            if self._device_record:
                self.gatherer.start()
            else:
                self.gatherer.start(obs, rew, done)
        return obs, rew, done, info

    def gathered(self):
        return None if self.gatherer is None else self.gatherer.wait()

    def close(self):
        self.env.close()
