"""Sharding a batch of independent envs over the GPUs of one node.

Instances never interact, so the data path has NO collective: rank r owns the contiguous global env
indices [start_r, start_r + count_r) and steps them with its own handle on its own GPU.  Per-env seeds and
the synthetic action hash are functions of the GLOBAL index (`env_index0`), so results do not depend on the
GPU count.  The only optional exchange is `gather_obs` — a whole-node observation gather for consumers that
want every row on every rank: one all-gather (RCCL over xGMI when the tensors are on GPUs, gloo on CPU
tensors in the tests), padded to the largest shard when the split is uneven.
"""
import os

import torch


def shard_range(total_envs, rank, world_size):
    """Contiguous, balanced split: the first (total % world) ranks get one extra env."""
    total_envs, rank, world_size = int(total_envs), int(rank), int(world_size)
    if not (0 <= rank < world_size) or total_envs < 0:
        raise ValueError("bad rank / world_size / total_envs")
    base, extra = divmod(total_envs, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def make_sharded(env_cls, total_envs, rank=None, world_size=None, local_rank=None, **kwargs):
    """Build this rank's shard of a `total_envs` batch: env_cls(count, env_index0=start, device=cuda:local_rank)."""
    r, w, lr = rank_world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    local_rank = lr if local_rank is None else local_rank
    start, count = shard_range(total_envs, rank, world_size)
    if count == 0:
        raise ValueError(f"rank {rank} would own no envs ({total_envs} envs over {world_size} ranks)")
    kwargs.setdefault("device", f"cuda:{local_rank}")
    env = env_cls(count, env_index0=start, **kwargs)
    env.global_num_envs = int(total_envs)
    env.shard = (start, count)
    return env


def gather_obs(local_obs, total_envs, group=None):
    """All-gather the per-rank observation shards into the full (total_envs, ...) tensor on every rank.
    Row order is the global env index.  Works for even and uneven shards."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    start, count = shard_range(total_envs, rank, world)
    if local_obs.shape[0] != count:
        raise ValueError(f"rank {rank} should hold {count} rows, got {local_obs.shape[0]}")
    counts = [shard_range(total_envs, r, world)[1] for r in range(world)]
    cmax = max(counts)
    tail = tuple(local_obs.shape[1:])
    if all(c == cmax for c in counts):
        out = torch.empty((total_envs,) + tail, dtype=local_obs.dtype, device=local_obs.device)
        dist.all_gather_into_tensor(out, local_obs.contiguous(), group=group)
        return out
    padded = torch.zeros((cmax,) + tail, dtype=local_obs.dtype, device=local_obs.device)
    padded[:count] = local_obs
    buf = torch.empty((world * cmax,) + tail, dtype=local_obs.dtype, device=local_obs.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * cmax: r * cmax + counts[r]] for r in range(world)], dim=0)


def max_over_ranks(seconds, device=None, group=None):
    """The bench contract's timing reduction: MAX of a per-rank wall time."""
    import torch.distributed as dist
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
