"""Multi-GPU path on CPU: world_size-2 `gloo` processes exercise the shard partition, the whole-node obs
gather (even and uneven shards) and the max-over-ranks timing reduction that bench.py uses.  The envs
themselves only run on a GPU; sharding invariance of their results is covered by the -m gpu tests
(test_million_env_config...: a handle owning the upper half reproduces the same rows)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from custom_gymnasium_environments_amd import sharding


def test_shard_range_partitions_exactly():
    for total in [0, 1, 7, 8, 9, 1000, 1 << 20, (1 << 20) + 3]:
        for world in [1, 2, 3, 4, 8]:
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, count = sharding.shard_range(total, rank, world)
        # a fake obs shard whose content encodes the global env index
        local = (torch.arange(start, start + count, dtype=torch.float32)[:, None] * 10 + torch.arange(5)[None, :]).contiguous()
        full = sharding.gather_obs(local, total)
        t = sharding.max_over_ranks(1.0 + rank)
        q.put((rank, full.numpy(), t, sharding.rank_world()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 65])
def test_gather_obs_and_timing_reduction_world2_gloo(total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.arange(total, dtype=np.float32)[:, None] * 10 + np.arange(5)[None, :]
    for rank, full, t, rw in results:
        assert np.array_equal(full, expect), rank
        assert t == 2.0                      # max over ranks of (1.0, 2.0)
        assert rw == (rank, world, rank)
