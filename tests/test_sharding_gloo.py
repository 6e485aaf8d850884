"""Multi-process (world_size 2 and 3, gloo, CPU) tests of the N>1 path:
shard partition + the single all-gather of per-set scores."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rrmpg_amd.sharding import allgather_scores, shard_bounds


def test_shard_bounds_partition():
    for n in (0, 1, 7, 8, 1000, 1_000_000, 1_000_003):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            for (a0, a1), (b0, b1) in zip(cuts, cuts[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(1_000_000, 8, 3) == (375_000, 500_000)
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_allgather_identity_without_group():
    x = torch.arange(5, dtype=torch.float64)
    assert allgather_scores(x) is x


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, num_sets, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = shard_bounds(num_sets, world, rank)
        # a per-set "score" that depends only on the global set index
        idx = torch.arange(a, b, dtype=torch.float64)
        local = idx * 0.5 + 1.0
        full = allgather_scores(local, num_sets)
        full2 = allgather_scores(local)          # num_sets inferred
        np.save(os.path.join(out_dir, "r%d.npy" % rank), full.numpy())
        assert torch.equal(full, full2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,num_sets", [(2, 1000), (2, 1001), (3, 1000)])
def test_allgather_scores_gloo(tmp_path, world, num_sets):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_sets, str(tmp_path)),
             nprocs=world, join=True)
    want = np.arange(num_sets, dtype=np.float64) * 0.5 + 1.0
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "r%d.npy" % r))
        assert np.array_equal(got, want)
