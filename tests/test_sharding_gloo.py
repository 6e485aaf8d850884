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
        # the same exchange in two halves, two of them under way at once (the
        # bench's pipelined steps): finished in the order they were begun
        from rrmpg_amd.sharding import allgather_scores_begin
        h1 = allgather_scores_begin(local, num_sets)
        h2 = allgather_scores_begin(local * 2, num_sets)
        assert torch.equal(h1.finish(), full)
        assert torch.equal(h2.finish(), full * 2)
        assert h1.finish() is h1.finish()
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


# ---- the library-level sharded sweep (rrmpg_amd.sharding.sweep /
# ResidentSweep) with a stand-in model: the partition, the score definitions
# and the one collective are what is tested here; the kernels behind
# Model._sweep have their own (GPU) tests.
class _LinearModel:
    """q[t, i] = a_i * base[t]: enough of a model for the sharding logic."""
    _dtype = np.dtype([("a", np.float64)])

    def _sweep(self, params, qobs, want_qsim, base):
        q = np.asarray(base, dtype=np.float64)[:, None] * params["a"][None, :]
        sse = ((np.asarray(qobs)[:, None] - q) ** 2).sum(0)
        return (q if want_qsim else None), sse


class _LinearEnsemble:
    """The same as a 'resident ensemble' (rrmpg_amd.device interface), on the
    CPU: run(params [n, 1], qsim, qobs=, sse=) fills sse."""
    def __init__(self, base):
        self.base = torch.as_tensor(base, dtype=torch.float64)
        self.device = torch.device("cpu")

    def run(self, params, qsim, storages=None, qobs=None, sse=None):
        q = self.base[:, None] * params[:, 0][None, :]
        if qsim is not None:
            qsim.copy_(q)
        sse.copy_(((qobs[:, None] - q) ** 2).sum(0))
        return sse


def _problem(num_sets):
    rng = np.random.default_rng(5)
    base = rng.uniform(0.1, 2.0, 50)
    qobs = 1.3 * base + rng.normal(0, 0.05, 50)
    params = np.zeros(num_sets, dtype=_LinearModel._dtype)
    params["a"] = rng.uniform(0.5, 2.0, num_sets)
    return base, qobs, params


def _sweep_worker(rank, world, port, num_sets, out_dir):
    from rrmpg_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base, qobs, params = _problem(num_sets)
        res = {}
        for score in ("mse", "nse"):
            out = sharding.sweep(_LinearModel(), params, qobs, score=score,
                                 return_qsim=True, base=base)
            a, b = sharding.shard_bounds(num_sets, world, rank)
            assert out["bounds"] == (a, b) and out["score"] == score
            assert out["qsim"].shape == (50, b - a)      # stays sharded
            res[score] = out["scores"]
            # the HBM-resident form of the same sweep (bench.py's loop)
            ens = _LinearEnsemble(base)
            block = torch.from_numpy(params["a"][a:b].copy())[:, None]
            rs = sharding.ResidentSweep(ens, block, torch.from_numpy(qobs),
                                        num_sets, score=score)
            res["resident_" + score] = rs.step().numpy()
        np.savez(os.path.join(out_dir, "s%d.npz" % rank), **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,num_sets", [(2, 501), (3, 100)])
def test_sharded_sweep_gloo_mse_and_nse(tmp_path, world, num_sets):
    """Every rank ends up with the scores of the WHOLE sweep, equal to the
    single-process sweep's -- MSE (the reference's monte_carlo score) and NSE
    (BASELINE configs[3]: all-gather of per-set NSE; calc_nse's definition,
    rrmpg/utils/metrics.py:61-75)."""
    from rrmpg_amd import sharding
    from rrmpg_amd.utils.metrics import calc_mse, calc_nse
    port = _free_port()
    mp.spawn(_sweep_worker, args=(world, port, num_sets, str(tmp_path)),
             nprocs=world, join=True)
    base, qobs, params = _problem(num_sets)
    q = base[:, None] * params["a"][None, :]
    want = {"mse": np.array([calc_mse(qobs, q[:, i]) for i in range(num_sets)]),
            "nse": np.array([calc_nse(qobs, q[:, i]) for i in range(num_sets)])}
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "s%d.npz" % r))
        for score in ("mse", "nse"):
            assert np.allclose(got[score], want[score], rtol=1e-12, atol=0)
            assert np.allclose(got["resident_" + score], want[score],
                               rtol=1e-12, atol=0)
    # constant observations: NSE raises calc_nse's error, MSE does not
    with pytest.raises(RuntimeError, match="Nash-Sutcliffe"):
        sharding.scores_from_sse(np.ones(3), np.full(10, 2.0), "nse")
    assert np.allclose(sharding.scores_from_sse(np.ones(3), np.full(10, 2.0),
                                                "mse"), 0.1)
    with pytest.raises(ValueError):
        sharding.scores_from_sse(np.ones(3), np.ones(4), "kge")
