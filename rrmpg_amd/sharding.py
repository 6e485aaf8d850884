"""Sharding of the parameter-set axis over the GPUs of one node.

The reference has no distributed code: its scalable axis, untouched beyond a
Python loop (reference: rrmpg/models/hbvedu.py:199-209), is the axis of
parameter realisations.  The sets are independent, the forcing is read-only
and replicated, so the sweep shards into contiguous blocks of sets, one block
per process/GPU, with NO collective on the data path.  The only exchange is
one all-gather of the per-set scores (8 B per set; 1 MB per rank at 1M sets
over 8 GPUs) at the end -- RCCL over xGMI when the tensors live on the GPUs
(backend "nccl"), gloo on the CPU.  The [timesteps, sets] discharge stays
sharded where it was written: gathering it (87.7 GB at 1M sets) over xGMI
would cost more than computing it.
"""

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(num_sets, world_size, rank):
    """[start, stop) of rank's contiguous block of the parameter-set axis.

    Blocks differ by at most one set; the first num_sets % world_size ranks
    hold the longer ones.
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(int(num_sets), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class ScoreExchange:
    """An all-gather of per-set scores under way (allgather_scores_begin).
    finish() makes the current stream wait for it -- the host does not -- and
    returns the [num_sets] tensor; until then the object keeps the buffers
    the collective reads and writes alive."""

    def __init__(self, result=None, work=None, out=None, lens=None,
                 keep=None):
        self._result, self._work, self._out = result, work, out
        self._lens, self._keep = lens, keep

    def finish(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
            out, lens = self._out, self._lens
            if lens is None:
                self._result = out
            else:
                longest = max(lens)
                self._result = torch.cat(
                    [out[r * longest:r * longest + lens[r]]
                     for r in range(len(lens))])
            self._out = self._keep = None
        return self._result


def allgather_scores_begin(local_scores, num_sets=None, group=None,
                           always_collective=False):
    """allgather_scores in two halves: the collective is ENQUEUED (on the
    process group's own stream, behind what the current stream holds so far)
    and the call returns a ScoreExchange; work enqueued on the current
    stream before its finish() -- the next sweep's kernel -- runs beside the
    exchange instead of behind it.  One process per GPU at eight GPUs: the
    8 MB all-gather of a million scores then hides under the next shard's
    2.4 ms instead of adding to every step."""
    if not (dist.is_available() and dist.is_initialized()):
        return ScoreExchange(result=local_scores)
    world = dist.get_world_size(group)
    if world == 1 and not always_collective:
        return ScoreExchange(result=local_scores)
    rank = dist.get_rank(group)
    if num_sets is None:
        n = torch.tensor([local_scores.numel()], device=local_scores.device)
        dist.all_reduce(n, group=group)
        num_sets = int(n.item())
    lens = [shard_bounds(num_sets, world, r) for r in range(world)]
    lens = [b - a for a, b in lens]
    if lens[rank] != local_scores.numel():
        raise ValueError("rank %d holds %d scores, expected %d"
                         % (rank, local_scores.numel(), lens[rank]))
    longest = max(lens)
    if min(lens) == longest:
        src = local_scores.contiguous()
        out = torch.empty(num_sets, dtype=src.dtype, device=src.device)
        work = dist.all_gather_into_tensor(out, src, group=group,
                                           async_op=True)
        return ScoreExchange(work=work, out=out, keep=src)
    # ragged: pad every block to the longest, gather, drop the padding
    padded = torch.zeros(longest, dtype=local_scores.dtype,
                         device=local_scores.device)
    padded[:local_scores.numel()] = local_scores
    out = torch.empty(world * longest, dtype=local_scores.dtype,
                      device=local_scores.device)
    work = dist.all_gather_into_tensor(out, padded, group=group,
                                       async_op=True)
    return ScoreExchange(work=work, out=out, lens=lens, keep=padded)


def allgather_scores(local_scores, num_sets=None, group=None,
                     always_collective=False):
    """All-gather the per-set scores of every rank's block, in set order.

    local_scores: 1-D tensor (this rank's block; blocks may differ in length
    by one).  Returns the concatenated [num_sets] tensor on every rank.
    Without an initialised process group (single process) it is the identity;
    so it is for a group of one rank unless always_collective is set (tests:
    the RCCL call itself on a single-GPU box).  (allgather_scores_begin and
    its finish() in one step.)
    """
    return allgather_scores_begin(local_scores, num_sets, group,
                                  always_collective).finish()


# ---------------------------------------------------------------------------
# scores of a sharded sweep
# ---------------------------------------------------------------------------
SCORES = ("mse", "nse")


def scores_from_sse(sse, qobs, score="mse"):
    """Per-set score from the kernels' fused squared-error sums.

    'mse': sse / T -- what the reference's monte_carlo computes
    (rrmpg/tools/monte_carlo.py:66-71, calc_mse rrmpg/utils/metrics.py:
    110-136); 'nse': 1 - sse / sum((obs - mean(obs))^2), calc_nse's
    definition (metrics.py:29-77) including its RuntimeError for constant
    observations.  sse: numpy array or torch tensor (any device), any shape;
    qobs: the observations the sums were taken against ([T] numpy / tensor;
    for a multi-catchment sweep [C, T] against sse [C, N])."""
    if score not in SCORES:
        raise ValueError("score must be one of %s" % (SCORES,))
    obs = qobs.detach().cpu().numpy() if isinstance(qobs, torch.Tensor) \
        else np.asarray(qobs, dtype=np.float64)
    t = obs.shape[-1]
    if score == "mse":
        return sse / t
    den = ((obs - obs.mean(axis=-1, keepdims=True)) ** 2).sum(axis=-1)
    if np.any(den == 0):
        raise RuntimeError(
            "The Nash-Sutcliffe-Efficiency coefficient is not defined for the "
            "case, that all values in the observations are equal. Maybe you "
            "should use the Mean-Squared-Error instead.")
    if isinstance(sse, torch.Tensor):
        den_t = torch.as_tensor(den, dtype=sse.dtype, device=sse.device)
        return 1 - sse / (den_t if den_t.dim() == 0 else den_t[:, None])
    return 1 - sse / (den if np.ndim(den) == 0 else den[:, None])


def _rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _group_on_device(group=None):
    """True if the group's collectives take device tensors only (backend
    nccl = RCCL): host scores must move to this rank's GPU first."""
    return "nccl" in str(dist.get_backend(group)).lower()


def sweep(model, params, qobs, score="mse", gpus=None, return_qsim=False,
          group=None, always_collective=False, **forcing):
    """One Monte-Carlo sweep of `model` over ALL rows of `params`, over
    several GPUs, scored per set -- the library call behind
    ``monte_carlo(..., gpus=...)`` and behind a torchrun job.

    * Inside an initialised ``torch.distributed`` process group (one process
      per GPU, the device chosen with ``rrmpg_amd._lib.set_device(local_rank)``
      or ``torch.cuda.set_device``): every rank passes the SAME `params`; rank
      r simulates the contiguous block ``shard_bounds(len(params), world, r)``
      and the per-set scores are exchanged with the job's one collective, the
      all-gather of 8 B per set.  The host-pointer call brings the scores to
      the host; a gloo group exchanges them there, an nccl-only group gets
      them back on this rank's GPU for the exchange over RCCL.  `gpus` is
      ignored.  (always_collective: run the collective for a group of one
      rank as well -- tests on a one-GPU box.)
    * Without a process group: the host-pointer call itself fans the
      parameter-set axis out over `gpus` devices (an int, or 'all'; None: the
      current device), one host thread per device inside librrhip
      (the call's rr_call_options, RR_OPT_HOST_SHARDS, include/rrhip.h), every device writing its columns
      of the one [T, N] result.

    Returns a dict: 'scores' ([N] numpy, the whole sweep on every rank),
    'score' (its name), 'bounds' ((first, stop) this process simulated) and,
    if return_qsim, 'qsim' ([T, stop - first]: the discharge stays sharded).
    """
    from . import _lib
    if score not in SCORES:
        raise ValueError("score must be one of %s" % (SCORES,))
    n = len(params)
    rank, world = _rank_world(group)
    in_group = dist.is_available() and dist.is_initialized()
    if world > 1 or (always_collective and in_group):
        first, stop = shard_bounds(n, world, rank)
        qsim, sse = model._sweep(params[first:stop], qobs, bool(return_qsim),
                                 **forcing)
        local = torch.from_numpy(np.ascontiguousarray(
            scores_from_sse(sse, qobs, score)))
        if _group_on_device(group):
            local = local.to(torch.device("cuda", torch.cuda.current_device()))
        scores = allgather_scores(
            local, n, group=group,
            always_collective=always_collective).cpu().numpy()
    else:
        first, stop = 0, n
        with _lib.call_options(host_shards=_lib.host_shards_of(gpus)):
            qsim, sse = model._sweep(params, qobs, bool(return_qsim),
                                     **forcing)
        scores = scores_from_sse(sse, qobs, score)
    out = {"scores": scores, "score": score, "bounds": (first, stop)}
    if return_qsim:
        out["qsim"] = qsim
    return out


class ResidentSweep:
    """This rank's share of a sharded sweep whose arrays never leave HBM (the
    *_simulate_dev family through rrmpg_amd.device) -- what ``bench.py`` times
    and what a multi-GPU job loops over.

    ens: a resident ensemble of rrmpg_amd.device (forcing in HBM) on this
    rank's GPU; params: this rank's block of the parameter sets, a device
    tensor [n, k] (multi-catchment ensembles: [c, n, k]); qobs: device
    tensor [T] ([c, T]); total_units: sets (x catchments) of the WHOLE sweep
    over all ranks.  ``launch()`` enqueues the sweep of the block (discharge /
    storages into the given buffers, or score-only), ``gather()`` turns the
    fused squared-error sums into `score` and runs the one collective of the
    job, the all-gather of the per-set scores (RCCL over xGMI when the process
    group's backend is nccl; `on_host` moves the scores to the host first for
    a gloo group)."""

    def __init__(self, ens, params, qobs, total_units, score="mse",
                 qsim=None, storages=None, on_host=False, group=None):
        if score not in SCORES:
            raise ValueError("score must be one of %s" % (SCORES,))
        self.ens, self.params, self.qobs = ens, params, qobs
        self.qsim, self.storages = qsim, storages
        self.total_units, self.score = int(total_units), score
        self.on_host, self.group = on_host, group
        self.sse = torch.empty(params.shape[:-1], dtype=torch.float64,
                               device=params.device)
        self._inv_den = None
        if score == "nse":          # 1 / sum((obs - mean)^2), once
            obs = qobs.detach().cpu().numpy()
            den = ((obs - obs.mean(axis=-1, keepdims=True)) ** 2).sum(axis=-1)
            if np.any(den == 0):
                scores_from_sse(np.zeros(1), obs, "nse")     # raises
            inv = torch.as_tensor(1.0 / den, dtype=torch.float64,
                                  device=params.device)
            self._inv_den = inv if inv.dim() == 0 else inv[:, None]

    @classmethod
    def from_sampler(cls, ens, model, sets_per_rank, total_sets, first, key,
                     **kw):
        """Every rank draws ITS rows [first, first + sets_per_rank) of one
        global population of `total_sets` sets in HBM (counter-based Philox:
        no communication, the same population for any number of GPUs;
        rrmpg_amd.device.sample_params)."""
        from . import device as rrdev
        params = rrdev.sample_params(model, sets_per_rank, key,
                                     n_total=total_sets, first=first,
                                     device=ens.device)
        return cls(ens, params, total_units=total_sets, **kw)

    def launch(self):
        kw = {"qobs": self.qobs, "sse": self.sse}
        if self.storages is not None:
            import inspect
            names = inspect.signature(self.ens.run).parameters
            kw["storages" if "storages" in names else "storage"] = \
                self.storages
        self.ens.run(self.params, self.qsim, **kw)

    def local_scores(self):
        if self.score == "mse":
            return self.sse / self.qobs.shape[-1]
        return 1 - self.sse * self._inv_den

    def gather(self):
        local = self.local_scores().reshape(-1)
        return allgather_scores(local.cpu() if self.on_host else local,
                                self.total_units, group=self.group)

    def gather_begin(self):
        """gather() in two halves (allgather_scores_begin): returns a
        ScoreExchange; a gloo group's host tensors are exchanged at once."""
        local = self.local_scores().reshape(-1)
        if self.on_host:
            return ScoreExchange(result=allgather_scores(
                local.cpu(), self.total_units, group=self.group))
        return allgather_scores_begin(local, self.total_units,
                                      group=self.group)

    def step(self):
        self.launch()
        return self.gather()
