"""Monte-Carlo simulation for rrmpg_amd.models.

Same function as the reference's rrmpg/tools/monte_carlo.py (:19-76): draw
`num` random parameter sets, simulate them all, and (with qobs) score each by
its mean squared error.  Here the whole sweep is ONE batched GPU call and the
per-set squared errors are accumulated inside the kernel's time loop, so the
reference's per-set Python loop over calc_mse (monte_carlo.py:66-71) and its
strided column reads disappear.
"""

import inspect

import numpy as np

from ..models.basemodel import BaseModel
from ..utils.array_checks import validate_array_input
from .. import _lib
from ..utils.metrics import mse_from_sse, nse_from_sse


class DeviceParams:
    """The parameter sets of a ``monte_carlo(..., sampler='device')`` sweep:
    they were drawn in HBM and stay there until somebody looks --
    ``np.asarray(p)``, ``p[i]``, ``p['x1']``, ``len(p)`` and ``p.dtype`` work
    as on the structured array ``get_random_params`` returns, the first
    access downloading the block once.  ``p.shards`` are the [n_j, k] device
    tensors in set order (one per shard of a ``gpus=G`` sweep, each on its
    shard's GPU; one tensor otherwise); ``p.tensor`` is the whole [num, k]
    block on the first shard's GPU (of a sharded sweep: gathered there on
    first access)."""

    def __init__(self, model, shards):
        if not isinstance(shards, (list, tuple)):
            shards = [shards]
        self._model, self.shards, self._host = model, list(shards), None
        self._tensor = self.shards[0] if len(self.shards) == 1 else None

    @property
    def tensor(self):
        if self._tensor is None:
            import torch
            dev = self.shards[0].device
            self._tensor = torch.cat([s.to(dev) for s in self.shards])
        return self._tensor

    def numpy(self):
        if self._host is None:
            flat = np.concatenate([s.cpu().numpy() for s in self.shards])
            rec = np.zeros(flat.shape[0], dtype=self._model._dtype)
            for j, name in enumerate(self._model._param_list):
                rec[name] = flat[:, j]
            self._host = rec
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        return self.numpy()[key]

    def __len__(self):
        return sum(int(s.shape[0]) for s in self.shards)

    @property
    def dtype(self):
        return self._model._dtype

    @property
    def shape(self):
        return (len(self),)


def _shard_devices(gpus):
    """The HIP device of every shard of a ``gpus=`` argument: shard j runs on
    device (current + j) % device_count, the rule of the host-pointer
    family's fan-out (csrc/api.hip host_fan_out) -- more shards than devices
    is allowed, they then share a GPU on streams of their own."""
    import torch
    shards = _lib.host_shards_of(gpus)
    if shards is None:
        return None
    ndev = max(1, _lib.device_count())
    cur = torch.cuda.current_device()
    count = ndev if shards < 0 else shards
    return [(cur + j) % ndev for j in range(count)]


_shard_streams = {}


def _shard_stream(dev, j):
    """Stream of shard j on `dev`, kept for the life of the process: torch's
    allocator caches blocks per stream, so a shard's parameter block, score
    vector and workspace come from the cache on every call but the first (a
    fresh stream per call meant fresh hipMallocs: 10 ms for eight shards)."""
    import torch
    key = (dev.index, j)
    if key not in _shard_streams:
        _shard_streams[key] = torch.cuda.Stream(dev)
    return _shard_streams[key]


def _resident_scores(sse, qobs, score):
    result = {'mse': mse_from_sse(sse, len(qobs))}
    if score == "nse":
        result['nse'] = nse_from_sse(sse, qobs)
    return result


_cliques = {}


def _clique(devices):
    """In-process RCCL communicators, one per device of `devices` (the C-ABI's
    rr_comm_init_all), kept for the life of the process."""
    import ctypes
    key = tuple(devices)
    if key not in _cliques:
        lib = _lib.load()
        comms = (ctypes.c_void_p * len(key))()
        devs = (ctypes.c_int * len(key))(*key)
        _lib.check(lib.rr_comm_init_all(comms, len(key), devs),
                   "rr_comm_init_all")
        _cliques[key] = [ctypes.c_void_p(c) for c in comms]
    return _cliques[key]


def _monte_carlo_resident(model, num, qobs, score, seed, gpus, kwargs,
                          exchange="host"):
    """sampler='device': sets drawn in HBM (numpy's Philox stream under
    `seed`, rrmpg_amd.device.sample_params), swept against the resident
    forcing, only the scores come back.

    The population is ONE counter-based stream of `num` rows whatever the
    number of GPUs: a shard draws rows [first, stop) of it
    (rr_sample_params_dev(first, n_total)), so the scores of a sharded sweep
    equal the single-GPU sweep's bit for bit.

    * ``gpus=G`` / ``'all'``: shard j runs on device (current + j) %
      device_count on a stream of its own; it draws its rows, sweeps them
      score-only against that device's replica of the forcing and sends
      its 8 B per set straight into its slice of ONE
      pinned host vector -- G copies of num/G x 8 B over G PCIe links beside
      each other, no GPU-to-GPU step: the scores' destination is the host
      (``exchange='host'``, the default).  ``exchange='rccl'``: the shards'
      sums are all-gathered over RCCL / xGMI first -- one in-process
      communicator per GPU (rr_comm_init_all), the G rr_allgather_metric
      calls of the one thread inside a group -- so that EVERY GPU holds all
      `num` sums (result['sse_device'], one tensor per GPU: a caller that
      goes on working on the GPUs, resampling around the best sets, say);
      the host's copy then comes from the first GPU alone.  One rank per
      device: G may not exceed the number of GPUs.
    * inside an initialised ``torch.distributed`` group of several ranks (one
      process per GPU under torchrun): this rank draws and sweeps its block
      of rrmpg_amd.sharding.shard_bounds and the one collective of the job,
      the all-gather of the per-set sums, gives every rank all of them (RCCL
      for an nccl group, gloo on the host otherwise); 'params' holds this
      rank's block."""
    import torch
    import torch.distributed as dist
    from .. import device as rrdev
    from .. import sharding
    if not hasattr(model, "_resident"):
        raise ValueError("sampler='device' is not available for %s"
                         % type(model).__name__)
    devices = _shard_devices(gpus)
    # on the CURRENT device (a rank of a torchrun job has selected its GPU
    # with torch.cuda.set_device / rrmpg_amd._lib.set_device), not on the
    # ensembles' default "cuda:0"
    ens = model._resident(
        device=torch.device("cuda", torch.cuda.current_device()), **kwargs)
    if seed is None:
        # no key given: one draw from numpy's global generator, so that
        # np.random.seed(s) in front of the call still fixes the sweep
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    seed = int(seed)
    if len(qobs) != ens.num_timesteps:
        raise ValueError("Arrays must have the same size.")

    world = (dist.get_world_size() if dist.is_available()
             and dist.is_initialized() else 1)
    if world > 1:
        # ONE key for the job: rank 0's (a key drawn from every rank's own
        # global generator would give every rank another population)
        key = [seed]
        dist.broadcast_object_list(key, src=0)
        seed = int(key[0])
        first, stop = sharding.shard_bounds(num, world, dist.get_rank())
        params = rrdev.sample_params(model, stop - first, seed, n_total=num,
                                     first=first, device=ens.device)
        q = torch.as_tensor(qobs, dtype=torch.float64, device=ens.device)
        sse = ens.run(params, None, qobs=q)
        if hasattr(ens, "check"):
            ens.check()
        on_host = dist.get_backend() != "nccl"
        sse = sharding.allgather_scores(sse.cpu() if on_host else sse, num)
        result = _resident_scores(sse.cpu().numpy(), qobs, score)
        result['params'] = DeviceParams(model, params)
        result['bounds'] = (first, stop)
        return result

    if exchange == "rccl" and devices is None:
        devices = [torch.cuda.current_device()]
    if exchange != "rccl" and (devices is None or len(devices) == 1
                               or num == 1):
        params = rrdev.sample_params(model, num, seed, device=ens.device)
        q = torch.as_tensor(qobs, dtype=torch.float64, device=ens.device)
        sse = ens.run(params, None, qobs=q)
        if hasattr(ens, "check"):
            ens.check()
        result = _resident_scores(sse.cpu().numpy(), qobs, score)
        result['params'] = DeviceParams(model, params)
        return result

    devices = devices[:num]
    count = len(devices)
    if exchange == "rccl" and len(set(devices)) != count:
        raise ValueError("exchange='rccl' needs one GPU per shard: gpus=%r "
                         "on %d device(s)" % (gpus, len(set(devices))))
    host = torch.empty(num, dtype=torch.float64).pin_memory()
    # one replica of the forcing and of the observations per GPU (the copies
    # are made here, behind the upload); shards sharing a GPU share both and
    # differ in workspace and stream only
    replicas, obs = {}, {}
    for d in devices:
        if d not in replicas:
            dev = torch.device("cuda", d)
            replicas[d] = ens if dev == ens.device else ens.replica(dev)
            obs[d] = torch.as_tensor(qobs, dtype=torch.float64).to(dev)
    for d in replicas:
        torch.cuda.synchronize(d)
    # Every step of a shard is asynchronous (draw, sweep, 8 B per set into
    # the pinned vector), so ONE host thread enqueues all shards, each on a
    # stream of its own on its GPU, and waits for them afterwards: the GPUs
    # run beside each other, and so do shards that share one.  (Host threads
    # per shard, the host-pointer family's way, bought nothing here but the
    # interpreter lock: 18.9 ms against 12.8 for eight shards on one GPU.)
    blocks, work = [], []
    for j, d in enumerate(devices):
        dev = torch.device("cuda", d)
        first, stop = sharding.shard_bounds(num, count, j)
        with torch.cuda.device(dev):
            mine = replicas[d].replica()
            stream = _shard_stream(dev, j)
            with torch.cuda.stream(stream):
                params = rrdev.sample_params(model, stop - first, seed,
                                             n_total=num, first=first,
                                             device=dev)
                sse = mine.run(params, None, qobs=obs[d])
                if exchange != "rccl":
                    host[first:stop].copy_(sse, non_blocking=True)
        blocks.append(params)
        work.append((dev, mine, stream, sse))
    gathered = None
    if exchange == "rccl":
        # every GPU gets all sums: the G collectives of the clique, issued by
        # this one thread inside a group, each on its shard's stream behind
        # the shard's sweep; the host's copy comes from the first GPU
        lib = _lib.load()
        comms = _clique(devices)
        gathered = []
        for dev, _, stream, _ in work:
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                gathered.append(torch.empty(num, dtype=torch.float64,
                                            device=dev))
        _lib.check(lib.rr_comm_group_start(), "rr_comm_group_start")
        try:
            for j, (dev, _, stream, sse) in enumerate(work):
                with torch.cuda.device(dev):
                    _lib.check(lib.rr_allgather_metric(
                        comms[j], sse.data_ptr(), sse.numel(),
                        gathered[j].data_ptr(), num, stream.cuda_stream),
                        "rr_allgather_metric")
        finally:
            _lib.check(lib.rr_comm_group_end(), "rr_comm_group_end")
        dev, _, stream, _ = work[0]
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            host.copy_(gathered[0], non_blocking=True)
    for dev, mine, stream, sse in work:
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            if hasattr(mine, "check"):
                mine.check()               # (waits for the shard's stream)
            stream.synchronize()
    result = _resident_scores(host.numpy(), qobs, score)
    result['params'] = DeviceParams(model, blocks)
    if gathered is not None:
        result['sse_device'] = gathered
    return result


def monte_carlo(model, num, qobs=None, return_qsim=True, gpus=None,
                score="mse", sampler="numpy", seed=None, exchange="host",
                **kwargs):
    """Perform Monte-Carlo-Simulation.

    Args:
        model: Any instance of a hydrological model of rrmpg_amd.models.
        num: Number of simulations.
        qobs: (optional) Array of observed streamflow.
        return_qsim: (optional, extension) set False to keep the
            [timesteps, num] discharge array on neither host nor device and
            return the per-set scores only (needs qobs) -- the mode for
            million-set sweeps.
        gpus: (optional, extension) spread the sweep over several GPUs of
            this node inside the one call: an int, or 'all' (None: the
            current device).  The parameter-set axis is cut into contiguous
            shards, one per GPU, each filling its columns of the one
            [timesteps, num] result (rrmpg_amd.sharding.sweep; a
            one-process-per-GPU job under torchrun calls that function).
        score: (optional, extension) 'mse' (default) or 'nse': with 'nse'
            the result also carries the Nash-Sutcliffe efficiency of every
            set (calc_nse's definition).
        sampler: (optional, extension) 'numpy' (default): the sets come from
            ``model.get_random_params`` -- numpy's global generator, the
            reference's contract: ``np.random.seed(s)`` in front of the call
            gives the reference's sets.  'device': the sets are drawn in the
            GPU's memory (numpy's Philox stream under `seed`; without one a
            key is taken from numpy's global generator) and never leave it
            unless looked at -- 'params' is then a DeviceParams; needs qobs
            and return_qsim=False.  The mode for million-set sweeps: at
            100,000 HBV-Edu sets drawing and uploading the sets on the host
            is three quarters of the call.  With ``gpus=G`` every GPU draws
            and sweeps its contiguous block of the ONE population `seed`
            names and only 8 B per set leave it: the scores equal the
            single-GPU call's bit for bit (BASELINE configs[3] as one call:
            ``monte_carlo(CemaneigeGR4J(), 1_000_000, qobs,
            return_qsim=False, score='nse', sampler='device', gpus=8)``).
            Called on every rank of a torchrun job (an initialised
            ``torch.distributed`` group), each rank sweeps its block and the
            scores are all-gathered (RCCL for an nccl group).
        seed: (optional) the key of sampler='device'.
        exchange: (optional, sampler='device' only) 'host' (default): every
            GPU sends its 8 B per set straight to the host; 'rccl': the
            per-set sums are first all-gathered over RCCL / xGMI between the
            GPUs of the call (in-process communicators), 'sse_device' in the
            result holds them on every GPU.
        **kwargs: Keyword arguments matching the inputs the model needs to
            perform a simulation; see help(model.simulate).

    Returns:
        A dictionary with the keys 'params' and 'qsim' (unless
        return_qsim=False) and, if qobs is given, 'mse': the
        mean-squared-error of each simulation (the reference's key and
        score, monte_carlo.py:66-71), plus 'nse' if score='nse'.

    Raises:
        ValueError: If any input contains invalid values.
        TypeError: If any of the inputs has a wrong datatype.
    """
    if not issubclass(model.__class__, BaseModel):
        raise TypeError("The model must be one of the models implemented in "
                        "the rrmpg.models module.")
    if not isinstance(num, int) or num < 1:
        raise TypeError("'n' must be a positive integer greate than zero.")
    if qobs is not None:
        qobs = validate_array_input(qobs, np.float64, 'qobs')
    elif not return_qsim:
        raise ValueError("return_qsim=False needs qobs to score the sets.")

    if score not in ("mse", "nse"):
        raise ValueError("score must be 'mse' or 'nse'")
    shards = _lib.host_shards_of(gpus)   # (checked before anything is drawn)
    if sampler not in ("numpy", "device"):
        raise ValueError("sampler must be 'numpy' or 'device'")
    if sampler == "device":
        if qobs is None or return_qsim:
            raise ValueError("sampler='device' scores resident sets: it "
                             "needs qobs and return_qsim=False")
        if exchange not in ("host", "rccl"):
            raise ValueError("exchange must be 'host' or 'rccl'")
        return _monte_carlo_resident(model, num, qobs, score, seed, gpus,
                                     kwargs, exchange)
    if exchange != "host":
        raise ValueError("exchange='rccl' needs sampler='device'")
    params = model.get_random_params(num=num)
    sweep = model._sweep
    accepted = inspect.signature(sweep).parameters
    if (not any(p.kind is p.VAR_KEYWORD for p in accepted.values())
            and any(k not in accepted for k in kwargs)):
        # a simulate() keyword the fused sweep does not take (return_storage
        # ...): go through simulate itself, as the reference does
        sweep = lambda *a, **kw: BaseModel._sweep(model, *a, **kw)  # noqa
    # per-call option of the host-pointer entry point (rr_<model>_simulate_opt,
    # include/rrhip.h): nothing process-wide is touched, so concurrent sweeps
    # from several threads keep their own shard counts
    with _lib.call_options(host_shards=shards):
        qsim, sse = sweep(params, qobs, bool(return_qsim), **kwargs)

    result = {'params': params}
    if return_qsim:
        result['qsim'] = qsim
    if qobs is not None:
        result['mse'] = mse_from_sse(sse, len(qobs))
        if score == "nse":
            result['nse'] = nse_from_sse(sse, qobs)
    return result
