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
    access downloading the block once; ``p.tensor`` is the [num, k] device
    tensor itself."""

    def __init__(self, model, tensor):
        self._model, self.tensor, self._host = model, tensor, None

    def numpy(self):
        if self._host is None:
            flat = self.tensor.cpu().numpy()
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
        return int(self.tensor.shape[0])

    @property
    def dtype(self):
        return self._model._dtype

    @property
    def shape(self):
        return (len(self),)


def _monte_carlo_resident(model, num, qobs, score, seed, kwargs):
    """sampler='device': sets drawn in HBM (numpy's Philox stream under
    `seed`, rrmpg_amd.device.sample_params), swept against the resident
    forcing, only the scores come back."""
    import torch
    from .. import device as rrdev
    if not hasattr(model, "_resident"):
        raise ValueError("sampler='device' is not available for %s"
                         % type(model).__name__)
    ens = model._resident(**kwargs)
    if seed is None:
        # no key given: one draw from numpy's global generator, so that
        # np.random.seed(s) in front of the call still fixes the sweep
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    params = rrdev.sample_params(model, num, int(seed), device=ens.device)
    q = torch.as_tensor(qobs, dtype=torch.float64, device=ens.device)
    if q.numel() != ens.num_timesteps:
        raise ValueError("Arrays must have the same size.")
    sse = ens.run(params, None, qobs=q)
    if hasattr(ens, "check"):
        ens.check()
    sse = sse.cpu().numpy()
    result = {'params': DeviceParams(model, params),
              'mse': mse_from_sse(sse, len(qobs))}
    if score == "nse":
        result['nse'] = nse_from_sse(sse, qobs)
    return result


def monte_carlo(model, num, qobs=None, return_qsim=True, gpus=None,
                score="mse", sampler="numpy", seed=None, **kwargs):
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
            is three quarters of the call.
        seed: (optional) the key of sampler='device'.
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
        if qobs is None or return_qsim or gpus is not None:
            raise ValueError("sampler='device' scores resident sets: it "
                             "needs qobs, return_qsim=False and gpus=None")
        return _monte_carlo_resident(model, num, qobs, score, seed, kwargs)
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
