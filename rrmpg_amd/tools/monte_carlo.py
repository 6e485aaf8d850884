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


def monte_carlo(model, num, qobs=None, return_qsim=True, gpus=None,
                score="mse", **kwargs):
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

    params = model.get_random_params(num=num)
    sweep = model._sweep
    accepted = inspect.signature(sweep).parameters
    if (not any(p.kind is p.VAR_KEYWORD for p in accepted.values())
            and any(k not in accepted for k in kwargs)):
        # a simulate() keyword the fused sweep does not take (return_storage
        # ...): go through simulate itself, as the reference does
        sweep = lambda *a, **kw: BaseModel._sweep(model, *a, **kw)  # noqa
    if score not in ("mse", "nse"):
        raise ValueError("score must be 'mse' or 'nse'")
    # per-call option of the host-pointer entry point (rr_<model>_simulate_opt,
    # include/rrhip.h): nothing process-wide is touched, so concurrent sweeps
    # from several threads keep their own shard counts
    with _lib.call_options(host_shards=_lib.host_shards_of(gpus)):
        qsim, sse = sweep(params, qobs, bool(return_qsim), **kwargs)

    result = {'params': params}
    if return_qsim:
        result['qsim'] = qsim
    if qobs is not None:
        result['mse'] = mse_from_sse(sse, len(qobs))
        if score == "nse":
            result['nse'] = nse_from_sse(sse, qobs)
    return result
