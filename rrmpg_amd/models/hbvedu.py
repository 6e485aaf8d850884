"""Interface to the educational version of the HBV model (GPU ensemble engine).

Same class surface as the reference's rrmpg/models/hbvedu.py (HBVEdu :21-307,
_loss :310-346); ``simulate`` evaluates ALL parameter sets with one call into
librrhip (rr_hbvedu_simulate) instead of a Python loop over run_hbvedu.
"""

import numpy as np

from .. import _lib
from ..utils.array_checks import check_for_negatives, validate_array_input
from .basemodel import BaseModel, new_outputs, out_ptr


class HBVEdu(BaseModel):
    """Interface to the educational version of the HBV model.

    Model of Aghakouchak & Habib, "Application of a conceptual hydrologic
    model in teaching hydrologic processes", IJEE 26.4 (2010).  Daily data
    only.  If no model parameters are passed upon initialization, a random
    parameter set is generated.

    Beyond the reference's interface (every default is the reference's
    behaviour): ``simulate(params=...)`` runs any number of parameter sets in
    ONE GPU call; ``fit(batched=True)`` evaluates a whole
    differential-evolution generation per GPU sweep (about 100 times faster,
    another optimiser trajectory than the reference's sequential search, which
    stays the default); ``rrmpg_amd.tools.monte_carlo(model, num, qobs,
    return_qsim=False, sampler='device', gpus=G)`` draws, sweeps and scores
    the sets in the GPUs' memory.

    Args:
        params: (optional) Dictionary containing all model parameters as
            separate key/value pairs.
    """

    _param_list = ['T_t', 'DD', 'FC', 'Beta', 'C', 'PWP', 'K_0', 'K_1',
                   'K_2', 'K_p', 'L']

    _default_bounds = {'T_t': (-1, 1),
                       'DD': (3, 7),
                       'FC': (100, 200),
                       'Beta': (1, 7),
                       'C': (0.01, 0.07),
                       'PWP': (90, 180),
                       'K_0': (0.05, 0.2),
                       'K_1': (0.01, 0.1),
                       'K_2': (0.01, 0.05),
                       'K_p': (0.01, 0.05),
                       'L': (2, 5)}

    _dtype = np.dtype([(name, np.float64) for name in _param_list])

    def __init__(self, params=None):
        super().__init__(params=params)

    def simulate(self, temp, prec, month, PE_m, T_m, snow_init=0, soil_init=0,
                 s1_init=0, s2_init=0, return_storage=False, params=None):
        """Simulate rainfall-runoff process for given input.

        Args:
            temp: Array of (mean) temperature for each timestep.
            prec: Array of (summed) precipitation for each timestep.
            month: Array of integers [1, ..., 12]: month of each timestep.
            PE_m: long-term mean monthly potential evapotranspiration (12).
            T_m: long-term mean monthly temperature (12).
            snow_init, soil_init, s1_init, s2_init: (optional) initial states
                of the four reservoirs.
            return_storage: (optional) also return the four storages.
            params: (optional) Numpy array of parameter sets of the model's
                custom dtype; all are evaluated at once on the GPU.

        Returns:
            qsim [timesteps, sets] and optionally snow, soil, s1, s2.

        Raises:
            ValueError: If one of the inputs contains invalid values.
            TypeError: If one of the inputs has an incorrect datatype.
            RuntimeError: If the monthly arrays are not of size 12 or the
                daily arrays differ in size.
        """
        forcing = _validate(temp, prec, month, PE_m, T_m)
        inits = tuple(float(v) for v in (snow_init, soil_init, s1_init,
                                         s2_init))
        params = self._resolve_params(params)
        out, _ = _run(forcing, inits, params, True, bool(return_storage), None)
        if return_storage:
            return tuple(out)
        return out[0]

    def fit(self, qobs, temp, prec, month, PE_m, T_m, snow_init=0.,
            soil_init=0., s1_init=0., s2_init=0., batched=False):
        """Fit the HBVEdu model to a timeseries of discharge.

        scipy differential evolution over the default bounds, as in the
        reference (hbvedu.py:216-307).

        batched: (extension) False (default): the reference's own call -- one
            candidate per loss evaluation, immediate updating -- which
            reproduces the reference's seeded runs evaluation by evaluation
            (tests/test_gpu_fit_reference.py).  True: scipy gets a vectorised
            loss and every generation's population is ONE GPU sweep
            (updating='deferred') -- about a hundred times faster, but a
            DIFFERENT optimiser trajectory than the reference's: a seeded
            fit ends in other (equally good) parameters.

        Returns:
            res: A scipy OptimizeResult class object.
        """
        qobs = validate_array_input(qobs, np.float64, 'observed discharge')
        forcing = _validate(temp, prec, month, PE_m, T_m)
        inits = tuple(float(v) for v in (snow_init, soil_init, s1_init,
                                         s2_init))
        args = (qobs,) + forcing + inits + (self._dtype,)
        return self._differential_evolution(_loss, args, batched)

    def _sweep(self, params, qobs, want_qsim, temp, prec, month, PE_m, T_m,
               snow_init=0, soil_init=0, s1_init=0, s2_init=0):
        forcing = _validate(temp, prec, month, PE_m, T_m)
        inits = tuple(float(v) for v in (snow_init, soil_init, s1_init,
                                         s2_init))
        params = self._resolve_params(params)
        out, sse = _run(forcing, inits, params, want_qsim, False, qobs)
        return out[0], sse


    def _resident(self, temp, prec, month, PE_m, T_m, snow_init=0,
                  soil_init=0, s1_init=0, s2_init=0, device=None):
        """The forcing of a simulate() call as an HBM-resident ensemble
        (rrmpg_amd.device.HBVEduEnsemble), after simulate()'s own input
        checks: what ``monte_carlo(..., sampler='device')`` sweeps."""
        from .. import device as rrdev
        temp, prec, month0, PE_m, T_m = _validate(temp, prec, month, PE_m,
                                                   T_m)
        return rrdev.HBVEduEnsemble(
            temp, prec, month0 + 1, PE_m, T_m, float(snow_init),
            float(soil_init), float(s1_init), float(s2_init),
            **({} if device is None else {"device": device}))


def _validate(temp, prec, month, PE_m, T_m):
    """Input checks of simulate()/fit() (reference: hbvedu.py:133-164)."""
    temp = validate_array_input(temp, np.float64, 'temperature')
    prec = validate_array_input(prec, np.float64, 'precipitation')
    if check_for_negatives(prec):
        raise ValueError("In the precipitation array are negative values.")
    month = validate_array_input(month, np.int8, 'month')
    if any(len(arr) != len(temp) for arr in [prec, month]):
        raise RuntimeError("The arrays of the temperature, precipitation and "
                           "month data must be of equal size.")
    PE_m = validate_array_input(PE_m, np.float64, 'PE_m')
    T_m = validate_array_input(T_m, np.float64, 'T_m')
    if any(len(arr) != 12 for arr in [PE_m, T_m]):
        raise RuntimeError("The monthly potential evapotranspiration and "
                           "temperature array must be of length 12.")
    if (np.min(month) < 1) or (np.max(month) > 12):
        raise ValueError("The month array must be between an integer1 (Jan) "
                         "and 12 (Dec).")
    # zero-based month index; `month` is our private copy
    month -= 1
    return temp, prec, month, PE_m, T_m


def _run(forcing, inits, params, want_qsim, want_storage, qobs):
    """One batched GPU call (include/rrhip.h: rr_hbvedu_simulate)."""
    temp, prec, month0, PE_m, T_m = forcing
    lib = _lib.load()
    _lib.require_gpu()
    block, p_ptr, n = _lib.params_block(params, 11)
    t = prec.shape[0]
    out = new_outputs((t, n), (want_qsim,) + (want_storage,) * 4)
    qobs_arr, qobs_ptr = _lib.f64(qobs)
    if qobs is not None and qobs_arr.shape[0] != t:
        raise ValueError("Arrays must have the same size.")
    sse = np.zeros(n) if qobs is not None else None
    month0 = np.ascontiguousarray(month0, dtype=np.int8)
    keep, (p_temp, p_prec, p_pe, p_tm) = _lib.f64s(temp, prec, PE_m, T_m)
    rc = lib.rr_hbvedu_simulate_opt(
        p_temp, p_prec, month0.ctypes.data_as(_lib._i8p), p_pe, p_tm, t,
        *inits, p_ptr, n, *[out_ptr(a) for a in out], qobs_ptr, out_ptr(sse),
        _lib.opts_ptr())
    del keep
    _lib.check(rc, "rr_hbvedu_simulate")
    return out, sse


def _loss(X, *args):
    """Return the loss value (MSE) for the current parameter set."""
    qobs = args[0]
    forcing = args[1:6]
    inits = args[6:10]
    dtype = args[10]
    params = HBVEdu._params_from_population(X)
    _, sse = _run(forcing, inits, params, False, False, qobs)
    mse = sse / forcing[1].shape[0]
    return mse if np.ndim(X) == 2 else mse[0]
