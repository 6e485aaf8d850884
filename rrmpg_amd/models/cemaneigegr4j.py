"""Interface to the Cemaneige + GR4J coupled model (GPU ensemble engine).

Same class surface as the reference's rrmpg/models/cemaneigegr4j.py
(CemaneigeGR4J :27-400, _loss :402-436); ``simulate`` evaluates ALL parameter
sets with one call into librrhip (rr_cemaneigegr4j_simulate): one fused
kernel, the snow routine's outflow feeds GR4J in registers.
"""

import numbers

import numpy as np

from .. import _lib
from ..utils.array_checks import validate_array_input
from .basemodel import BaseModel, new_outputs, out_ptr
from .cemaneige import prepare_snow_inputs


class CemaneigeGR4J(BaseModel):
    """Interface to the Cemaneige + GR4J coupled hydrological model.

    Cemaneige snow routine (Valery 2010) in front of GR4J (Perrin et al.
    2003).  Daily data only.  If no model parameters are passed upon
    initialization, a random parameter set is generated.

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

    _param_list = ['CTG', 'Kf', 'x1', 'x2', 'x3', 'x4']

    _default_bounds = {'CTG': (0, 1),
                       'Kf': (0, 10),
                       'x1': (100, 1200),
                       'x2': (-5, 3),
                       'x3': (20, 300),
                       'x4': (1.1, 2.9)}

    _dtype = np.dtype([(name, np.float64) for name in _param_list])

    def __init__(self, params=None):
        super().__init__(params=params)

    def simulate(self, prec, mean_temp, min_temp, max_temp, etp,
                 met_station_height, snow_pack_init=0, thermal_state_init=0,
                 s_init=0, r_init=0, altitudes=[], return_storages=False,
                 params=None):
        """Simulate the Cemaneige + GR4J coupled hydrological model.

        Args:
            prec: Array of daily precipitation sum [mm]
            mean_temp, min_temp, max_temp: Arrays of daily temperature [C]
            etp: Array of mean potential evapotranspiration [mm]
            met_station_height: Height of the meteorological station [m].
            snow_pack_init, thermal_state_init: (optional) initial snow states
            s_init, r_init: (optional) initial production / routing storage
                as fraction of x1 / x3.
            altitudes: (optional) List of median layer altitudes [m]
            return_storages: (optional) also return G, eTG
                [timesteps, layers, sets] and s_store, r_store.
            params: (optional) Numpy array of parameter sets of the model's
                custom dtype; all are evaluated at once on the GPU.

        Returns:
            qsim [timesteps, sets] and optionally G, eTG, s_store, r_store.

        Raises:
            ValueError: If one of the inputs contains invalid values.
            TypeError: If one of the inputs has an incorrect datatype.
            RuntimeError: If the meteorological arrays differ in size.
        """
        layers, inits = _prepare(prec, mean_temp, min_temp, max_temp, etp,
                                 met_station_height, snow_pack_init,
                                 thermal_state_init, s_init, r_init, altitudes)
        params = self._resolve_params(params)
        out, _ = _run(layers, inits, params, True, bool(return_storages),
                      None)
        if return_storages:
            return tuple(out)
        return out[0]

    def fit(self, obs, prec, mean_temp, min_temp, max_temp, etp,
            met_station_height, snow_pack_init=0, thermal_state_init=0,
            s_init=0, r_init=0, altitudes=[], batched=False):
        """Fit the Cemaneige + GR4J coupled model to an observed timeseries.

        scipy differential evolution over the default bounds, as in the
        reference (cemaneigegr4j.py:275-400).

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
        obs = validate_array_input(obs, np.float64, 'obs')
        layers, inits = _prepare(prec, mean_temp, min_temp, max_temp, etp,
                                 met_station_height, snow_pack_init,
                                 thermal_state_init, s_init, r_init, altitudes)
        args = (obs,) + layers + inits + (self._dtype,)
        return self._differential_evolution(_loss, args, batched)

    def _sweep(self, params, qobs, want_qsim, prec, mean_temp, min_temp,
               max_temp, etp, met_station_height, snow_pack_init=0,
               thermal_state_init=0, s_init=0, r_init=0, altitudes=[]):
        layers, inits = _prepare(prec, mean_temp, min_temp, max_temp, etp,
                                 met_station_height, snow_pack_init,
                                 thermal_state_init, s_init, r_init, altitudes)
        params = self._resolve_params(params)
        out, sse = _run(layers, inits, params, want_qsim, False, qobs)
        return out[0], sse


    def _resident(self, prec, mean_temp, min_temp, max_temp, etp,
                  met_station_height, snow_pack_init=0, thermal_state_init=0,
                  s_init=0, r_init=0, altitudes=[], device=None):
        """simulate()'s forcing as an HBM-resident ensemble
        (rrmpg_amd.device.CemaneigeGR4JEnsemble) after simulate()'s own
        checks and layer preprocessing."""
        from .. import device as rrdev
        layers, inits = _prepare(prec, mean_temp, min_temp, max_temp, etp,
                                 met_station_height, snow_pack_init,
                                 thermal_state_init, s_init, r_init, altitudes)
        return rrdev.CemaneigeGR4JEnsemble(
            layers[0], layers[1], layers[2], layers[3], *inits,
            **({} if device is None else {"device": device}))


def _prepare(prec, mean_temp, min_temp, max_temp, etp, met_station_height,
             snow_pack_init, thermal_state_init, s_init, r_init, altitudes):
    layers, snow_inits = prepare_snow_inputs(
        prec, mean_temp, min_temp, max_temp, met_station_height,
        snow_pack_init, thermal_state_init, altitudes, etp=etp)
    if not isinstance(s_init, numbers.Number):
        raise TypeError("'s1_init' must be a Number.")
    if not isinstance(r_init, numbers.Number):
        raise TypeError("'r_init' must be a Number.")
    return layers, snow_inits + (float(s_init), float(r_init))


def _run(layers, inits, params, want_qsim, want_storages, qobs):
    """One batched GPU call (include/rrhip.h: rr_cemaneigegr4j_simulate)."""
    prec, mean_temp, frac, etp = layers
    lib = _lib.load()
    _lib.require_gpu()
    block, p_ptr, n = _lib.params_block(params, 6)
    t, nl = prec.shape
    qsim, s_store, r_store = new_outputs(
        (t, n), (want_qsim, want_storages, want_storages))
    G, eTG = new_outputs((t, nl, n), (want_storages, want_storages))
    qobs_arr, qobs_ptr = _lib.f64(qobs)
    if qobs is not None and qobs_arr.shape[0] != t:
        raise ValueError("Arrays must have the same size.")
    sse = np.zeros(n) if qobs is not None else None
    keep, (p_prec, p_temp, p_etp, p_frac) = _lib.f64s(prec, mean_temp, etp,
                                                      frac)
    rc = lib.rr_cemaneigegr4j_simulate_opt(
        p_prec, p_temp, p_etp, p_frac, t, nl, *inits, p_ptr, n, out_ptr(qsim),
        out_ptr(G), out_ptr(eTG), out_ptr(s_store), out_ptr(r_store),
        qobs_ptr, out_ptr(sse),
        _lib.opts_ptr())
    del keep
    _lib.check(rc, "rr_cemaneigegr4j_simulate")
    return [qsim, G, eTG, s_store, r_store], sse


def _loss(X, *args):
    """Return the loss value (MSE) for the current parameter set."""
    obs = args[0]
    layers = args[1:5]
    inits = args[5:9]
    dtype = args[9]
    params = CemaneigeGR4J._params_from_population(X)
    _, sse = _run(layers, inits, params, False, False, obs)
    mse = sse / layers[0].shape[0]
    return mse if np.ndim(X) == 2 else mse[0]
