"""Interface to the Cemaneige snow routine (GPU ensemble engine).

Same class surface as the reference's rrmpg/models/cemaneige.py (Cemaneige
:26-359, _loss :362-386); ``simulate`` evaluates ALL parameter sets with one
call into librrhip (rr_cemaneige_simulate) instead of a Python loop over
run_cemaneige.  The parameter-independent forcing preprocessing
(cemaneige_utils) stays on the host.
"""

import numbers

import numpy as np

from .. import _lib
from ..utils.array_checks import check_for_negatives, validate_array_input
from .basemodel import BaseModel, new_outputs, out_ptr
from .cemaneige_utils import (calculate_solid_fraction,
                              extrapolate_precipitation,
                              extrapolate_temperature)


class Cemaneige(BaseModel):
    """Interface to the Cemaneige snow routine.

    Snow accounting routine of Valery (2010), see also Valery, Andreassian &
    Perrin, J. Hydrol. 517 (2014).  Daily data only.  If no model parameters
    are passed upon initialization, a random parameter set is generated.

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

    _param_list = ['CTG', 'Kf']

    _default_bounds = {'CTG': (0, 1),
                       'Kf': (0, 10)}

    _dtype = np.dtype([('CTG', np.float64),
                       ('Kf', np.float64)])

    def __init__(self, params=None):
        super().__init__(params=params)

    def simulate(self, prec, mean_temp, min_temp, max_temp, met_station_height,
                 snow_pack_init=0, thermal_state_init=0, altitudes=[],
                 return_storages=False, params=None):
        """Simulate the snow-routine of the Cemaneige model.

        If `altitudes` (median elevation of each equal-area layer) is given,
        the station series are extrapolated to every layer and the routine
        runs per layer; otherwise one layer at the station height is used.

        Args:
            prec: Array of daily precipitation sum [mm]
            mean_temp, min_temp, max_temp: Arrays of daily temperature [C]
            met_station_height: Height of the meteorological station [m].
            snow_pack_init: (optional) Initial value of the snow pack storage
            thermal_state_init: (optional) Initial thermal state of the pack
            altitudes: (optional) List of median layer altitudes [m]
            return_storages: (optional) also return G and eTG
                [timesteps, layers, sets].
            params: (optional) Numpy array of parameter sets of the model's
                custom dtype; all are evaluated at once on the GPU.

        Returns:
            outflow [timesteps, sets] and optionally G, eTG.

        Raises:
            ValueError: If one of the inputs contains invalid values.
            TypeError: If one of the inputs has an incorrect datatype.
            RuntimeError: If the meteorological arrays differ in size.
        """
        layers, inits = prepare_snow_inputs(
            prec, mean_temp, min_temp, max_temp, met_station_height,
            snow_pack_init, thermal_state_init, altitudes)
        params = self._resolve_params(params)
        out, _ = _run(layers, inits, params, True, bool(return_storages),
                      None)
        if return_storages:
            return tuple(out)
        return out[0]

    def fit(self, obs, prec, mean_temp, min_temp, max_temp,
            met_station_height, snow_pack_init=0, thermal_state_init=0,
            altitudes=[], batched=False):
        """Fit the Cemaneige model to an observed timeseries.

        scipy differential evolution over the default bounds, as in the
        reference (cemaneige.py:247-359).

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
        layers, inits = prepare_snow_inputs(
            prec, mean_temp, min_temp, max_temp, met_station_height,
            snow_pack_init, thermal_state_init, altitudes)
        args = (obs,) + layers + inits + (self._dtype,)
        return self._differential_evolution(_loss, args, batched)

    def _sweep(self, params, qobs, want_qsim, prec, mean_temp, min_temp,
               max_temp, met_station_height, snow_pack_init=0,
               thermal_state_init=0, altitudes=[]):
        layers, inits = prepare_snow_inputs(
            prec, mean_temp, min_temp, max_temp, met_station_height,
            snow_pack_init, thermal_state_init, altitudes)
        params = self._resolve_params(params)
        out, sse = _run(layers, inits, params, want_qsim, False, qobs)
        return out[0], sse


    def _resident(self, prec, mean_temp, min_temp, max_temp,
                  met_station_height, snow_pack_init=0, thermal_state_init=0,
                  altitudes=[], device=None):
        """simulate()'s forcing as an HBM-resident ensemble
        (rrmpg_amd.device.CemaneigeEnsemble) after simulate()'s own checks
        and layer preprocessing."""
        from .. import device as rrdev
        layers, inits = prepare_snow_inputs(
            prec, mean_temp, min_temp, max_temp, met_station_height,
            snow_pack_init, thermal_state_init, altitudes)
        return rrdev.CemaneigeEnsemble(
            layers[0], layers[1], layers[2], *inits,
            **({} if device is None else {"device": device}))


def prepare_snow_inputs(prec, mean_temp, min_temp, max_temp,
                        met_station_height, snow_pack_init,
                        thermal_state_init, altitudes, etp=None):
    """Validation + forcing preprocessing shared by the Cemaneige family.

    Follows the reference wrapper step by step (cemaneige.py:134-216,
    cemaneigegr4j.py:146-220): type/size checks with the same exceptions,
    then extrapolation to the elevation layers (or a single layer at station
    height) and the solid-precipitation fraction.

    Returns:
        (layer_prec, layer_mean_temp, frac_solid_prec) as [T, L] arrays and
        (snow_pack_init, thermal_state_init) as floats; with `etp` given the
        validated etp array is appended to the first tuple.
    """
    prec = validate_array_input(prec, np.float64, 'prec')
    mean_temp = validate_array_input(mean_temp, np.float64, 'mean_temp')
    min_temp = validate_array_input(min_temp, np.float64, 'min_temp')
    max_temp = validate_array_input(max_temp, np.float64, 'max_temp')
    series = [mean_temp, min_temp, max_temp]
    if etp is not None:
        etp = validate_array_input(etp, np.float64, 'pot. evapotranspiration')
        series.append(etp)
    if check_for_negatives(prec):
        raise ValueError("The precipitation array contains negative values.")
    if any(len(ar) != len(prec) for ar in series):
        raise RuntimeError("All meteorological input arrays must have the "
                           "same length.")

    if not isinstance(altitudes, list):
        raise TypeError("'altitudes' must be a list.")
    if len(altitudes) > 0:
        for val in altitudes:
            if not isinstance(val, numbers.Number):
                raise TypeError("All elements in 'altitudes must be numbers.")
        if met_station_height is None:
            raise ValueError("The height of the meteorological station is "
                             "missing.")
        if not isinstance(met_station_height, numbers.Number):
            raise TypeError("'met_station_height' must be a number.")
        altitudes = np.array(altitudes)
    if not isinstance(met_station_height, numbers.Number):
        raise TypeError("'met_station_height' must be a Number.")
    if not isinstance(snow_pack_init, numbers.Number):
        raise TypeError("'snow_pack_init' must be a Number.")
    if not isinstance(thermal_state_init, numbers.Number):
        raise TypeError("'thermal_state_init' must be a Number.")
    inits = (float(snow_pack_init), float(thermal_state_init))

    if len(altitudes) > 0:
        prec = extrapolate_precipitation(prec, altitudes, met_station_height)
        min_temp, mean_temp, max_temp = extrapolate_temperature(
            min_temp, mean_temp, max_temp, altitudes, met_station_height)
    else:
        prec = np.expand_dims(prec, axis=-1)
        mean_temp = np.expand_dims(mean_temp, axis=-1)
        min_temp = np.expand_dims(min_temp, axis=-1)
        max_temp = np.expand_dims(max_temp, axis=-1)
        altitudes = np.array([met_station_height])
    frac_solid_prec = calculate_solid_fraction(prec, altitudes, mean_temp,
                                               min_temp, max_temp)
    layers = (prec, mean_temp, frac_solid_prec)
    if etp is not None:
        layers = layers + (etp,)
    return layers, inits


def _run(layers, inits, params, want_outflow, want_storages, qobs):
    """One batched GPU call (include/rrhip.h: rr_cemaneige_simulate)."""
    prec, mean_temp, frac = layers
    lib = _lib.load()
    _lib.require_gpu()
    block, p_ptr, n = _lib.params_block(params, 2)
    t, nl = prec.shape
    outflow, = new_outputs((t, n), (want_outflow,))
    G, eTG = new_outputs((t, nl, n), (want_storages, want_storages))
    qobs_arr, qobs_ptr = _lib.f64(qobs)
    if qobs is not None and qobs_arr.shape[0] != t:
        raise ValueError("Arrays must have the same size.")
    sse = np.zeros(n) if qobs is not None else None
    keep, (p_prec, p_temp, p_frac) = _lib.f64s(prec, mean_temp, frac)
    rc = lib.rr_cemaneige_simulate_opt(p_prec, p_temp, p_frac, t, nl, inits[0],
                                   inits[1], p_ptr, n, out_ptr(outflow),
                                   out_ptr(G), out_ptr(eTG), qobs_ptr,
                                   out_ptr(sse),
        _lib.opts_ptr())
    del keep
    _lib.check(rc, "rr_cemaneige_simulate")
    return [outflow, G, eTG], sse


def _loss(X, *args):
    """Return the loss value (MSE) for the current parameter set."""
    obs = args[0]
    layers = args[1:4]
    inits = args[4:6]
    dtype = args[6]
    params = Cemaneige._params_from_population(X)
    _, sse = _run(layers, inits, params, False, False, obs)
    mse = sse / layers[0].shape[0]
    return mse if np.ndim(X) == 2 else mse[0]
