"""Interface to the IceMelt + Cemaneige + GR4J coupled model.

Same class surface as the reference's rrmpg/models/cemaneigegr4jice.py
(CemaneigeGR4JIce :28-417, _loss :419-455); all parameter sets of a
``simulate`` call run in one fused GPU kernel (rr_cemaneigegr4jice_simulate).
"""

import numpy as np

from . import _snowgr4j as core
from .basemodel import BaseModel


class CemaneigeGR4JIce(BaseModel):
    """Interface to the IceMelt + Cemaneige + GR4J coupled hydrological model.

    Degree-day ice melt (after Nepal et al. 2017) on the glaciated fraction
    of each elevation band, added to the Cemaneige outflow, routed by GR4J.
    Daily data only.  If no model parameters are passed upon initialization,
    a random parameter set is generated.

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

    _param_list = ['CTG', 'Kf', 'x1', 'x2', 'x3', 'x4', 'DDF']

    _default_bounds = {'CTG': (0, 1),
                       'Kf': (1, 15),
                       'x1': (100, 1200),
                       'x2': (-5, 3),
                       'x3': (20, 300),
                       'x4': (1.1, 2.9),
                       'DDF': (1, 30)}

    _dtype = np.dtype([(name, np.float64) for name in _param_list])

    def __init__(self, params=None):
        super().__init__(params=params)

    def simulate(self, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
                 met_station_height, snow_pack_init=0, thermal_state_init=0,
                 s_init=0, r_init=0, altitudes=[], return_storages=False,
                 params=None):
        """Simulate the IceMelt + Cemaneige + GR4J coupled model.

        Args:
            prec, mean_temp, min_temp, max_temp, etp: daily series
            frac_ice: fraction of glaciated area per elevation band [0 - 1]
            met_station_height: Height of the meteorological station [m]
            snow_pack_init, thermal_state_init: (optional) initial snow states
            s_init, r_init: (optional) initial production / routing storage
                as fraction of x1 / x3
            altitudes: (optional) List of median layer altitudes [m]
            return_storages: (optional) also return G, eTG
                [timesteps, layers, sets], s_store, r_store, ice_melt
            params: (optional) Numpy array of parameter sets of the model's
                custom dtype; all are evaluated at once on the GPU.

        Returns:
            qsim [timesteps, sets] and optionally G, eTG, s_store, r_store,
            ice_melt.
        """
        layers, fice, inits = core.prepare(
            False, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, 0,
            s_init, r_init, altitudes)
        params = self._resolve_params(params)
        out, _ = core.run(False, True, layers, fice, inits, params, True,
                          bool(return_storages), None)
        if return_storages:
            return (out["qsim"], out["G"], out["eTG"], out["s_store"],
                    out["r_store"], out["icemelt"])
        return out["qsim"]

    def _sweep(self, params, qobs, want_qsim, prec, mean_temp, min_temp,
               max_temp, etp, frac_ice, met_station_height, snow_pack_init=0,
               thermal_state_init=0, s_init=0, r_init=0,
               altitudes=[]):
        """monte_carlo's sweep: one GPU call, squared errors accumulated in
        the kernel."""
        layers, fice, inits = core.prepare(
            False, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, 0,
            s_init, r_init, altitudes)
        params = self._resolve_params(params)
        out, sse = core.run(False, True, layers, fice, inits, params,
                            want_qsim, False, qobs)
        return out["qsim"], sse

    def _resident(self, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
                  met_station_height, snow_pack_init=0, thermal_state_init=0,
                  s_init=0, r_init=0, altitudes=[], device=None):
        layers, fice, inits = core.prepare(
            False, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, 0,
            s_init, r_init, altitudes)
        return core.resident(False, True, layers, fice, inits, device)

    def fit(self, obs, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init=0, thermal_state_init=0,
            s_init=0, r_init=0, altitudes=[], batched=False):
        """Fit the model to an observed discharge series (scipy differential
        evolution on the MSE; reference: cemaneigegr4jice.py:290-417).

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
        obs = core.validated_obs(obs)
        layers, fice, inits = core.prepare(
            False, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, 0,
            s_init, r_init, altitudes)
        args = (obs, layers, fice, inits)
        return self._differential_evolution(_loss, args, batched)


def _loss(X, *args):
    """Return the loss value (MSE) for the current parameter set(s)."""
    obs, layers, fice, inits = args
    return core.loss_q(CemaneigeGR4JIce, False, True, False, X, obs, layers,
                       fice, inits, "mse")
