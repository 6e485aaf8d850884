"""Interface to the IceMelt + Cemaneige (hysteresis) + GR4J coupled model.

Same class surface as the reference's rrmpg/models/cemaneigehystgr4jice.py
(CemaneigeHystGR4JIce :30-593, _loss :595-638, _loss_Q_SCA :640-717); all
parameter sets of a ``simulate`` call run in one fused GPU kernel
(rr_cemaneigehystgr4jice_simulate).
"""

import numpy as np

from . import _snowgr4j as core
from .basemodel import BaseModel


class CemaneigeHystGR4JIce(BaseModel):
    """Interface to the IceMelt + Cemaneige Hysteresis + GR4J coupled model.

    SWE-SCA hysteresis snow routine (Riboust et al. 2019), degree-day ice melt
    on the glaciated fraction of each band (Nepal et al. 2017), GR4J routing.
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

    _param_list = ['CTG', 'Kf', 'Thacc', 'Rsp', 'x1', 'x2', 'x3', 'x4',
                   'DDF']

    _default_bounds = {'CTG': (0, 1),
                       'Kf': (0, 10),
                       'Thacc': (0, 1000),
                       'Rsp': (0, 1),
                       'x1': (10, 1200),
                       'x2': (-5, 3),
                       'x3': (20, 5000),
                       'x4': (1.1, 10),
                       'DDF': (0, 30)}

    _dtype = np.dtype([(name, np.float64) for name in _param_list])

    def __init__(self, params=None):
        super().__init__(params=params)

    def simulate(self, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
                 met_station_height, snow_pack_init=0, thermal_state_init=0,
                 sca_init=0, s_init=0, r_init=0, altitudes=[],
                 return_storages=False, params=None):
        """Simulate the IceMelt + Cemaneige Hysteresis + GR4J coupled model.

        Args: as CemaneigeHystGR4J.simulate plus
            frac_ice: fraction of glaciated area per elevation band [0 - 1]

        Returns:
            qsim [timesteps, sets] and optionally G, eTG, s_store, r_store,
            sca, ice_melt, snowmelt, rain.
        """
        layers, fice, inits = core.prepare(
            True, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        params = self._resolve_params(params)
        out, _ = core.run(True, True, layers, fice, inits, params, True,
                          bool(return_storages), None)
        if return_storages:
            return (out["qsim"], out["G"], out["eTG"], out["s_store"],
                    out["r_store"], out["sca"], out["icemelt"],
                    out["snowmelt"], core.rain_per_layer(layers, params.size))
        return out["qsim"]

    def _sweep(self, params, qobs, want_qsim, prec, mean_temp, min_temp,
               max_temp, etp, frac_ice, met_station_height, snow_pack_init=0,
               thermal_state_init=0, sca_init=0, s_init=0, r_init=0,
               altitudes=[]):
        """monte_carlo's sweep: one GPU call, squared errors accumulated in
        the kernel."""
        layers, fice, inits = core.prepare(
            True, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        params = self._resolve_params(params)
        out, sse = core.run(True, True, layers, fice, inits, params,
                            want_qsim, False, qobs)
        return out["qsim"], sse

    def _resident(self, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
                  met_station_height, snow_pack_init=0, thermal_state_init=0,
                  sca_init=0, s_init=0, r_init=0, altitudes=[], device=None):
        layers, fice, inits = core.prepare(
            True, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        return core.resident(True, True, layers, fice, inits, device)

    def fit(self, obs, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, loss_metric="mse", snow_pack_init=0,
            thermal_state_init=0, sca_init=0, s_init=0, r_init=0,
            altitudes=[], batched=False):
        """Fit the model to an observed discharge series (scipy differential
        evolution; loss_metric 'mse' or 'kge'; reference:
        cemaneigehystgr4jice.py:308-445).

        batched: (extension) False (default): the reference's own call -- one
            candidate per loss evaluation, immediate updating -- which
            reproduces the reference's seeded runs evaluation by evaluation
            (tests/test_gpu_fit_reference.py).  True: scipy gets a vectorised
            loss and every generation's population is ONE GPU sweep
            (updating='deferred') -- about a hundred times faster, but a
            DIFFERENT optimiser trajectory than the reference's: a seeded
            fit ends in other (equally good) parameters.

        Returns:
            res: A SciPy OptimizeResult object.
        """
        core.check_loss_metric(loss_metric)
        obs = core.validated_obs(obs)
        layers, fice, inits = core.prepare(
            True, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        args = (obs, layers, fice, inits, loss_metric)
        return self._differential_evolution(_loss, args, batched)

    def fit_Q_SCA(self, obs, prec, mean_temp, min_temp, max_temp, etp,
                  frac_ice, NDSI1, NDSI2, NDSI3, NDSI4, NDSI5,
                  met_station_height, loss_metric="mse", snow_pack_init=0,
                  thermal_state_init=0, sca_init=0, s_init=0, r_init=0,
                  altitudes=[], batched=False):
        """Fit to discharge AND the snow-covered area of five elevation bands
        (reference: cemaneigehystgr4jice.py:447-593).

        batched: (extension) False (default): the reference's own call -- one
            candidate per loss evaluation, immediate updating -- which
            reproduces the reference's seeded runs evaluation by evaluation
            (tests/test_gpu_fit_reference.py).  True: scipy gets a vectorised
            loss and every generation's population is ONE GPU sweep
            (updating='deferred') -- about a hundred times faster, but a
            DIFFERENT optimiser trajectory than the reference's: a seeded
            fit ends in other (equally good) parameters.

        Returns:
            res: A SciPy OptimizeResult object.
        """
        core.check_loss_metric(loss_metric)
        obs = core.validated_obs(obs)
        layers, fice, inits = core.prepare(
            True, True, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        ndsi = (NDSI1, NDSI2, NDSI3, NDSI4, NDSI5)
        # forcing, observations and NDSI series resident in HBM for the
        # whole optimisation; candidates are scored there
        scorer = core.QScaScorer(True, layers, fice, inits, obs, ndsi)
        args = (obs, layers, fice, ndsi, inits, loss_metric, scorer)
        return self._differential_evolution(_loss_Q_SCA, args, batched)


def _loss(X, *args):
    """Return the loss value for the current parameter set(s)."""
    obs, layers, fice, inits, loss_metric = args
    return core.loss_q(CemaneigeHystGR4JIce, True, True, False, X, obs,
                       layers, fice, inits, loss_metric)


def _loss_Q_SCA(X, *args):
    """Return the discharge + SCA loss for the current parameter set(s)."""
    obs, layers, fice, ndsi, inits, loss_metric = args[:6]
    return core.loss_q_sca(CemaneigeHystGR4JIce, True, X, obs, layers, fice,
                           ndsi, inits, loss_metric,
                           scorer=args[6] if len(args) > 6 else None)
