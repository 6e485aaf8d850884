"""Interface to the Cemaneige (SWE-SCA hysteresis) + GR4J coupled model.

Same class surface as the reference's rrmpg/models/cemaneigehystgr4j.py
(CemaneigeHystGR4J :27-571, _loss :573-613, _loss_Q_SCA :615-691); all
parameter sets of a ``simulate`` call run in one fused GPU kernel
(rr_cemaneigehystgr4j_simulate).
"""

import numpy as np

from . import _snowgr4j as core
from .basemodel import BaseModel


class CemaneigeHystGR4J(BaseModel):
    """Interface to the Cemaneige Hysteresis + GR4J coupled hydrological model.

    Cemaneige (Valery 2010) with the linear SWE-SCA hysteresis of Riboust et
    al. (2019) in front of GR4J (Perrin et al. 2003).  Daily data only.  If no
    model parameters are passed upon initialization, a random parameter set
    is generated.

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

    _param_list = ['CTG', 'Kf', 'Thacc', 'Rsp', 'x1', 'x2', 'x3', 'x4']

    _default_bounds = {'CTG': (0, 1),
                       'Kf': (0, 10),
                       'Thacc': (0, 1000),
                       'Rsp': (0, 1),
                       'x1': (10, 1200),
                       'x2': (-5, 3),
                       'x3': (20, 5000),
                       'x4': (1.1, 10)}

    _dtype = np.dtype([(name, np.float64) for name in _param_list])

    _HYST, _ICE = True, False

    def __init__(self, params=None):
        super().__init__(params=params)

    def simulate(self, prec, mean_temp, min_temp, max_temp, etp,
                 met_station_height, snow_pack_init=0, thermal_state_init=0,
                 sca_init=0, s_init=0, r_init=0, altitudes=[],
                 return_storages=False, params=None):
        """Simulate the Cemaneige Hysteresis + GR4J coupled model.

        Args:
            prec, mean_temp, min_temp, max_temp, etp: daily series
            met_station_height: Height of the meteorological station [m]
            snow_pack_init, thermal_state_init, sca_init: (optional) initial
                snow states (snow pack, thermal state, snow-covered area)
            s_init, r_init: (optional) initial production / routing storage
                as fraction of x1 / x3
            altitudes: (optional) List of median layer altitudes [m]
            return_storages: (optional) also return G, eTG
                [timesteps, layers, sets], s_store, r_store, sca, rain
            params: (optional) Numpy array of parameter sets of the model's
                custom dtype; all are evaluated at once on the GPU.

        Returns:
            qsim [timesteps, sets] and optionally G, eTG, s_store, r_store,
            sca, rain.

        Raises:
            ValueError, TypeError, RuntimeError: as the reference's wrapper.
        """
        layers, _, inits = core.prepare(
            True, False, prec, mean_temp, min_temp, max_temp, etp, None,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        params = self._resolve_params(params)
        out, _ = core.run(True, False, layers, None, inits, params, True,
                          bool(return_storages), None)
        if return_storages:
            return (out["qsim"], out["G"], out["eTG"], out["s_store"],
                    out["r_store"], out["sca"],
                    core.rain_per_layer(layers, params.size))
        return out["qsim"]

    def _sweep(self, params, qobs, want_qsim, prec, mean_temp, min_temp,
               max_temp, etp, met_station_height, snow_pack_init=0,
               thermal_state_init=0, sca_init=0, s_init=0, r_init=0,
               altitudes=[]):
        """monte_carlo's sweep: one GPU call, squared errors accumulated in
        the kernel."""
        layers, frac_ice, inits = core.prepare(
            True, False, prec, mean_temp, min_temp, max_temp, etp, None,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        params = self._resolve_params(params)
        out, sse = core.run(True, False, layers, frac_ice, inits, params,
                            want_qsim, False, qobs)
        return out["qsim"], sse

    def _resident(self, prec, mean_temp, min_temp, max_temp, etp,
                  met_station_height, snow_pack_init=0, thermal_state_init=0,
                  sca_init=0, s_init=0, r_init=0, altitudes=[], device=None):
        layers, frac_ice, inits = core.prepare(
            True, False, prec, mean_temp, min_temp, max_temp, etp, None,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        return core.resident(True, False, layers, None, inits, device)

    def fit(self, obs, prec, mean_temp, min_temp, max_temp, etp,
            met_station_height, loss_metric="mse", snow_pack_init=0,
            thermal_state_init=0, sca_init=0, s_init=0, r_init=0,
            altitudes=[], batched=False):
        """Fit the model to an observed discharge series (scipy differential
        evolution; loss_metric 'mse' or 'kge'; reference:
        cemaneigehystgr4j.py:292-424).

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
        layers, _, inits = core.prepare(
            True, False, prec, mean_temp, min_temp, max_temp, etp, None,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        args = (obs, layers, inits, loss_metric)
        return self._differential_evolution(_loss, args, batched)

    def fit_Q_SCA(self, obs, prec, mean_temp, min_temp, max_temp, etp, NDSI1,
                  NDSI2, NDSI3, NDSI4, NDSI5, met_station_height,
                  loss_metric="mse", snow_pack_init=0, thermal_state_init=0,
                  sca_init=0, s_init=0, r_init=0, altitudes=[],
                  batched=False):
        """Fit to discharge AND the snow-covered area of five elevation bands
        (NDSI1..NDSI5, in percent); 75 % / 5 x 5 % weighting (reference:
        cemaneigehystgr4j.py:427-570).

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
        layers, _, inits = core.prepare(
            True, False, prec, mean_temp, min_temp, max_temp, etp, None,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes)
        ndsi = (NDSI1, NDSI2, NDSI3, NDSI4, NDSI5)
        # forcing, observations and NDSI series resident in HBM for the
        # whole optimisation; candidates are scored there
        scorer = core.QScaScorer(False, layers, None, inits, obs, ndsi)
        args = (obs, layers, ndsi, inits, loss_metric, scorer)
        return self._differential_evolution(_loss_Q_SCA, args, batched)


def _loss(X, *args):
    """Return the loss value for the current parameter set(s)."""
    obs, layers, inits, loss_metric = args
    return core.loss_q(CemaneigeHystGR4J, True, False, True, X, obs, layers,
                       None, inits, loss_metric)


def _loss_Q_SCA(X, *args):
    """Return the discharge + SCA loss for the current parameter set(s)."""
    obs, layers, ndsi, inits, loss_metric = args[:5]
    return core.loss_q_sca(CemaneigeHystGR4J, False, X, obs, layers, None,
                           ndsi, inits, loss_metric,
                           scorer=args[5] if len(args) > 5 else None)
