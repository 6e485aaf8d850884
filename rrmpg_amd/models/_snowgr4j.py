"""Shared implementation of the three next-tier couplings: the SWE-SCA
hysteresis snow routine and/or the degree-day ice melt in front of GR4J.

The reference spells each model out in its own 450-700 line module
(rrmpg/models/cemaneigehystgr4j.py, cemaneigegr4jice.py,
cemaneigehystgr4jice.py); their simulate / fit / fit_Q_SCA bodies differ only
in which optional pieces (hysteresis parameters, frac_ice, sca_init) exist.
The public classes in those three modules here are thin and delegate to the
functions below; every parameter set is simulated by ONE call into librrhip
(rr_cemaneigehystgr4j_simulate & co.).
"""

import numbers

import numpy as np

from .. import _lib
from ..utils.array_checks import validate_array_input
from ..utils.metrics import calc_kge, calc_mse
from .basemodel import new_outputs, out_ptr
from .cemaneige import prepare_snow_inputs


def prepare(hyst, ice, prec, mean_temp, min_temp, max_temp, etp, frac_ice,
            met_station_height, snow_pack_init, thermal_state_init, sca_init,
            s_init, r_init, altitudes):
    """Validation + forcing preprocessing; same checks, exceptions and order
    as the reference wrappers (e.g. cemaneigehystgr4j.py:153-243).

    Returns (layers, frac_ice or None, inits) with layers = (layer_prec,
    layer_mean_temp, frac_solid_prec, etp) and inits = (snow_pack_init,
    thermal_state_init, sca_init, s_init, r_init) as floats.
    """
    layers, snow_inits = prepare_snow_inputs(
        prec, mean_temp, min_temp, max_temp, met_station_height,
        snow_pack_init, thermal_state_init, altitudes, etp=etp)
    if hyst and not isinstance(sca_init, numbers.Number):
        raise TypeError("'sca_init' must be a Number.")
    if not isinstance(s_init, numbers.Number):
        raise TypeError("'s_init' must be a Number." if (hyst and ice)
                        else "'s1_init' must be a Number.")
    if not isinstance(r_init, numbers.Number):
        raise TypeError("'r_init' must be a Number.")
    if ice:
        if isinstance(frac_ice, np.ndarray) and frac_ice.ndim != 1:
            raise ValueError("frac_ice must be a 1D array.")
        frac_ice = np.ascontiguousarray(np.asarray(frac_ice),
                                        dtype=np.float64).ravel()
        if frac_ice.shape[0] != layers[0].shape[1]:
            raise ValueError("frac_ice must hold one value per elevation "
                             "layer.")
    else:
        frac_ice = None
    inits = snow_inits + (float(sca_init) if hyst else 0.0, float(s_init),
                          float(r_init))
    return layers, frac_ice, inits


def rain_per_layer(layers, num_sets):
    """The reference's `rain` output: prec - prec * frac_solid_prec per layer
    (cemaneigehyst_model.py:89-90), identical for every parameter set."""
    prec, _, frac, _ = layers
    rain = prec - prec * frac
    return np.repeat(rain[:, :, None], num_sets, axis=2)


def run(hyst, ice, layers, frac_ice, inits, params, want_qsim, want_storages,
        qobs):
    """One batched GPU call.  Returns (dict of outputs, sse)."""
    prec, mean_temp, frac, etp = layers
    lib = _lib.load()
    _lib.require_gpu()
    k = 6 + (2 if hyst else 0) + (1 if ice else 0)
    block, p_ptr, n = _lib.params_block(params, k)
    t, nl = prec.shape
    qsim, s_store, r_store, icemelt, snowmelt = new_outputs(
        (t, n), (want_qsim, want_storages, want_storages,
                 want_storages and ice, want_storages and hyst and ice))
    G, eTG, sca = new_outputs((t, nl, n), (want_storages, want_storages,
                                           want_storages and hyst))
    qobs_arr, qobs_ptr = _lib.f64(qobs)
    if qobs is not None and qobs_arr.shape[0] != t:
        raise ValueError("Arrays must have the same size.")
    sse = np.zeros(n) if qobs is not None else None
    arrays = [prec, mean_temp, etp] + ([frac_ice] if ice else []) + [frac]
    keep, ptrs = _lib.f64s(*arrays)
    if hyst and ice:
        rc = lib.rr_cemaneigehystgr4jice_simulate_opt(
            *ptrs, t, nl, *inits, p_ptr, n, out_ptr(qsim), out_ptr(G),
            out_ptr(eTG), out_ptr(s_store), out_ptr(r_store), out_ptr(sca),
            out_ptr(icemelt), out_ptr(snowmelt), qobs_ptr, out_ptr(sse),
        _lib.opts_ptr())
        what = "rr_cemaneigehystgr4jice_simulate"
    elif hyst:
        rc = lib.rr_cemaneigehystgr4j_simulate_opt(
            *ptrs, t, nl, *inits, p_ptr, n, out_ptr(qsim), out_ptr(G),
            out_ptr(eTG), out_ptr(s_store), out_ptr(r_store), out_ptr(sca),
            qobs_ptr, out_ptr(sse),
        _lib.opts_ptr())
        what = "rr_cemaneigehystgr4j_simulate"
    else:
        rc = lib.rr_cemaneigegr4jice_simulate_opt(
            *ptrs, t, nl, inits[0], inits[1], inits[3], inits[4], p_ptr, n,
            out_ptr(qsim), out_ptr(G), out_ptr(eTG), out_ptr(s_store),
            out_ptr(r_store), out_ptr(icemelt), qobs_ptr, out_ptr(sse),
        _lib.opts_ptr())
        what = "rr_cemaneigegr4jice_simulate"
    del keep
    _lib.check(rc, what)
    out = dict(qsim=qsim, G=G, eTG=eTG, s_store=s_store, r_store=r_store,
               sca=sca, icemelt=icemelt, snowmelt=snowmelt)
    return out, sse


def resident(hyst, ice, layers, frac_ice, inits, device=None):
    """simulate()'s forcing -- after its own checks and layer preprocessing
    (`prepare`) -- as an HBM-resident ensemble
    (rrmpg_amd.device.SnowGR4JEnsemble): what
    ``monte_carlo(..., sampler='device')`` sweeps."""
    from .. import device as rrdev
    return rrdev.SnowGR4JEnsemble(
        hyst, ice, layers[0], layers[1], layers[2], layers[3],
        frac_ice=frac_ice if ice else None, snow_pack_init=inits[0],
        thermal_state_init=inits[1], sca_init=inits[2], s_init=inits[3],
        r_init=inits[4], **({} if device is None else {"device": device}))


def check_loss_metric(loss_metric):
    if loss_metric not in ("mse", "kge"):
        raise ValueError("Invalid loss_metric. Choose 'mse' or 'kge'.")


def loss_q(cls, hyst, ice, kge_as_is, X, obs, layers, frac_ice, inits,
           loss_metric):
    """Loss of one candidate (or, X 2-D, of a whole population) on discharge.

    mse: from the kernel's fused squared-error sum, no series leaves the GPU.
    kge: needs the series; kge_as_is reproduces CemaneigeHystGR4J's loss,
    which returns KGE itself rather than 1 - KGE (reference:
    cemaneigehystgr4j.py:608-609); CemaneigeHystGR4JIce uses 1 - KGE
    (cemaneigehystgr4jice.py:633-634).
    """
    check_loss_metric(loss_metric)
    params = cls._params_from_population(X)
    if loss_metric == "mse":
        _, sse = run(hyst, ice, layers, frac_ice, inits, params, False, False,
                     obs)
        loss = sse / layers[0].shape[0]
    else:
        out, _ = run(hyst, ice, layers, frac_ice, inits, params, True, False,
                     None)
        kge = np.array([calc_kge(obs, out["qsim"][:, j])
                        for j in range(params.size)])
        loss = kge if kge_as_is else 1 - kge
    return loss if np.ndim(X) == 2 else loss[0]


class QScaScorer:
    """The discharge + snow-covered-area loss of fit_Q_SCA evaluated where the
    series are: forcing, observed discharge and the five NDSI series are
    uploaded once, every candidate population is simulated with all state
    series left in HBM, and the six squared-error / KGE ingredients per
    candidate come from one column-sum pass each (rr_column_sums_dev) -- no
    [T, L, N] array crosses PCIe and no Python loop runs over candidates
    (the reference: one run + six metric calls per candidate,
    cemaneigehystgr4j.py:615-691)."""

    def __init__(self, ice, layers, frac_ice, inits, obs, ndsi):
        import torch
        from .. import device as rrdev
        self.torch, self.rrdev = torch, rrdev
        self.ens = rrdev.SnowGR4JEnsemble(
            True, ice, layers[0], layers[1], layers[2], layers[3],
            frac_ice=frac_ice if ice else None, snow_pack_init=inits[0],
            thermal_state_init=inits[1], sca_init=inits[2], s_init=inits[3],
            r_init=inits[4])
        self.obs = np.ascontiguousarray(obs, dtype=np.float64)
        # the model's sca is a fraction, the NDSI series are in percent:
        # compare sca with NDSI / 100 (MSE scales by 1e4, KGE not at all)
        self.ndsi = [np.ascontiguousarray(nd, dtype=np.float64).ravel() / 100
                     for nd in ndsi]
        dev = self.ens.device
        self.obs_d = torch.from_numpy(self.obs).to(dev)
        self.ndsi_d = [torch.from_numpy(nd).to(dev) for nd in self.ndsi]

    def losses(self, params, loss_metric):
        from ..utils.metrics import scores_from_sums
        ens = self.ens
        block = ens.upload_params(params)
        n = block.shape[0]
        qsim = ens.new_output(n)
        st = ens.new_storages(n)
        ens.run(block, qsim, storages=st)
        ens.check()
        key = "mse" if loss_metric == "mse" else "kge"
        # only the score that is asked for: the MSE is defined for any
        # observations (an NDSI band that is constantly 0, a constant
        # discharge), as calc_mse is in the reference's loss
        # (cemaneigehystgr4j.py:661-667); the KGE raises for them as
        # calc_kge does.  Moments about mean(obs) (column_sums' shift).
        def score(series, obs_d, obs_h):
            shift = float(np.mean(obs_h))
            sums = self.rrdev.column_sums(series, obs_d, shift).cpu().numpy()
            return scores_from_sums(sums, obs_h, only=(key,),
                                    shift=shift)[key]
        part = score(qsim, self.obs_d, self.obs)
        total = 0.75 * (part if key == "mse" else 1 - part)
        for b in range(5):
            part = score(st["sca"][:, b, :], self.ndsi_d[b], self.ndsi[b])
            total = total + 0.05 * (part * 1e4 if key == "mse" else 1 - part)
        return total


def loss_q_sca(cls, ice, X, obs, layers, frac_ice, ndsi, inits, loss_metric,
               scorer=None):
    """Multi-objective loss on discharge (75 %) and the snow-covered area of
    the five elevation bands (5 % each, in percent against the NDSI series);
    reference: cemaneigehystgr4j.py:615-691.  With a QScaScorer (what
    fit_Q_SCA builds) everything is scored in HBM; without one the series
    are brought to the host and scored with calc_mse / calc_kge one candidate
    at a time, as the reference does."""
    check_loss_metric(loss_metric)
    params = cls._params_from_population(X)
    if scorer is not None:
        losses = scorer.losses(params, loss_metric)
        return losses if np.ndim(X) == 2 else losses[0]
    out, _ = run(True, ice, layers, frac_ice, inits, params, True, True, None)
    losses = np.zeros(params.size)
    for j in range(params.size):
        outflow = out["qsim"][:, j]
        scas = [out["sca"][:, b, j].flatten() * 100 for b in range(5)]
        if loss_metric == "mse":
            parts = [calc_mse(obs, outflow)] + [calc_mse(nd, sc) for nd, sc
                                                in zip(ndsi, scas)]
        else:
            parts = [1 - calc_kge(obs, outflow)] + [1 - calc_kge(nd, sc)
                                                    for nd, sc
                                                    in zip(ndsi, scas)]
        losses[j] = 0.75 * parts[0] + sum(0.05 * v for v in parts[1:])
    return losses if np.ndim(X) == 2 else losses[0]


def validated_obs(obs):
    return validate_array_input(obs, np.float64, 'obs')
