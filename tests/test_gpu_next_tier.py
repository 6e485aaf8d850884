"""GPU parity of the next-tier models (SWE-SCA hysteresis snow routine, ice
melt, couplings with GR4J) against the reference's Excel known answers
(reference: test/test_models.py:293-310, 336-356), the golden fixtures and
the oracle.  Snow states (G, eTG, sca, ice melt, snow melt) involve no
transcendental and are asserted bit-exact; discharge and the GR4J stores at
1e-10 relative."""

import numpy as np
import pytest

from .conftest import golden, rel_err, snow_same

pytestmark = pytest.mark.gpu

RTOL = 1e-10
ALTS = [550, 620, 700, 785, 920]


@pytest.fixture(scope="module")
def models():
    from rrmpg_amd import _lib
    _lib.load()
    _lib.require_gpu()
    import rrmpg_amd.models as m
    return m


def _records(cls, flat):
    p = np.zeros(flat.shape[0], dtype=cls._dtype)
    for k, name in enumerate(cls._param_list):
        p[name] = flat[:, k]
    return p


def test_hyst_kat_excel(models):
    g = golden("kat_cemaneigehystgr4j")
    cls = models.CemaneigeHystGR4J
    m = cls(params=dict(zip(cls._param_list, g["params"].tolist())))
    qsim = m.simulate(g["prec"], g["mean_temp"], g["min_temp"], g["max_temp"],
                      g["etp"], met_station_height=700, altitudes=ALTS,
                      s_init=0.5, r_init=0.4)
    assert np.allclose(qsim.flatten(), g["qsim_excel"])  # the reference's test
    out = m.simulate(g["prec"], g["mean_temp"], g["min_temp"], g["max_temp"],
                     g["etp"], met_station_height=700, altitudes=ALTS,
                     s_init=0.5, r_init=0.4, return_storages=True)
    names = ["qsim", "G", "eTG", "s_store", "r_store", "sca", "rain"]
    for a, n in zip(out, names):
        assert rel_err(a, g["ref_" + n], floor=1e-6) < RTOL, n
    assert np.array_equal(out[6], g["ref_rain"])


def test_hystice_kat_excel(models):
    g = golden("kat_cemaneigehystgr4jice")
    cls = models.CemaneigeHystGR4JIce
    m = cls(params=dict(zip(cls._param_list, g["params"].tolist())))
    qsim = m.simulate(g["prec"], g["mean_temp"], g["min_temp"], g["max_temp"],
                      g["etp"], g["frac_ice"], met_station_height=700,
                      altitudes=ALTS, s_init=0.5, r_init=0.4, sca_init=0.2)
    assert np.allclose(qsim.flatten(), g["qsim_excel"])  # the reference's test
    out = m.simulate(g["prec"], g["mean_temp"], g["min_temp"], g["max_temp"],
                     g["etp"], g["frac_ice"], met_station_height=700,
                     altitudes=ALTS, s_init=0.5, r_init=0.4, sca_init=0.2,
                     return_storages=True)
    names = ["qsim", "G", "eTG", "s_store", "r_store", "sca", "icemelt",
             "snowmelt", "rain"]
    for a, n in zip(out, names):
        assert rel_err(a, g["ref_" + n], floor=1e-6) < RTOL, n


def _device_run(models, hyst, ice, g, forcing, frac_ice):
    """Through the host C-ABI with the [T, L] arrays of the fixture."""
    from rrmpg_amd.models import _snowgr4j as core
    cls = {(True, False): models.CemaneigeHystGR4J,
           (False, True): models.CemaneigeGR4JIce,
           (True, True): models.CemaneigeHystGR4JIce}[(hyst, ice)]
    layers = (forcing["layer_prec"], forcing["layer_mean"],
              forcing["frac_solid"], forcing["etp"])
    out, _ = core.run(hyst, ice, layers, frac_ice, tuple(g["inits"]),
                      _records(cls, g["params"]), True, True, None)
    out["rain"] = core.rain_per_layer(layers, g["params"].shape[0])
    return out, cls, layers


def _check(out, g, ref, keys, exact):
    idx = g["stride_idx"]
    for k in keys:
        a = out[k]
        if a.ndim == 3:
            assert rel_err(a[idx], g[k], floor=1e-6) < RTOL, k
            assert rel_err(a[-1], g[k + "_last"], floor=1e-6) < RTOL, k
        else:
            assert rel_err(a, g[k], floor=1e-6) < RTOL, k
        if k in exact:
            # vs the oracle: thermal state and snow-covered area bit for bit,
            # packs and layer means within conftest.SNOW_TOL
            snow_same(a, ref[k], exact=k in ("eTG", "sca"), what=k)
        else:
            assert rel_err(a, ref[k], floor=1e-9) < RTOL, k


def test_next_tier_golden_and_oracle(models, oracle):
    h = golden("syn_cemaneigehystgr4j")
    forcing = (h["layer_prec"], h["layer_mean"], h["etp"], h["frac_solid"])
    snow_exact = {"G", "eTG", "sca", "icemelt", "snowmelt", "rain"}

    out, _, _ = _device_run(models, True, False, h, h, None)
    ref = oracle.simulate_snow_gr4j(True, False, *forcing, h["inits"],
                                    h["params"], return_storages=True)
    _check(out, h, ref, ["qsim", "G", "eTG", "s_store", "r_store", "sca",
                         "rain"], snow_exact)

    g = golden("syn_cemaneigegr4jice")
    out, _, _ = _device_run(models, False, True, g, h, g["frac_ice"])
    ref = oracle.simulate_snow_gr4j(False, True, *forcing, g["inits"],
                                    g["params"], frac_ice=g["frac_ice"],
                                    return_storages=True)
    _check(out, g, ref, ["qsim", "G", "eTG", "s_store", "r_store", "icemelt"],
           snow_exact)

    g = golden("syn_cemaneigehystgr4jice")
    out, _, _ = _device_run(models, True, True, g, h, g["frac_ice"])
    ref = oracle.simulate_snow_gr4j(True, True, *forcing, g["inits"],
                                    g["params"], frac_ice=g["frac_ice"],
                                    return_storages=True)
    _check(out, g, ref, ["qsim", "G", "eTG", "s_store", "r_store", "sca",
                         "icemelt", "snowmelt", "rain"], snow_exact)


def test_next_tier_random_sweeps_and_scores(models, oracle):
    from rrmpg_amd.models import _snowgr4j as core
    from rrmpg_amd.utils.metrics import calc_mse
    h = golden("syn_cemaneigehystgr4j")
    t = 1500
    layers = tuple(h[k][:t] for k in ("layer_prec", "layer_mean",
                                      "frac_solid", "etp"))
    forcing = (layers[0], layers[1], layers[3], layers[2])
    fice = np.array([0.0, 0.1, 0.3, 0.6, 0.9])
    for (hyst, ice), cls in [((True, False), models.CemaneigeHystGR4J),
                             ((False, True), models.CemaneigeGR4JIce),
                             ((True, True), models.CemaneigeHystGR4JIce)]:
        np.random.seed(11)
        p = cls().get_random_params(193)          # x4 up to 10 -> LDS UH tier
        flat = np.stack([p[k] for k in cls._param_list], 1)
        inits = (3.0, -0.2, 0.4, 0.5, 0.6)
        ref = oracle.simulate_snow_gr4j(hyst, ice, *forcing, inits, flat,
                                        frac_ice=fice if ice else None,
                                        return_storages=True, nthreads=8)
        out, _ = core.run(hyst, ice, layers, fice if ice else None, inits, p,
                          True, True, None)
        for k, a in out.items():
            if a is None:
                continue
            if k in ("G", "eTG", "sca", "icemelt", "snowmelt"):
                snow_same(a, ref[k], exact=k in ("eTG", "sca"),
                          what=(hyst, ice, k))
            else:
                assert rel_err(a, ref[k], floor=1e-9) < RTOL, (hyst, ice, k)
        # fused score == MSE of the series; score-only == with series
        qobs = ref["qsim"][:, 0] * 1.05
        _, sse = core.run(hyst, ice, layers, fice if ice else None, inits, p,
                          False, False, qobs)
        for j in (0, 7, 192):
            want = calc_mse(qobs, out["qsim"][:, j])
            assert abs(sse[j] / t - want) <= 1e-10 * max(want, 1e-12)


def test_next_tier_validation_and_limits(models):
    m = models.CemaneigeHystGR4JIce()
    s = [1., 2., 3.]
    with pytest.raises(TypeError, match="'sca_init' must be a Number"):
        m.simulate(s, s, s, s, s, [0.1], 500, sca_init="x")
    with pytest.raises(TypeError, match="'s_init' must be a Number"):
        m.simulate(s, s, s, s, s, [0.1], 500, s_init="x")
    with pytest.raises(ValueError, match="frac_ice must be a 1D array"):
        m.simulate(s, s, s, s, s, np.ones((2, 2)), 500)
    with pytest.raises(TypeError, match="'s1_init' must be a Number"):
        models.CemaneigeHystGR4J().simulate(s, s, s, s, s, 500, s_init="x")
    with pytest.raises(ValueError, match="Invalid loss_metric"):
        models.CemaneigeHystGR4J().fit(s, s, s, s, s, s, 500,
                                       loss_metric="nse")


def test_next_tier_many_layers_vs_oracle(models, oracle):
    """More than 8 elevation layers run through the HBM-scratch kernel
    (snow_gr4j_dyn_kernel): every series against the oracle, snow states
    bit-exact, for the three couplings, with register- and LDS-tier unit
    hydrographs, ragged set counts and the fused error sum."""
    from rrmpg_amd.models import _snowgr4j as core
    from rrmpg_amd.models import cemaneige_utils as cu
    from rrmpg_amd.utils import synthetic as syn
    from rrmpg_amd.utils.metrics import calc_mse
    t = 700
    f = syn.make_forcing(t)
    for nl in (9, 13):
        alts = np.linspace(520, 3100, nl)
        lp = cu.extrapolate_precipitation(f["prec"], alts, 500)
        lmin, lmean, lmax = cu.extrapolate_temperature(
            f["tmin"] - 2, f["temp"] - 2, f["tmax"] - 2, alts, 500)
        fr = cu.calculate_solid_fraction(lp, alts, lmean, lmin, lmax)
        layers = (lp, lmean, fr, f["etp"])
        fice = np.linspace(0.0, 0.8, nl)
        inits = (2.0, -0.1, 0.3, 0.5, 0.6)
        for (hyst, ice), cls in [((True, False), models.CemaneigeHystGR4J),
                                 ((False, True), models.CemaneigeGR4JIce),
                                 ((True, True), models.CemaneigeHystGR4JIce)]:
            np.random.seed(31 + nl)
            n = 77
            p = cls().get_random_params(n)
            flat = np.stack([p[k] for k in cls._param_list], 1)
            if nl == 13:            # long unit hydrographs: LDS tier, or
                # (x4 beyond 20) the HBM scratch the host path sizes itself
                flat[:, cls._param_list.index("x4")] = \
                    np.random.uniform(0.6, 33.0 if hyst and ice else 14.0, n)
                for j, k in enumerate(cls._param_list):
                    p[k] = flat[:, j]
            out, _ = core.run(hyst, ice, layers, fice if ice else None, inits,
                              p, True, True, None)
            ref = oracle.simulate_snow_gr4j(
                hyst, ice, lp, lmean, f["etp"], fr, inits, flat,
                frac_ice=fice if ice else None, return_storages=True)
            for key in ("G", "eTG") + (("sca",) if hyst else ()):
                snow_same(out[key], ref[key], exact=key in ("eTG", "sca"),
                          what=(nl, hyst, ice, key))
            for key in ("qsim", "s_store", "r_store") + \
                    (("icemelt",) if ice else ()) + \
                    (("snowmelt",) if hyst and ice else ()):
                assert rel_err(out[key], ref[key], floor=1e-9) < 1e-10, \
                    (nl, hyst, ice, key)
            qobs = ref["qsim"][:, 5] * 1.05
            _, sse = core.run(hyst, ice, layers, fice if ice else None, inits,
                              p, False, False, qobs)
            for j in (0, 5, n - 1):
                want = calc_mse(qobs, out["qsim"][:, j])
                assert abs(sse[j] / t - want) <= 1e-10 * max(want, 1e-12)


def test_next_tier_fit_losses(models):
    """_loss / _loss_Q_SCA follow the reference's definitions, including
    CemaneigeHystGR4J's KGE loss that is KGE itself (quirk Q8)."""
    from rrmpg_amd.models import cemaneigehystgr4j as hmod
    from rrmpg_amd.models import cemaneigehystgr4jice as himod
    from rrmpg_amd.models import _snowgr4j as core
    from rrmpg_amd.utils.metrics import calc_kge, calc_mse
    h = golden("syn_cemaneigehystgr4j")
    t = 800
    layers = tuple(h[k][:t] for k in ("layer_prec", "layer_mean",
                                      "frac_solid", "etp"))
    inits = tuple(h["inits"])
    X = h["params"][2]
    p = _records(models.CemaneigeHystGR4J, X[None, :])
    out, _ = core.run(True, False, layers, None, inits, p, True, True, None)
    q = out["qsim"][:, 0]
    obs = q * 1.1 + 0.01
    assert abs(hmod._loss(X, obs, layers, inits, "mse")
               - calc_mse(obs, q)) < 1e-12
    assert abs(hmod._loss(X, obs, layers, inits, "kge")
               - calc_kge(obs, q)) < 1e-12                      # KGE itself
    ndsi = tuple(np.clip(out["sca"][:, b, 0] * 100 + 3, 0, 100)
                 for b in range(5))
    want = 0.75 * calc_mse(obs, q) + sum(
        0.05 * calc_mse(ndsi[b], out["sca"][:, b, 0] * 100) for b in range(5))
    got = hmod._loss_Q_SCA(X, obs, layers, ndsi, inits, "mse")
    assert abs(got - want) <= 1e-10 * want
    # what fit_Q_SCA really runs: the whole population scored in HBM
    # (QScaScorer) -- equal to the host scoring, both metrics, and for the
    # ice variant
    Xpop = np.stack([X, h["params"][3], h["params"][5]], 1)
    scorer = core.QScaScorer(False, layers, None, inits, obs, ndsi)
    for metric in ("mse", "kge"):
        host = hmod._loss_Q_SCA(Xpop, obs, layers, ndsi, inits, metric)
        dev = hmod._loss_Q_SCA(Xpop, obs, layers, ndsi, inits, metric, scorer)
        assert host.shape == dev.shape == (3,)
        assert np.max(np.abs(dev - host) / np.abs(host)) < 1e-9, metric
        one = hmod._loss_Q_SCA(Xpop[:, 1], obs, layers, ndsi, inits, metric,
                               scorer)
        assert abs(one - dev[1]) <= 1e-12 * abs(dev[1])
    # observations no KGE is defined for -- an NDSI band that is constantly 0
    # (a snow-free low band), a constant band -- are fine for the MSE loss, as
    # for the reference's calc_mse (cemaneigehystgr4j.py:661-667), and raise
    # calc_kge's RuntimeError for the KGE loss on either path
    flat_ndsi = (np.zeros(t), np.full(t, 40.0)) + ndsi[2:]
    sc_flat = core.QScaScorer(False, layers, None, inits, obs, flat_ndsi)
    host = hmod._loss_Q_SCA(Xpop, obs, layers, flat_ndsi, inits, "mse")
    dev = hmod._loss_Q_SCA(Xpop, obs, layers, flat_ndsi, inits, "mse", sc_flat)
    assert np.max(np.abs(dev - host) / np.abs(host)) < 1e-9
    for sc in (None, sc_flat):
        with pytest.raises(RuntimeError, match="KGE not definied"):
            hmod._loss_Q_SCA(Xpop, obs, layers, flat_ndsi, inits, "kge", sc)
    gi = golden("syn_cemaneigehystgr4jice")
    Xi2 = np.stack([gi["params"][1], gi["params"][4]], 1)
    sc_i = core.QScaScorer(True, layers, gi["frac_ice"], inits, obs, ndsi)
    host = himod._loss_Q_SCA(Xi2, obs, layers, gi["frac_ice"], ndsi, inits,
                             "mse")
    dev = himod._loss_Q_SCA(Xi2, obs, layers, gi["frac_ice"], ndsi, inits,
                            "mse", sc_i)
    assert np.max(np.abs(dev - host) / np.abs(host)) < 1e-9
    # population form
    pop = hmod._loss(np.stack([X, h["params"][3]], 1), obs, layers, inits,
                     "mse")
    assert pop.shape == (2,) and pop[0] == hmod._loss(X, obs, layers, inits,
                                                      "mse")
    # ice variant: 1 - KGE
    g = golden("syn_cemaneigehystgr4jice")
    Xi = g["params"][1]
    pi = _records(models.CemaneigeHystGR4JIce, Xi[None, :])
    oi, _ = core.run(True, True, layers, g["frac_ice"], inits, pi, True,
                     False, None)
    qi = oi["qsim"][:, 0]
    li = himod._loss(Xi, obs, layers, g["frac_ice"], inits, "kge")
    assert abs(li - (1 - calc_kge(obs, qi))) < 1e-12


def test_next_tier_resident_ensembles_match_host_path(models):
    """rrmpg_amd.device.SnowGR4JEnsemble (the *_dev entry points with torch
    memory) gives the same discharge and scores as the host-pointer path."""
    import torch
    from rrmpg_amd import device as rrdev
    from rrmpg_amd.models import _snowgr4j as core
    h = golden("syn_cemaneigehystgr4j")
    t = 900
    layers = tuple(h[k][:t] for k in ("layer_prec", "layer_mean",
                                      "frac_solid", "etp"))
    fice = np.array([0.0, 0.1, 0.3, 0.6, 0.9])
    inits = (3.0, -0.2, 0.4, 0.5, 0.6)
    for (hyst, ice), cls in [((True, False), models.CemaneigeHystGR4J),
                             ((False, True), models.CemaneigeGR4JIce),
                             ((True, True), models.CemaneigeHystGR4JIce)]:
        np.random.seed(21)
        p = cls().get_random_params(130)
        out, _ = core.run(hyst, ice, layers, fice if ice else None, inits, p,
                          True, False, None)
        qobs_h = out["qsim"][:, 3] * 0.9 + 0.05
        _, sse_h = core.run(hyst, ice, layers, fice if ice else None, inits,
                            p, False, False, qobs_h)
        ens = rrdev.SnowGR4JEnsemble(
            hyst, ice, layers[0], layers[1], layers[2], layers[3],
            frac_ice=fice if ice else None, snow_pack_init=inits[0],
            thermal_state_init=inits[1], sca_init=inits[2], s_init=inits[3],
            r_init=inits[4])
        params = ens.upload_params(p)
        q = ens.new_output(130)
        sse = ens.run(params, q, qobs=torch.from_numpy(qobs_h).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(q.cpu().numpy(), out["qsim"])
        assert np.array_equal(sse.cpu().numpy(), sse_h)


def test_next_tier_days_decided_from_the_record(models, oracle):
    """The days the kernels decide for the whole wave from the day's record
    (snow_core.h cema_day_io: frost in every layer; snownext_kernels.h: the
    hysteresis routine's idle days, the ice-melt loop skipped under frost):
    long frosts, dry spells, temperatures of exactly +0 and -0, layers that
    disagree -- and a wave that must NOT take the ice-melt shortcut, because
    one of its sets has a negative degree-day factor and melts ice in the
    frost.  Discharge against the oracle at 1e-10."""
    import torch
    from rrmpg_amd import device as rrdev
    rng = np.random.default_rng(5)
    T, L, n = 700, 5, 96
    base = 5 * np.sin(2 * np.pi * np.arange(T) / 180.0) + rng.normal(0, 2, T)
    temp = base[:, None] - np.linspace(0.2, 2.5, L)[None, :]
    temp[100:140] -= 12.0                        # a long frost
    temp[50] = 0.0
    temp[51] = -0.0
    temp[52, :2] = 0.4
    temp[52, 2:] = -0.4
    wet = rng.random(T) < 0.35
    wet[110:130] = False                         # ... and dry inside it
    prec = np.repeat(wet * rng.gamma(0.8, 6.0, T), L).reshape(T, L)
    frac = np.clip(0.5 - temp / 5.0, 0.0, 1.0)
    etp = np.clip(1.5 + 0.1 * base, 0, None)
    fice = np.array([0.0, 0.1, 0.3, 0.6, 0.9])
    inits = (2.0, -0.3, 0.4, 0.5, 0.6)
    for (hyst, ice), cls in [((True, False), models.CemaneigeHystGR4J),
                             ((False, True), models.CemaneigeGR4JIce),
                             ((True, True), models.CemaneigeHystGR4JIce)]:
        np.random.seed(33)
        p = cls().get_random_params(n)
        if ice:
            p["DDF"][70] = -0.5              # second wave: no shortcut
        flat = np.stack([p[k] for k in cls._param_list], axis=1)
        ref = oracle.simulate_snow_gr4j(hyst, ice, prec, temp, etp, frac,
                                        inits, flat,
                                        frac_ice=fice if ice else None)
        ens = rrdev.SnowGR4JEnsemble(
            hyst, ice, prec, temp, frac, etp,
            frac_ice=fice if ice else None, snow_pack_init=inits[0],
            thermal_state_init=inits[1], sca_init=inits[2], s_init=inits[3],
            r_init=inits[4])
        par = ens.upload_params(p)
        q = ens.new_output(n)
        ens.run(par, q)
        torch.cuda.synchronize()
        got = q.cpu().numpy()
        assert rel_err(got, ref, floor=1e-6) < RTOL, (hyst, ice)
        if ice:      # the set that melts ice in the frost is not a zero
            assert np.abs(got[100:140, 70] - got[100:140, 69]).max() > 0
        qobs = torch.from_numpy(np.ascontiguousarray(ref[:, 5])).cuda()
        sse = ens.run(par, None, qobs=qobs).cpu().numpy()
        want = ((ref - ref[:, 5:6]) ** 2).sum(0)
        assert np.allclose(sse, want, rtol=1e-8, atol=1e-12), (hyst, ice)


def test_score_only_sweeps_sorted_by_tier_keep_every_sets_bits(models):
    """A score-only sweep of four or more waves per SIMD takes its sets in
    the order of their ceil(x4) (csrc/gr4j.hip rr_gr4j_tier_sort_async) and
    lets every wave run in the narrowest hydrograph tier that holds its own
    sets (gr4j_core.h gr4j_wave_selects), the tiers' kernels side by side on
    streams of their own: the scores are those of the sweep that writes qsim
    (its order untouched, every wave in the launch's widest tier), set by
    set, bit for bit -- sets beyond the register tiers (x4 up to 18) and a
    ragged tail included."""
    import torch
    from rrmpg_amd import device as rrdev
    h = golden("syn_cemaneigehystgr4j")
    t = 120
    layers = tuple(h[k][:t] for k in ("layer_prec", "layer_mean",
                                      "frac_solid", "etp"))
    fice = np.array([0.0, 0.1, 0.3, 0.6, 0.9])
    n = 4 * 1024 * 64 + 777
    for (hyst, ice), cls, hi in [((True, False), models.CemaneigeHystGR4J, 10),
                                 ((True, True), models.CemaneigeHystGR4JIce,
                                  18)]:
        np.random.seed(5)
        p = cls().get_random_params(n)
        p["x4"] = np.random.uniform(1.1, hi, size=n)
        ens = rrdev.SnowGR4JEnsemble(
            hyst, ice, layers[0], layers[1], layers[2], layers[3],
            frac_ice=fice if ice else None, s_init=0.5, r_init=0.6)
        ens.max_x4 = float(hi)
        params = ens.upload_params(p)
        q = ens.new_output(n)
        ens.run(params, q)
        qobs = (q[:, 7] * 0.9 + 0.05).contiguous()
        sse_q = ens.run(params, q, qobs=qobs).clone()      # with qsim: plain
        sse_s = ens.run(params, None, qobs=qobs).clone()   # scores: by tier
        torch.cuda.synchronize()
        ens.check()
        assert torch.isfinite(sse_q).all()
        assert torch.equal(sse_q, sse_s)


def test_monte_carlo_every_model_class(models):
    """rrmpg_amd.tools.monte_carlo works for EVERY model class, as the
    reference's does (monte_carlo.py:64 goes through model.simulate): the
    fused score equals calc_mse of the returned columns, and a simulate()
    keyword the fused sweep does not know (return_storages) still works."""
    from rrmpg_amd.tools import monte_carlo
    from rrmpg_amd.utils.metrics import calc_mse
    from rrmpg_amd.utils import synthetic as syn
    f = syn.make_forcing(500)
    snow = dict(prec=f["prec"], mean_temp=f["temp"] - 3,
                min_temp=f["tmin"] - 3, max_temp=f["tmax"] - 3,
                met_station_height=500, altitudes=[550, 620, 700])
    cases = [
        (models.ABCModel, dict(prec=f["prec"])),
        (models.HBVEdu, dict(temp=f["temp"], prec=f["prec"],
                             month=f["month"], PE_m=f["PE_m"], T_m=f["T_m"],
                             soil_init=100.)),
        (models.GR4J, dict(prec=f["prec"], etp=f["etp"], s_init=.6)),
        (models.Cemaneige, dict(snow)),
        (models.CemaneigeGR4J, dict(snow, etp=f["etp"], s_init=.5)),
        (models.CemaneigeHystGR4J, dict(snow, etp=f["etp"], r_init=.4)),
        (models.CemaneigeGR4JIce, dict(snow, etp=f["etp"],
                                       frac_ice=[.1, .3, .6])),
        (models.CemaneigeHystGR4JIce, dict(snow, etp=f["etp"],
                                           frac_ice=[.1, .3, .6],
                                           sca_init=0.2)),
    ]
    qobs = np.abs(np.sin(np.arange(500) / 17.0)) * 3
    for cls, kw in cases:
        np.random.seed(4)
        res = monte_carlo(cls(), 70, qobs=qobs, **kw)
        assert res["qsim"].shape == (500, 70), cls.__name__
        want = np.array([calc_mse(qobs, res["qsim"][:, j])
                         for j in range(70)])
        assert np.max(np.abs(res["mse"] - want) / want) < 1e-12, cls.__name__
        np.random.seed(4)
        only = monte_carlo(cls(), 70, qobs=qobs, return_qsim=False, **kw)
        assert np.array_equal(only["mse"], res["mse"]), cls.__name__
        assert "qsim" not in only
    # a simulate() keyword outside the fused sweep: generic path
    np.random.seed(4)
    kw = dict(cases[5][1], return_storages=True)
    res = monte_carlo(models.CemaneigeHystGR4J(), 9, qobs=qobs, **kw)
    np.random.seed(4)
    ref = monte_carlo(models.CemaneigeHystGR4J(), 9, qobs=qobs, **cases[5][1])
    assert np.array_equal(res["qsim"], ref["qsim"])
    assert np.max(np.abs(res["mse"] - ref["mse"]) / ref["mse"]) < 1e-12
