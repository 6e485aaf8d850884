"""Pin the CPU oracle (oracle/rr_oracle.c) to the reference.

Two anchors per model:
  * the known-answer data the reference's own unit tests hold (MATLAB / Excel
    outputs; reference: test/test_models.py:142-174, 201-210, 227-236,
    258-268), asserted at the reference's own np.allclose tolerance;
  * outputs of the reference's own source executed in the build container
    (tests/golden/gen_golden.py), asserted at 1e-12 relative -- bit-exact
    for the models without transcendentals (ABC, Cemaneige).
"""

import numpy as np
import pytest

from .conftest import golden, rel_err

TOL = 1e-12


def _layers(g):
    """[T, L] Cemaneige inputs rebuilt the way the reference wrapper does
    (reference: rrmpg/models/cemaneige.py:189-212)."""
    from rrmpg_amd.models import cemaneige_utils as cu
    lp = cu.extrapolate_precipitation(g["prec"], g["altitudes"], g["station"])
    lmin, lmean, lmax = cu.extrapolate_temperature(
        g["min_temp"], g["mean_temp"], g["max_temp"], g["altitudes"],
        g["station"])
    frac = cu.calculate_solid_fraction(lp, g["altitudes"], lmean, lmin, lmax)
    return lp, lmean, frac


def test_abc_synthetic_bit_exact(oracle):
    g = golden("syn_abc")
    q, s = oracle.simulate_abc(g["prec"], g["initial_state"], g["params"],
                               return_storage=True)
    assert np.array_equal(q, g["qsim"])
    assert np.array_equal(s, g["storage"])


def test_hbvedu_kat_matlab(oracle):
    g = golden("kat_hbvedu")
    out = oracle.simulate_hbvedu(g["temp"], g["prec"], g["month"] - 1,
                                 g["PE_m"], g["T_m"], g["inits"], g["params"],
                                 return_storage=True)
    # rescale mm/d -> m3/s as the reference test does (test_models.py:172)
    q = (out[0] * g["area"] * 1000) / (24 * 60 * 60)
    assert np.allclose(q.ravel(), g["qsim_matlab"])
    assert rel_err(q.ravel(), g["qsim_matlab"]) < 1e-10
    for a, name in zip(out, ["qsim", "snow", "soil", "s1", "s2"]):
        assert rel_err(a.ravel(), g["ref_" + name].ravel()) < TOL, name


def test_hbvedu_synthetic(oracle):
    g = golden("syn_hbvedu")
    out = oracle.simulate_hbvedu(g["temp"], g["prec"], g["month"] - 1,
                                 g["PE_m"], g["T_m"], g["inits"], g["params"],
                                 return_storage=True)
    idx = g["stride_idx"]
    for a, name in zip(out, ["qsim", "snow", "soil", "s1", "s2"]):
        assert rel_err(a[idx], g[name + "_strided"]) < TOL, name
        assert rel_err(a[:, :4], g[name + "_full"]) < TOL, name
        assert rel_err(a[-1], g[name + "_last"]) < TOL, name
        assert rel_err(a.sum(0), g[name + "_sum"]) < 1e-11, name


def test_gr4j_kat_excel(oracle):
    g = golden("kat_gr4j")
    out = oracle.simulate_gr4j(g["prec"], g["etp"], g["inits"], g["params"],
                               return_storage=True)
    assert np.allclose(out[0].ravel(), g["qsim_excel"])
    for a, name in zip(out, ["qsim", "s_store", "r_store"]):
        assert rel_err(a.ravel(), g["ref_" + name].ravel()) < TOL, name


def test_gr4j_synthetic(oracle):
    g = golden("syn_gr4j")
    out = oracle.simulate_gr4j(g["prec"], g["etp"], g["inits"], g["params"],
                               return_storage=True)
    idx = g["stride_idx"]
    for a, name in zip(out, ["qsim", "s_store", "r_store"]):
        assert rel_err(a[idx], g[name + "_strided"]) < TOL, name
        assert rel_err(a[:, :4], g[name + "_full"]) < TOL, name
        assert rel_err(a[-1], g[name + "_last"]) < TOL, name


def test_cemaneige_kat_excel(oracle):
    g = golden("kat_cemaneige")
    lp, lmean, frac = _layers(g)
    o, G, e = oracle.simulate_cemaneige(lp, lmean, frac, (0., 0.), g["params"],
                                        return_storages=True)
    assert np.allclose(o.ravel(), g["liquid_outflow_excel"])
    assert rel_err(o, g["ref_outflow"]) < TOL
    assert rel_err(G, g["ref_G"]) < TOL
    assert rel_err(e, g["ref_eTG"], floor=1e-6) < TOL


def test_cemaneige_synthetic(oracle):
    g = golden("syn_cemaneige")
    o, G, e = oracle.simulate_cemaneige(g["layer_prec"], g["layer_mean"],
                                        g["frac_solid"], g["inits"],
                                        g["params"], return_storages=True)
    idx = g["stride_idx"]
    assert rel_err(o, g["outflow"]) < TOL
    assert rel_err(G[idx], g["G_strided"]) < TOL
    assert rel_err(e[idx], g["eTG_strided"], floor=1e-6) < TOL
    assert rel_err(G[:, :, :2], g["G_full"]) < TOL
    assert rel_err(G[-1], g["G_last"]) < TOL


def test_cemaneige_single_layer(oracle):
    g = golden("syn_cemaneige_l1")
    p = golden("syn_cemaneige_prep")
    o = oracle.simulate_cemaneige(p["prec"][:, None], p["temp"][:, None],
                                  g["frac_solid"], (0., 0.), g["params"])
    assert rel_err(o, g["outflow"]) < TOL


def test_cemaneigegr4j_kat_excel(oracle):
    g = golden("kat_cemaneigegr4j")
    lp, lmean, frac = _layers(g)
    out = oracle.simulate_cemaneigegr4j(lp, lmean, g["etp"], frac, g["inits"],
                                        g["params"], return_storages=True)
    assert np.allclose(out[0].ravel(), g["qsim_excel"])
    for a, name in zip(out, ["qsim", "G", "eTG", "s_store", "r_store"]):
        assert rel_err(a, g["ref_" + name], floor=1e-6) < TOL, name


def test_cemaneigegr4j_synthetic(oracle):
    g = golden("syn_cemaneigegr4j")
    out = oracle.simulate_cemaneigegr4j(
        g["layer_prec"], g["layer_mean"], g["etp"], g["frac_solid"],
        g["inits"], g["params"], return_storages=True)
    idx = g["stride_idx"]
    assert rel_err(out[0], g["qsim"]) < TOL
    assert rel_err(out[1][idx], g["G_strided"]) < TOL
    assert rel_err(out[2][idx], g["eTG_strided"], floor=1e-6) < TOL
    assert rel_err(out[3][idx], g["s_store_strided"]) < TOL
    assert rel_err(out[4][idx], g["r_store_strided"]) < TOL


def test_edge_cases(oracle):
    g = golden("edge")
    kat = golden("kat_hbvedu")
    m0 = g["month40"] - 1
    syn = golden("syn_hbvedu")
    # NaN propagation (FC < 0): same NaN pattern, same finite prefix
    q, _, soil, _, _ = oracle.simulate_hbvedu(
        g["temp40"], g["prec40"], m0, syn["PE_m"], syn["T_m"],
        (0., 100., 3., 10.), g["hbv_nan_params"], return_storage=True)
    assert rel_err(q.ravel(), g["hbv_nan_qsim"]) < TOL
    assert rel_err(soil.ravel(), g["hbv_nan_soil"]) < TOL
    assert np.isnan(q).any()
    # T = 1, 2, 3: the loop starts at t = 1 (quirk Q3)
    for tt in (1, 2, 3):
        q, snow, *_ = oracle.simulate_hbvedu(
            g["temp40"][:tt], g["prec40"][:tt], m0[:tt], syn["PE_m"],
            syn["T_m"], (1., 100., 3., 10.), kat["params"],
            return_storage=True)
        assert rel_err(q.ravel(), g["hbv_T%d_qsim" % tt]) < TOL
        assert rel_err(snow.ravel(), g["hbv_T%d_snow" % tt]) < TOL
    # GR4J x3 < 0: pow NaN is swallowed by max(0, .) exactly as numba does
    q, s, r = oracle.simulate_gr4j(g["prec40"], g["etp40"], (0.6, 0.7),
                                   g["gr4j_nan_params"], return_storage=True)
    assert rel_err(q.ravel(), g["gr4j_nan_qsim"]) < TOL
    assert rel_err(r.ravel(), g["gr4j_nan_r"]) < TOL
    q, s, r = oracle.simulate_gr4j(g["prec40"][:1], g["etp40"][:1], (0.6, 0.7),
                                   np.array([350., 0.5, 90., 1.7]),
                                   return_storage=True)
    assert rel_err(q.ravel(), g["gr4j_T1_qsim"]) < TOL
    # empty series
    assert oracle.simulate_abc(np.zeros(0), 0.,
                               np.array([.1, .2, .3])).shape == (0, 1)
    with pytest.raises(IndexError):
        oracle.simulate_gr4j(g["prec40"], g["etp40"], (0.6, 0.7),
                             np.array([350., 0.5, 90., -1.0]))


def test_threads_do_not_change_results(oracle):
    g = golden("syn_hbvedu")
    args = (g["temp"][:500], g["prec"][:500], g["month"][:500] - 1, g["PE_m"],
            g["T_m"], g["inits"], g["params"])
    a = oracle.simulate_hbvedu(*args, nthreads=1)
    b = oracle.simulate_hbvedu(*args, nthreads=4)
    assert np.array_equal(a, b)


# ------------------------------------------------------------ next tier
# hysteresis snow routine, ice melt and their GR4J couplings (reference:
# cemaneigehyst_model.py, icemelt_model.py, cemaneigehystgr4j_model.py,
# cemaneigegr4jice_model.py, cemaneigehystgr4jice_model.py); KATs:
# reference test/test_models.py:293-310 and :336-356.
def _check_next(out, g, keys, idx=None):
    for k in keys:
        a, b = out[k], g[k] if idx is not None else g["ref_" + k]
        if idx is not None and a.ndim == 3:
            assert rel_err(a[idx], b, floor=1e-6) < TOL, k
            assert rel_err(a[-1], g[k + "_last"], floor=1e-6) < TOL, k
        else:
            assert rel_err(a.reshape(b.shape), b, floor=1e-6) < TOL, k


def test_cemaneigehystgr4j_kat_excel(oracle):
    g = golden("kat_cemaneigehystgr4j")
    lp, lmean, frac = _layers(g)
    out = oracle.simulate_snow_gr4j(True, False, lp, lmean, g["etp"], frac,
                                    g["inits"], g["params"],
                                    return_storages=True)
    assert np.allclose(out["qsim"].ravel(), g["qsim_excel"])
    _check_next(out, g, ["qsim", "G", "eTG", "s_store", "r_store", "sca",
                         "rain"])


def test_cemaneigehystgr4jice_kat_excel(oracle):
    g = golden("kat_cemaneigehystgr4jice")
    lp, lmean, frac = _layers(g)
    out = oracle.simulate_snow_gr4j(True, True, lp, lmean, g["etp"], frac,
                                    g["inits"], g["params"],
                                    frac_ice=g["frac_ice"],
                                    return_storages=True)
    assert np.allclose(out["qsim"].ravel(), g["qsim_excel"])
    _check_next(out, g, ["qsim", "G", "eTG", "s_store", "r_store", "sca",
                         "icemelt", "snowmelt", "rain"])


def test_next_tier_synthetic(oracle):
    h = golden("syn_cemaneigehystgr4j")
    idx = h["stride_idx"]
    forcing = (h["layer_prec"], h["layer_mean"], h["etp"], h["frac_solid"])
    out = oracle.simulate_snow_gr4j(True, False, *forcing, h["inits"],
                                    h["params"], return_storages=True)
    _check_next(out, h, ["qsim", "G", "eTG", "s_store", "r_store", "sca",
                         "rain"], idx)
    g = golden("syn_cemaneigegr4jice")
    out = oracle.simulate_snow_gr4j(False, True, *forcing, g["inits"],
                                    g["params"], frac_ice=g["frac_ice"],
                                    return_storages=True)
    _check_next(out, g, ["qsim", "G", "eTG", "s_store", "r_store", "icemelt"],
                idx)
    g = golden("syn_cemaneigehystgr4jice")
    out = oracle.simulate_snow_gr4j(True, True, *forcing, g["inits"],
                                    g["params"], frac_ice=g["frac_ice"],
                                    return_storages=True)
    _check_next(out, g, ["qsim", "G", "eTG", "s_store", "r_store", "sca",
                         "icemelt", "snowmelt", "rain"], idx)
