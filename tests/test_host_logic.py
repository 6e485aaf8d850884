"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol
include/rrhip.h declares, the host mirror of the reference's interface
(validation, errors, parameter handling, sampling, preprocessing, metrics)
behaves like the reference, and nothing falls back to a CPU compute path.

The expectations follow the reference's own unit tests
(reference: test/test_models.py:20-140, test/test_utils.py:19-97,
test/test_tools.py:26-29).
"""

import os
import re

import numpy as np
import pandas as pd
import pytest

from .conftest import REPO, golden

from rrmpg_amd import _lib
from rrmpg_amd.models import (ABCModel, HBVEdu, GR4J, Cemaneige,
                              CemaneigeGR4J, CemaneigeHystGR4J,
                              CemaneigeGR4JIce, CemaneigeHystGR4JIce)
from rrmpg_amd.models import cemaneige_utils as cu
from rrmpg_amd.models.basemodel import BaseModel
from rrmpg_amd.tools import monte_carlo
from rrmpg_amd.utils.array_checks import (check_for_negatives,
                                          validate_array_input)
from rrmpg_amd.utils.metrics import (calc_alpha_nse, calc_beta_nse, calc_kge,
                                     calc_mse, calc_nse, calc_r, calc_rmse,
                                     mse_from_sse, nse_from_sse)

NO_GPU = _lib.device_count() == 0


# ------------------------------------------------------------- the C-ABI
def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "rrhip.h")).read()
    declared = set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.exported_names())
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rr_version() >= 100
    assert isinstance(lib.rr_device_count(), int)


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    assert lib.rr_hbvedu_workspace_bytes(10957, 1000) >= 10957 * 32
    assert lib.rr_gr4j_workspace_bytes(10957, 1000) >= 10957 * 16
    assert lib.rr_cemaneige_workspace_bytes(10957, 5, 10) >= 10957 * 15 * 8
    assert lib.rr_cemaneigegr4j_workspace_bytes(10957, 5, 10) >= 10957 * 16 * 8
    assert lib.rr_abc_workspace_bytes(10, 10) > 0


def test_argument_errors_are_reported_without_gpu():
    lib = _lib.load()
    # negative size / qobs without sse are caught before any HIP call
    rc = lib.rr_abc_simulate(None, -1, 0.0, None, 0, None, None, None, None)
    assert rc == -2 and b"negative size" in lib.rr_last_error()
    x = np.zeros(4)
    p = x.ctypes.data_as(_lib._f64p)
    rc = lib.rr_abc_simulate(p, 4, 0.0, p, 1, p, None, p, None)
    assert rc == -1 and b"qobs and sse" in lib.rr_last_error()
    # the kernels count days in 32 bits: more than 2e9 timesteps is a size error
    rc = lib.rr_abc_simulate(p, 2_000_000_001, 0.0, p, 1, p, None, None, None)
    assert rc == -2 and b"exceeds 2e9" in lib.rr_last_error()
    # storages come with the discharge, as the reference's return_storage
    # returns them (the kernels are built for those combinations only)
    rc = lib.rr_abc_simulate(p, 4, 0.0, p, 1, None, p, None, None)
    assert rc == -1 and b"come with qsim" in lib.rr_last_error()
    # empty problems succeed trivially
    assert lib.rr_abc_simulate(p, 0, 0.0, p, 1, p, None, None, None) == 0


@pytest.mark.skipif(not NO_GPU, reason="needs a box without a GPU")
def test_no_cpu_fallback():
    """Without a GPU every simulation must fail loudly."""
    with pytest.raises(RuntimeError, match="no CPU path"):
        ABCModel().simulate(np.ones(10))
    with pytest.raises(RuntimeError, match="no CPU path"):
        monte_carlo(ABCModel(), 4, prec=np.ones(10))
    x = np.ones(4)
    p = x.ctypes.data_as(_lib._f64p)
    lib = _lib.load()
    rc = lib.rr_abc_simulate(p, 4, 0.0, p, 1, p, None, None, None)
    assert rc == -5 and b"no CPU path" in lib.rr_last_error()


def test_product_never_touches_the_oracle():
    """rrmpg_amd/ must not import, load or mention the oracle."""
    for root, _, files in os.walk(os.path.join(REPO, "rrmpg_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")) or fn == "Makefile":
                text = open(os.path.join(root, fn), errors="replace").read()
                assert "oracle" not in text.lower(), os.path.join(root, fn)


# ------------------------------------------------ BaseModel (test_models.py)
class TestBaseModelFunctions:
    param_names = ['a', 'b', 'c']
    default_bounds = {'a': (0, 1), 'b': (0, 1), 'c': (0, 1)}
    dtype = np.dtype([('a', np.float64), ('b', np.float64),
                      ('c', np.float64)])

    def test_accessors(self):
        model = ABCModel()
        assert issubclass(model.__class__, BaseModel)
        assert model.get_parameter_names() == self.param_names
        assert model.get_default_bounds() == self.default_bounds
        assert model.get_dtype() == self.dtype
        for p in self.param_names:
            assert p in model.get_params()

    def test_random_params_in_default_bounds(self):
        model = ABCModel()
        params = model.get_random_params()
        for p in self.param_names:
            lo, hi = self.default_bounds[p]
            assert lo <= params[p][0] <= hi
        many = model.get_random_params(num=24)
        assert many.size == 24
        assert (many['a'] + many['b'] <= 1).all()

    def test_set_params(self):
        model = ABCModel()
        rand = model.get_random_params()
        params = {p: rand[p][0] for p in self.param_names}
        model.set_params(params)
        assert params == model.get_params()
        model.set_params(rand)            # ndarray of the model dtype
        model.set_params(rand[0])         # single record
        assert model.get_params()['a'] == rand['a'][0]
        with pytest.raises(AttributeError, match="Unknow parameter"):
            model.set_params({'zz': 1.0})
        with pytest.raises(ValueError, match="must be numerical"):
            model.set_params({'a': 'x'})
        with pytest.raises(TypeError, match="wrong data type"):
            model.set_params(np.zeros(1, dtype=GR4J._dtype))
        with pytest.raises(TypeError, match="Wrong input data type"):
            model.set_params([1, 2, 3])
        with pytest.raises(AttributeError, match="Missing the following"):
            ABCModel(params={'a': 0.1})

    def test_dtypes_are_packed_f64_records(self):
        for cls, k in [(ABCModel, 3), (HBVEdu, 11), (GR4J, 4), (Cemaneige, 2),
                       (CemaneigeGR4J, 6)]:
            assert cls._dtype.itemsize == 8 * k
            assert list(cls._dtype.names) == cls._param_list
            p = cls().get_random_params(5)
            flat, _, n = _lib.params_block(p, k)
            assert n == 5 and flat.shape == (5, k)
            for j, name in enumerate(cls._param_list):
                assert np.array_equal(flat[:, j], p[name])


def test_sampling_reproduces_the_reference_stream():
    g = golden("sampling")
    for name, cls in [("abc", ABCModel), ("hbvedu", HBVEdu), ("gr4j", GR4J),
                      ("cemaneige", Cemaneige),
                      ("cemaneigegr4j", CemaneigeGR4J)]:
        np.random.seed(1234)
        mdl = cls()
        pp = mdl.get_random_params(7)
        ctor = np.array([mdl.get_params()[k]
                         for k in mdl.get_parameter_names()])
        assert np.array_equal(ctor, g[name + "_ctor"]), name
        flat = np.stack([pp[k] for k in mdl.get_parameter_names()], 1)
        assert np.array_equal(flat, g[name + "_rand7"]), name


def test_next_tier_classes_sampling_and_metrics():
    g = golden("sampling_next")
    for name, cls, k in [("cemaneigehystgr4j", CemaneigeHystGR4J, 8),
                         ("cemaneigegr4jice", CemaneigeGR4JIce, 7),
                         ("cemaneigehystgr4jice", CemaneigeHystGR4JIce, 9)]:
        assert issubclass(cls, BaseModel) and cls._dtype.itemsize == 8 * k
        np.random.seed(1234)
        mdl = cls()
        pp = mdl.get_random_params(7)
        ctor = np.array([mdl.get_params()[n] for n in cls._param_list])
        assert np.array_equal(ctor, g[name + "_ctor"]), name
        flat = np.stack([pp[n] for n in cls._param_list], 1)
        assert np.array_equal(flat, g[name + "_rand7"]), name
    # KGE and the NSE decompositions (scores of a perfect / shifted series)
    obs = np.array([1., 2., 4., 3., 5.])
    assert calc_kge(obs, obs) == pytest.approx(1.0)
    assert calc_alpha_nse(obs, obs) == pytest.approx(1.0)
    assert calc_beta_nse(obs, obs + 1) == pytest.approx(1 / np.std(obs))
    assert calc_r(obs, 2 * obs)[0] == pytest.approx(1.0)
    r = np.corrcoef(obs, obs ** 2)[0, 1]
    want = 1 - np.sqrt((r - 1) ** 2 + (np.std(obs ** 2) / np.std(obs) - 1) ** 2
                       + (np.mean(obs ** 2) / np.mean(obs) - 1) ** 2)
    assert calc_kge(obs, obs ** 2) == pytest.approx(want)
    with pytest.raises(RuntimeError, match="mean of the observations"):
        calc_kge([-1., 1.], [1., 2.])
    with pytest.raises(RuntimeError, match="standard deviation"):
        calc_kge([2., 2.], [1., 2.])


# ---------------------------------------------- input validation & errors
def test_negative_rain_messages():
    with pytest.raises(ValueError,
                       match="In the precipitation array are negative values."):
        ABCModel().simulate([-1, 1, 1])
    with pytest.raises(ValueError,
                       match="In the precipitation array are negative values."):
        HBVEdu().simulate(temp=np.random.uniform(-15, 25, 100),
                          prec=np.arange(-1, 99),
                          month=np.random.randint(1, 12, 100),
                          PE_m=np.random.uniform(0, 4, 12),
                          T_m=np.random.uniform(-5, 15, 12))
    with pytest.raises(ValueError, match="contains negative values"):
        GR4J().simulate([-1., 2.], [1., 1.])
    with pytest.raises(ValueError, match="contains negative values"):
        Cemaneige().simulate([-1., 2.], [1., 1.], [0., 0.], [2., 2.], 500)


def test_hbvedu_validation():
    ok = dict(temp=np.zeros(10), prec=np.ones(10), month=np.ones(10),
              PE_m=np.ones(12), T_m=np.ones(12))
    m = HBVEdu()
    with pytest.raises(RuntimeError, match="must be of equal size"):
        m.simulate(**{**ok, "prec": np.ones(9)})
    with pytest.raises(RuntimeError, match="must be of length 12"):
        m.simulate(**{**ok, "PE_m": np.ones(11)})
    with pytest.raises(ValueError, match="month array must be between"):
        m.simulate(**{**ok, "month": np.zeros(10)})
    with pytest.raises(ValueError, match="month array must be between"):
        m.simulate(**{**ok, "month": np.full(10, 13)})
    with pytest.raises(TypeError, match="models own custom data type"):
        m.simulate(**ok, params=np.zeros(2, dtype=GR4J._dtype))
    with pytest.raises(TypeError, match="must be either a list"):
        m.simulate(**{**ok, "temp": (1, 2, 3)})
    with pytest.raises(ValueError, match="purely numerical"):
        m.simulate(**{**ok, "temp": ['a'] * 10})
    # the caller's month array is never modified (quirk Q7)
    month = np.full(10, 3)
    try:
        m.simulate(**{**ok, "month": month})
    except RuntimeError:
        pass
    assert (month == 3).all()


def test_gr4j_and_abc_validation():
    m = GR4J()
    with pytest.raises(RuntimeError, match="must be of the same size"):
        m.simulate([1., 2.], [1.])
    with pytest.raises(TypeError, match="'s1_init' must be a Number"):
        m.simulate([1., 2.], [1., 1.], s_init="a")
    with pytest.raises(ValueError, match="production storage must be in"):
        m.simulate([1., 2.], [1., 1.], s_init=1.5)
    with pytest.raises(ValueError, match="routing storage must be in"):
        m.simulate([1., 2.], [1., 1.], r_init=-0.1)
    a = ABCModel()
    with pytest.raises(TypeError, match="initial_state"):
        a.simulate([1., 2.], initial_state=-1)
    with pytest.raises(TypeError, match="return_storage arg must be a boolean"):
        a.simulate([1., 2.], return_storage=1)


def test_cemaneige_validation():
    m = CemaneigeGR4J()
    s = [1., 2., 3.]
    with pytest.raises(RuntimeError, match="must have the same length"):
        m.simulate(s, s, s, s, [1., 2.], 500)
    with pytest.raises(TypeError, match="'altitudes' must be a list"):
        m.simulate(s, s, s, s, s, 500, altitudes=(1, 2))
    with pytest.raises(TypeError, match="must be numbers"):
        m.simulate(s, s, s, s, s, 500, altitudes=['a'])
    with pytest.raises(TypeError, match="'met_station_height' must be a"):
        m.simulate(s, s, s, s, s, None)
    with pytest.raises(TypeError, match="'snow_pack_init' must be a Number"):
        m.simulate(s, s, s, s, s, 500, snow_pack_init="x")
    with pytest.raises(TypeError, match="'r_init' must be a Number"):
        m.simulate(s, s, s, s, s, 500, r_init="x")
    with pytest.raises(TypeError, match="'thermal_state_init'"):
        Cemaneige().simulate(s, s, s, s, 500, thermal_state_init=[1])


def test_monte_carlo_argument_checks():
    with pytest.raises(TypeError, match="must be one of the models"):
        monte_carlo(object(), 3, prec=[1.])
    with pytest.raises(TypeError, match="positive integer"):
        monte_carlo(ABCModel(), 0, prec=[1.])
    with pytest.raises(TypeError, match="positive integer"):
        monte_carlo(ABCModel(), 2.5, prec=[1.])
    with pytest.raises(ValueError, match="needs qobs"):
        monte_carlo(ABCModel(), 3, return_qsim=False, prec=[1.])


# ----------------------------------------------- preprocessing & metrics
def test_cemaneige_preprocessing_matches_reference():
    p = golden("syn_cemaneige_prep")
    lp = cu.extrapolate_precipitation(p["prec"], p["altitudes"], p["station"])
    lmin, lmean, lmax = cu.extrapolate_temperature(
        p["tmin"], p["temp"], p["tmax"], p["altitudes"], p["station"])
    assert np.array_equal(lp, p["layer_prec"])
    assert np.array_equal(lmin, p["layer_min"])
    assert np.array_equal(lmean, p["layer_mean"])
    assert np.array_equal(lmax, p["layer_max"])
    assert np.array_equal(
        cu.calculate_solid_fraction(lp, p["altitudes"], lmean, lmin, lmax),
        p["frac_solid"])
    # >= 1500 m and > 4000 m branches
    lp = cu.extrapolate_precipitation(p["prec"], p["altitudes_hi"],
                                      p["station_hi"])
    lmin, lmean, lmax = cu.extrapolate_temperature(
        p["tmin"], p["temp"], p["tmax"], p["altitudes_hi"], p["station_hi"])
    assert np.array_equal(lp, p["layer_prec_hi"])
    assert np.array_equal(lmean, p["layer_mean_hi"])
    assert np.array_equal(
        cu.calculate_solid_fraction(lp, p["altitudes_hi"], lmean, lmin, lmax),
        p["frac_solid_hi"])
    assert np.array_equal(
        cu.extrapolate_precipitation(p["prec"][:64], p["altitudes_hi"],
                                     p["station_vhi"]), p["layer_prec_vhi"])


def test_metrics_known_answers():
    assert calc_nse(obs=[1, 2, 3], sim=[1, 2, 3]) == 1
    assert calc_nse(obs=[1, 2, 3], sim=[2, 2, 2]) == 0
    with pytest.raises(RuntimeError, match="Maybe you should use the "
                                           "Mean-Squared-Error instead."):
        calc_nse(obs=[2, 2, 2], sim=[1, 2, 3])
    assert calc_rmse(obs=[1, 2, 3], sim=[1, 2, 3]) == 0
    assert calc_rmse(obs=[1, 1, 1], sim=[3, 3, 3]) == 2
    assert calc_mse(obs=[1, 2, 3], sim=[1, 2, 3]) == 0
    assert calc_mse(obs=[1, 1, 1], sim=[3, 3, 3]) == 4
    with pytest.raises(ValueError, match="same size"):
        calc_mse([1, 2], [1, 2, 3])
    obs = np.array([1., 2., 4.])
    sim = np.array([[1., 2.], [2., 2.], [3., 5.]])
    sse = ((obs[:, None] - sim) ** 2).sum(0)
    assert np.allclose(mse_from_sse(sse, 3),
                       [calc_mse(obs, sim[:, 0]), calc_mse(obs, sim[:, 1])])
    assert np.allclose(nse_from_sse(sse, obs),
                       [calc_nse(obs, sim[:, 0]), calc_nse(obs, sim[:, 1])])


def test_scores_from_sums_asks_only_for_what_is_defined():
    """scores_from_sums: every score from the four column sums (about any
    shift); a score raises for degenerate observations only when it is asked
    for, as the calc_* functions do."""
    from rrmpg_amd.utils import metrics as M
    rng = np.random.default_rng(3)
    obs = rng.uniform(0.5, 3, 200)
    sim = obs[:, None] * rng.uniform(0.8, 1.2, (200, 5)) + rng.normal(
        0, 0.1, (200, 5))
    for shift in (0.0, float(obs.mean())):
        qc, oc = sim - shift, obs - shift
        sums = np.stack([qc.sum(0), (qc * qc).sum(0), (qc * oc[:, None]).sum(0),
                         ((obs[:, None] - sim) ** 2).sum(0)], 1)
        sc = M.scores_from_sums(sums, obs, shift=shift)
        assert set(sc) == set(M.ALL_SCORES)
        for j in range(5):
            for name, fn in [("mse", M.calc_mse), ("rmse", M.calc_rmse),
                             ("nse", M.calc_nse), ("kge", M.calc_kge),
                             ("alpha", M.calc_alpha_nse),
                             ("beta", M.calc_beta_nse)]:
                assert abs(sc[name][j] - fn(obs, sim[:, j])) < 1e-11, name
            assert abs(sc["r"][j] - M.calc_r(obs, sim[:, j])[0]) < 1e-11
    zero = np.zeros(200)
    sums0 = np.stack([sim.sum(0), (sim * sim).sum(0), 0 * sim.sum(0),
                      (sim ** 2).sum(0)], 1)
    only = M.scores_from_sums(sums0, zero, only=("mse", "rmse"))
    assert set(only) == {"mse", "rmse"}
    assert np.allclose(only["mse"], [M.calc_mse(zero, sim[:, j])
                                     for j in range(5)])
    for key in ("kge", "nse", "alpha", "beta"):
        with pytest.raises(RuntimeError):
            M.scores_from_sums(sums0, zero, only=(key,))
    with pytest.raises(RuntimeError):
        M.scores_from_sums(sums0, zero)
    with pytest.raises(ValueError, match="unknown score"):
        M.scores_from_sums(sums0, obs, only=("nash",))


def test_array_checks():
    assert not check_for_negatives(np.array([1, 2, 3, 4, 5], dtype=np.float64))
    assert check_for_negatives(np.array([1, 2, -3, 4, 5], dtype=np.float64))
    vals = [1, 2, 3, 4]
    arr = validate_array_input(pd.Series(data=vals, dtype=np.float64),
                               np.float64, 'arr')
    assert arr.tolist() == np.array(vals, np.float64).tolist()
    assert validate_array_input([1., 2.], np.float64, 'arr').tolist() == [1., 2.]
    with pytest.raises(ValueError, match="The data in the parameter array "
                                         "'arr' must be purely numerical."):
        validate_array_input(['a', 'b', 1], np.float64, 'arr')
    with pytest.raises(TypeError, match="The array arr must be either a list, "
                                        "numpy.ndarray or pandas.Series"):
        validate_array_input((1, 2, 3), np.float64, 'arr')
    # always a flattened copy
    src = np.ones((2, 3))
    out = validate_array_input(src, np.float64, 'arr')
    assert out.shape == (6,) and out is not src


def test_debug_options_and_cache_release_need_no_gpu():
    """The measurement / test hooks are plain process-wide ints behind
    rr_debug_set_option (nothing in the library reads the environment), with
    range checks; releasing cached memory is a no-op without a device."""
    lib = _lib.load()
    opt = _lib.OPTIONS
    assert lib.rr_debug_get_option(opt["hbv_variant"]) == -1
    assert lib.rr_debug_get_option(opt["gr4j_force_lds"]) == 0
    assert lib.rr_debug_get_option(opt["max_block_cols"]) == 0
    with _lib.debug_option("hbv_variant", 3):
        assert lib.rr_debug_get_option(opt["hbv_variant"]) == 3
        with _lib.debug_option("max_block_cols", 512):
            assert lib.rr_debug_get_option(opt["max_block_cols"]) == 512
        assert lib.rr_debug_get_option(opt["max_block_cols"]) == 0
    assert lib.rr_debug_get_option(opt["hbv_variant"]) == -1
    assert lib.rr_debug_set_option(opt["hbv_variant"], 7) == -4    # RR_E_PARAM
    assert lib.rr_debug_get_option(opt["fused_variant"]) == 0
    assert lib.rr_debug_set_option(opt["fused_variant"], 6) == -4
    assert lib.rr_debug_get_option(opt["time_tiles"]) == -1
    assert lib.rr_debug_set_option(opt["time_tiles"], 1) == -4
    assert lib.rr_debug_get_option(opt["gr4j_variant"]) == 0
    assert lib.rr_debug_set_option(opt["gr4j_variant"], 2) == -4
    # the variants removed in round 6 are refused, not silently remapped
    assert lib.rr_debug_set_option(opt["hbv_variant"], 1) == -4
    assert lib.rr_debug_set_option(opt["hbv_variant"], 2) == -4
    assert lib.rr_debug_set_option(opt["fused_variant"], 4) == -4
    assert b"does not take" in lib.rr_last_error()
    assert lib.rr_debug_get_option(opt["warm_records"]) == -1
    assert lib.rr_debug_set_option(opt["warm_records"], 2) == -4
    assert lib.rr_debug_set_option(99, 0) == -4
    assert lib.rr_debug_get_option(99) == -2 ** 63
    assert lib.rr_release_cached_memory() == 0
    # per-call / per-thread options (rr_call_options): range-checked like the
    # process-wide hooks, and independent of them
    import ctypes
    o = _lib.CallOptions()
    lib.rr_call_options_init(ctypes.byref(o))
    assert o.struct_bytes == ctypes.sizeof(_lib.CallOptions)
    assert all(v == _lib.RR_OPT_UNSET for v in o.value)
    assert lib.rr_call_options_set(ctypes.byref(o), opt["host_shards"], 3) == 0
    assert o.value[opt["host_shards"]] == 3
    assert lib.rr_call_options_set(ctypes.byref(o), opt["host_shards"],
                                   -7) == -4
    assert lib.rr_call_options_set(ctypes.byref(o), 99, 0) == -4
    assert lib.rr_call_options_set(ctypes.byref(o), opt["host_shards"],
                                   _lib.RR_OPT_UNSET) == 0
    assert lib.rr_thread_options(ctypes.byref(o)) == 0
    assert lib.rr_thread_options(None) == 0
    raw = _lib.CallOptions()                       # never initialised
    assert lib.rr_call_options_set(ctypes.byref(raw), opt["host_shards"],
                                   2) == -4
    assert lib.rr_thread_options(ctypes.byref(raw)) == -4
    assert b"struct_bytes" in lib.rr_last_error()
    with _lib.call_options(host_shards=2):
        with _lib.call_options(max_block_cols=64):
            ptr = _lib.opts_ptr()
            assert ptr is not None
            cur = ptr._obj
            assert cur.value[opt["host_shards"]] == 2
            assert cur.value[opt["max_block_cols"]] == 64
        assert _lib.opts_ptr()._obj.value[opt["max_block_cols"]] \
            == _lib.RR_OPT_UNSET
    assert _lib.opts_ptr() is None
    # None in a nested block means "not set here": the outer value stays
    with _lib.call_options(host_shards=3):
        with _lib.call_options(host_shards=None, max_block_cols=8):
            cur = _lib.opts_ptr()._obj
            assert cur.value[opt["host_shards"]] == 3
            assert cur.value[opt["max_block_cols"]] == 8
    # standing options of the thread nest the same way, and the enclosing
    # block's options stand again behind an inner one
    with _lib.thread_options(hbv_variant=0):
        assert _lib._tls.standing == {"hbv_variant": 0}
        with _lib.thread_options(time_tiles=4, hbv_variant=None):
            assert _lib._tls.standing == {"hbv_variant": 0, "time_tiles": 4}
        assert _lib._tls.standing == {"hbv_variant": 0}
    assert _lib._tls.standing is None
    with pytest.raises(KeyError):
        _lib.thread_options(no_such_option=1)
    assert lib.rr_debug_get_option(opt["host_shards"]) == 0
    with pytest.raises(KeyError):
        _lib.call_options(no_such_option=1)
    for bad in (0, -2, 1.5, True, "most"):
        with pytest.raises(ValueError):
            _lib.host_shards_of(bad)
    assert _lib.host_shards_of(None) is None
    assert _lib.host_shards_of("all") == -1 and _lib.host_shards_of(4) == 4
    # no getenv anywhere in the library's sources
    csrc = os.path.join(REPO, "rrmpg_amd", "csrc")
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, name)).read(), name


def test_layer_preprocessing_argument_errors_without_gpu():
    lib = _lib.load()
    alt = np.array([600., 900.])
    pa = alt.ctypes.data_as(_lib._f64p)
    assert lib.rr_cemaneige_layers_workspace_bytes(5) >= 5 * 24
    rc = lib.rr_cemaneige_layers_dev(None, None, None, None, 10, pa, 2, 500.,
                                     None, None, None, None, None, 0, None)
    assert rc == -1 and b"NULL pointer" in lib.rr_last_error()
    rc = lib.rr_cemaneige_layers_dev(None, None, None, None, 10, pa, 0, 500.,
                                     None, None, None, None, None, 0, None)
    assert rc == -2
    assert lib.rr_cemaneige_layers_dev(None, None, None, None, 0, pa, 2, 500.,
                                       None, None, None, None, None, 0,
                                       None) == 0
    assert lib.rr_gr4j_plan_status(None, None) == -1


def test_fuzz_lost_days_rule():
    """tests/test_gpu_fuzz.py _lost_days, the one place where the GPU tests
    stop following the oracle: a WILD set whose own perturbed oracle runs
    drift 1e-3 apart (chaotic dynamics) is compared up to that day -- an
    in-bounds set never, and there is no rule for overflow, infinities or
    run-away stores (sets that are not civil run the reference's own
    sequence on the GPU)."""
    from . import test_gpu_fuzz as F
    assert not hasattr(F, "_overflow_horizon") and not hasattr(F, "RUNAWAY")
    t, n = 12, 4
    b = np.ones((t, n))
    p = b.copy()
    p[5:, 1] *= 1.01                 # wild set 1 drifts from day 5
    p[3:, 2] *= 1.5                  # in-bounds set 2 drifts: never excused
    p[8:, 3] = np.inf                # wild set 3: a probe overflows on day 8
    lost = F._lost_days(b, [p])
    assert lost.tolist() == [t, 5, t, 8]


def test_bench_socket_sampler_without_hwmon():
    """bench.py's power record is optional: no hwmon files (this container),
    no record -- and nothing raises."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = bench.SocketSampler(0)
    with s:
        pass
    assert s.record() is None or s.record()["samples"] >= 1


def test_resident_outputs_start_rows_on_128_byte_boundaries():
    """device.new_output hands out [T, N] views of a buffer whose row pitch
    is rounded up to 16 doubles (a wave's 512-byte store must not straddle
    partly written memory lines: profiles/r04_row_pitch.txt), and the shape
    checks of ``run`` accept such views.  No GPU needed: the logic is
    exercised on CPU tensors."""
    import types
    import torch
    from rrmpg_amd import device as dev
    ens = types.SimpleNamespace(num_timesteps=7, device=torch.device("cpu"),
                                ROW_PITCH=dev._Ensemble.ROW_PITCH)
    for n, pitch in ((1, 16), (15, 16), (16, 16), (99, 112), (1000, 1008),
                     (1_000_001, 1_000_016)):
        q = dev._Ensemble.new_output(ens, n)
        assert q.shape == (7, n) and q.stride() == (pitch, 1)
        assert q.dtype == torch.float64
    g = dev._Ensemble.new_output(ens, 99, 5)
    assert g.shape == (7, 5, 99) and g.stride() == (5 * 112, 112, 1)
    # one call's outputs share the pitch; a dense array beside a padded one
    # is refused
    q, s = (dev._Ensemble.new_output(ens, 99) for _ in range(2))
    ens._check_tensor = types.MethodType(dev._Ensemble._check_tensor, ens)
    assert dev._Ensemble._check_outputs(ens, 99, (q, s), (g,), 5) == 112
    with pytest.raises(ValueError, match="row stride"):
        dev._Ensemble._check_outputs(
            ens, 99, (q, torch.empty((7, 99), dtype=torch.float64)))


def test_counter_results_are_tied_to_the_kernel_sources(monkeypatch):
    """profiles/traffic.json carries the ids of the kernel sources its
    counters were collected on (rrmpg_amd/utils/buildid.py); bench.py quotes a
    model's HBM traffic / instruction count only while the tree's sources are
    those, and says "stale" otherwise."""
    import argparse
    import json
    import bench
    from rrmpg_amd.utils import buildid
    ids = {m: buildid.kernel_source_id(m) for m in buildid._KERNEL_FILE}
    assert all(len(v) == 16 and int(v, 16) >= 0 for v in ids.values())
    assert ids["hbvedu"] != ids["gr4j"] != ids["abc"]
    assert ids["cemaneige"] == ids["cemaneigegr4j"]      # one kernel file
    with open(os.path.join(REPO, "profiles", "traffic.json")) as fh:
        pmc = json.load(fh)
    assert set(pmc["_build"]) == set(ids)
    args = argparse.Namespace(model="hbvedu", mode="qsim", sets=1_000_000,
                              catchments=0)
    key = "hbvedu:qsim:1000000:10957"
    assert key in pmc and key + ":valu_instr_per_unit" in pmc
    # sources as stamped: the numbers are quoted ...
    monkeypatch.setattr(buildid, "kernel_source_id",
                        lambda m: pmc["_build"][m])
    traffic, valu = bench.traffic_record(args, 1_000_000, 10957)
    assert traffic == pmc[key] and valu == pmc[key + ":valu_instr_per_unit"]
    assert bench.traffic_is_stale("hbvedu") is False
    # ... sources changed since: withheld
    monkeypatch.setattr(buildid, "kernel_source_id", lambda m: "0" * 16)
    assert bench.traffic_record(args, 1_000_000, 10957) == (None, None)
    assert bench.traffic_is_stale("hbvedu") is True


def test_shard_bounds_and_the_collectives_argument_checks():
    """rr_shard_bounds (the C-ABI's partition of the parameter-set axis) is
    rrmpg_amd.sharding.shard_bounds; the RCCL entry points reject bad
    arguments before they touch the communication library (no GPU here)."""
    import ctypes
    from rrmpg_amd.sharding import shard_bounds
    lib = _lib.load()
    a, b = ctypes.c_int64(), ctypes.c_int64()
    for n in (0, 1, 7, 64, 1000, 1_000_003):
        for world in (1, 2, 3, 8, 64):
            prev = 0
            for r in range(world):
                assert lib.rr_shard_bounds(n, world, r, ctypes.byref(a),
                                           ctypes.byref(b)) == 0
                assert (a.value, b.value) == shard_bounds(n, world, r)
                assert a.value == prev
                prev = b.value
            assert prev == n
    assert lib.rr_shard_bounds(10, 3, 3, ctypes.byref(a),
                               ctypes.byref(b)) == -2          # RR_E_SIZE
    assert lib.rr_shard_bounds(10, 0, 0, ctypes.byref(a), ctypes.byref(b)) == -2
    assert lib.rr_shard_bounds(10, 2, 0, None, ctypes.byref(b)) == -1
    comm = ctypes.c_void_p()
    ident = (ctypes.c_char * 128)()
    assert lib.rr_comm_init(ctypes.byref(comm), 2, 2, ident) == -2
    assert lib.rr_comm_init(ctypes.byref(comm), 0, 0, ident) == -2
    assert lib.rr_comm_init(None, 1, 0, ident) == -1
    assert lib.rr_comm_init(ctypes.byref(comm), 1, 0, None) == -1
    assert lib.rr_comm_unique_id(None) == -1
    assert lib.rr_allgather_metric(None, None, 0, None, 0, None) == -1
    assert b"NULL" in lib.rr_last_error()
    assert lib.rr_comm_destroy(None) == 0


def test_bench_line_fits_the_drivers_tail():
    """bench.compact_line on the round's full record (and on its worst case:
    eight ranks' shards, every float at full precision, every extra
    configuration with every field): at most bench.LINE_LIMIT characters, the
    contract's fields and every BASELINE configuration's kernel_ms / frac /
    valu.frac / parity_spot present."""
    import json
    import bench
    with open(os.path.join(REPO, "profiles", "r06_bench_detail.json")) as fh:
        full = json.load(fh)
    line = bench.compact_line(full)
    assert len(json.dumps(line, separators=(",", ":"))) <= bench.LINE_LIMIT
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    ids = [e["id"] for e in line["extra_configs"]]
    assert ids[:4] == ["cfg1", "cfg2", "cfg3", "cfg4"]
    for e in line["extra_configs"][:4]:
        assert e["kernel_ms"] > 0 and "frac" in e and "parity_spot" in e
        assert e["valu"]["frac"] > 0
    worst = json.loads(json.dumps(full))
    worst["n_gpus"] = 8
    worst["config"]["shards"] = [[k * 125000 + 1, (k + 1) * 125000 + 1]
                                 for k in range(8)]
    worst["value"] = 6.123456789012345e11
    for e in worst["extra_configs"]:
        e["kernel_ms"] = 123.45678901234567
        e["parity_spot"] = 2.123456789012345e-13
    assert len(json.dumps(bench.compact_line(worst),
                          separators=(",", ":"))) <= bench.LINE_LIMIT


def test_monte_carlo_rejects_bad_exchange_before_touching_a_gpu():
    """`exchange` (how the shards' scores of a sampler='device' sweep reach
    the host) is validated with the other arguments, in front of any GPU
    work."""
    from rrmpg_amd.models import ABCModel
    from rrmpg_amd.tools import monte_carlo
    qobs = np.ones(10)
    prec = np.ones(10)
    with pytest.raises(ValueError, match="exchange"):
        monte_carlo(ABCModel(), 5, qobs=qobs, return_qsim=False,
                    sampler="device", exchange="mpi", prec=prec)
    with pytest.raises(ValueError, match="sampler='device'"):
        monte_carlo(ABCModel(), 5, qobs=qobs, exchange="rccl", prec=prec)
    with pytest.raises(ValueError, match="qobs"):
        monte_carlo(ABCModel(), 5, sampler="device", return_qsim=False,
                    prec=prec)
