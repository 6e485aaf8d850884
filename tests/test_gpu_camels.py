"""Real-basin workflow on the GPU: the packaged CAMELS basin (34 hydrological
years, 12,418 days) through GR4J and CemaneigeGR4J, against the CPU oracle on
the same arrays (tolerance 1e-10 relative, snow states bit-exact), and the
Monte-Carlo search the reference's tutorial runs on that basin."""

import numpy as np
import pytest

from .conftest import rel_err, snow_same

pytestmark = pytest.mark.gpu

RTOL = 1e-10


@pytest.fixture(scope="module")
def basin():
    from rrmpg_amd import _lib
    _lib.load()
    _lib.require_gpu()
    from rrmpg_amd.data import CAMELSLoader
    return CAMELSLoader().forcing("01031500")


def _flat(p, cls):
    return np.stack([p[n] for n in cls._param_list], axis=1)


def test_gr4j_on_camels_basin_vs_oracle(basin, oracle):
    from rrmpg_amd.models import GR4J
    np.random.seed(5)
    m = GR4J()
    p = m.get_random_params(96)
    q, s, r = m.simulate(basin["prec"], basin["etp"], 0.5, 0.4,
                         return_storage=True, params=p)
    ref = oracle.simulate_gr4j(basin["prec"], basin["etp"], (0.5, 0.4),
                               _flat(p, GR4J), return_storage=True)
    assert q.shape == (basin["prec"].size, 96)
    for a, b in zip((q, s, r), ref):
        assert rel_err(a, b, floor=1e-9) < RTOL


def test_cemaneigegr4j_on_camels_basin_vs_oracle(basin, oracle, fused_variant):
    from rrmpg_amd.models import CemaneigeGR4J
    from rrmpg_amd.models import cemaneige_utils as cu
    alts = [310., 420., 510., 640., 900.]
    h = basin["met_station_height"]
    np.random.seed(6)
    m = CemaneigeGR4J()
    p = m.get_random_params(80)
    out = m.simulate(basin["prec"], basin["mean_temp"], basin["min_temp"],
                     basin["max_temp"], basin["etp"], h, 0., 0., 0.5, 0.4,
                     altitudes=alts, return_storages=True, params=p)
    lp = cu.extrapolate_precipitation(basin["prec"], alts, h)
    lmin, lmean, lmax = cu.extrapolate_temperature(
        basin["min_temp"], basin["mean_temp"], basin["max_temp"], alts, h)
    fr = cu.calculate_solid_fraction(lp, np.array(alts), lmean, lmin, lmax)
    ref = oracle.simulate_cemaneigegr4j(lp, lmean, basin["etp"], fr,
                                        (0., 0., 0.5, 0.4),
                                        _flat(p, CemaneigeGR4J),
                                        return_storages=True)
    snow_same(out[1], ref[1])
    snow_same(out[2], ref[2], exact=True)
    for a, b in zip(out, ref):
        assert rel_err(a, b, floor=1e-9) < RTOL


def test_monte_carlo_on_camels_basin(basin, oracle):
    from rrmpg_amd.models import GR4J
    from rrmpg_amd.tools import monte_carlo
    from rrmpg_amd.utils.metrics import calc_mse, calc_nse
    np.random.seed(7)
    res = monte_carlo(GR4J(), num=2000, qobs=basin["qobs"],
                      prec=basin["prec"], etp=basin["etp"], s_init=0.5,
                      r_init=0.4)
    assert res["qsim"].shape == (basin["prec"].size, 2000)
    best = int(np.argmin(res["mse"]))
    assert rel_err(res["mse"][best],
                   calc_mse(basin["qobs"], res["qsim"][:, best])) < 1e-12
    # a random search over 2000 sets beats the mean-flow benchmark clearly
    # (GR4J has no snow routine and this is a snowy basin: NSE ~ 0.4)
    assert calc_nse(basin["qobs"], res["qsim"][:, best]) > 0.25
    ref = oracle.simulate_gr4j(
        basin["prec"], basin["etp"], (0.5, 0.4),
        _flat(res["params"][best:best + 1], GR4J))
    assert rel_err(res["qsim"][:, best:best + 1], ref, floor=1e-9) < RTOL
