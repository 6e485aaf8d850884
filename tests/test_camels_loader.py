"""CAMELS loader: same DataFrame as the
reference's rrmpg.data.CAMELSLoader (rrmpg/data/camelsloader.py:37-128),
pinned by tests/golden/camels_loader.npz (gen_golden_camels.py)."""

import gzip
import os

import numpy as np
import pandas as pd
import pytest

from rrmpg_amd.data import CAMELSLoader
from .conftest import golden


def test_packaged_basin_equals_reference_frame():
    g = golden("camels_loader")
    loader = CAMELSLoader()
    assert loader.get_basin_numbers() == list(g["basins"])
    assert loader.VALID_BASINS == list(g["basins"])
    basin = str(g["basin"])
    df = loader.load_basin(basin)
    assert list(df.columns) == list(g["columns"])
    assert isinstance(df.index, pd.DatetimeIndex)
    dates = df.index.to_numpy().astype("datetime64[D]").astype(np.int64)
    assert np.array_equal(dates, g["dates"])
    assert np.array_equal(df.to_numpy(dtype=np.float64), g["values"],
                          equal_nan=True)
    assert loader.get_station_height(basin) == float(g["station_height"])
    # complete hydrological years only
    assert (df.index[0].month, df.index[0].day) == (10, 1)
    assert (df.index[-1].month, df.index[-1].day) == (9, 30)


def test_invalid_basin_raises_like_the_reference():
    loader = CAMELSLoader()
    with pytest.raises(ValueError) as e:
        loader.load_basin("123")
    assert "Invalid basin number 123" in str(e.value)
    assert "['01031500']" in str(e.value)
    with pytest.raises(ValueError):
        loader.get_station_height("nope")


def _write_basin(path, basin, years=(1990, 1993), gz=False, height=123.5):
    days = pd.date_range(f"{years[0]}-01-01", f"{years[1]}-12-31")
    rng = np.random.default_rng(int(basin))
    n = len(days)
    lines = ["  44.00", f" {height:.2f}", " 1000000",
             "Year Mnth Day Hr dayl(s) prcp(mm/day) srad(W/m2) swe(mm) "
             "tmax(C) tmin(C) vp(Pa)"]
    prcp = np.round(rng.gamma(0.7, 5, n) * (rng.random(n) < 0.4), 2)
    tmax = np.round(10 + 10 * np.sin(np.arange(n) / 58.1), 2)
    for d, p, tx in zip(days, prcp, tmax):
        lines.append(f"{d.year} {d.month:02d} {d.day:02d} 12\t40000.00\t"
                     f"{p:.2f}\t200.00\t0.00\t{tx:.2f}\t{tx - 8:.2f}\t300.00")
    opener = gzip.open if gz else open
    ext = ".gz" if gz else ""
    with opener(os.path.join(path, f"{basin}_lump_cida_forcing_leap.txt{ext}"),
                "wt") as fp:
        fp.write("\n".join(lines) + "\n")
    # the model-output file starts at the first hydrological year
    out = ["YR MNTH DY HR SWE PRCP RAIM TAIR PET ET MOD_RUN OBS_RUN"]
    flow_days = days[days >= pd.Timestamp(f"{years[0]}-10-01")]
    pet = np.round(1 + rng.random(len(flow_days)), 4)
    obs = np.round(rng.random(len(flow_days)) * 3, 4)
    for d, pe, ob in zip(flow_days, pet, obs):
        out.append(f"{d.year} {d.month:02d} {d.day:02d} 12 0.0 0.0 0.0 5.0 "
                   f"{pe:.7f} 0.5 1.0 {ob:.7f}")
    with opener(os.path.join(path, f"{basin}_05_model_output.txt{ext}"),
                "wt") as fp:
        fp.write("\n".join(out) + "\n")
    return prcp, tmax, pet, obs, days, flow_days


def test_custom_directory_plain_and_gzip(tmp_path):
    a = _write_basin(str(tmp_path), "0100", gz=False, height=77.25)
    _write_basin(str(tmp_path), "0200", gz=True)
    # a forcing file without its model-output twin is not a loadable basin
    open(tmp_path / "0300_lump_cida_forcing_leap.txt", "w").close()
    loader = CAMELSLoader(data_dir=tmp_path)
    assert loader.get_basin_numbers() == ["0100", "0200"]
    assert loader.get_station_height("0100") == 77.25
    df = loader.load_basin("0100")
    prcp, tmax, pet, obs, days, flow_days = a
    assert df.index[0] == pd.Timestamp("1990-10-01")
    assert df.index[-1] == pd.Timestamp("1993-09-30")
    sel = (days >= df.index[0]) & (days <= df.index[-1])
    assert np.array_equal(df["prcp(mm/day)"].to_numpy(), prcp[sel])
    fsel = (flow_days >= df.index[0]) & (flow_days <= df.index[-1])
    assert np.array_equal(df["PET"].to_numpy(), pet[fsel])
    assert np.array_equal(df["QObs(mm/d)"].to_numpy(), obs[fsel])
    assert loader.load_basin("0200").shape == df.shape


def test_forcing_arrays_are_model_ready(tmp_path):
    _write_basin(str(tmp_path), "0100")
    f = CAMELSLoader(tmp_path).forcing("0100")
    n = f["prec"].size
    for key in ("prec", "mean_temp", "min_temp", "max_temp", "etp", "qobs"):
        assert f[key].dtype == np.float64 and f[key].flags.c_contiguous
        assert f[key].shape == (n,)
    assert f["month"].dtype == np.int8
    assert f["month"][0] == 10 and f["month"][-1] == 9
    assert np.array_equal(f["mean_temp"], (f["max_temp"] + f["min_temp"]) / 2)
    assert str(f["dates"][0]) == "1990-10-01"
    assert f["met_station_height"] == 123.5
