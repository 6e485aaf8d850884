"""CAMELS basin files -> daily forcing and observed discharge.

Mirrors the reference's ``rrmpg.data.CAMELSLoader``
(reference: rrmpg/data/camelsloader.py:14-128): ``load_basin`` returns the
same pandas DataFrame (same columns, index, hydrological-year window),
``get_basin_numbers`` and ``get_station_height`` behave the same and the same
ValueError is raised for an unknown basin.  Two things are added for the
ensemble engine this package is about:

* ``CAMELSLoader(data_dir=...)`` reads any directory of CAMELS-format files
  (``<basin>_lump_cida_forcing_leap.txt`` + ``<basin>_05_model_output.txt``,
  plain or ``.gz``), not only the toy basin that ships with the package;
* ``forcing(basin)`` hands back contiguous float64 arrays named like the
  model arguments (prec, mean_temp, min_temp, max_temp, etp, month, qobs),
  i.e. what ``Model.simulate`` / ``monte_carlo`` upload once per sweep.

The toy basin (01031500) is the one the reference distributes; it is CAMELS
data (Addor et al. 2017, doi:10.5065/D6G73C3Q), stored gzipped under
``rrmpg_amd/data/camels/``.
"""

import os
import re

import numpy as np
import pandas as pd

_PACKAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "camels")
_MET = "{}_lump_cida_forcing_leap.txt"
_FLOW = "{}_05_model_output.txt"
_MET_RE = re.compile(r"^(\d+)_lump_cida_forcing_leap\.txt(\.gz)?$")


def _existing(path):
    for cand in (path, path + ".gz"):
        if os.path.isfile(cand):
            return cand
    return None


def _date_index(year, month, day):
    return pd.DatetimeIndex(pd.to_datetime(
        dict(year=np.asarray(year), month=np.asarray(month),
             day=np.asarray(day))))


class CAMELSLoader(object):
    """Read basins of the CAMELS data set (reference:
    rrmpg/data/camelsloader.py:14-34)."""

    def __init__(self, data_dir=None):
        self.data_dir = os.fspath(data_dir) if data_dir else _PACKAGED
        basins = []
        if os.path.isdir(self.data_dir):
            for name in sorted(os.listdir(self.data_dir)):
                m = _MET_RE.match(name)
                if m and _existing(os.path.join(self.data_dir,
                                                _FLOW.format(m.group(1)))):
                    if m.group(1) not in basins:
                        basins.append(m.group(1))
        #: basins that can be loaded (the reference's class attribute of the
        #: same name lists its one packaged basin, camelsloader.py:32)
        self.VALID_BASINS = basins

    # -- helpers -----------------------------------------------------------
    def _check(self, basin_number):
        if basin_number not in self.VALID_BASINS:
            # same text as camelsloader.py:55-57 / :113-115
            raise ValueError(f"Invalid basin number {basin_number}. Must be "
                             f"one of {self.VALID_BASINS}.")

    def _files(self, basin_number):
        met = _existing(os.path.join(self.data_dir, _MET.format(basin_number)))
        flow = _existing(os.path.join(self.data_dir,
                                      _FLOW.format(basin_number)))
        return met, flow

    # -- reference surface -------------------------------------------------
    def get_basin_numbers(self):
        """List of the basins available (camelsloader.py:97-99)."""
        return self.VALID_BASINS

    def get_station_height(self, basin_number):
        """Elevation of the basin's meteorological station: the second header
        line of the forcing file (camelsloader.py:101-128)."""
        self._check(basin_number)
        met, _ = self._files(basin_number)
        head = pd.read_csv(met, header=None, nrows=3, sep=r"\s+")
        return float(head.iloc[1, 0])

    def load_basin(self, basin_number):
        """DataFrame of daily meteorology + PET + observed discharge, cut to
        complete hydrological years (camelsloader.py:37-95).

        Columns, in order: the forcing file's ``dayl(s) prcp(mm/day)
        srad(W/m2) swe(mm) tmax(C) tmin(C) vp(Pa)``, then ``PET`` and
        ``QObs(mm/d)`` taken from the model-output file by date.
        """
        self._check(basin_number)
        met, flow = self._files(basin_number)
        # three header lines (latitude, elevation, area) precede the column
        # names of the forcing file
        frame = pd.read_csv(met, sep=r"\s+", skiprows=3)
        frame.index = _date_index(frame["Year"], frame["Mnth"], frame["Day"])
        out = pd.read_csv(flow, sep=r"\s+")
        out.index = _date_index(out["YR"], out["MNTH"], out["DY"])
        frame = frame.drop(columns=["Year", "Mnth", "Day", "Hr"])
        # aligned by date: days the model-output file lacks become NaN, then
        # fall outside the window below
        frame["PET"] = out["PET"]
        frame["QObs(mm/d)"] = out["OBS_RUN"]
        first = pd.Timestamp(year=frame.index[0].year, month=10, day=1)
        last = pd.Timestamp(year=frame.index[-1].year, month=9, day=30)
        return frame[first:last]

    # -- engine-facing -----------------------------------------------------
    def forcing(self, basin_number):
        """Contiguous float64 arrays named like the model arguments.

        Keys: prec, mean_temp, min_temp, max_temp, etp, qobs, month (int8,
        1..12), dates (datetime64[D]) and met_station_height.  mean_temp is
        (tmax + tmin) / 2, as the reference's examples compute it.
        """
        df = self.load_basin(basin_number)
        col = lambda name: np.ascontiguousarray(df[name].to_numpy(),
                                                dtype=np.float64)
        tmax, tmin = col("tmax(C)"), col("tmin(C)")
        return dict(
            prec=col("prcp(mm/day)"), mean_temp=(tmax + tmin) / 2,
            min_temp=tmin, max_temp=tmax, etp=col("PET"),
            qobs=col("QObs(mm/d)"),
            month=df.index.month.to_numpy().astype(np.int8),
            dates=df.index.to_numpy().astype("datetime64[D]"),
            met_station_height=self.get_station_height(basin_number))
