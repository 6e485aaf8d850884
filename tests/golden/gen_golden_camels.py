#!/usr/bin/env python3
"""Golden fixture for the CAMELS loader: what the REFERENCE's
rrmpg.data.CAMELSLoader returns for its packaged basin (build container only;
data only -- the DataFrame's values, column names and date range).

Usage:  python tests/golden/gen_golden_camels.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import REF, _install_numba_stub   # noqa: E402


def main():
    if not os.path.isdir(REF):
        print("no /root/reference here; nothing to do")
        return 0
    _install_numba_stub()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import warnings
    warnings.filterwarnings("ignore")
    from rrmpg.data import CAMELSLoader
    loader = CAMELSLoader()
    basin = loader.get_basin_numbers()[0]
    df = loader.load_basin(basin)
    np.savez_compressed(
        os.path.join(HERE, "camels_loader.npz"),
        basin=np.array(basin), basins=np.array(loader.get_basin_numbers()),
        columns=np.array(list(df.columns)),
        values=df.to_numpy(dtype=np.float64),
        dates=df.index.to_numpy().astype("datetime64[D]").astype(np.int64),
        station_height=np.array(loader.get_station_height(basin)))
    print("camels_loader.npz", df.shape, df.index[0], df.index[-1])
    return 0


if __name__ == "__main__":
    sys.exit(main())
