#!/usr/bin/env python3
"""Golden fixtures for the next-tier models (hysteresis snow routine, ice
melt and their couplings with GR4J), generated from the REFERENCE itself --
same method and caveats as gen_golden.py (build container only; no-op numba
stub; data only).  KATs: reference test/test_models.py:270-356.

Usage:  python tests/golden/gen_golden_next.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import REF, REPO, _install_numba_stub   # noqa: E402


def main():
    if not os.path.isdir(REF):
        print("no /root/reference here; nothing to do")
        return 0
    _install_numba_stub()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    sys.dont_write_bytecode = True
    import warnings
    warnings.filterwarnings("ignore")
    import pandas as pd
    from rrmpg.models import (CemaneigeHystGR4J, CemaneigeGR4JIce,
                              CemaneigeHystGR4JIce)
    from rrmpg.models import cemaneige_utils as cu
    from rrmpg.models.cemaneigehystgr4j_model import run_cemaneigehystgr4j
    from rrmpg.models.cemaneigegr4jice_model import run_cemaneigegr4jice
    from rrmpg.models.cemaneigehystgr4jice_model import run_cemaneigehystgr4jice
    from rrmpg_amd.utils import synthetic as syn

    tdata = os.path.join(REF, "test", "data")
    alts = [550, 620, 700, 785, 920]
    hyst_params = dict(Thacc=18.6, Rsp=0.22, CTG=0.78, Kf=4.02, x1=546,
                       x2=0.53, x3=276, x4=1.32)

    def cols(df):
        return dict(prec=df.precipitation.to_numpy(np.float64),
                    mean_temp=df.mean_temp.to_numpy(np.float64),
                    min_temp=df.min_temp.to_numpy(np.float64),
                    max_temp=df.max_temp.to_numpy(np.float64),
                    etp=df.pe.to_numpy(np.float64),
                    qsim_excel=df.qsim.to_numpy(np.float64))

    # CemaneigeHystGR4J KAT (test_models.py:293-310)
    df = pd.read_csv(os.path.join(tdata, "cemaneigehystgr4j_validation_data.csv"),
                     index_col=0)
    m = CemaneigeHystGR4J(params=hyst_params)
    out = m.simulate(df.precipitation, df.mean_temp, df.min_temp, df.max_temp,
                     df.pe, met_station_height=700, altitudes=alts, s_init=0.5,
                     r_init=0.4, return_storages=True)
    names = ["qsim", "G", "eTG", "s_store", "r_store", "sca", "rain"]
    np.savez_compressed(
        os.path.join(HERE, "kat_cemaneigehystgr4j.npz"), **cols(df),
        altitudes=np.array(alts, dtype=np.float64), station=np.float64(700),
        params=np.array([hyst_params[k] for k in m.get_parameter_names()]),
        inits=np.array([0., 0., 0., 0.5, 0.4]),
        **{"ref_" + n: a for n, a in zip(names, out)})

    # CemaneigeHystGR4JIce KAT (test_models.py:336-356)
    df = pd.read_csv(os.path.join(tdata,
                                  "cemaneigehystgr4jice_validation_data.csv"),
                     index_col=0)
    ice_params = dict(hyst_params, DDF=5)
    frac_ice = np.array([0.02, 0.04, 0.25, 0.51, 0.71])
    m = CemaneigeHystGR4JIce(params=ice_params)
    out = m.simulate(df.precipitation, df.mean_temp, df.min_temp, df.max_temp,
                     df.pe, frac_ice, met_station_height=700, altitudes=alts,
                     s_init=0.5, r_init=0.4, sca_init=0.2,
                     return_storages=True)
    names = ["qsim", "G", "eTG", "s_store", "r_store", "sca", "icemelt",
             "snowmelt", "rain"]
    np.savez_compressed(
        os.path.join(HERE, "kat_cemaneigehystgr4jice.npz"), **cols(df),
        altitudes=np.array(alts, dtype=np.float64), station=np.float64(700),
        frac_ice=frac_ice,
        params=np.array([ice_params[k] for k in m.get_parameter_names()],
                        dtype=np.float64),
        inits=np.array([0., 0., 0.2, 0.5, 0.4]),
        **{"ref_" + n: a for n, a in zip(names, out)})

    # synthetic multi-set sweeps, 10-year series, L = 5
    f = syn.make_forcing(syn.T_10YR)
    alt = np.array(syn.ALTITUDES)
    lprec = cu.extrapolate_precipitation(f["prec"], alt, syn.STATION_HEIGHT)
    lmin, lmean, lmax = cu.extrapolate_temperature(
        f["tmin"] - 3, f["temp"] - 3, f["tmax"] - 3, alt, syn.STATION_HEIGHT)
    frac = cu.calculate_solid_fraction(lprec, alt, lmean, lmin, lmax)
    M = 12
    idx = np.arange(0, syn.T_10YR, 53)
    inits = (4.0, -0.3, 0.35, 0.6, 0.7)

    def as2d(p, names):
        return np.stack([p[n] for n in names], axis=1)

    def pack(keys, runs):
        d = {}
        for j, k in enumerate(keys):
            a = np.stack([r[j] for r in runs], axis=-1)
            d[k] = a if a.ndim == 2 else a[idx]
            if a.ndim == 3:
                d[k + "_last"] = a[-1]
        return d

    np.random.seed(1)
    mdl = CemaneigeHystGR4J()
    p = mdl.get_random_params(M)
    p["x4"][:4] = np.random.uniform(1.1, 2.9, 4)     # register UH tier too
    runs = [run_cemaneigehystgr4j(lprec, lmean, f["etp"], frac, *inits, p[i])
            for i in range(M)]
    np.savez_compressed(
        os.path.join(HERE, "syn_cemaneigehystgr4j.npz"), layer_prec=lprec,
        layer_mean=lmean, frac_solid=frac, etp=f["etp"],
        inits=np.array(inits), params=as2d(p, mdl.get_parameter_names()),
        stride_idx=idx,
        **pack(["qsim", "G", "eTG", "s_store", "r_store", "sca", "rain"],
               runs))

    np.random.seed(2)
    mdl = CemaneigeGR4JIce()
    p = mdl.get_random_params(M)
    runs = [run_cemaneigegr4jice(lprec, lmean, f["etp"], frac_ice, frac,
                                 inits[0], inits[1], inits[3], inits[4], p[i])
            for i in range(M)]
    np.savez_compressed(
        os.path.join(HERE, "syn_cemaneigegr4jice.npz"), frac_ice=frac_ice,
        inits=np.array(inits), params=as2d(p, mdl.get_parameter_names()),
        stride_idx=idx,
        **pack(["qsim", "G", "eTG", "s_store", "r_store", "icemelt"], runs))

    np.random.seed(3)
    mdl = CemaneigeHystGR4JIce()
    p = mdl.get_random_params(M)
    p["x4"][:4] = np.random.uniform(1.1, 2.9, 4)
    runs = [run_cemaneigehystgr4jice(lprec, lmean, f["etp"], frac_ice, frac,
                                     *inits, p[i]) for i in range(M)]
    np.savez_compressed(
        os.path.join(HERE, "syn_cemaneigehystgr4jice.npz"), frac_ice=frac_ice,
        inits=np.array(inits), params=as2d(p, mdl.get_parameter_names()),
        stride_idx=idx,
        **pack(["qsim", "G", "eTG", "s_store", "r_store", "sca", "icemelt",
                "snowmelt", "rain"], runs))

    # sampling streams of the three classes
    samp = {}
    for name, cls in [("cemaneigehystgr4j", CemaneigeHystGR4J),
                      ("cemaneigegr4jice", CemaneigeGR4JIce),
                      ("cemaneigehystgr4jice", CemaneigeHystGR4JIce)]:
        np.random.seed(1234)
        mdl = cls()
        pp = mdl.get_random_params(7)
        samp[name + "_ctor"] = np.array(
            [mdl.get_params()[k] for k in mdl.get_parameter_names()])
        samp[name + "_rand7"] = as2d(pp, mdl.get_parameter_names())
    np.savez_compressed(os.path.join(HERE, "sampling_next.npz"), **samp)

    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz") and ("hyst" in fn or "ice" in fn
                                    or "next" in fn):
            print("%-34s %8.1f KB" % (fn, os.path.getsize(
                os.path.join(HERE, fn)) / 1024))
    return 0


if __name__ == "__main__":
    sys.exit(main())
