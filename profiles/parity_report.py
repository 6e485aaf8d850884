#!/usr/bin/env python3
"""Observed GPU-vs-oracle deviations per model (run on the GPU box).
1000 parameter sets x 10,957 days each; prints max relative deviation
(|a-b| / max(|b|, 1e-9)) per output and the fraction of bit-identical values."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle                                   # noqa: E402
from rrmpg_amd import models                                  # noqa: E402
from rrmpg_amd.models.cemaneige import prepare_snow_inputs    # noqa: E402
from rrmpg_amd.utils import synthetic as syn                  # noqa: E402


def dev(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-9)))


def flat(p, cls):
    return np.stack([p[n] for n in cls._param_list], 1)


def main():
    f = syn.make_forcing()
    n = 1000
    np.random.seed(int(os.environ.get("RR_PARITY_SEED", "7")))
    rows = []
    p = models.ABCModel().get_random_params(n)
    out = models.ABCModel().simulate(f["prec"], 2.0, True, p)
    ref = pyoracle.simulate_abc(f["prec"], 2.0, flat(p, models.ABCModel), True, 8)
    rows.append(("ABC", ["qsim", "storage"], out, ref))
    p = models.HBVEdu().get_random_params(n)
    out = models.HBVEdu().simulate(f["temp"], f["prec"], f["month"], f["PE_m"],
                                   f["T_m"], 0., 100., 3., 10., True, p)
    ref = pyoracle.simulate_hbvedu(f["temp"], f["prec"], f["month"] - 1,
                                   f["PE_m"], f["T_m"], (0., 100., 3., 10.),
                                   flat(p, models.HBVEdu), True, 8)
    rows.append(("HBV-Edu", ["qsim", "snow", "soil", "s1", "s2"], out, ref))
    p = models.GR4J().get_random_params(n)
    out = models.GR4J().simulate(f["prec"], f["etp"], 0.6, 0.7, True, p)
    ref = pyoracle.simulate_gr4j(f["prec"], f["etp"], (0.6, 0.7),
                                 flat(p, models.GR4J), True, 8)
    rows.append(("GR4J", ["qsim", "s_store", "r_store"], out, ref))
    layers, _ = prepare_snow_inputs(f["prec"], f["temp"], f["tmin"], f["tmax"],
                                    syn.STATION_HEIGHT, 0, 0,
                                    list(syn.ALTITUDES), etp=f["etp"])
    p = models.Cemaneige().get_random_params(n)
    out = models.Cemaneige().simulate(f["prec"], f["temp"], f["tmin"], f["tmax"],
                                      syn.STATION_HEIGHT, 2.0, -0.3,
                                      list(syn.ALTITUDES), True, p)
    ref = pyoracle.simulate_cemaneige(layers[0], layers[1], layers[2],
                                      (2.0, -0.3), flat(p, models.Cemaneige),
                                      True, 8)
    rows.append(("Cemaneige(L=5)", ["outflow", "G", "eTG"], out, ref))
    p = models.CemaneigeGR4J().get_random_params(n)
    out = models.CemaneigeGR4J().simulate(
        f["prec"], f["temp"], f["tmin"], f["tmax"], f["etp"],
        syn.STATION_HEIGHT, 2.0, -0.3, 0.6, 0.7, list(syn.ALTITUDES), True, p)
    ref = pyoracle.simulate_cemaneigegr4j(
        layers[0], layers[1], layers[3], layers[2], (2.0, -0.3, 0.6, 0.7),
        flat(p, models.CemaneigeGR4J), True, 8)
    rows.append(("CemaneigeGR4J(L=5)", ["qsim", "G", "eTG", "s_store",
                                        "r_store"], out, ref))
    print("| model | output | max rel. deviation | bit-identical |")
    print("|---|---|---|---|")
    for name, outs, a, b in rows:
        for o, x, y in zip(outs, a, b):
            print("| %s | %s | %.2e | %.1f %% |"
                  % (name, o, dev(x, y), 100 * np.mean(x == y)))


if __name__ == "__main__":
    main()
