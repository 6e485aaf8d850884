"""Deterministic synthetic daily forcing for parity tests and the benchmark.

There is no network (no CAMELS download) on the build or GPU boxes, so every
measured workload uses the series below.  The shapes
follow the reference's own inputs: ``prec``/``temp``/``etp`` daily series, a
1..12 ``month`` vector and the two 12-vectors HBV-Edu wants
(reference: rrmpg/models/hbvedu.py:82-214), plus min/max temperature and
elevation-layer altitudes for the Cemaneige family
(reference: rrmpg/models/cemaneige.py:81-245).
"""

import numpy as np

#: 1981-01-01 .. 2010-12-31
T_30YR = 10957
#: ten years of daily steps
T_10YR = 3653

FORCING_SEED = 20260928

PE_M = np.array([.2, .3, .8, 1.6, 2.6, 3.3, 3.6, 3.1, 2., 1., .4, .2])
T_M = np.array([-2., -1., 3., 8., 13., 16., 18., 17., 13., 8., 3., -1.])

STATION_HEIGHT = 500
ALTITUDES = [550, 620, 700, 785, 920]

HBV_INITS = dict(snow_init=0., soil_init=100., s1_init=3., s2_init=10.)
GR4J_INITS = dict(s_init=0.6, r_init=0.7)


def make_forcing(num_timesteps=T_30YR, seed=FORCING_SEED):
    """Return a dict of synthetic daily forcing series of length num_timesteps.

    Keys: temp, prec, etp, month (1..12, int8), tmin, tmax, PE_m, T_m.
    """
    rng = np.random.default_rng(seed)
    t = np.arange(num_timesteps)
    doy = t % 365.25
    season = np.sin(2 * np.pi * (doy - 110) / 365.25)
    temp = 8 + 12 * season + rng.normal(0, 3, num_timesteps)
    wet = rng.random(num_timesteps) < 0.4
    prec = wet * rng.gamma(0.8, 6.0, num_timesteps)
    etp = np.clip(2.0 + 1.8 * season + rng.normal(0, 0.2, num_timesteps),
                  0, None)
    month = (np.floor(doy / 30.4375).astype(np.int64) % 12 + 1).astype(np.int8)
    return dict(temp=temp, prec=prec, etp=etp, month=month,
                tmin=temp - 4, tmax=temp + 5,
                PE_m=PE_M.copy(), T_m=T_M.copy())


def make_qobs(qsim_truth, seed=FORCING_SEED + 7):
    """Synthetic 'observed' discharge: a truth run with 10 % noise."""
    rng = np.random.default_rng(seed)
    q = np.asarray(qsim_truth, dtype=np.float64).ravel()
    return np.clip(q * (1 + 0.1 * rng.normal(0, 1, q.size)), 0, None)
