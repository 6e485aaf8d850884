#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer family (numpy in / numpy out):

  * HBVEdu.simulate for 100k sets x 30 yr (BASELINE configs[1]), qsim only;
  * the same for a sweep whose qsim (35 GB at 400k sets) is streamed through
    double-buffered column blocks instead of being held in HBM;
  * monte_carlo(return_qsim=False): scores only;
  * one-candidate loss evaluations as Model.fit(batched=False) makes them, on
    the reference's 10-year series length, and a whole fit either way.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrmpg_amd.models import HBVEdu, hbvedu as hbv_mod
from rrmpg_amd.tools import monte_carlo
from rrmpg_amd.utils import synthetic as syn

f = syn.make_forcing()
np.random.seed(1)
m = HBVEdu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
big = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
p = m.get_random_params(max(n, big))
kw = dict(temp=f["temp"], prec=f["prec"], month=f["month"], PE_m=f["PE_m"],
          T_m=f["T_m"], **syn.HBV_INITS)
m.simulate(params=p[:1000], **kw)
for count in (n, n, n, big):
    q = None                      # free the previous result outside the clock
    t0 = time.perf_counter()
    q = m.simulate(params=p[:count], **kw)
    dt = time.perf_counter() - t0
    print("simulate  N=%d: %.3f s  %.3e model-timesteps/s  (%.2f GB of qsim "
          "to host, %.2f GB/s)" % (count, dt, count * syn.T_30YR / dt,
                                   q.nbytes / 1e9, q.nbytes / 1e9 / dt))
qobs = q[:, 0].copy()
del q
for rep in range(3):
    t0 = time.perf_counter()
    res = monte_carlo(m, n, qobs=qobs, return_qsim=False, **kw)
    dt = time.perf_counter() - t0
    print("monte_carlo(return_qsim=False) N=%d: %.3f s  %.3e model-timesteps/s"
          % (n, dt, n * syn.T_30YR / dt))
t0 = time.perf_counter()
pr = m.get_random_params(n)
print("   (of which get_random_params: %.3f s)" % (time.perf_counter() - t0))

# calibration: the reference's loss call, one candidate at a time
t10 = syn.T_10YR
g = syn.make_forcing(t10)
truth = HBVEdu(params=dict(zip(HBVEdu._param_list,
                               [0., 4., 150., 3., .04, 120., .1, .05, .03,
                                .02, 3.])))
kw10 = dict(temp=g["temp"], prec=g["prec"], month=g["month"], PE_m=g["PE_m"],
            T_m=g["T_m"], **syn.HBV_INITS)
obs = truth.simulate(**kw10).ravel()
X = np.array([0.3, 4.5, 140., 2.5, .03, 110., .12, .04, .02, .03, 2.5])
from rrmpg_amd.utils.array_checks import validate_array_input
args = (obs, validate_array_input(g["temp"], np.float64, 'temp'),
        validate_array_input(g["prec"], np.float64, 'prec'),
        (g["month"] - 1).astype(np.int8), g["PE_m"], g["T_m"], 0., 100., 3.,
        10., HBVEdu._dtype)
try:
    hbv_mod._loss(X, *args)
    t0 = time.perf_counter()
    for _ in range(2000):
        hbv_mod._loss(X, *args)
    dt = (time.perf_counter() - t0) / 2000
    print("_loss, one candidate, %d days: %.3f ms per evaluation" % (t10, dt * 1e3))
except Exception as e:                      # signature drift: report, go on
    print("_loss timing skipped:", repr(e))
for batched in (True, False):
    np.random.seed(5)
    t0 = time.perf_counter()
    r = HBVEdu().fit(obs, g["temp"], g["prec"], g["month"], g["PE_m"],
                     g["T_m"], soil_init=100., s1_init=3., s2_init=10.,
                     batched=batched)
    dt = time.perf_counter() - t0
    print("fit(batched=%s): %.2f s, %d evaluations, final MSE %.3e"
          % (batched, dt, r.nfev, r.fun))
