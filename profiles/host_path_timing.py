#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer family (numpy in / numpy out):
HBVEdu.simulate for 100k sets x 30 yr (BASELINE configs[1]), qsim only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrmpg_amd.models import HBVEdu
from rrmpg_amd.tools import monte_carlo
from rrmpg_amd.utils import synthetic as syn

f = syn.make_forcing()
np.random.seed(1)
m = HBVEdu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
p = m.get_random_params(n)
kw = dict(temp=f["temp"], prec=f["prec"], month=f["month"], PE_m=f["PE_m"],
          T_m=f["T_m"], **syn.HBV_INITS)
m.simulate(params=p[:1000], **kw)
for rep in range(3):
    q = None                      # free the previous result outside the clock
    t0 = time.perf_counter()
    q = m.simulate(params=p, **kw)
    dt = time.perf_counter() - t0
    print("simulate  N=%d: %.3f s  %.3e model-timesteps/s  (%.2f GB of qsim "
          "to host, %.2f GB/s)" % (n, dt, n * syn.T_30YR / dt, q.nbytes / 1e9,
                                   q.nbytes / 1e9 / dt))
qobs = q[:, 0].copy()
del q
t0 = time.perf_counter()
res = monte_carlo(m, n, qobs=qobs, return_qsim=False, **kw)
dt = time.perf_counter() - t0
print("monte_carlo(return_qsim=False) N=%d: %.3f s  %.3e model-timesteps/s"
      % (n, dt, n * syn.T_30YR / dt))
