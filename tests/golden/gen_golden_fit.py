#!/usr/bin/env python3
"""Golden fixture for ``Model.fit``: the REFERENCE's own optimiser runs.

Runs only in the build container (needs /root/reference).  The reference's
``fit`` (e.g. rrmpg/models/hbvedu.py:216-307) is scipy's differential
evolution over its ``_loss`` with the default arguments -- immediate updating,
one candidate per loss evaluation, and with ``seed=None`` scipy draws from
numpy's global RandomState, so ``np.random.seed(k)`` before ``fit`` fixes the
whole run.  Executed here under the no-op numba stub
on a short synthetic series, with the module's ``_loss`` wrapped to log every
evaluation.

Stored per model (data only): the forcing, observations, initial states, the
seed, res.x / res.fun / res.nfev / res.nit and the logged loss of every
evaluation in call order (the optimiser's trajectory).
``rrmpg_amd``'s ``fit(batched=False)`` claims the same call shape; the GPU
test ``tests/test_gpu_fit_reference.py`` holds it to these numbers.

Usage:  python tests/golden/gen_golden_fit.py   (writes tests/golden/fit_ref.npz)
"""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import REF, REPO, _install_numba_stub   # noqa: E402


def main():
    if not os.path.isdir(REF):
        print("no /root/reference here - generated in the build container only")
        return 0
    _install_numba_stub()
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    sys.dont_write_bytecode = True
    import warnings
    warnings.filterwarnings("ignore")
    from rrmpg.models import ABCModel, GR4J, HBVEdu
    from rrmpg.models import abcmodel as abc_mod
    from rrmpg.models import gr4j as gr4j_mod
    from rrmpg.models import hbvedu as hbv_mod
    from rrmpg_amd.utils import synthetic as syn

    t = 120
    f = syn.make_forcing(400)
    sl = slice(200, 200 + t)            # a stretch with rain, frost and thaw
    rng = np.random.default_rng(20260929)
    out = {"T": t}

    def logged(mod):
        log = []
        inner = mod._loss

        def wrapper(X, *args):
            v = inner(X, *args)
            log.append(float(v))
            return v
        mod._loss = wrapper
        return log

    # ---- HBV-Edu
    truth = dict(T_t=0.2, DD=4.0, FC=180.0, Beta=2.5, C=0.04, PWP=120.0,
                 K_0=0.1, K_1=0.05, K_2=0.02, K_p=0.03, L=3.5)
    m = HBVEdu(params=truth)
    forcing = dict(temp=f["temp"][sl], prec=f["prec"][sl], month=f["month"][sl],
                   PE_m=f["PE_m"], T_m=f["T_m"])
    inits = dict(snow_init=0.0, soil_init=100.0, s1_init=3.0, s2_init=10.0)
    q = m.simulate(**forcing, **inits).ravel()
    qobs = q * (1 + 0.1 * rng.standard_normal(t))
    log = logged(hbv_mod)
    np.random.seed(11)
    t0 = time.time()
    res = HBVEdu().fit(qobs, **forcing, **inits)
    print("HBVEdu.fit: nfev %d nit %d fun %.6e (%.0f s)"
          % (res.nfev, res.nit, res.fun, time.time() - t0))
    # (HBVEdu() itself draws random parameters first: part of the seeded run)
    out.update(hbv_qobs=qobs, hbv_temp=forcing["temp"], hbv_prec=forcing["prec"],
               hbv_month=forcing["month"], hbv_PE_m=f["PE_m"], hbv_T_m=f["T_m"],
               hbv_inits=np.array(list(inits.values())), hbv_seed=11,
               hbv_x=res.x, hbv_fun=res.fun, hbv_nfev=res.nfev,
               hbv_nit=res.nit, hbv_losses=np.array(log))

    # ---- GR4J
    m = GR4J(params=dict(x1=320.0, x2=0.8, x3=75.0, x4=1.7))
    gf = dict(prec=f["prec"][sl], etp=f["etp"][sl])
    gi = dict(s_init=0.6, r_init=0.7)
    q = m.simulate(**gf, **gi, return_storage=True)[0].ravel()
    qobs = q * (1 + 0.1 * rng.standard_normal(t))
    log = logged(gr4j_mod)
    np.random.seed(12)
    t0 = time.time()
    res = GR4J().fit(qobs, **gf, **gi)
    print("GR4J.fit: nfev %d nit %d fun %.6e (%.0f s)"
          % (res.nfev, res.nit, res.fun, time.time() - t0))
    out.update(gr4j_qobs=qobs, gr4j_prec=gf["prec"], gr4j_etp=gf["etp"],
               gr4j_inits=np.array([0.6, 0.7]), gr4j_seed=12, gr4j_x=res.x,
               gr4j_fun=res.fun, gr4j_nfev=res.nfev, gr4j_nit=res.nit,
               gr4j_losses=np.array(log))

    # ---- ABC (bit-exact kernel: the tightest pin of the call shape)
    m = ABCModel(params=dict(a=0.3, b=0.2, c=0.1))
    q = m.simulate(f["prec"][sl], initial_state=2.0).ravel()
    qobs = q * (1 + 0.1 * rng.standard_normal(t))
    log = logged(abc_mod)
    np.random.seed(13)
    res = ABCModel().fit(qobs, f["prec"][sl], initial_state=2.0)
    print("ABCModel.fit: nfev %d nit %d fun %.6e" % (res.nfev, res.nit, res.fun))
    out.update(abc_qobs=qobs, abc_prec=f["prec"][sl], abc_init=2.0,
               abc_seed=13, abc_x=res.x, abc_fun=res.fun, abc_nfev=res.nfev,
               abc_nit=res.nit, abc_losses=np.array(log))

    np.savez_compressed(os.path.join(HERE, "fit_ref.npz"), **out)
    print("wrote fit_ref.npz")
    return 0


if __name__ == "__main__":
    sys.exit(main())
