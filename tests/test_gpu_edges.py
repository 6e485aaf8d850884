"""Edge shapes through the resident (*_dev) entry points: one to three time
steps, set counts around the 64-lane wave (1, 63, 64, 65, 129), and output
rows wider than the set count (ld > N) whose padding columns must not be
touched -- the row stores are buffer stores whose descriptor drops the
columns past N, this pins that.  Results against the CPU oracle (1e-10
relative; ABC bit-exact)."""

import numpy as np
import pytest

from .conftest import rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-10
SENTINEL = -12345.678


@pytest.fixture(scope="module")
def env():
    import torch
    from rrmpg_amd import _lib
    _lib.load()
    _lib.require_gpu()
    from rrmpg_amd import device as rrdev
    from rrmpg_amd.utils import synthetic as syn
    import rrmpg_amd.models as models
    return torch, rrdev, syn, models


def _flat(p, cls):
    return np.stack([p[n] for n in cls._param_list], axis=1)


def _padded(torch, t, n, pad):
    buf = torch.full((t, n + pad), SENTINEL, dtype=torch.float64,
                     device="cuda")
    return buf, buf[:, :n]


@pytest.mark.parametrize("t", [1, 2, 3, 400])
def test_hbv_gr4j_abc_edge_shapes(env, oracle, t, hbv_variant, gr4j_variant):
    torch, rrdev, syn, models = env
    f = syn.make_forcing(max(t, 2))
    f = {k: (v[:t] if getattr(v, "shape", (0,))[0] >= t and k not in
             ("PE_m", "T_m") else v) for k, v in f.items()}
    for n in (1, 63, 64, 65, 129):
        np.random.seed(100 + n)
        # ---- HBV-Edu: qsim + one storage + fused SSE
        p = models.HBVEdu().get_random_params(n)
        ens = rrdev.HBVEduEnsemble(f["temp"], f["prec"], f["month"], f["PE_m"],
                                   f["T_m"], **syn.HBV_INITS)
        qbuf, q = _padded(torch, t, n, 7)
        sbufs = [_padded(torch, t, n, 7) for _ in range(4)]
        qobs = torch.linspace(0.1, 2.0, t, dtype=torch.float64, device="cuda")
        sse = ens.run(ens.upload_params(p), q, tuple(v for _, v in sbufs),
                      qobs=qobs)
        torch.cuda.synchronize()
        ref = oracle.simulate_hbvedu(
            f["temp"], f["prec"], f["month"] - 1, f["PE_m"], f["T_m"],
            [syn.HBV_INITS[k] for k in ("snow_init", "soil_init", "s1_init",
                                        "s2_init")],
            _flat(p, models.HBVEdu), return_storage=True)
        got = [q.cpu().numpy()] + [v.cpu().numpy() for _, v in sbufs]
        for a, b in zip(got, ref):
            assert rel_err(a, b, floor=1e-9) < RTOL, (t, n)
        want_sse = ((qobs.cpu().numpy()[:, None] - got[0]) ** 2).sum(axis=0)
        assert np.allclose(sse.cpu().numpy(), want_sse, rtol=1e-12, atol=0)
        for buf in [qbuf] + [b for b, _ in sbufs]:
            assert bool((buf[:, n:] == SENTINEL).all()), (t, n)
        # ---- GR4J: qsim + both stores
        p = models.GR4J().get_random_params(n)
        ens = rrdev.GR4JEnsemble(f["prec"], f["etp"], **syn.GR4J_INITS)
        qbuf, q = _padded(torch, t, n, 5)
        sb, s = _padded(torch, t, n, 5)
        rb, r = _padded(torch, t, n, 5)
        ens.run(ens.upload_params(p), q, (s, r))
        torch.cuda.synchronize()
        ref = oracle.simulate_gr4j(
            f["prec"], f["etp"], (syn.GR4J_INITS["s_init"],
                                  syn.GR4J_INITS["r_init"]),
            _flat(p, models.GR4J), return_storage=True)
        for a, b in zip((q, s, r), ref):
            assert rel_err(a.cpu().numpy(), b, floor=1e-9) < RTOL, (t, n)
        for buf in (qbuf, sb, rb):
            assert bool((buf[:, n:] == SENTINEL).all()), (t, n)
        # ---- ABC (both store shapes: even and odd ld)
        p = models.ABCModel().get_random_params(n)
        ens = rrdev.ABCEnsemble(f["prec"], 1.5)
        for pad in (3, 4):
            qbuf, q = _padded(torch, t, n, pad)
            sb, s = _padded(torch, t, n, pad)
            ens.run(ens.upload_params(p), q, s)
            torch.cuda.synchronize()
            ref = oracle.simulate_abc(f["prec"], 1.5, _flat(p, models.ABCModel),
                                      return_storage=True)
            assert np.array_equal(q.cpu().numpy(), ref[0])
            assert np.array_equal(s.cpu().numpy(), ref[1])
            assert bool((qbuf[:, n:] == SENTINEL).all())
            assert bool((sb[:, n:] == SENTINEL).all())
