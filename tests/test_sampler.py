"""HBM-resident parameter sampling.

The device sampler's stream is numpy's Philox4x64-10 bit generator.  The CPU
tests pin that definition (Random123's known-answer vector, numpy's counter
convention, the uniform mapping); the GPU tests require the HIP kernel to
reproduce the host population bit for bit."""

import numpy as np
import pytest

M0, M1 = 0xD2E7470EE14C6C93, 0xCA5A826395121157
W0, W1 = 0x9E3779B97F4A7C15, 0xBB67AE8584CAA73B
MASK = (1 << 64) - 1


def philox4x64_10(ctr, key):
    """Salmon et al. (SC'11), Philox4x64 with 10 rounds -- plain restatement."""
    c, k = list(ctr), list(key)
    for r in range(10):
        if r:
            k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 64) ^ c[1] ^ k[0], p1 & MASK,
             (p0 >> 64) ^ c[3] ^ k[1], p0 & MASK]
    return c


def stream_uniform(key, e, lo, hi):
    w = philox4x64_10([e // 4 + 1, 0, 0, 0], [key, 0])[e % 4]
    return lo + (hi - lo) * ((w >> 11) * (1.0 / 9007199254740992.0))


def test_philox_known_answer_and_numpy_convention():
    # Random123 kat_vectors: philox4x64-10, counter 0, key 0
    assert philox4x64_10([0] * 4, [0] * 2) == [
        0x16554d9eca36314c, 0xdb20fe9d672d0fdc, 0xd7e772cee186176b,
        0x7e68b68aec7ba23b]
    key = 20260928
    raw = np.random.Philox(key=key).random_raw(12)
    mine = sum((philox4x64_10([b, 0, 0, 0], [key, 0]) for b in (1, 2, 3)), [])
    assert [int(x) for x in raw] == mine
    u = np.random.Generator(np.random.Philox(key=key)).uniform(-2.5, 7, 9)
    assert u.tolist() == [stream_uniform(key, e, -2.5, 7) for e in range(9)]


def test_host_population_is_the_documented_stream():
    from rrmpg_amd import device
    from rrmpg_amd.models import ABCModel, GR4J
    key, n = 77, 50
    pop = device.host_population(GR4J(), n, key)
    m = GR4J()
    for j, name in enumerate(m._param_list):
        lo, hi = m._default_bounds[name]
        for i in (0, 1, 17, 49):
            assert pop[i, j] == stream_uniform(key, j * n + i, lo, hi)
    abc = device.host_population(ABCModel(), n, key)
    a, b, c = abc.T
    assert np.all(a + b <= 1) and np.all(b >= 0)
    # draw order a, c, b: b is the third block of the stream
    for i in (0, 31):
        assert c[i] == stream_uniform(key, n + i, *ABCModel._default_bounds["c"])
        assert b[i] == stream_uniform(key, 2 * n + i, 0, 1 - a[i])


@pytest.mark.gpu
def test_device_sampler_equals_host_population_bit_for_bit():
    import torch
    from rrmpg_amd import device
    import rrmpg_amd.models as M
    for cls in (M.ABCModel, M.HBVEdu, M.GR4J, M.Cemaneige, M.CemaneigeGR4J,
                M.CemaneigeHystGR4J, M.CemaneigeGR4JIce,
                M.CemaneigeHystGR4JIce):
        model = cls()
        for n, key in ((1, 3), (1001, 2**63 + 11)):
            pop = device.host_population(model, n, key)
            got = device.sample_params(model, n, key).cpu().numpy()
            assert got.shape == (n, len(model._param_list))
            assert np.array_equal(got, pop), cls.__name__
    # shards of one population, drawn independently
    model = M.HBVEdu()
    pop = device.host_population(model, 5000, 9)
    parts = [device.sample_params(model, hi - lo, 9, n_total=5000, first=lo)
             for lo, hi in ((0, 1700), (1700, 1701), (1701, 5000))]
    assert np.array_equal(torch.cat(parts).cpu().numpy(), pop)


@pytest.mark.gpu
def test_sampled_block_feeds_the_ensemble():
    """sample in HBM -> sweep -> scores, no host parameter array at all;
    identical to uploading the host population."""
    import torch
    from rrmpg_amd import device
    from rrmpg_amd.models import HBVEdu
    from rrmpg_amd.utils import synthetic as syn
    f = syn.make_forcing(900)
    ens = device.HBVEduEnsemble(f["temp"], f["prec"], f["month"], f["PE_m"],
                                f["T_m"], **syn.HBV_INITS)
    n = 640
    params = device.sample_params(HBVEdu(), n, 123)
    q = ens.new_output(n)
    ens.run(params, q)
    q2 = ens.new_output(n)
    host = device.host_population(HBVEdu(), n, 123)
    ens.run(torch.from_numpy(host).cuda(), q2)
    torch.cuda.synchronize()
    assert torch.equal(q, q2)


@pytest.mark.gpu
def test_sampler_argument_errors():
    import ctypes
    from rrmpg_amd import _lib
    lib = _lib.load()
    _lib.require_gpu()
    lo = (ctypes.c_double * 3)(0, 0, 0)
    hi = (ctypes.c_double * 3)(1, 1, 1)
    bad = (ctypes.c_int * 3)(0, 0, 1)
    assert lib.rr_sample_params_dev(1, 3, lo, hi, bad, 0, 4, 0, 4, None,
                                    None) == -1          # RR_E_NULL
    assert lib.rr_sample_params_dev(1, 17, lo, hi, None, 0, 4, 0, 4, None,
                                    None) == -2          # RR_E_SIZE
    assert lib.rr_sample_params_dev(1, 3, lo, hi, None, 0, 4, 2, 4, None,
                                    None) == -2
    import torch
    out = torch.empty((4, 3), dtype=torch.float64, device="cuda")
    assert lib.rr_sample_params_dev(1, 3, lo, hi, bad, 0, 4, 0, 4,
                                    out.data_ptr(), None) == -4   # RR_E_PARAM
    assert b"permutation" in lib.rr_last_error()
    assert lib.rr_sample_params_dev(1, 3, lo, hi, None, 0, 4, 0, 0, None,
                                    None) == 0           # empty shard
