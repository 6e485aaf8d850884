"""The *_simulate_dev family is enqueue-and-return: no call synchronises its
stream (the GR4J family chooses its unit-hydrograph storage on the GPU and
reports unusable x4 values through rr_gr4j_plan_status), sweeps on two
streams overlap, and the resident ensembles reject tensors whose dtype /
device / shape / layout the raw-pointer kernels could not survive."""

import time

import numpy as np
import pytest

from .conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from rrmpg_amd import _lib, device, models
    from rrmpg_amd.utils import synthetic as syn
    _lib.load()
    _lib.require_gpu()
    return torch, device, models, syn, syn.make_forcing(syn.T_30YR)


def _ensembles(env, which):
    torch, dev, models, syn, f = env
    if which == "gr4j":
        return (dev.GR4JEnsemble(f["prec"], f["etp"], **syn.GR4J_INITS),
                models.GR4J)
    from rrmpg_amd.models.cemaneige import prepare_snow_inputs
    layers, _ = prepare_snow_inputs(f["prec"], f["temp"], f["tmin"],
                                    f["tmax"], syn.STATION_HEIGHT, 0, 0,
                                    list(syn.ALTITUDES), etp=f["etp"])
    if which == "cemaneigegr4j":
        return (dev.CemaneigeGR4JEnsemble(layers[0], layers[1], layers[2],
                                          layers[3], 0., 0., .6, .7),
                models.CemaneigeGR4J)
    return (dev.SnowGR4JEnsemble(True, False, layers[0], layers[1], layers[2],
                                 layers[3], s_init=.6, r_init=.7),
            models.CemaneigeHystGR4J)


@pytest.mark.parametrize("which", ["gr4j", "cemaneigegr4j", "hystgr4j"])
def test_gr4j_family_run_does_not_block(env, which):
    """Second call on a stream that is still busy with the first returns long
    before the first finishes (the old plan read the x4 scan back and waited
    for the stream)."""
    torch = env[0]
    ens, cls = _ensembles(env, which)
    n = 400_000
    params = env[1].sample_params(cls(), n, 99)
    qobs = torch.rand(ens.num_timesteps, dtype=torch.float64, device="cuda")
    sse = ens.run(params, None, qobs=qobs)            # warm-up / sizing
    torch.cuda.synchronize()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    e0.record()
    ens.run(params, None, qobs=qobs, sse=sse)
    e1.record()
    torch.cuda.synchronize()
    kernel_ms = e0.elapsed_time(e1)
    assert kernel_ms > 5.0
    ens.run(params, None, qobs=qobs, sse=sse)
    t0 = time.perf_counter()
    ens.run(params, None, qobs=qobs, sse=sse)         # stream busy
    host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    assert host_ms < 0.25 * kernel_ms, (host_ms, kernel_ms)
    ens.check()                                       # no complaint


def test_two_streams_overlap(env):
    """Two quarter-chip GR4J sweeps on two streams take about as long as one,
    not twice as long."""
    torch, dev, models, syn, f = env
    n = 16_384                       # 256 waves: a quarter of the SIMDs
    a = dev.GR4JEnsemble(f["prec"], f["etp"], **syn.GR4J_INITS)
    b = dev.GR4JEnsemble(f["prec"], f["etp"], **syn.GR4J_INITS)
    pa = dev.sample_params(models.GR4J(), n, 1)
    pb = dev.sample_params(models.GR4J(), n, 2)
    qa, qb = a.new_output(n), b.new_output(n)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def both(stream_b):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(sa):
            a.run(pa, qa)
        with torch.cuda.stream(stream_b):
            b.run(pb, qb)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    both(sb), both(sa)                                # warm-up
    serial = min(both(sa) for _ in range(3))
    overlapped = min(both(sb) for _ in range(3))
    assert overlapped < 0.85 * serial, (overlapped, serial)
    # and the results are those of a plain single-stream run
    q1 = a.new_output(n)
    a.run(pa, q1)
    torch.cuda.synchronize()
    assert torch.equal(q1, qa)


@pytest.mark.timeout(300)
def test_tiled_sweeps_share_the_gpu_without_deadlock(env):
    """Time-tiled sweeps draw their work items as tickets from an atomic
    counter (csrc/common.h "the time axis in pieces"), so an item only ever
    waits for one that a RUNNING wave holds -- no assumption about the order
    in which the dispatcher starts workgroups, or about who else is on the
    GPU.  Here two tiled GR4J sweeps and a tiled Cemaneige and HBV-Edu sweep
    run on four streams while a fifth holds every CU busy with a spin kernel
    (rrdbg_spin_dev) for the first 30 ms: all complete, with the bits of the
    untiled runs.  (Rounds 1-3 took the item from blockIdx and could only
    bound the wait; the launch failed after 2^26 polls.)"""
    import ctypes
    torch, dev, models, syn, f = env
    from rrmpg_amd import _lib
    from rrmpg_amd.models.cemaneige import prepare_snow_inputs
    lib = _lib.load()
    spin = lib.rrdbg_spin_dev
    spin.restype = ctypes.c_int
    spin.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
    n = 70_000                       # ~1100 waves: a bit more than one round
    layers, _ = prepare_snow_inputs(f["prec"], f["temp"] - 3, f["tmin"] - 3,
                                    f["tmax"] - 3, syn.STATION_HEIGHT, 0, 0,
                                    list(syn.ALTITUDES))
    jobs = []
    for k in range(2):
        jobs.append((dev.GR4JEnsemble(f["prec"], f["etp"], **syn.GR4J_INITS),
                     dev.sample_params(models.GR4J(), n, 10 + k)))
    jobs.append((dev.CemaneigeEnsemble(layers[0], layers[1], layers[2]),
                 dev.sample_params(models.Cemaneige(), n, 12)))
    jobs.append((dev.HBVEduEnsemble(f["temp"], f["prec"], f["month"],
                                    f["PE_m"], f["T_m"], **syn.HBV_INITS),
                 dev.sample_params(models.HBVEdu(), n, 13)))
    with _lib.debug_option("time_tiles", 0), \
            _lib.debug_option("hbv_variant", 0):
        plain = []
        for ens, p in jobs:
            q = ens.new_output(n)
            ens.run(p, q)
            plain.append(q)
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in jobs]
    hold = torch.cuda.Stream()
    for pieces in (4, 7):
        outs = [ens.new_output(n) for ens, _ in jobs]
        torch.cuda.synchronize()
        with _lib.debug_option("time_tiles", pieces), \
                _lib.debug_option("hbv_variant", 0):
            _lib.check(spin(4096, 30_000, hold.cuda_stream), "rrdbg_spin_dev")
            for (ens, p), q, st in zip(jobs, outs, streams):
                with torch.cuda.stream(st):
                    ens.run(p, q)
            _lib.check(spin(2048, 5_000, hold.cuda_stream), "rrdbg_spin_dev")
        torch.cuda.synchronize()
        for (ens, _), q, want in zip(jobs, outs, plain):
            if hasattr(ens, "check"):
                ens.check()
            assert torch.equal(q, want), (pieces, type(ens).__name__)


def test_unusable_x4_is_reported_by_check(env, oracle):
    torch, dev, models, syn, f = env
    t = 400
    ens = dev.GR4JEnsemble(f["prec"][:t], f["etp"][:t], **syn.GR4J_INITS)
    flat = np.array([[300., .5, 80., 1.7], [250., -1., 60., 2.4],
                     [400., 1., 90., 0.9]])
    good = ens.upload_params(flat)
    q = torch.full((t, 3), -7.0, dtype=torch.float64, device="cuda")
    ens.run(good, q)
    ens.check()
    ref = oracle.simulate_gr4j(f["prec"][:t], f["etp"][:t], (.6, .7), flat)
    assert rel_err(q.cpu().numpy(), ref) < 1e-10
    for bad_x4, msg in ((-0.5, "ceil"), (float("nan"), "ceil"),
                        (25.0, "unit-hydrograph scratch")):
        bad = flat.copy()
        bad[1, 3] = bad_x4
        q.fill_(-7.0)
        sse = ens.run(ens.upload_params(bad), q,      # returns, no error yet
                      qobs=torch.zeros(t, dtype=torch.float64, device="cuda"))
        with pytest.raises(RuntimeError, match="RR_E_PARAM") as ei:
            ens.check()
        assert msg in str(ei.value)
        assert bool((q == -7.0).all())                # nothing was written
        assert bool(torch.isnan(sse).all())           # and the scores say so
    ens.run(good, q)                                  # and it recovers
    ens.check()
    assert rel_err(q.cpu().numpy(), ref) < 1e-10
    # with a workspace sized for it (ens.max_x4) the long hydrograph runs,
    # as in the reference
    ens.max_x4 = 30.0
    long = flat.copy()
    long[1, 3] = 25.0
    ens.run(ens.upload_params(long), q)
    ens.check()
    ref = oracle.simulate_gr4j(f["prec"][:t], f["etp"][:t], (.6, .7), long)
    assert rel_err(q.cpu().numpy(), ref) < 1e-10
    long[2, 3] = 31.0                                 # beyond THIS workspace
    q.fill_(-7.0)
    ens.run(ens.upload_params(long), q)
    with pytest.raises(RuntimeError, match="unit-hydrograph scratch"):
        ens.check()
    assert bool((q == -7.0).all())


def test_resident_ensembles_validate_their_tensors(env):
    torch, dev, models, syn, f = env
    t = 300
    ens = dev.HBVEduEnsemble(f["temp"][:t], f["prec"][:t], f["month"][:t],
                             f["PE_m"], f["T_m"], **syn.HBV_INITS)
    p = dev.sample_params(models.HBVEdu(), 100, 5)
    q = ens.new_output(100)
    ens.run(p, q)
    with pytest.raises(TypeError, match="float64"):
        ens.run(p.float(), q)
    with pytest.raises(TypeError, match="float64"):
        ens.run(p, q.float())
    with pytest.raises(ValueError, match="shape"):
        ens.run(p, ens.new_output(99))
    with pytest.raises(ValueError, match="shape"):
        ens.run(p, q[:-1])
    with pytest.raises(ValueError, match="contiguous"):
        ens.run(p, torch.empty((100, t), dtype=torch.float64,
                               device="cuda").t())
    with pytest.raises(ValueError, match="contiguous"):
        ens.run(torch.empty((11, 100), dtype=torch.float64,
                            device="cuda").t(), q)
    with pytest.raises(ValueError, match="lives on"):
        ens.run(p, q.cpu())
    with pytest.raises(ValueError, match="shape"):
        ens.run(p, q, qobs=torch.zeros(t - 1, dtype=torch.float64,
                                       device="cuda"))
    with pytest.raises(ValueError, match="row stride"):
        wide = torch.empty((t, 228), dtype=torch.float64, device="cuda")
        ens.run(p, q, (wide[:, :100], q.clone(), q.clone(), q.clone()))
    with pytest.raises(RuntimeError, match="same size"):
        dev.HBVEduEnsemble(f["temp"][:t], f["prec"][:t - 1], f["month"][:t],
                           f["PE_m"], f["T_m"])
    with pytest.raises(RuntimeError, match="length 12"):
        dev.HBVEduEnsemble(f["temp"][:t], f["prec"][:t], f["month"][:t],
                           f["PE_m"][:11], f["T_m"])
    with pytest.raises(RuntimeError, match="same size"):
        dev.GR4JEnsemble(f["prec"][:t], f["etp"][:t + 1])
    # a column block of a wider array is fine (ld > N)
    wide = torch.zeros((t, 228), dtype=torch.float64, device="cuda")
    ens.run(p, wide[:, 28:128])
    torch.cuda.synchronize()
    assert torch.equal(wide[:, 28:128], q) and float(wide[:, :28].sum()) == 0
    # new_output's rows start on 128-byte boundaries whatever N is; a dense
    # [T][N] array takes the same values
    q99 = ens.new_output(99)
    assert q99.shape == (t, 99) and q99.stride() == (112, 1)
    assert q99.data_ptr() % 128 == 0
    dense = torch.empty((t, 99), dtype=torch.float64, device="cuda")
    ens.run(p[:99].contiguous(), q99)
    ens.run(p[:99].contiguous(), dense)
    torch.cuda.synchronize()
    assert torch.equal(q99, dense) and torch.equal(q99, q[:, :99])


def test_device_snow_layers_match_reference_preprocessing(env):
    """rr_cemaneige_layers_dev (forcing preprocessing on the GPU) against the
    reference's own cemaneige_utils outputs (golden, generated by the
    reference), including the >= 1500 m solid-fraction rule and the 4000 m
    precipitation cap -- bit for bit."""
    from .conftest import golden
    torch, dev = env[0], env[1]
    g = golden("syn_cemaneige_prep")
    series = (g["prec"], g["temp"], g["tmin"], g["tmax"])
    lp, lm, fr = dev.snow_layers(*series, float(g["station"]),
                                 list(g["altitudes"]))
    assert np.array_equal(lp.cpu().numpy(), g["layer_prec"])
    assert np.array_equal(lm.cpu().numpy(), g["layer_mean"])
    assert np.array_equal(fr.cpu().numpy(), g["frac_solid"])
    lp, lm, fr = dev.snow_layers(*series, float(g["station_hi"]),
                                 list(g["altitudes_hi"]))
    assert np.array_equal(lp.cpu().numpy(), g["layer_prec_hi"])
    assert np.array_equal(lm.cpu().numpy(), g["layer_mean_hi"])
    assert np.array_equal(fr.cpu().numpy(), g["frac_solid_hi"])
    lp, _, _ = dev.snow_layers(*(s[:64] for s in series),
                               float(g["station_vhi"]),
                               list(g["altitudes_hi"]))
    assert np.array_equal(lp.cpu().numpy(), g["layer_prec_vhi"])
    # the C library's exp instead of numpy's: the factor may differ in its
    # last bit, the product by two
    lp2, _, _ = dev.snow_layers(*series, float(g["station"]),
                                list(g["altitudes"]), numpy_exp=False)
    a, b = lp2.cpu().numpy(), g["layer_prec"]
    assert np.all(np.abs(a - b) <= 2 * np.spacing(np.abs(b)))
    # no elevation layers: one layer at station height; feeds an ensemble
    lp1, lm1, fr1 = dev.snow_layers(*series, 500.0)
    assert lp1.shape == (series[0].size, 1)
    assert np.array_equal(lp1[:, 0].cpu().numpy(), g["prec"])
    ens = dev.CemaneigeEnsemble(lp1, lm1, fr1)
    from rrmpg_amd import models
    p = dev.sample_params(models.Cemaneige(), 64, 3)
    q = ens.new_output(64)
    ens.run(p, q)
    rec = np.zeros(64, dtype=models.Cemaneige._dtype)
    for k, name in enumerate(models.Cemaneige._param_list):
        rec[name] = p[:, k].cpu().numpy()
    host = models.Cemaneige().simulate(
        g["prec"], g["temp"], g["tmin"], g["tmax"], met_station_height=500,
        params=rec)
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), host)


def test_host_family_context_and_staged_gather(env, oracle):
    """The host-pointer family's per-device context: an input whose bytes
    changed IN PLACE is uploaded again (the hash, not the pointer, decides);
    a result above 64 MiB goes through the pinned staging ring in several
    double-buffered column blocks and equals the small-block result; the
    cache can be released and is rebuilt on demand."""
    torch, dev, models, syn, f = env
    from rrmpg_amd import _lib
    t = 2000
    temp, prec = f["temp"][:t].copy(), f["prec"][:t].copy()
    kw = dict(month=f["month"][:t], PE_m=f["PE_m"], T_m=f["T_m"],
              soil_init=100., s1_init=3., s2_init=10.)
    np.random.seed(2)
    m = models.HBVEdu()
    p = m.get_random_params(6000)                # 2000 x 6000 x 8 B = 96 MB
    q1 = m.simulate(temp, prec, params=p, **kw)
    flat = np.stack([p[k] for k in m.get_parameter_names()], 1)
    cols = np.array([0, 17, 2999, 3000, 5999])
    ref = oracle.simulate_hbvedu(temp, prec, f["month"][:t] - 1, f["PE_m"],
                                 f["T_m"], (0., 100., 3., 10.), flat[cols])
    assert rel_err(q1[:, cols], ref) < 1e-10
    with _lib.debug_option("max_block_cols", 640):       # ten ragged blocks
        q2 = m.simulate(temp, prec, params=p, **kw)
    assert np.array_equal(q1, q2)
    # same buffer, new content: must not be served from the cached copy
    prec *= 1.5
    q3 = m.simulate(temp, prec, params=p[:64], **kw)
    ref3 = oracle.simulate_hbvedu(temp, prec, f["month"][:t] - 1, f["PE_m"],
                                  f["T_m"], (0., 100., 3., 10.), flat[:64])
    assert rel_err(q3, ref3) < 1e-10
    assert not np.array_equal(q3, q1[:, :64])
    assert _lib.load().rr_release_cached_memory() == 0
    q4 = m.simulate(temp, prec, params=p[:64], **kw)
    assert np.array_equal(q3, q4)
    q5 = m.simulate(temp, prec.copy(), params=p, **kw)        # ring again
    assert np.array_equal(q5[:, :64], q3)


@pytest.mark.parametrize("which", ["hbvedu", "gr4j", "hystgr4j"])
def test_dev_sweeps_capture_into_a_hip_graph(env, which):
    """Nothing in a *_simulate_dev call allocates, copies back or waits: a
    whole sweep -- pre-pass, plan, the tier kernels (for a sorted score-only
    sweep of the hysteresis coupling: on side streams forked from and joined
    to the caller's with events), the reference kernel -- captures into ONE
    HIP graph, and its replays give the bits of the plain call."""
    torch, dev, models, syn, f = env
    t = 400
    if which == "hbvedu":
        ens = dev.HBVEduEnsemble(f["temp"][:t], f["prec"][:t], f["month"][:t],
                                 f["PE_m"], f["T_m"], **syn.HBV_INITS)
        cls, n, with_q = models.HBVEdu, 70_001, True
    elif which == "gr4j":
        ens = dev.GR4JEnsemble(f["prec"][:t], f["etp"][:t], **syn.GR4J_INITS)
        cls, n, with_q = models.GR4J, 30_011, True
    else:
        from rrmpg_amd.models.cemaneige import prepare_snow_inputs
        layers, _ = prepare_snow_inputs(f["prec"][:t], f["temp"][:t],
                                        f["tmin"][:t], f["tmax"][:t],
                                        syn.STATION_HEIGHT, 0, 0,
                                        list(syn.ALTITUDES), etp=f["etp"][:t])
        ens = dev.SnowGR4JEnsemble(True, False, layers[0], layers[1],
                                   layers[2], layers[3], s_init=.6, r_init=.7)
        cls, n, with_q = models.CemaneigeHystGR4J, 4 * 1024 * 64 + 5, False
    side = torch.cuda.Stream()
    params = dev.sample_params(cls(), n, 5)
    qobs = torch.rand(t, dtype=torch.float64, device="cuda")
    q_plain = ens.new_output(n) if with_q else None
    q_graph = ens.new_output(n) if with_q else None
    sse_graph = torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        # plain call first: workspace, side streams, the comparison
        sse_plain = ens.run(params, q_plain, qobs=qobs).clone()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ens.run(params, q_graph, qobs=qobs, sse=sse_graph)
        side.synchronize()
        assert not bool(sse_graph.any())       # captured, not run
        for _ in range(3):
            sse_graph.zero_()
            g.replay()
            side.synchronize()
            assert torch.equal(sse_graph, sse_plain)
            if with_q:
                assert torch.equal(q_graph, q_plain)


def test_record_prefetch_changes_nothing_but_the_time(env):
    """RR_OPT_WARM_RECORDS (the waves read a share of the day records into
    their XCD's L2 when they start, common.h rr_warm_l2): a prefetch only --
    the same bits with it pinned on, pinned off and chosen by sweep size, for
    HBV-Edu (every loop variant's launch path), the fused CemaneigeGR4J
    kernels and GR4J (where it only runs on request)."""
    torch, dev, models, syn, f = env
    from rrmpg_amd import _lib
    t = 500
    hbv = dev.HBVEduEnsemble(f["temp"][:t], f["prec"][:t], f["month"][:t],
                             f["PE_m"], f["T_m"], **syn.HBV_INITS)
    gr = dev.GR4JEnsemble(f["prec"][:t], f["etp"][:t], **syn.GR4J_INITS)
    from rrmpg_amd.models.cemaneige import prepare_snow_inputs
    layers, _ = prepare_snow_inputs(f["prec"][:t], f["temp"][:t],
                                    f["tmin"][:t], f["tmax"][:t],
                                    syn.STATION_HEIGHT, 0, 0,
                                    list(syn.ALTITUDES), etp=f["etp"][:t])
    fused = dev.CemaneigeGR4JEnsemble(layers[0], layers[1], layers[2],
                                      layers[3], 0., 0., .6, .7)
    for ens, cls, n in ((hbv, models.HBVEdu, 1000), (hbv, models.HBVEdu, 70_001),
                        (gr, models.GR4J, 9_000),
                        (fused, models.CemaneigeGR4J, 5_003),
                        (fused, models.CemaneigeGR4J, 140_000)):
        params = dev.sample_params(cls(), n, 17)
        qobs = torch.rand(t, dtype=torch.float64, device="cuda")
        got = []
        for pin in (-1, 0, 1):
            q = ens.new_output(n)
            with _lib.debug_option("warm_records", pin):
                sse = ens.run(params, q, qobs=qobs).clone()
                sse_only = ens.run(params, None, qobs=qobs).clone()
            torch.cuda.synchronize()
            got.append((q, sse, sse_only))
        for q, sse, sse_only in got[1:]:
            assert torch.equal(q, got[0][0]), (cls.__name__, n)
            assert torch.equal(sse, got[0][1])
            assert torch.equal(sse_only, got[0][2])
        assert torch.equal(got[0][1], got[0][2])
