"""GPU tests at BASELINE.json's sizes through the HBM-resident *_dev entry
points: size-independent properties plus oracle spot checks.

configs[0] ABC 1 set x 10 yr          -> bit-exact vs oracle
configs[1] HBV-Edu 100k sets x 30 yr  -> column permutation invariance,
                                         block-split invariance (ld > N),
                                         fused SSE == SSE of the written qsim,
                                         64 random columns vs the oracle
configs[2] GR4J 1M sets x 30 yr       -> score-only sweep; block-split
                                         invariance; random columns vs oracle
configs[3] CemaneigeGR4J, one GPU's shard (125k sets) of the 1M-set sweep,
           per-set NSE                 -> random columns vs oracle
"""

import numpy as np
import pytest

from .conftest import rel_err, snow_same

pytestmark = pytest.mark.gpu

RTOL = 1e-10


@pytest.fixture(scope="module")
def env():
    import torch
    from rrmpg_amd import _lib, device, models
    from rrmpg_amd.utils import synthetic as syn
    _lib.load()
    _lib.require_gpu()
    return dict(torch=torch, device=device, models=models, syn=syn,
                f=syn.make_forcing(syn.T_30YR))


def _flat(p, cls):
    return np.stack([p[n] for n in cls._param_list], axis=1)


def test_config0_abc_single_set(env, oracle):
    syn, torch = env["syn"], env["torch"]
    f = env["f"]
    prec = f["prec"][:syn.T_10YR]
    ens = env["device"].ABCEnsemble(prec, 1.5)
    flat = np.array([[0.3, 0.2, 0.1]])
    q, s = ens.new_output(1), ens.new_output(1)
    ens.run(ens.upload_params(flat), q, s)
    ref = oracle.simulate_abc(prec, 1.5, flat, return_storage=True)
    assert np.array_equal(q.cpu().numpy(), ref[0])
    assert np.array_equal(s.cpu().numpy(), ref[1])


def test_config1_hbv_100k_properties(env, oracle):
    torch, syn, f = env["torch"], env["syn"], env["f"]
    HBV = env["models"].HBVEdu
    n, t = 100_000, syn.T_30YR
    np.random.seed(1)
    flat = _flat(HBV().get_random_params(n), HBV)
    ens = env["device"].HBVEduEnsemble(f["temp"], f["prec"], f["month"],
                                       f["PE_m"], f["T_m"], **syn.HBV_INITS)
    params = ens.upload_params(flat)
    qsim = ens.new_output(n)
    inits = [syn.HBV_INITS[k] for k in ("snow_init", "soil_init", "s1_init",
                                        "s2_init")]
    truth = oracle.simulate_hbvedu(f["temp"], f["prec"], f["month"] - 1,
                                   f["PE_m"], f["T_m"], inits, flat[:1])
    qobs = torch.from_numpy(syn.make_qobs(truth)).cuda()
    sse = ens.run(params, qsim, qobs=qobs)
    torch.cuda.synchronize()
    assert qsim.shape == (t, n)
    assert bool((qsim[0] == 0).all())                   # quirk Q3
    assert bool(torch.isfinite(qsim).all())
    # (a) fused SSE == SSE recomputed from the qsim that was written
    sse2 = ((qobs[:, None] - qsim) ** 2).sum(0)
    assert float(((sse - sse2).abs() / sse2).max()) < 1e-12
    # (b) 64 random columns against the oracle
    rng = np.random.default_rng(0)
    cols = np.sort(rng.choice(n, 64, replace=False))
    ref = oracle.simulate_hbvedu(f["temp"], f["prec"], f["month"] - 1,
                                 f["PE_m"], f["T_m"], inits, flat[cols],
                                 nthreads=8)
    got = qsim[:, torch.from_numpy(cols).cuda()].cpu().numpy()
    assert rel_err(got, ref) < RTOL
    # (c) permuting the parameter sets permutes the columns, bit for bit
    perm = torch.randperm(n, device="cuda")
    q2 = ens.new_output(n)
    sse_p = ens.run(params[perm].contiguous(), q2, qobs=qobs)
    torch.cuda.synchronize()
    assert torch.equal(q2, qsim[:, perm])
    assert torch.equal(sse_p, sse[perm])
    # (d) two column blocks written into one [T, N] array through ld > N give
    # the same array as one launch (how several GPUs / host blocks assemble)
    q3 = ens.new_output(n)
    h = 37_001                                       # odd split point
    left = q3[:, :h]
    right = q3[:, h:]
    assert left.stride(0) == n and right.stride(0) == n
    ens.run(params[:h].contiguous(), left)
    ens.run(params[h:].contiguous(), right)
    torch.cuda.synchronize()
    assert torch.equal(q3, qsim)
    # (e) score-only mode gives the same scores
    sse_only = ens.run(params, None, qobs=qobs)
    torch.cuda.synchronize()
    assert torch.equal(sse_only, sse)
    # (f) all four storages, 5k sets: final states vs oracle
    m = 5_000
    st = tuple(ens.new_output(m) for _ in range(4))
    qm = ens.new_output(m)
    ens.run(params[:m].contiguous(), qm, st)
    torch.cuda.synchronize()
    assert torch.equal(qm, qsim[:, :m])
    ref = oracle.simulate_hbvedu(f["temp"], f["prec"], f["month"] - 1,
                                 f["PE_m"], f["T_m"], inits, flat[:32],
                                 return_storage=True)
    for a, b in zip(st, ref[1:]):
        assert rel_err(a[:, :32].cpu().numpy(), b) < RTOL


def test_metric_config_hbv_1m(env, oracle):
    """The workload BASELINE.json's metric is quoted on (bench.py default):
    HBV-Edu, 1,000,000 sets x 10,957 days, qsim[T,N] written + fused SSE --
    the kernel variant bench.py times (one scalar load per day), checked
    against the oracle on 64 random columns, and bit for bit against the
    next-day-prefetch variant the small fixtures run."""
    from rrmpg_amd import _lib
    torch, syn, f = env["torch"], env["syn"], env["f"]
    HBV = env["models"].HBVEdu
    n, t = 1_000_000, syn.T_30YR
    ens = env["device"].HBVEduEnsemble(f["temp"], f["prec"], f["month"],
                                       f["PE_m"], f["T_m"], **syn.HBV_INITS)
    params = env["device"].sample_params(HBV(), n, syn.FORCING_SEED)
    inits = [syn.HBV_INITS[k] for k in ("snow_init", "soil_init", "s1_init",
                                        "s2_init")]
    rng = np.random.default_rng(2)
    cols = np.sort(rng.choice(n, 64, replace=False))
    cols[0], cols[-1] = 0, n - 1                 # first and last (tail wave)
    tcols = torch.from_numpy(cols).cuda()
    flat = params[tcols].cpu().numpy()
    ref = oracle.simulate_hbvedu(f["temp"], f["prec"], f["month"] - 1,
                                 f["PE_m"], f["T_m"], inits, flat, nthreads=8)
    qobs_h = syn.make_qobs(ref[:, :1])
    qobs = torch.from_numpy(qobs_h).cuda()
    assert _lib.load().rr_debug_get_option(_lib.OPTIONS["hbv_variant"]) == -1
    qsim = ens.new_output(n)
    sse = ens.run(params, qsim, qobs=qobs)       # heuristic: variant 0 here
    torch.cuda.synchronize()
    got = qsim[:, tcols].cpu().numpy()
    assert rel_err(got, ref) < RTOL
    ref_sse = ((qobs_h[:, None] - ref) ** 2).sum(0)
    assert np.max(np.abs(sse[tcols].cpu().numpy() - ref_sse) / ref_sse) < 1e-10
    assert bool((qsim[0] == 0).all()) and bool(torch.isfinite(sse).all())
    with _lib.debug_option("hbv_variant", 0):
        q0 = ens.new_output(n)
        sse0 = ens.run(params, q0, qobs=qobs)
        torch.cuda.synchronize()
        assert torch.equal(q0, qsim) and torch.equal(sse0, sse)
        del q0
        # all four storages on this variant (5,000 sets): vs oracle
        m = 5_000
        st = tuple(ens.new_output(m) for _ in range(4))
        qm = ens.new_output(m)
        ens.run(params[:m].contiguous(), qm, st)
        torch.cuda.synchronize()
        assert torch.equal(qm, qsim[:, :m])
        refs = oracle.simulate_hbvedu(
            f["temp"], f["prec"], f["month"] - 1, f["PE_m"], f["T_m"], inits,
            params[:32].cpu().numpy(), return_storage=True)
        for a, b in zip(st, refs[1:]):
            assert rel_err(a[:, :32].cpu().numpy(), b) < RTOL
        assert np.array_equal(st[0][:, :32].cpu().numpy(), refs[1])  # snow
        del st, qm
    with _lib.debug_option("hbv_variant", 3):
        q2 = ens.new_output(n)
        sse2 = ens.run(params, q2, qobs=qobs)
        torch.cuda.synchronize()
        assert torch.equal(q2, qsim) and torch.equal(sse2, sse)
        del q2
    # one GPU's shard of the million-set sweep on eight GPUs, and the sizes
    # either side of the kernel-variant thresholds: same columns, bit for bit
    for m in (124_999, 65_536, 65_537, 131_073, 655_361):
        qs = ens.new_output(m)
        ss = ens.run(params[:m].contiguous(), qs, qobs=qobs)
        torch.cuda.synchronize()
        assert torch.equal(qs, qsim[:, :m]) and torch.equal(ss, sse[:m]), m
        del qs


def test_config2_gr4j_1m_scores(env, oracle):
    torch, syn, f = env["torch"], env["syn"], env["f"]
    GR4J = env["models"].GR4J
    n = 1_000_000
    np.random.seed(1)
    flat = _flat(GR4J().get_random_params(n), GR4J)
    ens = env["device"].GR4JEnsemble(f["prec"], f["etp"], **syn.GR4J_INITS)
    params = ens.upload_params(flat)
    truth = oracle.simulate_gr4j(f["prec"], f["etp"], (0.6, 0.7), flat[:1])
    qobs_h = syn.make_qobs(truth)
    qobs = torch.from_numpy(qobs_h).cuda()
    sse = ens.run(params, None, qobs=qobs)
    torch.cuda.synchronize()
    assert sse.shape == (n,) and bool(torch.isfinite(sse).all())
    # block-split invariance of the scores
    a, b = 123_457, 123_457 + 4_099
    part = ens.run(params[a:b].contiguous(), None, qobs=qobs)
    torch.cuda.synchronize()
    assert torch.equal(part, sse[a:b])
    # random columns vs oracle (scores and series)
    rng = np.random.default_rng(1)
    cols = np.sort(rng.choice(n, 32, replace=False))
    ref = oracle.simulate_gr4j(f["prec"], f["etp"], (0.6, 0.7), flat[cols],
                               nthreads=8)
    ref_sse = ((qobs_h[:, None] - ref) ** 2).sum(0)
    got = sse[torch.from_numpy(cols).cuda()].cpu().numpy()
    assert rel_err(got, ref_sse) < RTOL
    q = ens.new_output(32)
    ens.run(ens.upload_params(flat[cols]), q)
    assert rel_err(q.cpu().numpy(), ref) < RTOL
    # the 1M-set qsim mode (what the bench's extra config and the profiles
    # time): 64 columns of the resident 87.7-GB array, first and last
    # included, against the oracle; its fused sums equal the score-only
    # sweep's bit for bit; and the kernel variants agree on them
    from rrmpg_amd import _lib
    qsim = ens.new_output(n)
    sse_q = ens.run(params, qsim, qobs=qobs)
    ens.check()
    assert torch.equal(sse_q, sse)
    cols = np.unique(np.concatenate([[0, n - 1], rng.choice(n, 62,
                                                            replace=False)]))
    ref = oracle.simulate_gr4j(f["prec"], f["etp"], (0.6, 0.7), flat[cols],
                               nthreads=8)
    tcols = torch.from_numpy(cols).cuda()
    got = qsim[:, tcols].cpu().numpy()
    assert rel_err(got, ref) < RTOL
    for v in (1,):
        with _lib.debug_option("gr4j_variant", v):
            qsim.fill_(-1.0)
            sse_v = ens.run(params, qsim, qobs=qobs)
            torch.cuda.synchronize()
        assert torch.equal(sse_v, sse), v
        assert np.array_equal(qsim[:, tcols].cpu().numpy(), got), v
    del qsim


def test_config3_cemaneigegr4j_shard_nse(env, oracle, fused_variant):
    torch, syn, f = env["torch"], env["syn"], env["f"]
    from rrmpg_amd.models.cemaneige import prepare_snow_inputs
    from rrmpg_amd.sharding import shard_bounds
    from rrmpg_amd.utils.metrics import calc_nse, nse_from_sse
    CG = env["models"].CemaneigeGR4J
    total = 1_000_000
    a, b = shard_bounds(total, 8, 3)                # the shard of GPU 3 of 8
    n = b - a
    assert n == 125_000
    np.random.seed(1)
    flat_all = _flat(CG().get_random_params(total), CG)
    flat = flat_all[a:b]
    layers, inits = prepare_snow_inputs(
        f["prec"], f["temp"], f["tmin"], f["tmax"], syn.STATION_HEIGHT, 0., 0.,
        list(syn.ALTITUDES), etp=f["etp"])
    ens = env["device"].CemaneigeGR4JEnsemble(*layers, 0., 0., 0.6, 0.7)
    params = ens.upload_params(flat)
    truth = oracle.simulate_cemaneigegr4j(layers[0], layers[1], layers[3],
                                          layers[2], (0., 0., 0.6, 0.7),
                                          flat_all[:1])
    qobs_h = syn.make_qobs(truth)
    qobs = torch.from_numpy(qobs_h).cuda()
    sse = ens.run(params, None, qobs=qobs)
    torch.cuda.synchronize()
    nse = nse_from_sse(sse.cpu().numpy(), qobs_h)
    assert nse.shape == (n,) and np.isfinite(nse).all()
    rng = np.random.default_rng(2)
    cols = np.sort(rng.choice(n, 16, replace=False))
    ref = oracle.simulate_cemaneigegr4j(layers[0], layers[1], layers[3],
                                        layers[2], (0., 0., 0.6, 0.7),
                                        flat[cols], nthreads=8)
    for j, c in enumerate(cols):
        want = calc_nse(qobs_h, ref[:, j])
        assert abs(nse[c] - want) <= 1e-10 * max(1.0, abs(want))


def test_config4_multi_catchment_hbv(env, oracle):
    """BASELINE configs[4], one GPU's share: 125 catchments x 10,000 sets x
    10,957 days in ONE launch (score-only at full size); a 6-catchment subset
    with qsim must equal six single-catchment launches bit for bit, and
    random columns must match the oracle."""
    torch, syn = env["torch"], env["syn"]
    HBV = env["models"].HBVEdu
    dev = env["device"]
    C, n, t = 125, 10_000, syn.T_30YR
    fs = [syn.make_forcing(t, seed=syn.FORCING_SEED + c) for c in range(C)]
    temp = np.stack([f["temp"] for f in fs])
    prec = np.stack([f["prec"] for f in fs])
    month = np.stack([f["month"] for f in fs])
    rng = np.random.default_rng(5)
    PE_m = np.stack([syn.PE_M * rng.uniform(0.8, 1.2) for _ in range(C)])
    T_m = np.stack([syn.T_M + rng.uniform(-2, 2) for _ in range(C)])
    inits = np.stack([[0., 100. + c % 7, 3., 10.] for c in range(C)])
    np.random.seed(1)
    flat = _flat(HBV().get_random_params(C * n), HBV).reshape(C, n, 11)
    qobs_h = np.stack([syn.make_qobs(oracle.simulate_hbvedu(
        temp[c], prec[c], month[c] - 1, PE_m[c], T_m[c], inits[c],
        flat[c, :1])) for c in range(C)])
    ens = dev.HBVEduCatchments(temp, prec, month, PE_m, T_m, inits)
    params = torch.from_numpy(flat).cuda()
    qobs = torch.from_numpy(qobs_h).cuda()
    sse = ens.run(params, None, qobs=qobs)
    torch.cuda.synchronize()
    assert sse.shape == (C, n) and bool(torch.isfinite(sse).all())
    # subset with qsim == single-catchment launches, bit for bit
    sub = [0, 1, 17, 63, 99, 124]
    ens6 = dev.HBVEduCatchments(temp[sub], prec[sub], month[sub], PE_m[sub],
                                T_m[sub], inits[sub])
    q6 = ens6.new_output(n)
    sse6 = ens6.run(params[sub].contiguous(), q6, qobs=qobs[sub].contiguous())
    torch.cuda.synchronize()
    for j, c in enumerate(sub):
        one = dev.HBVEduEnsemble(temp[c], prec[c], month[c], PE_m[c], T_m[c],
                                 *inits[c])
        q1 = one.new_output(n)
        s1 = one.run(params[c].contiguous(), q1, qobs=qobs[c].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(q1, q6[j])
        assert torch.equal(s1, sse6[j])
        assert torch.equal(s1, sse[c])
        cols = np.sort(rng.choice(n, 8, replace=False))
        ref = oracle.simulate_hbvedu(temp[c], prec[c], month[c] - 1, PE_m[c],
                                     T_m[c], inits[c], flat[c, cols])
        got = q1[:, torch.from_numpy(cols).cuda()].cpu().numpy()
        assert rel_err(got, ref) < RTOL


def test_column_scores_match_metric_functions(env, oracle):
    """rr_column_sums_dev + scores_from_sums == calc_mse / rmse / nse / kge /
    alpha / beta / r evaluated column by column (reference:
    rrmpg/utils/metrics.py:29-299), here for HBV-Edu 20k sets x 30 yr."""
    torch, syn, f = env["torch"], env["syn"], env["f"]
    from rrmpg_amd.utils import metrics as M
    HBV = env["models"].HBVEdu
    n = 20_000
    np.random.seed(4)
    flat = _flat(HBV().get_random_params(n), HBV)
    ens = env["device"].HBVEduEnsemble(f["temp"], f["prec"], f["month"],
                                       f["PE_m"], f["T_m"], **syn.HBV_INITS)
    qsim = ens.new_output(n)
    params = ens.upload_params(flat)
    ens.run(params, qsim)
    torch.cuda.synchronize()
    qobs_h = syn.make_qobs(qsim[:, 0].cpu().numpy())
    qobs = torch.from_numpy(qobs_h).cuda()
    sums = env["device"].column_sums(qsim, qobs)
    sse = ens.run(params, None, qobs=qobs)
    torch.cuda.synchronize()
    assert float(((sums[:, 3] - sse).abs() / sse).max()) < 1e-12
    sc = M.scores_from_sums(sums.cpu().numpy(), qobs_h)
    rng = np.random.default_rng(9)
    for c in rng.choice(n, 12, replace=False):
        q = qsim[:, c].cpu().numpy()
        for name, fn in [("mse", M.calc_mse), ("rmse", M.calc_rmse),
                         ("nse", M.calc_nse), ("kge", M.calc_kge),
                         ("alpha", M.calc_alpha_nse),
                         ("beta", M.calc_beta_nse)]:
            want = fn(qobs_h, q)
            assert abs(sc[name][c] - want) <= 1e-9 * max(1.0, abs(want)), name
        assert abs(sc["r"][c] - M.calc_r(qobs_h, q)[0]) < 1e-9
    # a column view of a wider array (ld > N)
    part = env["device"].column_sums(qsim[:, 100:164], qobs)
    assert torch.equal(part, sums[100:164])
    # moments about mean(obs): the same scores, and still the right ones for
    # a series that is large and nearly constant (about 0 the one-pass
    # variance loses every digit there)
    shift = float(qobs_h.mean())
    sums_c = env["device"].column_sums(qsim, qobs, shift)
    sc_c = M.scores_from_sums(sums_c.cpu().numpy(), qobs_h, shift=shift)
    for name in M.ALL_SCORES:
        assert np.allclose(sc_c[name], sc[name], rtol=1e-9, atol=1e-9), name
    big = qsim[:, :64] * 1e-3 + 1e3
    big_obs_h = qobs_h * 1e-3 + 1e3
    big_obs = torch.from_numpy(big_obs_h).cuda()
    shift = float(big_obs_h.mean())
    sc_b = M.scores_from_sums(
        env["device"].column_sums(big, big_obs, shift).cpu().numpy(),
        big_obs_h, only=("kge", "alpha", "r"), shift=shift)
    for c in (0, 17, 63):
        q = big[:, c].cpu().numpy()
        assert abs(sc_b["kge"][c] - M.calc_kge(big_obs_h, q)) < 1e-8
        assert abs(sc_b["alpha"][c] - M.calc_alpha_nse(big_obs_h, q)) < 1e-8
        assert abs(sc_b["r"][c] - M.calc_r(big_obs_h, q)[0]) < 1e-8


def test_column_blocks_with_3d_storages(env, oracle):
    """ld > N with the [T][L][ld] storages: two launches fill the two column
    blocks of one set of wide arrays; result == one launch == oracle (snow
    states bit for bit)."""
    torch, syn, f = env["torch"], env["syn"], env["f"]
    from rrmpg_amd.models.cemaneige import prepare_snow_inputs
    CG = env["models"].CemaneigeGR4J
    t = 1200
    layers, _ = prepare_snow_inputs(
        f["prec"][:t], f["temp"][:t] - 4, f["tmin"][:t] - 4, f["tmax"][:t] - 4,
        syn.STATION_HEIGHT, 0., 0., list(syn.ALTITUDES), etp=f["etp"][:t])
    n, h = 517, 200
    np.random.seed(8)
    flat = _flat(CG().get_random_params(n), CG)
    ens = env["device"].CemaneigeGR4JEnsemble(*layers, 2.0, -0.1, 0.6, 0.7)
    params = ens.upload_params(flat)

    def outputs():
        return (ens.new_output(n), ens.new_output(n, 5), ens.new_output(n, 5),
                ens.new_output(n), ens.new_output(n))
    q, G, E, S, R = outputs()
    ens.run(params, q, (G, E, S, R))
    q2, G2, E2, S2, R2 = outputs()
    ens.run(params[:h].contiguous(), q2[:, :h],
            (G2[:, :, :h], E2[:, :, :h], S2[:, :h], R2[:, :h]))
    ens.run(params[h:].contiguous(), q2[:, h:],
            (G2[:, :, h:], E2[:, :, h:], S2[:, h:], R2[:, h:]))
    torch.cuda.synchronize()
    for a, b in ((q, q2), (G, G2), (E, E2), (S, S2), (R, R2)):
        assert torch.equal(a, b)
    ref = oracle.simulate_cemaneigegr4j(layers[0], layers[1], layers[3],
                                        layers[2], (2.0, -0.1, 0.6, 0.7), flat,
                                        return_storages=True, nthreads=8)
    snow_same(G.cpu().numpy(), ref[1])
    snow_same(E.cpu().numpy(), ref[2], exact=True)
    assert rel_err(q.cpu().numpy(), ref[0]) < RTOL
    # snow routine alone, same exercise
    C = env["models"].Cemaneige
    ens2 = env["device"].CemaneigeEnsemble(layers[0], layers[1], layers[2],
                                           2.0, -0.1)
    p2 = ens2.upload_params(flat[:, :2].copy())
    o = ens2.new_output(n)
    g, e = ens2.new_output(n, 5), ens2.new_output(n, 5)
    ens2.run(p2[:h].contiguous(), o[:, :h], (g[:, :, :h], e[:, :, :h]))
    ens2.run(p2[h:].contiguous(), o[:, h:], (g[:, :, h:], e[:, :, h:]))
    torch.cuda.synchronize()
    refc = oracle.simulate_cemaneige(layers[0], layers[1], layers[2],
                                     (2.0, -0.1), flat[:, :2],
                                     return_storages=True, nthreads=8)
    for k, (a, b) in enumerate(zip((o, g, e), refc)):
        snow_same(a.cpu().numpy(), b, exact=(k == 2))
