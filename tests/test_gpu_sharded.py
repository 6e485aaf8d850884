"""The parameter-set axis over several GPUs as a LIBRARY call: the in-process
fan-out of the host-pointer family (the call's rr_call_options,
RR_OPT_HOST_SHARDS, through rr_<model>_simulate_opt: one host thread and
one device context per shard, every shard filling its columns of the caller's
[T, N] arrays) behind ``monte_carlo(..., gpus=...)`` / ``sharding.sweep``, and
the HBM-resident form ``sharding.ResidentSweep`` that bench.py loops over.
On this one-GPU box the shards share the device (more shards than devices is
allowed exactly for that): what must hold is bit-equality with the single
launch, for every output shape, ragged shard sizes included."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from rrmpg_amd import _lib, device, models, sharding
    from rrmpg_amd.utils import synthetic as syn
    _lib.load()
    _lib.require_gpu()
    return dict(torch=torch, lib=_lib, device=device, models=models, syn=syn,
                sharding=sharding, f=syn.make_forcing(1500))


def _same(a, b):
    assert type(a) is type(b)
    if isinstance(a, tuple):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    else:
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shards", [2, 3, 7])
def test_host_family_fans_out_bit_identically(env, shards):
    m, f, lib = env["models"], env["f"], env["lib"]
    np.random.seed(31)
    n = 1001                                    # ragged shards
    # HBV-Edu: qsim + four storages
    p = m.HBVEdu().get_random_params(n)
    args = (f["temp"], f["prec"], f["month"], f["PE_m"], f["T_m"])
    one = m.HBVEdu().simulate(*args, return_storage=True, params=p,
                              **env["syn"].HBV_INITS)
    with lib.call_options(host_shards=shards):
        many = m.HBVEdu().simulate(*args, return_storage=True, params=p,
                                   **env["syn"].HBV_INITS)
    _same(one, many)
    # GR4J (the deferred x4 check runs per shard) and ABC
    p = m.GR4J().get_random_params(n)
    one = m.GR4J().simulate(f["prec"], f["etp"], 0.6, 0.7,
                            return_storage=True, params=p)
    with lib.call_options(host_shards=shards):
        many = m.GR4J().simulate(f["prec"], f["etp"], 0.6, 0.7,
                                 return_storage=True, params=p)
        bad = p.copy()
        bad["x4"][n - 2] = -1.0                 # lands in the last shard
        with pytest.raises(RuntimeError, match="RR_E_PARAM"):
            m.GR4J().simulate(f["prec"], f["etp"], params=bad)
    _same(one, many)
    p = m.ABCModel().get_random_params(n)
    one = m.ABCModel().simulate(f["prec"], 2.0, return_storage=True, params=p)
    with lib.call_options(host_shards=shards):
        many = m.ABCModel().simulate(f["prec"], 2.0, return_storage=True,
                                     params=p)
    _same(one, many)
    # CemaneigeGR4J: [T, L, N] storages, column blocks inside every shard
    p = m.CemaneigeGR4J().get_random_params(300)
    kw = dict(prec=f["prec"], mean_temp=f["temp"], min_temp=f["tmin"],
              max_temp=f["tmax"], etp=f["etp"],
              met_station_height=env["syn"].STATION_HEIGHT,
              altitudes=list(env["syn"].ALTITUDES), s_init=0.6, r_init=0.7,
              return_storages=True, params=p)
    one = m.CemaneigeGR4J().simulate(**kw)
    with lib.call_options(host_shards=shards, max_block_cols=64):
        many = m.CemaneigeGR4J().simulate(**kw)
    _same(one, many)


def test_monte_carlo_gpus_and_sharding_sweep(env):
    from rrmpg_amd.tools import monte_carlo
    from rrmpg_amd.utils.metrics import calc_mse, calc_nse
    m, f, sh = env["models"], env["f"], env["sharding"]
    kw = dict(temp=f["temp"], prec=f["prec"], month=f["month"],
              PE_m=f["PE_m"], T_m=f["T_m"], **env["syn"].HBV_INITS)
    np.random.seed(7)
    base = monte_carlo(m.HBVEdu(), 500, qobs=None, **kw)
    qobs = base["qsim"][:, 3] * 1.05 + 0.02
    np.random.seed(7)
    one = monte_carlo(m.HBVEdu(), 500, qobs=qobs, score="nse", **kw)
    np.random.seed(7)
    two = monte_carlo(m.HBVEdu(), 500, qobs=qobs, gpus=2, score="nse", **kw)
    np.random.seed(7)
    allg = monte_carlo(m.HBVEdu(), 500, qobs=qobs, gpus="all",
                       return_qsim=False, **kw)
    for key in ("qsim", "mse", "nse"):
        assert np.array_equal(one[key], two[key]), key
    assert np.array_equal(allg["mse"], one["mse"]) and "qsim" not in allg
    for j in (0, 250, 499):
        assert abs(one["mse"][j] - calc_mse(qobs, one["qsim"][:, j])) < 1e-12
        assert abs(one["nse"][j] - calc_nse(qobs, one["qsim"][:, j])) < 1e-12
    # the same through sharding.sweep (no process group: in-process fan-out)
    out = sh.sweep(m.HBVEdu(), one["params"], qobs, score="nse", gpus=3, **kw)
    assert out["bounds"] == (0, 500) and out["score"] == "nse"
    assert np.array_equal(out["scores"], one["nse"])
    with pytest.raises(ValueError):
        monte_carlo(m.HBVEdu(), 10, qobs=qobs, score="kge", **kw)
    for bad in (-3, 0, 2.5, "some"):
        with pytest.raises(ValueError):
            monte_carlo(m.HBVEdu(), 10, qobs=qobs, gpus=bad, **kw)


def test_monte_carlo_over_eight_shards_with_qsim(env):
    """monte_carlo(gpus=8) as an eight-GPU node would run it -- eight shards
    of a ragged total inside the one host-pointer call, all on this box's one
    device --: discharge columns and scores of the single launch."""
    from rrmpg_amd.tools import monte_carlo
    m, f = env["models"], env["f"]
    kw = dict(temp=f["temp"], prec=f["prec"], month=f["month"],
              PE_m=f["PE_m"], T_m=f["T_m"], **env["syn"].HBV_INITS)
    n = 20011
    np.random.seed(11)
    base = monte_carlo(m.HBVEdu(), 8, qobs=None, **kw)
    qobs = base["qsim"][:, 5] * 0.97 + 0.01
    np.random.seed(12)
    one = monte_carlo(m.HBVEdu(), n, qobs=qobs, score="nse", **kw)
    np.random.seed(12)
    eight = monte_carlo(m.HBVEdu(), n, qobs=qobs, gpus=8, score="nse",
                        return_qsim=True, **kw)
    assert eight["qsim"].shape == (len(qobs), n)
    for key in ("qsim", "mse", "nse"):
        assert np.array_equal(one[key], eight[key]), key
    assert np.array_equal(one["params"], eight["params"])


def test_monte_carlo_device_sampler(env):
    """monte_carlo(sampler='device'): the sets are drawn in HBM (numpy's
    Philox stream under `seed`) and scored against the resident forcing; the
    scores are those of the same population run through the default path,
    'params' downloads on first access, and without a seed the key comes
    from numpy's global generator (np.random.seed still fixes the sweep)."""
    from rrmpg_amd.tools import monte_carlo
    from rrmpg_amd.tools.monte_carlo import DeviceParams
    m, f, dev = env["models"], env["f"], env["device"]
    n = 3001
    for cls, kw in (
            (m.HBVEdu, dict(temp=f["temp"], prec=f["prec"], month=f["month"],
                            PE_m=f["PE_m"], T_m=f["T_m"],
                            **env["syn"].HBV_INITS)),
            (m.GR4J, dict(prec=f["prec"], etp=f["etp"], s_init=0.6,
                          r_init=0.7)),
            (m.ABCModel, dict(prec=f["prec"], initial_state=2.0)),
            (m.CemaneigeGR4J, dict(
                prec=f["prec"], mean_temp=f["temp"], min_temp=f["tmin"],
                max_temp=f["tmax"], etp=f["etp"],
                met_station_height=env["syn"].STATION_HEIGHT,
                altitudes=list(env["syn"].ALTITUDES), s_init=0.6,
                r_init=0.7))):
        model = cls()
        pop = dev.host_population(model, n, 77)
        rec = np.zeros(n, dtype=model._dtype)
        for j, name in enumerate(model._param_list):
            rec[name] = pop[:, j]
        base = model.simulate(params=rec[:8], **kw)
        qobs = np.asarray(base)[:, 3] * 0.95 + 0.02
        qs, sse = model._sweep(rec, qobs, False, **kw)
        want = sse / len(qobs)
        got = monte_carlo(model, n, qobs=qobs, return_qsim=False,
                          sampler="device", seed=77, score="nse", **kw)
        assert isinstance(got["params"], DeviceParams)
        assert np.array_equal(got["mse"], want)
        assert len(got["params"]) == n and got["params"].dtype == model._dtype
        assert np.array_equal(np.asarray(got["params"]), rec)
        assert np.array_equal(got["params"]["%s" % model._param_list[0]],
                              rec[model._param_list[0]])
        assert got["nse"].shape == (n,)
    np.random.seed(3)
    a = monte_carlo(model, 500, qobs=qobs, return_qsim=False,
                    sampler="device", **kw)
    np.random.seed(3)
    b = monte_carlo(model, 500, qobs=qobs, return_qsim=False,
                    sampler="device", **kw)
    assert np.array_equal(a["mse"], b["mse"])
    for bad in (dict(return_qsim=True), dict(gpus=0, return_qsim=False)):
        with pytest.raises(ValueError):
            monte_carlo(model, 10, qobs=qobs, sampler="device", **bad, **kw)
    with pytest.raises(ValueError):
        monte_carlo(model, 10, qobs=qobs, sampler="gpu", **kw)


def _fused_kw(env):
    f = env["f"]
    return dict(prec=f["prec"], mean_temp=f["temp"], min_temp=f["tmin"],
                max_temp=f["tmax"], etp=f["etp"],
                met_station_height=env["syn"].STATION_HEIGHT,
                altitudes=list(env["syn"].ALTITUDES), s_init=0.6, r_init=0.7)


@pytest.mark.parametrize("model_name", ["HBVEdu", "CemaneigeGR4J"])
def test_monte_carlo_device_sampler_over_eight_gpus_in_one_call(env,
                                                                model_name):
    """BASELINE configs[3] as a user writes it (reference seam:
    rrmpg/tools/monte_carlo.py:47-76): ``monte_carlo(model, 1_000_003, qobs,
    return_qsim=False, score='nse', sampler='device', seed=s, gpus=8)`` --
    every shard draws ITS rows of the one Philox population in HBM, sweeps
    them score-only against its GPU's replica of the forcing and sends 8 B
    per set to the host.  On this one-GPU box the eight shards share the
    device (a stream each); the scores must equal the single sweep's bit for
    bit, ragged blocks included, and so must the parameter sets."""
    from rrmpg_amd.tools import monte_carlo
    m, f = env["models"], env["f"]
    model = getattr(m, model_name)()
    kw = (_fused_kw(env) if model_name == "CemaneigeGR4J" else
          dict(temp=f["temp"], prec=f["prec"], month=f["month"],
               PE_m=f["PE_m"], T_m=f["T_m"], **env["syn"].HBV_INITS))
    rec = model.get_random_params(4)
    qobs = np.asarray(model.simulate(params=rec, **kw))[:, 1] * 0.9 + 0.05
    n = 1_000_003
    call = dict(qobs=qobs, return_qsim=False, score="nse", sampler="device",
                seed=20260930)
    one = monte_carlo(model, n, **call, **kw)
    eight = monte_carlo(model, n, gpus=8, **call, **kw)
    assert eight["mse"].shape == (n,) and np.isfinite(eight["mse"]).all()
    assert np.array_equal(one["mse"], eight["mse"])
    assert np.array_equal(one["nse"], eight["nse"])
    assert len(eight["params"].shards) == 8 and len(eight["params"]) == n
    sizes = [int(s.shape[0]) for s in eight["params"].shards]
    assert sizes == [125001] * 3 + [125000] * 5
    assert env["torch"].equal(one["params"].tensor, eight["params"].tensor)
    # ragged small sweeps, 'all' (= the one device here), more shards than sets
    small = monte_carlo(model, 1001, **call, **kw)
    for g in (3, 7, "all", 2000):
        got = monte_carlo(model, 1001, gpus=g, **call, **kw)
        assert np.array_equal(got["mse"], small["mse"]), g
        assert np.array_equal(np.asarray(got["params"]),
                              np.asarray(small["params"])), g
    with pytest.raises(ValueError):
        monte_carlo(model, 10, gpus=2, qobs=qobs[:-1], return_qsim=False,
                    sampler="device", **kw)


@pytest.mark.parametrize("model_name", ["ABCModel", "GR4J", "Cemaneige",
                                        "CemaneigeHystGR4J",
                                        "CemaneigeGR4JIce",
                                        "CemaneigeHystGR4JIce"])
def test_every_model_sweeps_resident_over_shards(env, model_name):
    """sampler='device' (and with it gpus=G) for every model class: the
    couplings of the next tier included, whose single sweep of 300,007 sets
    sorts its sets by hydrograph tier and runs the tiers' kernels side by
    side, while its eight shards of 37,501 keep their order -- the same
    scores, bit for bit."""
    from rrmpg_amd.tools import monte_carlo
    m, f, syn = env["models"], env["f"], env["syn"]
    model = getattr(m, model_name)()
    snow = dict(prec=f["prec"], mean_temp=f["temp"] - 3,
                min_temp=f["tmin"] - 3, max_temp=f["tmax"] - 3,
                met_station_height=syn.STATION_HEIGHT,
                altitudes=list(syn.ALTITUDES))
    if model_name == "ABCModel":
        kw = dict(prec=f["prec"], initial_state=2.0)
    elif model_name == "GR4J":
        kw = dict(prec=f["prec"], etp=f["etp"], s_init=0.6, r_init=0.7)
    elif model_name == "Cemaneige":
        kw = snow
    else:
        kw = dict(snow, etp=f["etp"], s_init=0.6, r_init=0.7)
        if "Ice" in model_name:
            kw["frac_ice"] = np.array([0.02, 0.04, 0.25, 0.51, 0.71])
    rec = model.get_random_params(3)
    base = model.simulate(params=rec, **kw)
    qobs = np.asarray(base[0] if isinstance(base, tuple) else base)[:, 1] \
        * 0.9 + 0.05
    n = 300_007
    call = dict(qobs=qobs, return_qsim=False, score="nse", sampler="device",
                seed=77)
    one = monte_carlo(model, n, **call, **kw)
    eight = monte_carlo(model, n, gpus=8, **call, **kw)
    assert np.isfinite(one["mse"]).all()
    assert np.array_equal(one["mse"], eight["mse"])
    assert np.array_equal(one["nse"], eight["nse"])
    # ... and they are the scores of the host path on the same population
    some = np.asarray(one["params"])[:500]
    _, sse = model._sweep(some, qobs, False, **kw)
    assert np.array_equal(sse / len(qobs), one["mse"][:500])


def test_ensemble_replica_shares_forcing_not_workspace(env):
    dev, f, torch = env["device"], env["f"], env["torch"]
    ens = dev.GR4JEnsemble(f["prec"], f["etp"], s_init=0.6, r_init=0.7)
    p = dev.sample_params(env["models"].GR4J(), 500, 5)
    q = torch.as_tensor(f["prec"] * 0.3, device=ens.device)
    a = ens.run(p, None, qobs=q).clone()
    twin = ens.replica()
    assert twin.prec.data_ptr() == ens.prec.data_ptr() and twin._ws is None
    b = twin.run(p, None, qobs=q)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and twin._ws.data_ptr() != ens._ws.data_ptr()


def test_concurrent_sweeps_keep_their_own_options(env):
    """The shard count travels with the CALL (rr_<model>_simulate_opt's
    rr_call_options), not through a process-wide switch: two threads running
    monte_carlo(gpus=1) and monte_carlo(gpus=3) at the same time -- and a
    third sweeping GR4J in small column blocks -- get the bits of the same
    calls made one after the other.  (The reference's seam:
    rrmpg/tools/monte_carlo.py:61-71.)"""
    import threading
    from rrmpg_amd.tools import monte_carlo
    m, f, lib = env["models"], env["f"], env["lib"]
    kw = dict(temp=f["temp"], prec=f["prec"], month=f["month"],
              PE_m=f["PE_m"], T_m=f["T_m"], **env["syn"].HBV_INITS)
    np.random.seed(11)
    pa = m.HBVEdu().get_random_params(3001)
    pb = m.HBVEdu().get_random_params(2777)
    pg = m.GR4J().get_random_params(1500)
    qobs = m.HBVEdu().simulate(f["temp"], f["prec"], f["month"], f["PE_m"],
                               f["T_m"], params=pa[:1],
                               **env["syn"].HBV_INITS)[:, 0] * 1.03

    def fixed(model, params):
        # monte_carlo draws from numpy's global stream: hand each thread its
        # own pre-drawn population instead
        model.get_random_params = lambda num=1: params[:num]
        return model

    def job_a():
        return monte_carlo(fixed(m.HBVEdu(), pa), len(pa), qobs=qobs,
                           gpus=1, score="nse", **kw)

    def job_b():
        return monte_carlo(fixed(m.HBVEdu(), pb), len(pb), qobs=qobs,
                           gpus=3, **kw)

    def job_g():
        with lib.call_options(max_block_cols=192, host_shards=2):
            return m.GR4J().simulate(f["prec"], f["etp"], 0.6, 0.7,
                                     return_storage=True, params=pg)

    want = [job_a(), job_b(), job_g()]
    # nothing process-wide was touched
    assert lib.load().rr_debug_get_option(lib.OPTIONS["host_shards"]) == 0
    assert lib.load().rr_debug_get_option(lib.OPTIONS["max_block_cols"]) == 0
    for _ in range(3):
        got, errs = [None] * 3, []

        def run(k, fn):
            try:
                got[k] = fn()
            except Exception as e:            # pragma: no cover
                errs.append(e)
        th = [threading.Thread(target=run, args=(k, fn))
              for k, fn in enumerate((job_a, job_b, job_g))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for key in ("qsim", "mse", "nse"):
            assert np.array_equal(got[0][key], want[0][key]), key
        for key in ("qsim", "mse"):
            assert np.array_equal(got[1][key], want[1][key]), key
        _same(got[2], want[2])
    # a thread's options never leak into another thread's calls
    seen = {}

    def probe():
        seen["ptr"] = lib.opts_ptr()
    with lib.call_options(host_shards=5):
        t = threading.Thread(target=probe)
        t.start()
        t.join()
        assert lib.opts_ptr() is not None
    assert seen["ptr"] is None and lib.opts_ptr() is None


def test_resident_sweep_scores(env):
    """sharding.ResidentSweep (what bench.py times): MSE and NSE of a resident
    block equal the host-path scores; the device sampler gives every rank its
    rows of ONE population (two 'ranks' drawn here by hand)."""
    torch, dev, m, f, sh = (env["torch"], env["device"], env["models"],
                            env["f"], env["sharding"])
    from rrmpg_amd.utils.metrics import calc_mse, calc_nse
    ens = dev.GR4JEnsemble(f["prec"], f["etp"], s_init=0.6, r_init=0.7)
    total, key = 999, 1234
    whole = dev.sample_params(m.GR4J(), total, key)
    q = ens.new_output(total)
    ens.run(whole, q)
    torch.cuda.synchronize()
    qobs_h = q[:, 5].cpu().numpy() * 0.9 + 0.05
    qobs = torch.from_numpy(qobs_h).cuda()
    got = {}
    for score in ("mse", "nse"):
        parts = []
        for r in range(2):
            a, b = sh.shard_bounds(total, 2, r)
            rs = sh.ResidentSweep.from_sampler(ens, m.GR4J(), b - a, total, a,
                                               key, qobs=qobs, score=score)
            assert torch.equal(rs.params, whole[a:b])
            parts.append(rs.step())        # no process group: this block's
            assert parts[-1].numel() == b - a
        got[score] = torch.cat(parts).cpu().numpy()
    qh = q.cpu().numpy()
    for j in (0, 499, 500, 998):
        assert abs(got["mse"][j] - calc_mse(qobs_h, qh[:, j])) \
            <= 1e-12 * max(1.0, calc_mse(qobs_h, qh[:, j]))
        assert abs(got["nse"][j] - calc_nse(qobs_h, qh[:, j])) <= 1e-10
