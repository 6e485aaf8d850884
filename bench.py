#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: model-timesteps/s of an ensemble sweep.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full Monte-Carlo sweep of the workload on every rank: all of
the rank's parameter sets x all daily timesteps, with the [T, N] discharge
array materialised in HBM AND the per-set squared error accumulated in the
kernel (what rrmpg.tools.monte_carlo computes: simulate + per-set MSE),
followed (N > 1) by the single all-gather of the per-set scores over RCCL.
Inputs (forcing, parameter block, output buffers) are resident in HBM before
the timed region starts.

Launch: one process per GPU over RCCL.  Under torchrun (WORLD_SIZE set) the
ranks are taken from the environment; started plainly with --gpus N > 1 the
script spawns its own N ranks (127.0.0.1 rendezvous) and rank 0 prints the
line.  --scaling strong (default): --sets is the size of the WHOLE sweep,
sharded into contiguous blocks of sets (rrmpg_amd.sharding.shard_bounds;
1M sets -> 125k per GPU at N=8, BASELINE.json's "HBV 1M-param MC at 1/2/4/8
MI355X"); --scaling weak: --sets per GPU.

Default workload = the one BASELINE.json's metric is quoted on: HBV-Edu,
1,000,000 parameter sets, 10,957 daily steps (30 years), fp64.
Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# Algorithmic bytes per model-timestep: 8 B, the qsim
# element, in the default mode.  Forcing (40 B/day shared by all sets) and the
# parameter block (88 B/set, read once) amortise to ~0.


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="hbvedu",
                    choices=["hbvedu", "abc", "gr4j", "cemaneige",
                             "cemaneigegr4j",
                             "cemaneigehystgr4j", "cemaneigegr4jice",
                             "cemaneigehystgr4jice"])
    ap.add_argument("--sets", type=int, default=1_000_000,
                    help="parameter sets: of the whole sweep (--scaling "
                         "strong) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: --sets is the global count, sharded over "
                         "the GPUs; weak: every GPU gets --sets")
    ap.add_argument("--days", type=int, default=10957)
    ap.add_argument("--catchments", type=int, default=0,
                    help="HBV-Edu only: C independent catchments x --sets "
                         "parameter sets each in one launch (BASELINE "
                         "configs[4]; 125 x 10000 is one GPU's share)")
    ap.add_argument("--mode", default="qsim",
                    choices=["qsim", "metric", "storages"],
                    help="qsim: materialise qsim[T,N] + fused per-set SSE "
                         "(default); metric: fused per-set SSE only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) is the product "
                         "path; gloo + --share-gpu lets several ranks share "
                         "ONE GPU to rehearse the multi-rank code path on a "
                         "single-GPU box (scores gathered on the host)")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--sampler", default="device", choices=["device", "host"],
                    help="device: every rank draws its shard of ONE global "
                         "population in HBM (rr_sample_params_dev, numpy's "
                         "Philox stream); host: Model.get_random_params + "
                         "upload, as the reference does")
    ap.add_argument("--hbv-variant", type=int, default=-1,
                    help="measurement hook: pin the HBV-Edu kernel variant "
                         "(rr_debug_set_option RR_OPT_HBV_VARIANT)")
    ap.add_argument("--row-pitch", type=int, default=0,
                    help="measurement hook: row pitch of the output arrays in "
                         "doubles is rounded up to this (default: "
                         "rrmpg_amd.device's 16 = 128 B; 1 = dense [T][N])")
    ap.add_argument("--time-tiles", type=int, default=-1,
                    help="measurement hook: RR_OPT_TIME_TILES (0 untiled, k > 1 "
                         "pieces of the time axis; default by sweep size)")
    ap.add_argument("--fused-variant", type=int, default=0,
                    help="measurement hook: pin the CemaneigeGR4J kernel "
                         "variant (RR_OPT_FUSED_VARIANT: 1 many-waves, "
                         "2 small-sweep, 3 small-sweep optimistic)")
    ap.add_argument("--gr4j-variant", type=int, default=0,
                    help="measurement hook: pin the GR4J kernel variant "
                         "(RR_OPT_GR4J_VARIANT: 1 every vote decided on "
                         "the spot)")
    ap.add_argument("--no-parity-spot", action="store_true")
    ap.add_argument("--no-power-soak", action="store_true",
                    help="skip the 2.5-s steady-state soak after the timed "
                         "region that reads socket power and shader clock "
                         "(roofline.power)")
    ap.add_argument("--soak-scale", type=float, default=1.0,
                    help="scale of the soaks' durations (2.5 s for the "
                         "headline, 1.2 s per extra configuration); tests "
                         "pass 0.3: the sensor then still sees the sweep, "
                         "with fewer samples")
    ap.add_argument("--score", default="mse", choices=["mse", "nse"],
                    help="per-set score that is all-gathered: mse (the "
                         "reference's monte_carlo) or nse (BASELINE "
                         "configs[3])")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="N = 1 only: skip the short timings of the other "
                         "BASELINE configurations after the headline")
    ap.add_argument("--extra-steps", type=int, default=3)
    ap.add_argument("--live-counters", default="all",
                    choices=["all", "headline", "none"],
                    help="N = 1 only: three short rocprofv3 --pmc passes "
                         "(FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU -- one pass "
                         "each) of a child run of the workload, which measure "
                         "roofline.traffic and the instruction count on THIS "
                         "box: all = the headline and BASELINE configs[1]-[4] "
                         "(default), headline, none (the committed "
                         "profiles/traffic.json then stands in, as it does "
                         "for the other extra configurations)")
    ap.add_argument("--no-live-counters", dest="live_counters",
                    action="store_const", const="none")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="N = 1 only: skip the PCIe-inclusive timing of the "
                         "host-pointer family (HBVEdu.simulate, 100k sets)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start N copies of this script, one
    rank per GPU, rendezvous on 127.0.0.1.  Rank 0 inherits stdout (its one
    JSON line is ours); any failing rank fails the job."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
            env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs and rc == 0:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = code
            time.sleep(0.05)
    finally:
        for p in procs:                 # a rank failed: stop the others
            p.kill()
        for p in procs:
            p.wait()
    return rc


def build_workload(args, device, rank, n, first, total_sets):
    """Resident ensemble + parameter block + output buffers for one rank:
    sets [first, first + n) of the total_sets-set sweep."""
    import torch
    from rrmpg_amd import device as rrdev
    from rrmpg_amd import models
    from rrmpg_amd.utils import synthetic as syn

    f = syn.make_forcing(args.days)
    np.random.seed(1 + rank)            # each rank: its own block of sets
    if args.model == "hbvedu" and args.catchments > 0:
        return build_catchments(args, device, rank)
    if args.model == "hbvedu":
        cls = models.HBVEdu
        ens = rrdev.HBVEduEnsemble(f["temp"], f["prec"], f["month"], f["PE_m"],
                                   f["T_m"], device=device, **syn.HBV_INITS)
        name = "HBV-Edu"
    elif args.model == "abc":
        cls = models.ABCModel
        ens = rrdev.ABCEnsemble(f["prec"], 2.5, device=device)
        name = "ABC"
    elif args.model == "gr4j":
        cls = models.GR4J
        ens = rrdev.GR4JEnsemble(f["prec"], f["etp"], device=device,
                                 **syn.GR4J_INITS)
        name = "GR4J"
    elif args.model == "cemaneige":
        cls = models.Cemaneige
        from rrmpg_amd.models.cemaneige import prepare_snow_inputs
        layers, inits = prepare_snow_inputs(
            f["prec"], f["temp"] - 3, f["tmin"] - 3, f["tmax"] - 3,
            syn.STATION_HEIGHT, 0, 0, list(syn.ALTITUDES))
        ens = rrdev.CemaneigeEnsemble(layers[0], layers[1], layers[2],
                                      device=device)
        name = "Cemaneige(L=5)"
    elif args.model in ("cemaneigehystgr4j", "cemaneigegr4jice",
                        "cemaneigehystgr4jice"):
        from rrmpg_amd.models.cemaneige import prepare_snow_inputs
        hyst, ice = "hyst" in args.model, "ice" in args.model
        cls = {"cemaneigehystgr4j": models.CemaneigeHystGR4J,
               "cemaneigegr4jice": models.CemaneigeGR4JIce,
               "cemaneigehystgr4jice": models.CemaneigeHystGR4JIce}[args.model]
        layers, inits = prepare_snow_inputs(
            f["prec"], f["temp"] - 3, f["tmin"] - 3, f["tmax"] - 3,
            syn.STATION_HEIGHT, 0, 0, list(syn.ALTITUDES), etp=f["etp"])
        ens = rrdev.SnowGR4JEnsemble(
            hyst, ice, layers[0], layers[1], layers[2], layers[3],
            frac_ice=SNOW_NEXT_FRAC_ICE if ice else None,
            s_init=0.6, r_init=0.7, device=device)
        name = cls.__name__ + "(L=5)"
    else:
        cls = models.CemaneigeGR4J
        from rrmpg_amd.models.cemaneige import prepare_snow_inputs
        layers, inits = prepare_snow_inputs(
            f["prec"], f["temp"], f["tmin"], f["tmax"], syn.STATION_HEIGHT, 0,
            0, list(syn.ALTITUDES), etp=f["etp"])
        ens = rrdev.CemaneigeGR4JEnsemble(layers[0], layers[1], layers[2],
                                          layers[3], 0., 0., 0.6, 0.7,
                                          device=device)
        name = "CemaneigeGR4J(L=5)"
    if args.sampler == "device":
        # every rank draws ITS rows of one global population (counter-based
        # Philox: no communication, same population for any number of GPUs)
        params = rrdev.sample_params(cls(), n, syn.FORCING_SEED,
                                     n_total=total_sets, first=first,
                                     device=device)
        params_host = params[:min(n, 400000)].cpu().numpy()
        p0 = rrdev.sample_params(cls(), 1, syn.FORCING_SEED,
                                 n_total=total_sets, first=0, device=device)
    else:
        rec = cls().get_random_params(n)
        params = ens.upload_params(rec)
        params_host = np.stack([rec[k] for k in cls._param_list], 1)
        p0 = params[:1].contiguous()
    # synthetic observations: the sweep's first set's run + 10 % noise (the
    # same series on every rank with the device sampler)
    if getattr(args, "row_pitch", 0) > 0:
        ens.ROW_PITCH = args.row_pitch
    q0 = ens.new_output(1)
    ens.run(p0, q0)
    torch.cuda.synchronize(device)
    qobs = torch.from_numpy(syn.make_qobs(q0.cpu().numpy())).to(device)
    qsim = ens.new_output(n) if args.mode != "metric" else None
    storages = None
    if args.mode == "storages":          # every state series as well
        if args.model == "hbvedu":
            storages = tuple(ens.new_output(n) for _ in range(4))
        elif args.model == "gr4j":
            storages = tuple(ens.new_output(n) for _ in range(2))
        elif args.model == "abc":
            storages = ens.new_output(n)
        elif args.model == "cemaneige":
            storages = (ens.new_output(n, 5), ens.new_output(n, 5))
        else:
            storages = (ens.new_output(n, 5), ens.new_output(n, 5),
                        ens.new_output(n), ens.new_output(n))
    sse = torch.empty(n, dtype=torch.float64, device=device)
    return ens, params, params_host, qsim, storages, qobs, sse, name, f


def build_catchments(args, device, rank):
    """C catchments x N sets each (score-only unless --mode qsim)."""
    import torch
    from rrmpg_amd import device as rrdev
    from rrmpg_amd import models
    from rrmpg_amd.utils import synthetic as syn
    c, n = args.catchments, args.sets
    fs = [syn.make_forcing(args.days, seed=syn.FORCING_SEED + rank * c + k)
          for k in range(c)]
    temp = np.stack([f["temp"] for f in fs])
    prec = np.stack([f["prec"] for f in fs])
    month = np.stack([f["month"] for f in fs])
    pe = np.tile(syn.PE_M, (c, 1))
    tm = np.tile(syn.T_M, (c, 1))
    inits = np.tile([0., 100., 3., 10.], (c, 1))
    ens = rrdev.HBVEduCatchments(temp, prec, month, pe, tm, inits,
                                 device=device)
    np.random.seed(1 + rank)
    cls = models.HBVEdu
    rec = cls().get_random_params(c * n)
    flat = np.stack([rec[k] for k in cls._param_list], 1)
    params_host = flat
    params = torch.from_numpy(flat.reshape(c, n, 11)).to(device)
    q0 = ens.new_output(1)
    ens.run(params[:, :1].contiguous(), q0)
    torch.cuda.synchronize(device)
    qobs = torch.from_numpy(np.stack(
        [syn.make_qobs(q0[k].cpu().numpy()) for k in range(c)])).to(device)
    qsim = ens.new_output(n) if args.mode != "metric" else None
    sse = torch.empty((c, n), dtype=torch.float64, device=device)
    name = "HBV-Edu x %d catchments" % c
    return ens, params, params_host, qsim, None, qobs, sse, name, fs[0]


def _oracle_sweep(args, f):
    """(name, k, fn): fn(flat, nthreads) runs the CPU oracle over the rows of
    `flat` for the model of `args` on the forcing `f` -- reference-shaped (one
    run per set, fresh [T] arrays, column scatter into qsim[T, N])."""
    from oracle import pyoracle
    from rrmpg_amd.utils import synthetic as syn
    if args.model == "hbvedu":
        m0 = (f["month"] - 1).astype(np.int8)
        inits = [syn.HBV_INITS[k] for k in ("snow_init", "soil_init",
                                            "s1_init", "s2_init")]
        return "HBV-Edu", 11, lambda flat, nt: pyoracle.simulate_hbvedu(
            f["temp"], f["prec"], m0, f["PE_m"], f["T_m"], inits, flat,
            nthreads=nt)
    if args.model == "abc":
        return "ABC", 3, lambda flat, nt: pyoracle.simulate_abc(
            f["prec"], 2.5, flat, nthreads=nt)
    if args.model == "gr4j":
        return "GR4J", 4, lambda flat, nt: pyoracle.simulate_gr4j(
            f["prec"], f["etp"], (syn.GR4J_INITS["s_init"],
                                  syn.GR4J_INITS["r_init"]), flat,
            nthreads=nt)
    if args.model == "cemaneigegr4j":
        from rrmpg_amd.models.cemaneige import prepare_snow_inputs
        layers, _ = prepare_snow_inputs(
            f["prec"], f["temp"], f["tmin"], f["tmax"], syn.STATION_HEIGHT, 0,
            0, list(syn.ALTITUDES), etp=f["etp"])
        return "CemaneigeGR4J(L=5)", 6, \
            lambda flat, nt: pyoracle.simulate_cemaneigegr4j(
                layers[0], layers[1], layers[3], layers[2],
                (0., 0., 0.6, 0.7), flat, nthreads=nt)
    if args.model in SNOW_NEXT:
        hyst, ice = SNOW_NEXT[args.model]
        layers, fi = _snow_next_inputs(f, ice)
        return SNOW_NEXT_NAMES[args.model] + "(L=5)", \
            6 + (2 if hyst else 0) + (1 if ice else 0), \
            lambda flat, nt: pyoracle.simulate_snow_gr4j(
                hyst, ice, layers[0], layers[1], layers[3], layers[2],
                (0., 0., 0., 0.6, 0.7), flat, frac_ice=fi, nthreads=nt)
    return None


# the hysteresis / ice-melt couplings: model -> (hyst, ice)
SNOW_NEXT = {"cemaneigehystgr4j": (True, False),
             "cemaneigegr4jice": (False, True),
             "cemaneigehystgr4jice": (True, True)}
SNOW_NEXT_NAMES = {"cemaneigehystgr4j": "CemaneigeHystGR4J",
                   "cemaneigegr4jice": "CemaneigeGR4JIce",
                   "cemaneigehystgr4jice": "CemaneigeHystGR4JIce"}
SNOW_NEXT_FRAC_ICE = [0.02, 0.04, 0.25, 0.51, 0.71]


def _snow_next_inputs(f, ice):
    """The layer forcing build_workload gives the next-tier ensembles."""
    from rrmpg_amd.models.cemaneige import prepare_snow_inputs
    from rrmpg_amd.utils import synthetic as syn
    layers, _ = prepare_snow_inputs(
        f["prec"], f["temp"] - 3, f["tmin"] - 3, f["tmax"] - 3,
        syn.STATION_HEIGHT, 0, 0, list(syn.ALTITUDES), etp=f["etp"])
    return layers, (SNOW_NEXT_FRAC_ICE if ice else None)


# the reference's own published single-thread numba rates (BASELINE.md /
# docs/source/examples/speed_comparision.rst:205-210), model-timesteps/s
PUBLISHED_NUMBA = {"abc": 3.0e8}


def cpu_baseline(args, f, params_host, seconds=10.0):
    """The CPU oracle (oracle/rr_oracle.c, kind "port") timed on this box's
    host cores on a bounded sample of the same workload (about `seconds` of
    all-core work).  Checker code: it is only timed here, never used to
    produce the GPU result."""
    from oracle import pyoracle
    sweep = _oracle_sweep(args, f)
    if sweep is None or args.catchments > 0:
        return None
    name, k, run = sweep
    cores = pyoracle.max_threads()
    flat = np.ascontiguousarray(params_host).reshape(-1, k)

    def timed(nsets, nthreads):
        t0 = time.perf_counter()
        run(flat[:nsets], nthreads)
        return time.perf_counter() - t0

    # single thread, reference-shaped: a second or two
    n1 = min(2000 if seconds >= 10 else 500, flat.shape[0])
    t1 = timed(n1, 1)
    rate1 = n1 * args.days / t1
    # all host cores: calibrate, then ~`seconds` of work
    ncal = min(flat.shape[0], 250 * cores)
    tcal = timed(ncal, cores)
    nall = int(min(flat.shape[0], max(ncal, ncal * seconds / max(tcal, 1e-3))))
    tall = timed(nall, cores) if nall > ncal else tcal
    rec = {
        "value": nall * args.days / tall,
        "unit": "model-timesteps/s",
        "cores": cores,
        "kind": "port",
        "sample_short": "%d sets x %d d, %d threads, %.1f s"
                        % (nall, args.days, cores, tall),
        "sample": "%d %s parameter sets x %d days, all %d host threads "
                  "(OpenMP over sets), reference-shaped: one run per set, "
                  "fresh [T] arrays, column scatter into qsim[T,N]; %.1f s"
                  % (nall, name, args.days, cores, tall),
        "value_1thread": rate1,
        "sample_1thread": "%d sets x %d days, 1 thread, %.1f s"
                          % (n1, args.days, t1),
    }
    if args.model in PUBLISHED_NUMBA:
        rec["published_numba_1thread"] = PUBLISHED_NUMBA[args.model]
    return rec


def parity_spot(args, f, params_host, qsim, sse, qobs, n_cols=16):
    """After the timed region: columns of the RESIDENT result against the CPU
    oracle (checker only).  Returns the max relative error, or None where the
    oracle has no direct entry for the workload."""
    import torch
    from oracle import pyoracle
    from rrmpg_amd.utils import synthetic as syn
    if args.model not in ("hbvedu", "gr4j", "abc", "cemaneigegr4j") \
            and args.model not in SNOW_NEXT:
        return None
    if args.catchments > 0:
        # catchment 0 of the launch: its forcing is `f`, its parameter sets
        # the first rows of the block
        m = args.sets
        params_host = params_host[:m]
        sse = sse[0]
        qobs = qobs[0]
        qsim = qsim[0] if qsim is not None else None
    m = params_host.shape[0]
    cols = np.unique(np.linspace(0, m - 1, n_cols).astype(np.int64))
    flat = np.ascontiguousarray(params_host[cols])
    if args.model == "hbvedu":
        inits = [syn.HBV_INITS[k] for k in ("snow_init", "soil_init",
                                            "s1_init", "s2_init")]
        ref = pyoracle.simulate_hbvedu(f["temp"], f["prec"], f["month"] - 1,
                                       f["PE_m"], f["T_m"], inits, flat,
                                       nthreads=4)
    elif args.model == "gr4j":
        ref = pyoracle.simulate_gr4j(f["prec"], f["etp"],
                                     (syn.GR4J_INITS["s_init"],
                                      syn.GR4J_INITS["r_init"]), flat)
    elif args.model == "cemaneigegr4j":
        from rrmpg_amd.models.cemaneige import prepare_snow_inputs
        layers, _ = prepare_snow_inputs(
            f["prec"], f["temp"], f["tmin"], f["tmax"], syn.STATION_HEIGHT, 0,
            0, list(syn.ALTITUDES), etp=f["etp"])
        ref = pyoracle.simulate_cemaneigegr4j(
            layers[0], layers[1], layers[3], layers[2], (0., 0., 0.6, 0.7),
            flat, nthreads=4)
    elif args.model in SNOW_NEXT:
        ref = _oracle_sweep(args, f)[2](flat, 4)
    else:
        ref = pyoracle.simulate_abc(f["prec"], 2.5, flat)
    if isinstance(ref, tuple):
        ref = ref[0]
    tcols = torch.from_numpy(cols).to(sse.device)
    if qsim is not None:
        got = qsim[:, tcols].cpu().numpy()
        err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-9)
    else:                       # score-only mode: the fused sums themselves
        qo = qobs.cpu().numpy()
        want = ((qo[:, None] - ref) ** 2).sum(0)
        err = np.abs(sse[tcols].cpu().numpy() - want) / want
    if not np.all(np.isfinite(err)):
        return float("nan")
    return float(err.max())


MODEL_CLASSES = {"hbvedu": "HBVEdu", "abc": "ABCModel", "gr4j": "GR4J",
                 "cemaneigegr4j": "CemaneigeGR4J",
                 "cemaneigehystgr4j": "CemaneigeHystGR4J",
                 "cemaneigegr4jice": "CemaneigeGR4JIce",
                 "cemaneigehystgr4jice": "CemaneigeHystGR4JIce"}


def parity_spot_host_population(args, ens, f, n=4096, n_cols=16):
    """SURVEY.md section 8d's population -- ``np.random.seed(1)`` then
    ``Model.get_random_params(n)``, the reference's own sampler -- run
    through the resident ensemble after the timed region and compared with
    the CPU oracle column by column (checker only).  The timed sweep draws
    its sets on the device (numpy's Philox stream); this shows the legacy
    population gives the same agreement.  Returns the max relative error."""
    import torch
    from rrmpg_amd import models
    sweep = _oracle_sweep(args, f)
    if sweep is None or args.catchments > 0:
        return None
    _, k, run = sweep
    cls = getattr(models, MODEL_CLASSES[args.model])
    np.random.seed(1)
    rec = cls().get_random_params(n)
    params = ens.upload_params(rec)
    q = ens.new_output(n)
    ens.run(params, q)
    torch.cuda.synchronize()
    cols = np.unique(np.linspace(0, n - 1, n_cols).astype(np.int64))
    flat = np.stack([rec[name] for name in cls._param_list], 1)
    ref = run(np.ascontiguousarray(flat[cols]), 4)
    if isinstance(ref, tuple):
        ref = ref[0]
    got = q[:, torch.from_numpy(cols).to(q.device)].cpu().numpy()
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-9)
    return float(err.max()) if np.all(np.isfinite(err)) else float("nan")


def end_to_end(args, n=100_000):
    """The host-pointer family as a reference user calls it -- numpy in, numpy
    out, PCIe and the first touch of the result's pages included (never
    `value`): HBVEdu.simulate of BASELINE configs[1] (100k sets x 30 years =
    8.77 GB of qsim into a numpy array; replaces the loop of the reference's
    rrmpg/models/hbvedu.py:190-214) and monte_carlo(return_qsim=False) of the
    same sets (scores only: 0.8 MB back)."""
    from rrmpg_amd import _lib, models
    from rrmpg_amd.tools import monte_carlo
    from rrmpg_amd.utils import synthetic as syn
    f = syn.make_forcing(args.days)
    kw = dict(temp=f["temp"], prec=f["prec"], month=f["month"],
              PE_m=f["PE_m"], T_m=f["T_m"], **syn.HBV_INITS)
    np.random.seed(1)
    m = models.HBVEdu()
    p = m.get_random_params(n)
    m.simulate(params=p[:64], **kw)                  # context, forcing upload
    t0 = time.perf_counter()
    q = m.simulate(params=p, **kw)
    t_sim = time.perf_counter() - t0
    qobs = syn.make_qobs(q[:, :1].copy())
    finite = bool(np.isfinite(q[-1]).all())
    nbytes = q.nbytes
    del q
    t_mc = None
    for _ in range(3):                  # (best of three: one call is 6 ms)
        np.random.seed(1)
        t0 = time.perf_counter()
        mc = monte_carlo(m, n, qobs=qobs, return_qsim=False, **kw)
        dt = time.perf_counter() - t0
        t_mc = dt if t_mc is None else min(t_mc, dt)
    # the same call with the sets drawn in HBM (sampler='device'), at the
    # configuration's 100k sets and at the headline's million: what the
    # reference's monte_carlo(model, num, qobs) costs a user end to end
    monte_carlo(m, 1000, qobs=qobs, return_qsim=False, sampler="device",
                seed=1, **kw)                              # ensemble upload
    dev_times = {}
    for nn in (n, 1_000_000):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            r_dev = monte_carlo(m, nn, qobs=qobs, return_qsim=False,
                                sampler="device", seed=7, **kw)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        dev_times[nn] = (best, bool(np.isfinite(r_dev["mse"]).all()))
    # ... and the same million as ONE call over eight shards (gpus=8: on this
    # one GPU the shards share it, a stream each; on a node each has its
    # own): against eight single-shard calls back to back
    def best_of(fn, k=3):
        best = None
        for _ in range(k):
            t0 = time.perf_counter()
            res = fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, res
    t_shard, _ = best_of(lambda: monte_carlo(
        m, 125_000, qobs=qobs, return_qsim=False, sampler="device", seed=7,
        **kw))
    t_g8, r_g8 = best_of(lambda: monte_carlo(
        m, 1_000_000, qobs=qobs, return_qsim=False, sampler="device", seed=7,
        gpus=8, **kw))
    g8_equal = bool(np.array_equal(r_g8["mse"], r_dev["mse"]))
    # BASELINE configs[3] as the user writes it: CemaneigeGR4J, 1M sets,
    # per-set NSE, eight shards, one call
    fm = models.CemaneigeGR4J()
    fkw = dict(prec=f["prec"], mean_temp=f["temp"], min_temp=f["tmin"],
               max_temp=f["tmax"], etp=f["etp"],
               met_station_height=syn.STATION_HEIGHT,
               altitudes=list(syn.ALTITUDES), s_init=0.6, r_init=0.7)
    fq = syn.make_qobs(np.asarray(fm.simulate(
        params=fm.get_random_params(1), **fkw))[:, :1].copy())
    fcall = dict(qobs=fq, return_qsim=False, score="nse", sampler="device",
                 seed=7)
    monte_carlo(fm, 1000, **fcall, **fkw)
    tf_shard, _ = best_of(lambda: monte_carlo(fm, 125_000, **fcall, **fkw), 2)
    tf_one, rf_one = best_of(lambda: monte_carlo(fm, 1_000_000, **fcall,
                                                 **fkw), 2)
    tf_g8, rf_g8 = best_of(lambda: monte_carlo(fm, 1_000_000, gpus=8, **fcall,
                                               **fkw), 2)
    fused_equal = bool(np.array_equal(rf_g8["nse"], rf_one["nse"]))
    _lib.load().rr_release_cached_memory()
    steps = n * args.days
    return {
        "workload": "HBVEdu.simulate, %d sets x %d days, host pointers: "
                    "numpy in, qsim[T,N] (%.2f GB) in a fresh numpy array out"
                    % (n, args.days, nbytes / 1e9),
        "seconds": t_sim,
        "model_timesteps_per_s": steps / t_sim,
        "gb_per_s_into_numpy": nbytes / t_sim / 1e9,
        "finite": finite,
        "monte_carlo_scores_only": {
            "workload": "monte_carlo(HBVEdu, %d, qobs, return_qsim=False): "
                        "sets drawn on the host (numpy's global generator, "
                        "the reference's contract: ~3.8 ms of the call), "
                        "upload, sweep, per-set MSE back; best of three "
                        "calls" % n,
            "seconds": t_mc,
            "model_timesteps_per_s": steps / t_mc,
            "finite": bool(np.isfinite(mc["mse"]).all())},
        "monte_carlo_device_sampler": {
            "workload": "monte_carlo(HBVEdu, num, qobs, return_qsim=False, "
                        "sampler='device'): forcing validated and uploaded, "
                        "sets drawn in HBM, sweep, per-set MSE back (best of "
                        "three calls)",
            "seconds_100k": dev_times[n][0],
            "seconds_1m": dev_times[1_000_000][0],
            "model_timesteps_per_s_1m":
                1_000_000 * args.days / dev_times[1_000_000][0],
            "finite": dev_times[n][1] and dev_times[1_000_000][1],
            # gpus=8 in ONE call (the eight shards share this GPU) against
            # eight single-shard calls back to back
            "seconds_125k": t_shard,
            "seconds_1m_gpus8": t_g8,
            "gpus8_over_8_shards": t_g8 / (8 * t_shard),
            "gpus8_equals_one_gpu": g8_equal},
        "monte_carlo_configs3": {
            "workload": "monte_carlo(CemaneigeGR4J(), 1_000_000, qobs, "
                        "return_qsim=False, score='nse', sampler='device', "
                        "gpus=8): BASELINE configs[3] as one call, the eight "
                        "shards sharing this GPU (best of two)",
            "seconds_125k": tf_shard,
            "seconds_1m": tf_one,
            "seconds_1m_gpus8": tf_g8,
            "gpus8_over_8_shards": tf_g8 / (8 * tf_shard),
            "gpus8_equals_one_gpu": fused_equal},
        "note": "PCIe-inclusive, single GPU; never the line's `value`",
    }


# fp64 VALU issue roof (profiles/README.md, profiles/ubench/valu_cost.hip): a
# wave64 fp64 instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs;
# 2.4 GHz nominal engine clock.
VALU_CYCLES_PER_INSTR = 4
SIMDS = 1024
CLOCK_GHZ_NOMINAL = 2.4

ALL_OUT_BYTES = {"hbvedu": 40, "abc": 16, "gr4j": 24, "cemaneigegr4j": 104,
                 "cemaneige": 88, "cemaneigehystgr4j": 0,
                 "cemaneigegr4jice": 0, "cemaneigehystgr4jice": 0}


def traffic_record(args, n, t):
    """HBM traffic and VALU instructions per model-timestep of this workload
    from the committed counter passes (profiles/traffic.json, written by
    profiles/summarize.py from separate rocprofv3 --pmc runs): NOT measured
    in this run -- the line says so in "source"."""
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    key = "%s:%s:%d:%d" % (args.model, args.mode, n, t)
    if args.catchments > 0:
        key = "%s:%s:%dx%d:%d" % (args.model, args.mode, args.catchments,
                                  args.sets, t)
    try:
        from rrmpg_amd.utils.buildid import kernel_source_id
        with open(tpath) as fh:
            pmc = json.load(fh)
        if pmc.get("_build", {}).get(args.model) != kernel_source_id(args.model):
            # the kernels have changed since the counters were collected
            return None, None
        return pmc.get(key), pmc.get(key + ":valu_instr_per_unit")
    except Exception:
        return None, None


def traffic_is_stale(model):
    """True if profiles/traffic.json was collected on other kernel sources
    than this tree's (rrmpg_amd.utils.buildid)."""
    try:
        from rrmpg_amd.utils.buildid import kernel_source_id
        with open(os.path.join(REPO, "profiles", "traffic.json")) as fh:
            pmc = json.load(fh)
        return pmc.get("_build", {}).get(model) != kernel_source_id(model)
    except Exception:
        return True


def live_counters(args, n, t, per_pass_timeout=120):
    """HBM traffic per launch and vector instructions per model-timestep of
    THIS workload measured on THIS box: three rocprofv3 passes (--kernel-trace
    --pmc, one counter group each: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU) over
    a short child run of this script, corrected as MI355X_MICROARCH.md's HBM
    section prescribes (KiB -> x1024; gfx950's FETCH_SIZE counts 64 B per
    128-B request: x2) -- profiles/collect.sh's recipe, without the committed
    file in between.  The sweep kernel is the one with the largest total
    duration; only its launches at its largest grid count (the one-set helper
    launch does not).  Returns a dict or None (no rocprofv3, a pass that
    fails or runs into its timeout: the committed numbers then stand in)."""
    import csv
    import glob
    import shutil
    import signal
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    # not under a profiler ourselves (profiles/collect.sh, a driver's own
    # rocprofv3 around this script): a profiler inside a profiled process is
    # asking for trouble -- the committed numbers then stand in
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in
           os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1",
             "--steps", "3", "--warmup", "1", "--model", args.model, "--mode",
             args.mode, "--sets", str(args.sets), "--days", str(args.days),
             "--catchments", str(args.catchments), "--score", args.score,
             "--sampler", args.sampler, "--no-cpu-baseline",
             "--no-extra-configs", "--no-power-soak", "--no-parity-spot",
             "--no-end-to-end", "--live-counters", "none"]
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    t0 = time.perf_counter()
    for group in (["FETCH_SIZE"], ["WRITE_SIZE"],
                  ["SQ_INSTS_VALU", "SQ_WAVES"]):
        out = tempfile.mkdtemp(prefix="rr_pmc_")
        cmd = [exe, "--kernel-trace", "--pmc", *group, "--output-format",
               "csv", "-d", out, "--"] + child
        try:
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env,
                                    stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=per_pass_timeout)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)   # the group WE started
                proc.wait()
                return None
            if rc != 0:
                return None
            rows = []
            for f in glob.glob(os.path.join(out, "*",
                                            "*_counter_collection.csv")):
                with open(f) as fh:
                    rows += list(csv.DictReader(fh))
        finally:
            shutil.rmtree(out, ignore_errors=True)
        if not rows:
            return None
        total = {}
        for r in rows:
            if r["Counter_Name"] == group[0]:
                total[r["Kernel_Name"]] = total.get(r["Kernel_Name"], 0) + (
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if not total:
            return None
        kern = max(total, key=total.get)
        grid = max(float(r["Grid_Size"]) for r in rows
                   if r["Kernel_Name"] == kern)
        for name in group:
            vals = [float(r["Counter_Value"]) for r in rows
                    if r["Kernel_Name"] == kern and r["Counter_Name"] == name
                    and float(r["Grid_Size"]) == grid]
            if not vals:
                return None
            got[name] = sum(vals) / len(vals)
        got["kernel"] = kern.split("(")[0][:100]
    jobs = -(-args.sets // 64) * max(1, args.catchments)
    return {"traffic": got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024,
            "read_bytes_corrected": got["FETCH_SIZE"] * 1024 * 2,
            "write_bytes": got["WRITE_SIZE"] * 1024,
            "valu_instr_per_unit": got["SQ_INSTS_VALU"] / (jobs * t),
            "waves": got["SQ_WAVES"], "kernel": got["kernel"],
            "seconds": time.perf_counter() - t0,
            "source": "rocprofv3 --kernel-trace --pmc, three passes of a "
                      "3-step child run on this box (FETCH_SIZE x 2048 + "
                      "WRITE_SIZE x 1024 bytes per launch)"}


class SocketSampler:
    """Socket power and shader clock of one GPU, read from its hwmon files
    (power1_input in microwatts, freq1_input in hertz, power1_cap) every few
    milliseconds on a thread of its own: which roof the chip is at.
    Everything is optional -- no files, no record."""

    def __init__(self, device_index):
        import glob
        import threading
        self.dir = None
        want = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id,
                                        pr.pci_device_id)
        except Exception:
            pass
        cands = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.path.exists(os.path.join(d, "power1_input")):
                cands.append(d)
        for d in cands:
            if want and want in os.path.realpath(os.path.join(d, "..", "..")):
                self.dir = d
        if self.dir is None and len(cands) == 1:
            self.dir = cands[0]
        self.power, self.clock = [], []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as fp:
                return float(fp.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            p, c = self._read("power1_input"), self._read("freq1_input")
            if p is not None:
                self.power.append(p * 1e-6)
            if c is not None:
                self.clock.append(c * 1e-6)
            time.sleep(0.004)

    def __enter__(self):
        if self.dir:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.dir:
            self._thread.join(timeout=1.0)

    def record(self):
        if not self.power:
            return None
        cap = self._read("power1_cap")
        return {"socket_w": float(np.mean(self.power)),
                "sclk_mhz": float(np.mean(self.clock)) if self.clock else None,
                "cap_w": cap * 1e-6 if cap else None,
                "samples": len(self.power),
                "source": "hwmon power1_input / freq1_input of this GPU"}


# What the headline's ingredients cost at the socket (profiles/r06_energy.txt,
# profiles/ubench/energy.hip): picojoules per fp64 lane-operation and per
# byte stored with `nt sc1`, measured with each ingredient alone
ENERGY_PJ = {"fp64_lane_op": 33.0, "stored_byte": 140.0}


def energy_budget(valu_per_unit, bytes_per_unit):
    """Dynamic energy of one model-timestep by component (nJ) at the
    measured prices: which ingredient the power cap is spent on."""
    if not valu_per_unit:
        return None
    v = valu_per_unit * ENERGY_PJ["fp64_lane_op"] * 1e-3
    b = bytes_per_unit * ENERGY_PJ["stored_byte"] * 1e-3
    return {"valu_nj": v, "store_nj": b, "valu_share": v / (v + b),
            "source": "profiles/r06_energy.txt"}


# what a SIMD with four or more waves needs per fp64 vector instruction
# (profiles/ubench/valu_cost.hip): 4 is the hardware's rate,
# 4.3 what a stream of dependent-free fp64 instructions is measured at
VALU_CYCLES_MEASURED = 4.3


def valu_at_clock(floor_ms_at_clock, kernel_ms):
    """The fp64 issue roof at the clock the chip sustained under the sweep:
    at the hardware's 4 cycles per instruction and at the measured 4.3."""
    issue_ms = floor_ms_at_clock * VALU_CYCLES_MEASURED / VALU_CYCLES_PER_INSTR
    return {"floor_ms_at_measured_clock": floor_ms_at_clock,
            "frac_at_measured_clock": floor_ms_at_clock / kernel_ms,
            "cycles_per_instr_measured": VALU_CYCLES_MEASURED,
            "issue_ms_at_measured_clock": issue_ms,
            "issue_frac_at_measured_clock": issue_ms / kernel_ms}


def power_soak(sweep, per_step_s, device, world, seconds=2.5):
    """Socket power and shader clock of the sweep in steady state: the same
    step repeated for `seconds` AFTER the timed region (the sensor is an
    average over about a second; the timed region of a default run lasts a
    tenth of one), samples of the second half only.  Every rank runs the
    same number of steps (the step holds a collective)."""
    import torch
    import torch.distributed as dist
    count = max(4, int(seconds / max(per_step_s, 1e-4)))
    sampler = SocketSampler(device.index if getattr(device, "index", None)
                            is not None else 0)
    with sampler:
        t0 = time.perf_counter()
        for k in range(count):
            if k == count // 2:
                torch.cuda.synchronize(device)
                half = (len(sampler.power), len(sampler.clock))
            sweep.step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
    sampler.power = sampler.power[half[0]:]
    sampler.clock = sampler.clock[half[1]:]
    rec = sampler.record()
    if rec:
        rec["soak_steps"] = count
        rec["soak_ms_per_step"] = wall / count * 1e3
        rec["source"] = ("hwmon power1_input / freq1_input of this GPU, "
                         "second half of a %.1f-s soak of the same sweep "
                         "after the timed region" % wall)
    return rec


def run_workload(args, device, rank, world, on_host, steps, warmup,
                 score="mse", settle_s=0.0, min_timed_s=0.0):
    """Build this rank's share of the workload `args` names, time `steps`
    sweeps (barrier + synchronize on both sides) and return the measurements.
    The sweep itself -- kernel launch, score, the one all-gather -- is
    rrmpg_amd.sharding.ResidentSweep's."""
    import torch
    import torch.distributed as dist
    from rrmpg_amd.sharding import ResidentSweep, shard_bounds

    scaling = "weak" if args.catchments > 0 else args.scaling
    if scaling == "strong":
        total_sets = args.sets
        first, stop = shard_bounds(total_sets, world, rank)
        n_sets = stop - first
    else:
        n_sets = args.sets
        total_sets = n_sets * world
        first = rank * n_sets
    (ens, params, params_host, qsim, storages, qobs, _sse, name,
     f) = build_workload(args, device, rank, n_sets, first, total_sets)
    cmul = max(1, args.catchments)
    n, t = n_sets * cmul, args.days
    total_units = total_sets * cmul
    sweep = ResidentSweep(ens, params, qobs, total_units, score=score,
                          qsim=qsim, storages=storages, on_host=on_host)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(warmup):
        scores = sweep.step()
    fence()
    if settle_s > 0:
        # the extra configurations: each follows seconds of CPU work (the
        # previous one's oracle sample), during which the GPU has clocked
        # down, and a sweep of a few milliseconds is over before the clock is
        # back -- round 4's driver run timed HBV-Edu 100k at 2.83 ms where
        # sweeps in steady state take 2.4-2.6.  Untimed sweeps until `settle_s`
        # seconds have passed, then enough timed ones to fill `min_timed_s`.
        t_s = time.perf_counter()
        k_s = 0
        while time.perf_counter() - t_s < settle_s:
            scores = sweep.step()
            torch.cuda.synchronize(device)
            k_s += 1
        per = (time.perf_counter() - t_s) / max(k_s, 1)
        steps = int(min(200, max(steps, np.ceil(min_timed_s / max(per, 1e-6)))))
    # kernel time: HIP events on the stream the kernel is launched on (torch's
    # current stream), bracketing only the library call of each step; a third
    # event closes the score exchange
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
          for _ in range(steps)]
    fence()
    t0 = time.perf_counter()
    # The score exchange of step k is enqueued behind its sweep and finished
    # after step k + 1's sweep has been enqueued (sharding.ScoreExchange): on
    # several GPUs the all-gather runs beside the next shard's kernel.  Every
    # step's exchange completes inside the timed region.
    pending = None
    for k in range(steps):
        ev[k][0].record()
        sweep.launch()
        ev[k][1].record()
        exchange = sweep.gather_begin()
        if pending is not None:
            scores = pending.finish()
        pending = exchange
        ev[k][2].record()
    scores = pending.finish()
    fence()
    elapsed = time.perf_counter() - t0
    if hasattr(ens, "check"):
        ens.check()       # GR4J family: an unusable x4 would have written nothing

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in ev]))
    gather_ms = float(np.mean([b.elapsed_time(c) for _, b, c in ev]))
    k_min = k_max = kernel_ms
    if world > 1:
        red = torch.tensor([elapsed, kernel_ms, -kernel_ms, gather_ms],
                           dtype=torch.float64,
                           device="cpu" if on_host else device)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        elapsed, k_max, k_min, gather_ms = (float(red[0]), float(red[1]),
                                            -float(red[2]), float(red[3]))
    assert scores.numel() == total_units
    finite = bool(torch.isfinite(scores).all().item())
    # the whole sweep's scores as every rank holds them after the all-gather
    # (same population for any number of ranks with the device sampler: a
    # sharded run must reproduce the single-rank digest)
    import hashlib
    digest = hashlib.sha1(scores.detach().cpu().numpy().tobytes()).hexdigest()
    bytes_per_step = {"qsim": 8, "metric": 0,
                      "storages": ALL_OUT_BYTES[args.model]}[args.mode]
    return dict(steps=steps, digest=digest, first=first, ens=ens, sweep=sweep,
                params_host=params_host, qsim=qsim,
                qobs=qobs, f=f, name=name, scaling=scaling, n=n, t=t,
                total_units=total_units, elapsed=elapsed, kernel_ms=kernel_ms,
                k_min=k_min, k_max=k_max, gather_ms=gather_ms, finite=finite,
                bytes_per_step=bytes_per_step,
                achieved=bytes_per_step * n * t / (kernel_ms * 1e-3) / 1e9)


# The other BASELINE.json configurations (and the HBM-bound modes), timed in
# the same driver run after the headline: label -> bench arguments.  One
# GPU's share where the config names eight.
EXTRA_CONFIGS = [
    # (id, label, bench arguments); BASELINE configs[1]..[4] first: the
    # driver keeps the END of stdout, and the line is short enough to fit
    ("cfg1", "HBV-Edu 100k sets, qsim + MSE (configs[1])",
     dict(model="hbvedu", mode="qsim", sets=100_000)),
    ("cfg2", "GR4J 1M sets, qsim + MSE (configs[2])",
     dict(model="gr4j", mode="qsim", sets=1_000_000)),
    ("cfg3", "CemaneigeGR4J 125k sets = one GPU's shard of 1M over 8, "
     "per-set NSE (configs[3])",
     dict(model="cemaneigegr4j", mode="metric", sets=125_000, score="nse")),
    ("cfg4", "HBV-Edu 125 catchments x 10k sets = one GPU's share of 1000 x "
     "10k, per-set MSE (configs[4])",
     dict(model="hbvedu", mode="metric", sets=10_000, catchments=125)),
    ("hbv5out", "HBV-Edu 400k sets, all five outputs (40 B per "
     "model-timestep)",
     dict(model="hbvedu", mode="storages", sets=400_000)),
    ("abc", "ABC 1M sets, qsim + MSE (8 B per model-timestep)",
     dict(model="abc", mode="qsim", sets=1_000_000)),
    # the hysteresis / ice-melt couplings, L = 5, default bounds
    ("hyst", "CemaneigeHystGR4J 1M sets, per-set MSE (next tier)",
     dict(model="cemaneigehystgr4j", mode="metric", sets=1_000_000)),
    ("ice", "CemaneigeGR4JIce 1M sets, per-set MSE (next tier)",
     dict(model="cemaneigegr4jice", mode="metric", sets=1_000_000)),
    ("hystice", "CemaneigeHystGR4JIce 1M sets, per-set MSE (next tier)",
     dict(model="cemaneigehystgr4jice", mode="metric", sets=1_000_000)),
]


def extra_configs(args, device):
    """Each of EXTRA_CONFIGS for a few sweeps on this GPU: kernel ms, rate,
    fraction of the HBM peak (where the mode writes bytes) and the parity
    spot against the oracle."""
    import copy
    import gc
    import torch
    out = []
    for ident, label, spec in EXTRA_CONFIGS:
        a = copy.copy(args)
        a.catchments, a.scaling, a.sampler = 0, "strong", args.sampler
        score = spec.get("score", "mse")
        for k, v in spec.items():
            if k != "score":
                setattr(a, k, v)
        try:
            r = run_workload(a, device, 0, 1, False, args.extra_steps, 1,
                             score=score, settle_s=0.3, min_timed_s=0.06)
            rec = {"id": ident, "workload": label,
                   "kernel_ms": r["kernel_ms"], "steps": r["steps"],
                   "model_timesteps_per_s": r["n"] * r["t"]
                   / (r["kernel_ms"] * 1e-3),
                   "bytes_per_unit": r["bytes_per_step"],
                   "frac": (r["achieved"] / HBM_PEAK_GBPS
                            if r["bytes_per_step"] else None),
                   "score": score, "scores_finite": r["finite"],
                   "parity_spot": (None if args.no_parity_spot else
                                   parity_spot(a, r["f"], r["params_host"],
                                               r["qsim"], r["sweep"].sse,
                                               r["qobs"]))}
            traffic, valu = traffic_record(a, r["n"], r["t"])
            rec["traffic_from"] = "profiles/traffic.json" if traffic else None
            if (args.live_counters == "all" and ident.startswith("cfg")
                    and r["bytes_per_step"] * r["n"] * r["t"] * 1.1 + (4 << 30)
                    < torch.cuda.mem_get_info(device)[0]):
                # BASELINE configs[1]-[4]: this box's own counter passes
                try:
                    live = live_counters(a, r["n"], r["t"])
                except Exception:
                    live = None
                if live:
                    traffic, valu = live["traffic"], live["valu_instr_per_unit"]
                    rec["traffic_from"] = "this run"
                    rec["live_counters"] = live
            if traffic:
                rec["traffic"] = traffic
            # clock and socket power under THIS sweep (a 1.2-s soak): the
            # issue roof below is taken at that clock as well
            pw = (None if args.no_power_soak else
                  power_soak(r["sweep"], r["kernel_ms"] * 1e-3, device, 1,
                             seconds=1.2 * args.soak_scale))
            if pw:
                rec["power"] = {k: pw[k] for k in
                                ("socket_w", "sclk_mhz", "cap_w")}
            if valu:
                floor_ms = (valu / 64.0 * r["n"] * r["t"]
                            * VALU_CYCLES_PER_INSTR / SIMDS
                            / (CLOCK_GHZ_NOMINAL * 1e9) * 1e3)
                rec["valu"] = {"instr_per_unit": valu, "floor_ms": floor_ms,
                               "frac": floor_ms / r["kernel_ms"]}
                if pw and pw.get("sclk_mhz"):
                    f_ms = floor_ms * CLOCK_GHZ_NOMINAL * 1e3 / pw["sclk_mhz"]
                    rec["valu"].update(valu_at_clock(f_ms, r["kernel_ms"]))
            if not args.no_cpu_baseline:
                # the oracle on the host cores for THIS model (a short
                # sample); the HBV-Edu configurations share the headline's
                rec["cpu_baseline"] = (
                    "the headline's (same model)" if a.model == "hbvedu"
                    else cpu_baseline(a, r["f"], r["params_host"],
                                      seconds=2.5))
        except Exception as exc:                # keep the headline line
            rec = {"id": ident, "workload": label, "error": "%s: %s"
                   % (type(exc).__name__, exc)}
            r = None
        out.append(rec)
        del r
        gc.collect()
        torch.cuda.empty_cache()
    return out


def _sig(x, digits=5):
    """Floats of a record to `digits` significant digits (ints, strings,
    None and bools stay)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


LINE_LIMIT = 7500        # the driver keeps 8 KB of stdout


def compact_line(d):
    """The ONE line bench.py prints, from the full record `d` (which goes to
    the side file the line names under "detail"): numbers and short keys
    only -- every key is explained in profiles/BENCH_KEYS.md.  The contract's
    fields keep their names; headline numbers keep full precision."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup",
            "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: d[k] for k in keep}
    c = d["config"]
    out["config"] = {k: c[k] for k in
                     ("workload", "model", "sets_total", "sets_per_gpu",
                      "timesteps", "mode", "score", "sampler") if k in c}
    if "shards" in c and d["n_gpus"] > 1:
        out["config"]["shards"] = c["shards"]
    r = d["roofline"]
    roof = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac",
                              "traffic", "kernel_ms")}
    roof["traffic_from"] = ("this run (rocprofv3 --pmc passes)"
                            if r.get("live_counters")
                            else "stale" if "stale" in r["source"]
                            else "profiles/traffic.json")
    if r.get("live_counters") and r.get("committed_counters", {}).get(
            "traffic"):
        roof["traffic_committed"] = r["committed_counters"]["traffic"]
    roof["valu_instr_per_unit"] = r.get("valu_instr_per_unit")
    if "valu" in r:
        roof["valu"] = _sig({k: r["valu"].get(k) for k in
                             ("instr_per_unit", "floor_ms", "frac",
                              "frac_at_measured_clock",
                              "issue_frac_at_measured_clock")})
    if r.get("power"):
        roof["power"] = _sig({k: r["power"].get(k) for k in
                              ("socket_w", "sclk_mhz", "cap_w")})
        bud = r["power"].get("budget")
        if bud:
            roof["power"]["budget_nj"] = _sig(
                {"valu": bud["valu_nj"], "store": bud["store_nj"]}, 4)
    if "binding_roof" in r:
        roof["binding_roof"] = r["binding_roof"]
    out["roofline"] = roof
    for k in ("kernel_ms_per_rank", "allgather_ms", "scores_finite",
              "scores_digest", "parity_spot", "parity_spot_host_population"):
        if k in d:
            out[k] = d[k]
    cb = d.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _sig({
            "value": cb["value"], "unit": cb["unit"], "cores": cb["cores"],
            "kind": cb["kind"], "sample": cb["sample_short"],
            "value_1thread": cb["value_1thread"]}, 6)
    ex = []
    for e in d.get("extra_configs", []):
        if "error" in e:
            ex.append({"id": e["id"], "error": e["error"][:120]})
            continue
        rec = {"id": e["id"], "kernel_ms": e["kernel_ms"],
               "rate": e["model_timesteps_per_s"], "B": e["bytes_per_unit"],
               "frac": e["frac"], "score": e["score"],
               "finite": e["scores_finite"], "parity_spot": e["parity_spot"]}
        if e.get("traffic"):
            rec["traffic"] = e["traffic"]
            rec["live"] = e.get("traffic_from") == "this run"
        if e.get("power"):
            rec["w"] = e["power"].get("socket_w")
            rec["mhz"] = e["power"].get("sclk_mhz")
        if e.get("valu"):
            v = e["valu"]
            rec["valu"] = {"n": v["instr_per_unit"], "frac": v["frac"],
                           "frac_clk": v.get("frac_at_measured_clock"),
                           "issue_clk": v.get("issue_frac_at_measured_clock")}
        cpu = e.get("cpu_baseline")
        if isinstance(cpu, dict):
            rec["cpu"] = [cpu["value"], cpu["cores"], cpu["value_1thread"]]
        ex.append(_sig(rec, 4))
    if ex:
        out["extra_configs"] = ex
    ee = d.get("end_to_end")
    if ee and "error" in ee:
        out["end_to_end"] = {"error": ee["error"][:200]}
    elif ee:
        mh, md = ee["monte_carlo_scores_only"], ee["monte_carlo_device_sampler"]
        m3 = ee["monte_carlo_configs3"]
        out["end_to_end"] = _sig({
            "simulate_100k_s": ee["seconds"],
            "gb_per_s_into_numpy": ee["gb_per_s_into_numpy"],
            "mc_host_100k_s": mh["seconds"],
            "mc_dev_100k_s": md["seconds_100k"],
            "mc_dev_1m_s": md["seconds_1m"],
            "mc_dev_125k_s": md["seconds_125k"],
            "mc_dev_1m_gpus8_s": md["seconds_1m_gpus8"],
            "gpus8_over_8_shards": md["gpus8_over_8_shards"],
            "cfg3_125k_s": m3["seconds_125k"], "cfg3_1m_s": m3["seconds_1m"],
            "cfg3_1m_gpus8_s": m3["seconds_1m_gpus8"],
            "cfg3_gpus8_over_8_shards": m3["gpus8_over_8_shards"],
            "gpus8_equal": bool(md["gpus8_equals_one_gpu"]
                                and m3["gpus8_equals_one_gpu"]),
            "finite": bool(ee["finite"] and mh["finite"] and md["finite"])}, 4)
    out["keys"] = "profiles/BENCH_KEYS.md"
    if d.get("detail"):
        out["detail"] = d["detail"]
    return out


def write_detail(full):
    """The full record (prose included) beside the line: gpurun_out/ of the
    tree bench.py runs from (merged back by gpurun), else the temp dir."""
    import tempfile
    for folder in (os.path.join(REPO, "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(folder, exist_ok=True)
            path = os.path.join(folder, "bench_detail.json")
            with open(path, "w") as fh:
                json.dump(full, fh, indent=1)
            return os.path.relpath(path, REPO) if path.startswith(REPO) \
                else path
        except OSError:
            continue
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import gc
    import torch
    import torch.distributed as dist
    from rrmpg_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE"
              % (args.gpus, world), file=sys.stderr)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    on_host = world > 1 and args.backend == "gloo"
    for opt, val in (("hbv_variant", args.hbv_variant if args.hbv_variant >= 0
                      else None),
                     ("time_tiles", args.time_tiles if args.time_tiles >= 0
                      else None),
                     ("gr4j_variant", args.gr4j_variant or None),
                     ("fused_variant", args.fused_variant or None)):
        if val is not None:
            _lib.check(_lib.load().rr_debug_set_option(_lib.OPTIONS[opt], val),
                       "rr_debug_set_option")

    r = run_workload(args, device, rank, world, on_host, args.steps,
                     args.warmup, score=args.score)
    n, t, kernel_ms = r["n"], r["t"], r["kernel_ms"]
    r["power"] = (None if args.no_power_soak else
                  power_soak(r["sweep"], r["elapsed"] / args.steps, device,
                             world, seconds=2.5 * args.soak_scale))

    if rank == 0:
        value = r["total_units"] * t * args.steps / r["elapsed"]
        traffic, valu = traffic_record(args, n, t)
        committed = {"traffic": traffic, "valu_instr_per_unit": valu}
        live = None
        # (the child allocates the sweep's outputs a second time beside ours)
        out_bytes = r["bytes_per_step"] * n * t
        free_b = torch.cuda.mem_get_info(device)[0]
        if (world == 1 and args.live_counters != "none"
                and out_bytes * 1.1 + (4 << 30) < free_b):
            # this box's own counter passes (the GPU is ours: nothing else of
            # this script runs meanwhile; the sweep's buffers stay allocated,
            # the child's fit beside them)
            try:
                live = live_counters(args, n, t)
            except Exception:
                live = None
            if live:
                traffic, valu = live["traffic"], live["valu_instr_per_unit"]
        what = {"qsim": "qsim[T,N] written to HBM + fused per-set MSE",
                "metric": "fused per-set MSE only",
                "storages": "qsim and every state series written to HBM + "
                            "fused per-set MSE"}[args.mode]
        if args.score == "nse":
            what = what.replace("MSE", "NSE")
        if r["scaling"] == "strong":
            size = ("%d parameter sets in total (contiguous shards of %d per "
                    "GPU)" % (r["total_units"], n))
        else:
            size = "%d parameter sets per GPU" % n
        roof = {
            "bound": "hbm",
            "achieved": r["achieved"],
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": r["achieved"] / HBM_PEAK_GBPS,
            "traffic": traffic,
            "kernel_ms": kernel_ms,
            "kernel": "%s ensemble kernel, %d B/model-timestep "
                      "algorithmic" % (r["name"], r["bytes_per_step"]),
            # traffic and instruction count: committed counter passes of this
            # workload, not measured in this run
            "source": {"achieved": "HIP events, this run",
                       "traffic": (live["source"] if live else
                                   "profiles/traffic.json (rocprofv3 --pmc "
                                   "passes, profiles/collect.sh)"),
                       "valu.instr_per_unit": ("the same passes "
                                               "(SQ_INSTS_VALU)" if live else
                                               "profiles/traffic.json "
                                               "(SQ_INSTS_VALU pass)")},
            "live_counters": live,
            "committed_counters": committed,
            "valu_instr_per_unit": valu,
            # socket power and shader clock of the same sweep in steady
            # state (rank 0's GPU): at the cap, the kernel is power-bound
            "power": r["power"],
        }
        if not live and traffic_is_stale(args.model):
            roof["source"]["stale"] = (
                "profiles/traffic.json was collected on other kernel sources "
                "than this tree's (rrmpg_amd/utils/buildid.py): traffic and "
                "instruction count withheld until profiles/collect.sh + "
                "make_traffic.py have run again")
        if valu:
            # the roof that actually binds the 8 B/unit mode: fp64 VALU issue.
            # floor = instructions x 4 cycles on 1024 SIMDs at the nominal
            # clock; frac = floor / this run's kernel time
            floor_ms = (valu / 64.0 * n * t * VALU_CYCLES_PER_INSTR
                        / SIMDS / (CLOCK_GHZ_NOMINAL * 1e9) * 1e3)
            roof["valu"] = {"instr_per_unit": valu,
                            "cycles_per_instr": VALU_CYCLES_PER_INSTR,
                            "simds": SIMDS,
                            "clock_ghz_nominal": CLOCK_GHZ_NOMINAL,
                            "floor_ms": floor_ms,
                            "frac": floor_ms / kernel_ms}
            pw = r["power"] or {}
            if pw.get("sclk_mhz"):
                # ... and at the clock the chip actually sustains under this
                # sweep (the soak's mean shader clock): what is left in the
                # kernel once the socket's power cap has been paid
                f_ms = floor_ms * CLOCK_GHZ_NOMINAL * 1e3 / pw["sclk_mhz"]
                roof["valu"].update(valu_at_clock(f_ms, kernel_ms))
        if r["power"] and valu:
            # per-component budget of a model-timestep's dynamic energy
            roof["power"]["budget"] = energy_budget(valu, r["bytes_per_step"])
        pw = r["power"] or {}
        if pw.get("socket_w") and pw.get("cap_w"):
            # `bound` stays what the contract asks for (the HBM roof the
            # fraction is taken against); this names the roof that BINDS
            roof["binding_roof"] = (
                "socket power" if pw["socket_w"] >= 0.97 * pw["cap_w"]
                else ("hbm" if roof["frac"] >= 0.7 else "fp64 issue"))
        out = {
            "metric": "model-timesteps/s",
            "value": value,
            "unit": "model-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": r["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": r["scaling"],
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s Monte-Carlo sweep, %s x %d daily steps, %s, "
                            "RCCL all-gather of per-set %s"
                            % (r["name"], size, t, what, args.score.upper()),
                "model": args.model,
                "sets_total": r["total_units"],
                "sets_per_gpu": n,
                "timesteps": t,
                "mode": args.mode,
                "score": args.score,
                "sharding": "parameter sets, one contiguous block per GPU "
                            "(rrmpg_amd.sharding.ResidentSweep)",
                "sampler": args.sampler if args.catchments == 0 else "host",
            },
            "roofline": roof,
            # HIP-event time of the sweep kernel on the slowest / fastest
            # rank, and what the score exchange costs the compute stream
            # (score + enqueue; an RCCL all-gather itself runs beside the
            # next step's sweep, sharding.ScoreExchange)
            "kernel_ms_per_rank": {"min": r["k_min"], "max": r["k_max"]},
            "allgather_ms": r["gather_ms"],
            "allgather": "overlapped with the next step's sweep",
            "scores_finite": r["finite"],
            "scores_digest": r["digest"],
        }
        if r["scaling"] == "strong":
            from rrmpg_amd.sharding import shard_bounds
            out["config"]["shards"] = [list(shard_bounds(r["total_units"],
                                                         world, k))
                                       for k in range(world)]
            assert out["config"]["shards"][0] == [r["first"], r["first"] + n]
        if not args.no_parity_spot:
            # columns of the resident result vs the CPU oracle, after timing
            out["parity_spot"] = parity_spot(args, r["f"], r["params_host"],
                                             r["qsim"], r["sweep"].sse,
                                             r["qobs"])
            # ... and SURVEY 8d's own population (np.random.seed(1) +
            # get_random_params) through the same resident ensemble
            out["parity_spot_host_population"] = parity_spot_host_population(
                args, r["ens"], r["f"])
        if (not args.no_cpu_baseline and world == 1
                and args.catchments == 0):
            out["cpu_baseline"] = cpu_baseline(args, r["f"], r["params_host"])
        if world == 1 and not args.no_extra_configs:
            r = None
            gc.collect()
            torch.cuda.empty_cache()
            out["extra_configs"] = extra_configs(args, device)
            gc.collect()
            torch.cuda.empty_cache()
            try:
                if not args.no_end_to_end:
                    out["end_to_end"] = end_to_end(args)
            except Exception as exc:            # keep the headline line
                out["end_to_end"] = {"error": "%s: %s"
                                     % (type(exc).__name__, exc)}
        out["detail"] = write_detail(out)
        line = json.dumps(compact_line(out), separators=(",", ":"))
        if len(line) > LINE_LIMIT:
            print("warning: bench line is %d characters (limit %d)"
                  % (len(line), LINE_LIMIT), file=sys.stderr)
        print(line, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
