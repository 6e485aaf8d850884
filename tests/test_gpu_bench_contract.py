"""bench.py's output contract, checked on a small workload: one JSON line on
stdout with the keys the driver reads, roofline and cpu_baseline objects, and
the two-rank path (gloo, both ranks on the one GPU of the test box) giving
twice the single-rank units."""

import json
import os
import subprocess
import sys

import pytest

from .conftest import REPO

pytestmark = pytest.mark.gpu


def _run(cmd, env=None):
    e = dict(os.environ)
    for k, v in (env or {}).items():
        if v is None:
            e.pop(k, None)
        else:
            e[k] = v
    out = subprocess.run(cmd, cwd=REPO, env=e, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    d["_line_chars"] = len(lines[0])
    return d


def test_single_gpu_line():
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3",
              "--warmup", "1", "--sets", "20000", "--days", "800",
              "--no-extra-configs"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "model-timesteps/s" and d["unit"] == d["metric"]
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong"
    assert d["config"]["sets_total"] == 20000
    assert d["config"]["sets_per_gpu"] == 20000
    # columns of the resident qsim against the oracle, after the timed region
    assert 0 <= d["parity_spot"] < 1e-10
    assert d["kernel_ms_per_rank"]["min"] == d["kernel_ms_per_rank"]["max"]
    assert d["allgather_ms"] >= 0
    assert d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["data"] == "synthetic" and "workload" in d["config"]
    assert d["scores_finite"] is True
    # value is units / time
    units = 20000 * 800 * 3
    assert abs(d["value"] - units / (d["ms_per_step"] * 3e-3)) \
        <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - 8 * 20000 * 800 / (r["kernel_ms"] * 1e-3) / 1e9) \
        <= 1e-6 * r["achieved"]
    # HBM traffic: this run's own rocprofv3 --pmc passes (three short child
    # runs) where rocprofv3 is there -- the qsim rows plus a little reading --,
    # else nothing (the committed counters cover the default workload only)
    if r["traffic_from"].startswith("this run"):
        assert 0.98 < r["traffic"] / (8 * 20000 * 800) < 1.15, r["traffic"]
        assert 30 < r["valu_instr_per_unit"] < 45
    else:
        assert r["traffic"] is None
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert c["unit"] == "model-timesteps/s" and "sets" in c["sample"]


def test_default_line_carries_every_config_and_the_valu_roof():
    """The driver's command (default workload = BASELINE.json's metric
    configuration): besides the headline fields the line holds the other
    BASELINE configurations timed in the same run -- configs[1]..[4] first --
    each with its parity spot against the oracle, and the fp64-issue roof
    computed from this run's kernel time.  The WHOLE line fits the 8 KB of
    stdout the driver keeps (numbers and short keys, profiles/BENCH_KEYS.md;
    the prose is in the side file the line names)."""
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3",
              "--warmup", "1", "--extra-steps", "2", "--no-cpu-baseline",
              "--soak-scale", "0.3", "--live-counters", "headline"])
    assert d["_line_chars"] <= 7500, d["_line_chars"]
    assert d["config"]["sets_total"] == 1_000_000
    assert d["config"]["timesteps"] == 10957 and d["config"]["mode"] == "qsim"
    assert 0 <= d["parity_spot"] < 1e-10
    r = d["roofline"]
    if r["traffic_from"].startswith("this run"):
        # measured on this box, in this run: within a per cent of the
        # algorithmic bytes, and of the committed passes where those are fresh
        assert 0.995 < r["traffic"] / 87.656e9 < 1.03, r["traffic"]
        assert 34 < r["valu_instr_per_unit"] < 37
        if r.get("traffic_committed"):
            assert abs(r["traffic"] / r["traffic_committed"] - 1) < 0.01
        assert 0.4 < r["valu"]["frac"] < 1.0
    elif r["traffic_from"] == "stale":
        # the kernels have changed since the committed counter passes
        # (rrmpg_amd/utils/buildid.py): their numbers are withheld
        assert r["traffic"] is None and r["valu_instr_per_unit"] is None
        assert "valu" not in r
    else:
        assert r["traffic_from"] == "profiles/traffic.json"
        assert r["traffic"] and r["valu_instr_per_unit"]
        v = r["valu"]
        assert abs(v["frac"] - v["floor_ms"] / r["kernel_ms"]) < 1e-3
        assert 0.4 < v["frac"] < 1.0
        if r.get("power") and r["power"]["sclk_mhz"]:
            # the same floor at the clock the chip sustains under the sweep,
            # and at the 4.3 cycles an fp64 instruction is measured at
            assert v["frac"] <= v["frac_at_measured_clock"] < 1.05
            assert (v["frac_at_measured_clock"]
                    < v["issue_frac_at_measured_clock"] < 1.1)
            assert r["binding_roof"] in ("socket power", "hbm", "fp64 issue")
    assert 0.3 < r["frac"] < 1.0
    # socket power / shader clock of the same sweep in steady state, read
    # after the timed region (absent where the GPU has no hwmon files)
    pw = r.get("power")
    assert pw is None or 50 < pw["socket_w"] < 3000
    ex = d["extra_configs"]
    assert [e["id"] for e in ex] == ["cfg1", "cfg2", "cfg3", "cfg4", "hbv5out",
                                     "abc", "hyst", "ice", "hystice"]
    for e in ex:
        assert "error" not in e, e
        assert e["kernel_ms"] > 0 and e["finite"] is True
        # every configuration -- the hysteresis / ice couplings included --
        # carries its parity spot against the oracle
        assert 0 <= e["parity_spot"] < 1e-10, e
        # its own clock and socket power, and with them its issue roof
        if e.get("mhz") and "valu" in e:
            assert 500 < e["mhz"] < 3000
            assert 0.0 < e["valu"]["issue_clk"] < 1.1, e
        if e["B"]:
            assert 0 < e["frac"] < 1
    by = {e["id"]: e for e in ex}
    assert by["abc"]["frac"] > 0.6                  # the HBM-bound kernels
    assert by["hbv5out"]["frac"] > 0.6
    assert by["cfg3"]["score"] == "nse"
    # the full record (prose, every field of round 5's line) is beside it
    with open(os.path.join(REPO, d["detail"])) as fh:
        full = json.load(fh)
    assert len(full["extra_configs"]) == 9
    assert "configs[2]" in full["extra_configs"][1]["workload"]
    assert full["roofline"]["valu"]["cycles_per_instr"] == 4 \
        if "valu" in full["roofline"] else True
    # one call over eight shards (sampler='device', gpus=8): the scores of
    # the single sweep, within 1.3 x eight single-shard calls
    ee = d["end_to_end"]
    assert "error" not in ee, ee
    assert ee["gpus8_equal"] is True and ee["finite"] is True
    # (measured 0.87 ... 1.09 over six runs, the review's mark is 1.3; the
    # assertion leaves room for a noisy box)
    assert ee["gpus8_over_8_shards"] < 1.6
    assert ee["cfg3_gpus8_over_8_shards"] < 1.6


def test_eight_ranks_rehearsal_on_one_gpu():
    """As far as one GPU allows towards the 8-GPU job: the driver's launch
    line with eight ranks sharing the device over gloo, a ragged total of
    1,000,003 sets, strong scaling.  The partition is shard_bounds', and the
    all-gathered scores are those of the single-rank sweep, bit for bit (the
    ranks draw their rows of ONE Philox population)."""
    from rrmpg_amd.sharding import shard_bounds
    common = ["--steps", "2", "--warmup", "1", "--sets", "1000003", "--mode",
              "metric", "--no-extra-configs", "--no-cpu-baseline",
              "--no-parity-spot", "--no-power-soak", "--live-counters", "none"]
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common)
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
              "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
              "--master-port", "29583", "bench.py", "--gpus", "8",
              "--backend", "gloo", "--share-gpu"] + common,
             env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["sets_total"] == 1000003
    assert d["config"]["shards"] == [list(shard_bounds(1000003, 8, r))
                                     for r in range(8)]
    assert d["config"]["sets_per_gpu"] == 125001       # rank 0: a longer block
    assert d["scores_finite"] is True
    assert d["scores_digest"] == one["scores_digest"]
    units = 1000003 * 10957 * 2
    assert abs(d["value"] - units / (d["ms_per_step"] * 2e-3)) \
        <= 1e-6 * d["value"]


def test_two_ranks_share_the_gpu_over_gloo():
    """The driver's launch line (torch.distributed.run), weak scaling."""
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
              "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29577", "bench.py", "--gpus", "2", "--steps",
              "2", "--warmup", "1", "--sets", "20000", "--days", "800",
              "--backend", "gloo", "--share-gpu", "--scaling", "weak"],
             env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert d["scores_finite"] is True and d["scaling"] == "weak"
    assert d["config"]["sets_total"] == 40000
    units = 2 * 20000 * 800 * 2
    assert abs(d["value"] - units / (d["ms_per_step"] * 2e-3)) \
        <= 1e-6 * d["value"]


def test_self_launched_ranks_strong_scaling():
    """`python bench.py --gpus 2` with no launcher spawns its own two ranks;
    strong scaling shards --sets (odd, so the blocks are ragged)."""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2",
              "--warmup", "1", "--sets", "30001", "--days", "800",
              "--backend", "gloo", "--share-gpu"],
             env={"WORLD_SIZE": None, "RANK": None, "LOCAL_RANK": None})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["sets_total"] == 30001
    assert d["config"]["sets_per_gpu"] == 15001        # rank 0: longer block
    assert d["scores_finite"] is True and 0 <= d["parity_spot"] < 1e-10
    units = 30001 * 800 * 2
    assert abs(d["value"] - units / (d["ms_per_step"] * 2e-3)) \
        <= 1e-6 * d["value"]
    k = d["kernel_ms_per_rank"]
    assert 0 < k["min"] <= k["max"]


def test_self_launched_ranks_over_rccl():
    """The product path: one GPU per rank, backend nccl (= RCCL)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2",
              "--warmup", "1", "--sets", "40000", "--days", "800"],
             env={"WORLD_SIZE": None, "RANK": None, "LOCAL_RANK": None})
    assert d["n_gpus"] == 2 and d["scores_finite"] is True
    assert 0 <= d["parity_spot"] < 1e-10


def test_rccl_collectives_on_one_rank(tmp_path):
    """backend "nccl" (= RCCL) initialises and runs the sweep's collectives
    (barrier, max-reduce of the timings, all-gather of the scores) on this
    software stack -- with the one rank a single-GPU box can host."""
    script = tmp_path / "one_rank.py"
    script.write_text(
        "import os, sys, torch\n"
        "import torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from rrmpg_amd.sharding import allgather_scores\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29611',\n"
        "                  RANK='0', WORLD_SIZE='1')\n"
        "torch.cuda.set_device(0)\n"
        "dev = torch.device('cuda', 0)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "dist.barrier()\n"
        "x = torch.arange(1001, dtype=torch.float64, device=dev) * 0.5\n"
        "y = allgather_scores(x, 1001, always_collective=True)\n"
        "t = torch.tensor([3.0, 1.0], dtype=torch.float64, device=dev)\n"
        "dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(x, y) and y.data_ptr() != x.data_ptr()\n"
        "assert t.tolist() == [3.0, 1.0]\n"
        "# the exchange in two halves (the bench's pipelined steps): two\n"
        "# under way, kernels enqueued in between, finished in order\n"
        "from rrmpg_amd.sharding import allgather_scores_begin\n"
        "h1 = allgather_scores_begin(x, 1001, always_collective=True)\n"
        "z = (x * 3).sum()\n"
        "h2 = allgather_scores_begin(x * 2, 1001, always_collective=True)\n"
        "y1 = h1.finish(); w = (y1 - x).abs().sum(); y2 = h2.finish()\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(y1, x) and torch.equal(y2, x * 2)\n"
        "assert float(w) == 0.0 and float(z) == float((x * 3).sum())\n"
        "dist.barrier(); dist.destroy_process_group(); print('rccl ok')\n"
        % REPO)
    out = subprocess.run([sys.executable, str(script)], capture_output=True,
                         text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-2000:]


def test_sharding_sweep_and_monte_carlo_under_an_nccl_only_group(tmp_path):
    """A torchrun job initialised with backend "nccl" alone: sharding.sweep
    gets its host scores back onto the GPU for the exchange (round 5 handed
    RCCL a CPU tensor there), and monte_carlo(sampler='device') called inside
    the group draws, sweeps and all-gathers over RCCL -- with the one rank a
    single-GPU box can host (always_collective runs the collective anyway)."""
    script = tmp_path / "nccl_sweep.py"
    script.write_text(
        "import os, sys, numpy as np, torch\n"
        "import torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from rrmpg_amd import models, sharding\n"
        "from rrmpg_amd.tools import monte_carlo\n"
        "from rrmpg_amd.utils import synthetic as syn\n"
        "f = syn.make_forcing(900)\n"
        "kw = dict(prec=f['prec'], etp=f['etp'], s_init=0.6, r_init=0.7)\n"
        "m = models.GR4J()\n"
        "np.random.seed(4); p = m.get_random_params(777)\n"
        "qobs = m.simulate(params=p[:2], **kw)[:, 1] * 0.9\n"
        "want = sharding.sweep(m, p, qobs, score='nse', **kw)['scores']\n"
        "ref = monte_carlo(m, 777, qobs=qobs, return_qsim=False,\n"
        "                  sampler='device', seed=9, **kw)\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29613',\n"
        "                  RANK='0', WORLD_SIZE='1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "got = sharding.sweep(m, p, qobs, score='nse', always_collective=True,"
        " **kw)\n"
        "assert got['bounds'] == (0, 777)\n"
        "assert np.array_equal(got['scores'], want)\n"
        "mc = monte_carlo(m, 777, qobs=qobs, return_qsim=False,\n"
        "                 sampler='device', seed=9, **kw)\n"
        "assert np.array_equal(mc['mse'], ref['mse'])\n"
        "dist.barrier(); dist.destroy_process_group(); print('nccl sweep ok')\n"
        % REPO)
    out = subprocess.run([sys.executable, str(script)], capture_output=True,
                         text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "nccl sweep ok" in out.stdout, \
        (out.stdout + out.stderr)[-2000:]


def test_monte_carlo_device_sampler_inside_a_two_rank_group(tmp_path):
    """monte_carlo(sampler='device') called on every rank of a job (two
    ranks sharing this GPU, gloo): each draws and sweeps its block of the one
    population, the per-set sums are all-gathered, every rank returns the
    scores of the single-process call; 'params' is the rank's block."""
    script = tmp_path / "mc_ranks.py"
    script.write_text(
        "import os, sys, numpy as np, torch\n"
        "import torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from rrmpg_amd import models\n"
        "from rrmpg_amd.tools import monte_carlo\n"
        "from rrmpg_amd.utils import synthetic as syn\n"
        "rank = int(sys.argv[1])\n"
        "f = syn.make_forcing(700)\n"
        "kw = dict(temp=f['temp'], prec=f['prec'], month=f['month'],\n"
        "          PE_m=f['PE_m'], T_m=f['T_m'], **syn.HBV_INITS)\n"
        "m = models.HBVEdu()\n"
        "qobs = f['prec'] * 0.4 + 0.1\n"
        "call = dict(qobs=qobs, return_qsim=False, sampler='device', seed=11,\n"
        "            score='nse')\n"
        "ref = monte_carlo(m, 1001, **call, **kw)\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2],\n"
        "                  RANK=str(rank), WORLD_SIZE='2')\n"
        "dist.init_process_group('gloo')\n"
        "got = monte_carlo(m, 1001, **call, **kw)\n"
        "a, b = got['bounds']\n"
        "assert (a, b) == ((0, 501) if rank == 0 else (501, 1001))\n"
        "assert np.array_equal(got['mse'], ref['mse'])\n"
        "assert np.array_equal(got['nse'], ref['nse'])\n"
        "assert np.array_equal(np.asarray(got['params']),\n"
        "                      np.asarray(ref['params'])[a:b])\n"
        "# no seed: the ranks' own generators differ, the job's key is rank 0's\n"
        "np.random.seed(100 + rank)\n"
        "del call['seed']\n"
        "free = monte_carlo(m, 1001, **call, **kw)\n"
        "both = [None, None]\n"
        "dist.all_gather_object(both, free['mse'].tobytes())\n"
        "assert both[0] == both[1]\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "print('rank', rank, 'ok')\n" % REPO)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in o, o[-2000:]


def test_monte_carlo_exchange_rccl_in_process(tmp_path):
    """monte_carlo(sampler='device', exchange='rccl'): the per-set sums
    all-gathered between the GPUs of the call through in-process RCCL
    communicators (rr_comm_init_all, one thread, one group) before the host
    gets them -- with the one GPU of this box: a clique of one.  The scores
    are the host exchange's, 'sse_device' holds all sums on the GPU, and two
    shards on one device are refused (one rank per GPU)."""
    script = tmp_path / "mc_rccl.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from rrmpg_amd import models\n"
        "from rrmpg_amd.tools import monte_carlo\n"
        "from rrmpg_amd.utils import synthetic as syn\n"
        "f = syn.make_forcing(900)\n"
        "kw = dict(prec=f['prec'], etp=f['etp'], s_init=0.6, r_init=0.7)\n"
        "m = models.GR4J()\n"
        "qobs = f['prec'] * 0.3 + 0.1\n"
        "call = dict(qobs=qobs, return_qsim=False, sampler='device', seed=5,\n"
        "            score='nse')\n"
        "ref = monte_carlo(m, 100_003, **call, **kw)\n"
        "for g in (None, 1, 'all'):\n"
        "    got = monte_carlo(m, 100_003, gpus=g, exchange='rccl', **call, **kw)\n"
        "    assert np.array_equal(got['mse'], ref['mse']), g\n"
        "    assert np.array_equal(got['nse'], ref['nse']), g\n"
        "    dev = got['sse_device']\n"
        "    assert len(dev) == 1 and dev[0].is_cuda\n"
        "    assert np.array_equal(dev[0].cpu().numpy() / len(qobs), ref['mse'])\n"
        "if torch.cuda.device_count() == 1:\n"
        "    try:\n"
        "        monte_carlo(m, 1000, gpus=2, exchange='rccl', **call, **kw)\n"
        "        raise SystemExit('two ranks on one GPU were accepted')\n"
        "    except ValueError as e:\n"
        "        assert 'one GPU per shard' in str(e)\n"
        "try:\n"
        "    monte_carlo(m, 10, qobs=qobs, exchange='rccl', **kw)\n"
        "    raise SystemExit('exchange without the device sampler')\n"
        "except ValueError:\n"
        "    pass\n"
        "print('mc rccl ok')\n" % REPO)
    out = subprocess.run([sys.executable, str(script)], capture_output=True,
                         text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "mc rccl ok" in out.stdout, \
        (out.stdout + out.stderr)[-2000:]


def test_c_abi_allgather_over_rccl_on_one_rank(tmp_path):
    """The collective of the C-ABI itself (include/rrhip.h rr_comm_* /
    rr_allgather_metric: RCCL opened at first use, one group of broadcasts,
    rank r the root of block r) -- with the one rank a single-GPU box can
    host: communicator from a unique id, out of place and in place, the
    block-size check, destroy.  In a process of its own (RCCL and a second
    communicator next to torch's are not what the other tests need)."""
    script = tmp_path / "abi_one_rank.py"
    script.write_text(
        "import ctypes, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from rrmpg_amd import _lib\n"
        "lib = _lib.load()\n"
        "torch.cuda.set_device(0)\n"
        "ident = (ctypes.c_char * 128)()\n"
        "assert lib.rr_comm_unique_id(ident) == 0, lib.rr_last_error()\n"
        "comm = ctypes.c_void_p()\n"
        "assert lib.rr_comm_init(ctypes.byref(comm), 1, 0, ident) == 0, "
        "lib.rr_last_error()\n"
        "n = 1001\n"
        "x = torch.arange(n, dtype=torch.float64, device='cuda') * 0.25\n"
        "y = torch.zeros(n, dtype=torch.float64, device='cuda')\n"
        "st = torch.cuda.current_stream().cuda_stream\n"
        "assert lib.rr_allgather_metric(comm, x.data_ptr(), n, y.data_ptr(),"
        " n, st) == 0, lib.rr_last_error()\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(x, y)\n"
        "z = x.clone()\n"
        "assert lib.rr_allgather_metric(comm, z.data_ptr(), n, z.data_ptr(),"
        " n, st) == 0\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(x, z)\n"
        "assert lib.rr_allgather_metric(comm, x.data_ptr(), n - 1, "
        "y.data_ptr(), n, st) == -2\n"
        "assert b'holds 1000 scores' in lib.rr_last_error()\n"
        "assert lib.rr_comm_destroy(comm) == 0\n"
        "print('abi rccl ok')\n" % REPO)
    out = subprocess.run([sys.executable, str(script)], capture_output=True,
                         text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "abi rccl ok" in out.stdout, \
        (out.stdout + out.stderr)[-2000:]


def test_c_abi_allgather_over_rccl_two_ranks(tmp_path):
    """Two processes, one GPU each, ragged blocks (1001 scores: 501 + 500):
    the id travels through a file, every rank ends with the whole vector."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = tmp_path / "abi_two_ranks.py"
    script.write_text(
        "import ctypes, os, sys, time, torch\n"
        "sys.path.insert(0, %r)\n"
        "from rrmpg_amd import _lib\n"
        "rank, idfile = int(sys.argv[1]), sys.argv[2]\n"
        "lib = _lib.load()\n"
        "torch.cuda.set_device(rank)\n"
        "ident = (ctypes.c_char * 128)()\n"
        "if rank == 0:\n"
        "    assert lib.rr_comm_unique_id(ident) == 0, lib.rr_last_error()\n"
        "    open(idfile + '.tmp', 'wb').write(bytes(ident))\n"
        "    os.rename(idfile + '.tmp', idfile)\n"
        "else:\n"
        "    while not os.path.exists(idfile): time.sleep(0.05)\n"
        "    ident = (ctypes.c_char * 128).from_buffer_copy("
        "open(idfile, 'rb').read())\n"
        "comm = ctypes.c_void_p()\n"
        "assert lib.rr_comm_init(ctypes.byref(comm), 2, rank, ident) == 0, "
        "lib.rr_last_error()\n"
        "n = 1001\n"
        "a, b = ctypes.c_int64(), ctypes.c_int64()\n"
        "lib.rr_shard_bounds(n, 2, rank, ctypes.byref(a), ctypes.byref(b))\n"
        "whole = torch.arange(n, dtype=torch.float64, device='cuda') * 0.5\n"
        "mine = whole[a.value:b.value].clone()\n"
        "out = torch.zeros(n, dtype=torch.float64, device='cuda')\n"
        "st = torch.cuda.current_stream().cuda_stream\n"
        "assert lib.rr_allgather_metric(comm, mine.data_ptr(), mine.numel(), "
        "out.data_ptr(), n, st) == 0, lib.rr_last_error()\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(out, whole)\n"
        "assert lib.rr_comm_destroy(comm) == 0\n"
        "print('rank', rank, 'ok')\n" % REPO)
    idfile = str(tmp_path / "rccl.id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), idfile],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in o, o[-2000:]
