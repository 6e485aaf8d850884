#!/usr/bin/env python3
"""Kernel time of one resident sweep with a given build of the library:

    python profiles/ab_time.py --lib exp/librrhip_x.so --model hbvedu \
        --mode qsim --sets 100000 [--steps 30] [--tag x]

prints one line: tag, sets, mean / min kernel ms over the timed sweeps (HIP
events around the library call, bench.py's own measurement) and a checksum
of the scores and of qsim (two builds that claim the same bits must print
the same checksums).  Measurement tool: nothing of the product imports it.
"""
import argparse
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--tag", default=None)
    ap.add_argument("--model", default="hbvedu")
    ap.add_argument("--mode", default="qsim")
    ap.add_argument("--sets", type=int, nargs="+", default=[100000])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--days", type=int, default=10957)
    ap.add_argument("--score", default="mse")
    ap.add_argument("--hbv-variant", type=int, default=-1)
    ap.add_argument("--time-tiles", type=int, default=-1)
    ap.add_argument("--fused-variant", type=int, default=0)
    ap.add_argument("--gr4j-variant", type=int, default=0)
    ap.add_argument("--catchments", type=int, default=0)
    ap.add_argument("--warm-records", type=int, default=-1)
    ap.add_argument("--force-lds", type=int, default=0)
    a = ap.parse_args()
    from rrmpg_amd import _lib
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    import numpy as np
    import torch
    import bench
    lib = _lib.load()
    tag = a.tag or (os.path.basename(a.lib) if a.lib else "default")
    for n in a.sets:
        args = argparse.Namespace(
            gpus=1, model=a.model, mode=a.mode, sets=n, days=a.days,
            catchments=a.catchments, scaling="strong", sampler="device",
            row_pitch=0, score=a.score, hbv_variant=a.hbv_variant,
            time_tiles=a.time_tiles, fused_variant=a.fused_variant,
            gr4j_variant=a.gr4j_variant)
        if a.hbv_variant >= 0:
            lib.rr_debug_set_option(_lib.OPTIONS["hbv_variant"], a.hbv_variant)
        if a.time_tiles >= 0:
            lib.rr_debug_set_option(_lib.OPTIONS["time_tiles"], a.time_tiles)
        if a.fused_variant:
            lib.rr_debug_set_option(_lib.OPTIONS["fused_variant"],
                                    a.fused_variant)
        lib.rr_debug_set_option(_lib.OPTIONS["warm_records"], a.warm_records)
        lib.rr_debug_set_option(_lib.OPTIONS["gr4j_force_lds"], a.force_lds)
        if a.gr4j_variant:
            lib.rr_debug_set_option(_lib.OPTIONS["gr4j_variant"],
                                    a.gr4j_variant)
        dev = torch.device("cuda:0")
        r = bench.run_workload(args, dev, 0, 1, False, a.steps, a.warmup,
                               score=a.score)
        sweep = r["sweep"]
        ev = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        for e0, e1 in ev:
            e0.record(); sweep.launch(); e1.record()
        torch.cuda.synchronize()
        ms = np.array([e0.elapsed_time(e1) for e0, e1 in ev])
        scores = sweep.gather().cpu().numpy()
        h = hashlib.sha1(scores.tobytes()).hexdigest()[:10]
        hq = "-"
        if r["qsim"] is not None:
            q = r["qsim"]
            cols = q[:, :: max(1, q.shape[1] // 64)].contiguous().cpu().numpy()
            hq = hashlib.sha1(cols.tobytes()).hexdigest()[:10]
        tag_ = tag + ("" if a.warm_records < 0 else "/warm%d" % a.warm_records) \
            + ("" if a.hbv_variant < 0 else "/v%d" % a.hbv_variant)
        print("AB tag=%s model=%s mode=%s sets=%d mean_ms=%.4f min_ms=%.4f "
              "first_run_mean=%.4f scores=%s qsim=%s"
              % (tag_, a.model, a.mode, n, ms.mean(), ms.min(),
                 r["kernel_ms"], h, hq), flush=True)
        del r, sweep
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
