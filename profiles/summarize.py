#!/usr/bin/env python3
"""Condense a profiles/collect.sh run (gpurun_out/prof_<tag>/) into the small
tracked summary profiles/<tag>_summary.{json,md}.

HBM traffic per launch follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are reported in KiB (x1024); on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so the read side is doubled.  Each
counter comes from its own pass."""
import collections
import csv
import glob
import json
import os
import sys


MIN_GRID = 0      # launches with fewer work-items are ignored (argv[4])


def counters(path, match):
    acc = collections.defaultdict(list)
    dur = []
    meta = {}
    for f in glob.glob(os.path.join(path, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"] and float(r["Grid_Size"]) >= MIN_GRID:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size",
                                          "LDS_Block_Size", "Scratch_Size",
                                          "VGPR_Count", "SGPR_Count")}
    return ({k: sum(v) / len(v) for k, v in acc.items()},
            (sum(dur) / len(dur) / 1e6) if dur else None, meta)


def trace_stats(root, match):
    """Launch durations of the matching kernel from the kernel trace (the
    stats CSV cannot tell a one-set helper launch of the same kernel from the
    sweep)."""
    dur = []
    name = None
    for f in glob.glob(os.path.join(root, "trace", "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"] and float(r["Grid_Size_X"]) >= MIN_GRID:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                name = r["Kernel_Name"]
    if not dur:
        return None
    return {"name": name, "calls": len(dur), "avg_ms": sum(dur) / len(dur) / 1e6,
            "min_ms": min(dur) / 1e6, "max_ms": max(dur) / 1e6}


def main():
    global MIN_GRID
    tag, match = sys.argv[1], sys.argv[2]
    algo_bytes = float(sys.argv[3]) if len(sys.argv) > 3 else None
    MIN_GRID = float(sys.argv[4]) if len(sys.argv) > 4 else 0
    # argv[5]: the collection to read if it is not the tag's own (several
    # kernels of one run, each with a summary of its own)
    root = os.path.join("gpurun_out", "prof_" + (sys.argv[5] if len(sys.argv)
                                                  > 5 else tag))
    out = {"tag": tag, "kernel_match": match}
    stats = glob.glob(os.path.join(root, "trace", "*", "*_kernel_stats.csv"))
    rows = []
    if stats:
        for r in csv.DictReader(open(stats[0])):
            rows.append(r)
            if match in r["Name"] and "kernel_stats" not in out:
                out["kernel_stats"] = {
                    "name": r["Name"], "calls": int(r["Calls"]),
                    "avg_ms": float(r["AverageNs"]) / 1e6,
                    "min_ms": float(r["MinNs"]) / 1e6,
                    "max_ms": float(r["MaxNs"]) / 1e6,
                    "pct_of_gpu_time": float(r["Percentage"])}
    if MIN_GRID:
        ts = trace_stats(root, match)
        if ts:
            out["kernel_stats"] = ts
    fetch, d1, meta = counters(os.path.join(root, "pmc_fetch"), match)
    write, d2, _ = counters(os.path.join(root, "pmc_write"), match)
    sq, d3, _ = counters(os.path.join(root, "pmc_sq"), match)
    out["launch"] = meta
    if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
        rd = fetch["FETCH_SIZE"] * 1024 * 2     # gfx950: x2 (see docstring)
        wr = write["WRITE_SIZE"] * 1024
        out["hbm"] = {"FETCH_SIZE_KiB": fetch["FETCH_SIZE"],
                      "WRITE_SIZE_KiB": write["WRITE_SIZE"],
                      "read_bytes_corrected": rd, "write_bytes": wr,
                      "traffic_bytes_per_launch": rd + wr,
                      "avg_ms_fetch_pass": d1, "avg_ms_write_pass": d2}
        if algo_bytes:
            out["hbm"]["algorithmic_bytes_per_launch"] = algo_bytes
            out["hbm"]["traffic_over_algorithmic"] = (rd + wr) / algo_bytes
    if sq:
        out["sq"] = dict(sq)
        out["sq"]["avg_ms"] = d3
        if "SQ_INSTS_VALU" in sq and "SQ_WAVES" in sq:
            out["sq"]["valu_insts_per_wave"] = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"]
        if "SQ_ACTIVE_INST_VALU" in sq and d3:
            # quad-cycles -> cycles; 1024 SIMDs; assumes 2.4 GHz (upper bound
            # on available cycles -> a lower bound on utilisation)
            cyc = sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024
            out["sq"]["valu_active_cycles_per_simd"] = cyc
            out["sq"]["valu_utilisation_at_2.4GHz"] = cyc / (d3 * 1e-3 * 2.4e9)
    with open(os.path.join("profiles", tag + "_summary.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join("profiles", tag + "_kernel_stats.md"), "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats (%s)\n\n" % tag)
        fh.write("| kernel | calls | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|\n")
        for r in rows[:8]:
            fh.write("| `%s` | %s | %.3f | %.3f | %.3f | %s |\n" % (
                r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6,
                float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6,
                r["Percentage"]))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
