#!/usr/bin/env python3
"""profiles/traffic.json from the round's counter summaries
(profiles/<tag>_summary.json, written by summarize.py from the separate
rocprofv3 --pmc passes of collect.sh): HBM bytes per launch and VALU
instructions per set-day for every workload bench.py reports.

    python profiles/make_traffic.py r04
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from rrmpg_amd.utils.buildid import kernel_source_id  # noqa: E402
T = 10957
# tag suffix -> (traffic.json key, waves' worth of sets in one launch)
WORKLOADS = {
    "": ("hbvedu:qsim:1000000:%d" % T, 15625),
    "_hbv125k": ("hbvedu:qsim:125000:%d" % T, 1954),
    "_hbv100k": ("hbvedu:qsim:100000:%d" % T, 1563),
    "_hbv400ks": ("hbvedu:storages:400000:%d" % T, 6250),
    "_hbvcat": ("hbvedu:metric:125x10000:%d" % T, 125 * 157),
    "_gr4j": ("gr4j:qsim:1000000:%d" % T, 15625),
    "_gr4j125k": ("gr4j:metric:125000:%d" % T, 1954),
    "_fused125k": ("cemaneigegr4j:metric:125000:%d" % T, 1954),
    "_cema": ("cemaneige:qsim:1000000:%d" % T, 15625),
    "_abc": ("abc:qsim:1000000:%d" % T, 15625),
    "_hyst": ("cemaneigehystgr4j:metric:1000000:%d" % T, 15625),
    "_ice": ("cemaneigegr4jice:metric:1000000:%d" % T, 15625),
    "_hystice": ("cemaneigehystgr4jice:metric:1000000:%d" % T, 15625),
}


def main():
    tag = sys.argv[1]
    out = {"_source": "profiles/%s*_summary.json (rocprofv3 --pmc FETCH_SIZE / "
                      "WRITE_SIZE, separate passes; bytes = (2*FETCH_SIZE + "
                      "WRITE_SIZE)*1024, FETCH x2 per the gfx950 note of "
                      "MI355X_MICROARCH.md); includes the time tiles' state "
                      "hand-over" % tag,
           "_source_valu": "the same summaries: SQ_INSTS_VALU of the sweep "
                           "kernel (x SQ_WAVES) / the waves' worth of sets / "
                           "%d days (rocprofv3 --pmc, own pass)" % T}
    # the kernel sources the counters were collected on (bench.py quotes a
    # model's numbers only while its sources are these)
    out["_build"] = {m: kernel_source_id(m) for m in sorted(
        {key.split(":")[0] for key, _ in WORKLOADS.values()})}
    for suffix, (key, jobs) in WORKLOADS.items():
        path = os.path.join(HERE, "%s%s_summary.json" % (tag, suffix))
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        hbm, sq = d.get("hbm"), d.get("sq", {})
        if hbm:
            out[key] = hbm["traffic_bytes_per_launch"]
        if "SQ_INSTS_VALU" in sq:
            valu = sq["SQ_INSTS_VALU"]
            # a sweep whose waves run in several tier kernels side by side
            # (the hysteresis couplings): the other tiers' summaries
            for extra in ("_t3", "_t5"):
                pe = os.path.join(HERE, "%s%s%s_summary.json"
                                  % (tag, suffix, extra))
                if suffix and os.path.exists(pe):
                    valu += json.load(open(pe)).get("sq", {}).get(
                        "SQ_INSTS_VALU", 0)
            out[key + ":valu_instr_per_unit"] = round(valu / (jobs * T), 2)
    with open(os.path.join(HERE, "traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
