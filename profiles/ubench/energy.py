#!/usr/bin/env python3
"""Driver of profiles/ubench/energy.hip: builds it, runs every mode for a few
seconds and reads the GPU's socket power and shader clock from hwmon (bench.py's
SocketSampler) over the second half of each run.

    python profiles/ubench/energy.py [seconds] > profiles/r06_energy.txt
"""
import os
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import bench  # noqa: E402

MODES = {0: "32 v_fma_f64 per trip",
         1: "+ scalar mix (64-B s_load burst, 12 SALU)",
         2: "+ one nt sc1 8-B store per lane and trip",
         3: "+ two LDS table reads per trip",
         4: "the stores alone (no arithmetic)",
         5: "24 FMAs + scalar mix + stores",
         6: "stores alone, 16 B per lane (1 KiB per wave-row)",
         7: "two sets per lane: 64 FMAs + scalar + 16-B store",
         10: "32 v_add_f64 per trip", 11: "32 v_mul_f64 per trip",
         12: "32 v_max_f64 per trip", 13: "32 v_cndmask_b32 (VOP3) per trip",
         14: "32 v_add_u32 per trip", 15: "8 v_rcp_f64 per trip"}
INSTR_PER_TRIP = {0: 32, 10: 32, 11: 32, 12: 32, 13: 32, 14: 32, 15: 8}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    exe = os.path.join("/tmp", "rr_energy")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-o", exe,
                           os.path.join(HERE, "energy.hip")])
    idle = bench.SocketSampler(0)
    with idle:
        time.sleep(1.5)
    rec = idle.record()
    print("# idle: %s" % (rec and {k: round(v, 1) for k, v in rec.items()
                                   if isinstance(v, float)}))
    print("# mode | what | socket W | sclk MHz | wave-trips/s | stored GB/s |"
          " cycles per trip per SIMD")
    only = [int(m) for m in sys.argv[2:]]
    for mode, what in MODES.items():
        if only and mode not in only:
            continue
        s = bench.SocketSampler(0)
        out = {}

        def work():
            out["txt"] = subprocess.run([exe, str(mode), str(seconds)],
                                        capture_output=True,
                                        text=True).stdout.strip()
        th = threading.Thread(target=work)
        with s:
            th.start()
            time.sleep(seconds * 0.5)
            half = (len(s.power), len(s.clock))
            th.join()
        s.power, s.clock = s.power[half[0]:], s.clock[half[1]:]
        r = s.record() or {}
        f = dict(zip(out["txt"].split()[0::2], out["txt"].split()[1::2]))
        wt = float(f.get("wave_trips_per_s", "nan"))
        mhz = r.get("sclk_mhz") or float("nan")
        # 1024 SIMDs: cycles a SIMD spends per wave-trip
        cyc = mhz * 1e6 * 1024 / wt if wt else float("nan")
        extra = ""
        if mode in INSTR_PER_TRIP and rec and wt:
            # dynamic power / lane-operations per second
            pj = ((r.get("socket_w", 0) - rec["socket_w"])
                  / (wt * INSTR_PER_TRIP[mode] * 64) * 1e12)
            extra = " | %.1f pJ per lane-op, %.2f cycles per instr" % (
                pj, cyc / INSTR_PER_TRIP[mode])
        print("%d | %-44s | %7.1f | %7.1f | %.4e | %7.1f | %6.1f%s"
              % (mode, what, r.get("socket_w", float("nan")), mhz, wt,
                 float(f.get("stored_GBps", "nan")), cyc, extra), flush=True)
        time.sleep(1.0)


if __name__ == "__main__":
    main()
