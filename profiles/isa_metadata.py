#!/usr/bin/env python3
"""Per-kernel resource usage straight from the code object hipcc produces
(rocprofv3's VGPR_Count field is not the allocation): compiles every .hip of
rrmpg_amd/csrc to gfx950 assembly with the library's own flags and lists
VGPRs, SGPRs, LDS, scratch, the resulting waves per SIMD, the number of SGPR
spill moves (v_readlane / v_writelane) and VGPR spills to scratch.

    python profiles/isa_metadata.py [tag]      -> profiles/<tag>_isa_metadata.md
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "rrmpg_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fPIC", "-S", "--cuda-device-only"]


def demangle(names):
    out = subprocess.run(["c++filt"],
                         input="\n".join(names), capture_output=True,
                         text=True).stdout.splitlines()
    return dict(zip(names, out))


def kernels_of(path):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", *FLAGS, "-o", asm, path], check=True,
                       capture_output=True)
        text = open(asm).read()
    rows = []
    # one function body per .amdhsa_kernel; the "; Kernel info" comment block
    # follows the metadata
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)\.amdhsa_kernel \1\n(.*?)"
                         r"\.end_amdhsa_kernel(.*?); COMPUTE_PGM_RSRC2:SCRATCH_EN",
                         text, flags=re.S | re.M):
        name, body, meta, info = m.groups()

        def num(pat, src):
            g = re.search(pat, src)
            return int(g.group(1)) if g else None
        rows.append(dict(
            name=name,
            vgpr=num(r"\.amdhsa_next_free_vgpr (\d+)", meta),
            sgpr=num(r"; TotalNumSgprs: (\d+)", info),
            lds=num(r"\.amdhsa_group_segment_fixed_size (\d+)", meta),
            scratch=num(r"; ScratchSize: (\d+)", info),
            occupancy=num(r"; Occupancy: (\d+)", info),
            lane_moves=len(re.findall(r"v_(?:read|write)lane_b32", body)),
            code=num(r"; codeLenInByte = (\d+)", info)))
    return rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r00"
    wanted = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    lines = ["# Kernel resources from the gfx950 code object (%s)" % tag, "",
             "hipcc flags: `%s`.  Occupancy = waves per SIMD the register "
             "allocation permits (512 VGPRs per lane and SIMD); `lane moves` = "
             "v_readlane/v_writelane instructions in the kernel (SGPR spills "
             "to VGPR lanes; each executed one is a VALU slot)." %
             " ".join(FLAGS[:-2]), "",
             "| file | kernel | VGPRs | SGPRs | LDS B | scratch B | waves/SIMD "
             "| lane moves | code B |", "|---|---|---|---|---|---|---|---|---|"]
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        rows = kernels_of(os.path.join(CSRC, f))
        names = demangle([r["name"] for r in rows])
        for r in rows:
            nice = names[r["name"]]
            nice = re.sub(r"\(.*", "", nice).replace("void ", "")
            if wanted and not wanted.search(nice):
                continue
            lines.append("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s |" % (
                f, nice, r["vgpr"], r["sgpr"], r["lds"], r["scratch"],
                r["occupancy"], r["lane_moves"], r["code"]))
    out = os.path.join(REPO, "profiles", "%s_isa_metadata.md" % tag)
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print(out)


if __name__ == "__main__":
    main()
