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


def loop_lane_moves(body):
    """(all, hot): v_readlane / v_writelane instructions inside the kernel's
    INNERMOST loops (the day loops) -- in any of their blocks, and in the
    blocks laid out between a loop's header and its back edge only (the
    unlikely slow blocks are placed behind the back edge, out of line).  The
    asm comments of hipcc name every block's loop: "=>This Inner Loop Header"
    / "in Loop: Header=BBx_y"."""
    blocks, cur = [], None
    for line in body.splitlines():
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", line)
        m2 = re.match(r"^; %bb\.\d+:(.*)$", line)
        if m or m2:
            cur = {"label": m.group(1)[2:] if m else None,      # "BBx_y"
                   "note": (m or m2).group(2) if m else m2.group(1),
                   "lines": []}
            blocks.append(cur)
        elif cur is not None:
            cur["lines"].append(line)
    inner = set()
    for k, b in enumerate(blocks):
        note = b["note"]
        # the comment may continue on the following comment line
        if "This Inner Loop Header" in note or (
                b["lines"] and "This Inner Loop Header" in b["lines"][0]):
            inner.add(b["label"])
    total = hot = 0
    for hdr in inner:
        members = [k for k, b in enumerate(blocks)
                   if b["label"] == hdr or
                   ("Header=%s " % hdr) in (b["note"] + " ") or
                   any(("Header=%s " % hdr) in (l + " ") for l in b["lines"][:1])]
        if not members:
            continue
        latch = None
        for k in members:
            for l in blocks[k]["lines"]:
                # (labels are ".LBBx_y", `hdr` is "BBx_y"; until round 5 the
                # pattern lacked the L, no latch was ever found and every block
                # of a loop counted as its straight path)
                if re.search(r"s_c?branch\w*\s+\.L%s\b" % re.escape(hdr), l):
                    latch = k if latch is None else max(latch, k)
        for k in members:
            n = sum(1 for l in blocks[k]["lines"]
                    if re.search(r"v_(?:read|write)lane_b32", l))
            total += n
            if latch is None or k <= latch:
                hot += n
    return total, hot


def outer_loop_moves(body):
    """Lane moves of every OUTERMOST loop of more than 200 instructions --
    the copies of a kernel's time loop (a kernel carries two: the one its
    sane / tame waves run, and the general one) --, in code order: which copy
    the moves of `in day loops` sit in."""
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    spans = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            spans.append((labels[m.group(1)], i))
    spans.sort(key=lambda ab: (ab[0], -ab[1]))
    outer, end = [], -1
    for a, b in spans:
        if a > end:
            outer.append([a, b])
            end = b
        elif b > end:               # overlapping back edges: one region
            outer[-1][1] = b
            end = b
    out = []
    for a, b in outer:
        seg = lines[a:b + 1]
        if sum(1 for l in seg if re.match(r"\s+[vs]_", l)) > 200:
            out.append(sum(1 for l in seg
                           if re.search(r"v_(?:read|write)lane_b32", l)))
    return out


# per-file flags of rrmpg_amd/csrc/Makefile (FLAGS_<file>)
FILE_FLAGS = {"snownext_hyst.hip": ["-mllvm",
                                    "-amdgpu-sched-strategy=iterative-ilp"]}


def kernels_of(path):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", *FLAGS,
                        *FILE_FLAGS.get(os.path.basename(path), []), "-o",
                        asm, path], check=True, capture_output=True)
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
            loop_moves=loop_lane_moves(body),
            outer_moves=outer_loop_moves(body),
            scratch_instrs=len(re.findall(r"\bscratch_(?:load|store)", body)),
            code=num(r"; codeLenInByte = (\d+)", info)))
    return rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r00"
    wanted = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    lines = ["# Kernel resources from the gfx950 code object (%s)" % tag, "",
             "hipcc flags: `%s`.  Occupancy = waves per SIMD the register "
             "allocation permits (512 VGPRs per lane and SIMD); `lane moves` = "
             "v_readlane/v_writelane instructions in the kernel (SGPR spills "
             "to VGPR lanes; each executed one is a VALU slot); `in day "
             "loops` = those of them inside the innermost loops, all blocks "
             "/ the blocks on the loops' straight path only (the unlikely "
             "slow blocks lie behind the back edge): what a wave executes "
             "per day, against what it executes once per launch or work "
             "item.  `scratch B (instr)`: the private segment the kernel "
             "reserves and the scratch_load / scratch_store instructions in "
             "its code (a reservation nobody touches costs nothing).  `per "
             "time-loop copy`: lane moves inside each outermost loop of more "
             "than 200 instructions, in code order -- a kernel with a sane / "
             "tame copy of its time loop runs the first one for every wave "
             "whose sets and forcing are civil." %
             " ".join(FLAGS[:-2]), "",
             "| file | kernel | VGPRs | SGPRs | LDS B | scratch B (instr) | "
             "waves/SIMD | lane moves | in day loops (all / straight path) | "
             "per time-loop copy | code B |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    count, code_total = 0, 0
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
            count += 1
            code_total += r["code"] or 0
            lines.append("| %s | `%s` | %s | %s | %s | %s (%d) | %s | %s | "
                         "%d / %d | %s | %s |" % (
                f, nice, r["vgpr"], r["sgpr"], r["lds"], r["scratch"],
                r["scratch_instrs"], r["occupancy"], r["lane_moves"],
                r["loop_moves"][0], r["loop_moves"][1],
                " / ".join(str(n) for n in r["outer_moves"]) or "-",
                r["code"]))
    cond = 0
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, f)) as fh:
                cond += len(re.findall(r"^\s*#\s*(?:if|ifdef|ifndef|elif)\b",
                                       fh.read(), flags=re.M))
    lines += ["", "%d kernel instantiations, %d bytes of code; %d preprocessor "
              "conditionals (#if / #ifdef / #ifndef / #elif) in "
              "rrmpg_amd/csrc (include guards and the host / device "
              "branches of fastmath.h and invdiv.h among them)."
              % (count, code_total, cond)]
    out = os.path.join(REPO, "profiles", "%s_isa_metadata.md" % tag)
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print(out)


if __name__ == "__main__":
    main()
