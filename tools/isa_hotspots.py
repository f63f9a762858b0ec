#!/usr/bin/env python
"""Static ISA patterns that cost a lone wave dearly, attributed to source lines (no GPU needed).

    python tools/isa_hotspots.py inst_H [kernel index] [--top N]

Compiles the translation unit device-only with the build's flags + -gline-tables-only and reports, per source line of
myosim_engine_kernel.hpp:
  masked-load blocks   basic blocks of <= 12 instructions that hold a load and an s_waitcnt: `cond ? table[i] : 0` / `if (flag[i])
                       x = table[j]` compile to a branch around the load with a full wait behind it -- one serialised LDS / L2
                       round trip per block (round 3: 5.5 k of the 6.3 k cycles of a dense solve were 36 of these)
  SGPR spill restores  v_readlane from the SGPR-spill VGPRs (lane masks shared between inlined copies of a function and kept alive
                       across stages: round 3's dense factor had 690 of them)
Counts are static (loops count once, unrolled code per copy): read them against the stage profile (tools/gpu_perf.py on the
MM_STAGE_PROF build).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_amd import engine as E   # flags only

HDR = os.path.join(E.CSRC, "myosim_engine_kernel.hpp")


def assembly(unit):
    src = os.path.join(E.CSRC, f"myosim_{unit}.hip")
    base = os.path.basename(src)
    sched = E.SCHED_STRATEGY.get(base, E.SCHED_STRATEGY["default"])
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), unit + ".s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "--no-gpu-bundle-output", "-gline-tables-only"] + \
          E.EXTRA_FLAGS + E.FILE_FLAGS.get(base, []) + ["-mllvm", f"-amdgpu-sched-strategy={sched}", "-S", "-o", out, src]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernels(lines):
    start = [i for i, l in enumerate(lines) if l.startswith("_Z8k_engine")] + [len(lines)]
    return [(lines[start[i]].split(":")[0], lines[start[i]:start[i + 1]]) for i in range(len(start) - 1)]


def masked_loads(body):
    blocks, cur, line = [], [], None
    for l in body:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            line = int(m.group(2)); continue
        if not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
            continue
        if t.startswith(".LBB"):
            if cur: blocks.append(cur); cur = []
            continue
        cur.append((t, line))
        if t.startswith(("s_cbranch", "s_branch")):
            blocks.append(cur); cur = []
    c = collections.Counter()
    for b in blocks:
        ins = [x[0] for x in b]
        loads = [x for x in b if x[0].startswith(("ds_read", "global_load", "scratch_load"))]
        if len(ins) <= 12 and loads and any(x.startswith("s_waitcnt") for x in ins):
            c[loads[0][1]] += 1
    return c


def sgpr_restores(body):
    wl = collections.Counter()
    for l in body:
        m = re.match(r"\s+v_writelane_b32 (v\d+),", l)
        if m: wl[m.group(1)] += 1
    spill = {v for v, n in wl.items() if n >= 8}
    line, last, rd = None, None, collections.Counter()
    for l in body:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            line = int(m.group(2))
            if line: last = line
            continue
        m = re.match(r"v_readlane_b32 s\d+, (v\d+), \d+", t)
        if m and m.group(1) in spill: rd[last] += 1
    return rd, sum(n for v, n in wl.items() if v in spill)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 20
    unit = args[0] if args else "inst_H"
    ks = kernels(assembly(unit))
    src = open(HDR).read().split("\n")
    for ki, (name, body) in enumerate(ks):
        if len(args) > 1 and ki != int(args[1]): continue
        try:
            name = subprocess.check_output(["c++filt", name], text=True).strip()
        except Exception:
            pass
        ml = masked_loads(body)
        rd, nsp = sgpr_restores(body)
        print(f"== [{ki}] {name}: {sum(ml.values())} masked-load blocks, {nsp} SGPR spills / {sum(rd.values())} restores")
        for title, c in (("masked-load blocks", ml), ("SGPR spill restores", rd)):
            print(f"  -- {title}")
            for ln, n in c.most_common(top):
                print(f"  {n:5d}  {ln}: {src[ln - 1].strip()[:120] if ln and ln <= len(src) else ''}")
