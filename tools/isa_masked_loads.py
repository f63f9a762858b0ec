#!/usr/bin/env python
"""Small basic blocks that hold an LDS / global load and a wait -- `cond ? table[i] : 0` compiled as a branch around the load: one
serialised round trip each -- by source line of myosim_engine_body.inc (static count; the assembly is tools/isa_class_histogram.py's).
    python tools/isa_masked_loads.py inst_B "32, 24, true" [--top N]"""
import collections, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_class_histogram as H
args = [a for a in sys.argv[1:] if not a.startswith("--")]
unit, pick = args[0], args[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
lines = H.assembly(unit)
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+(?:"([^"]*)"\s+)?"([^"]*)"', l)
    if m: files[int(m.group(1))] = m.group(3)
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*8k_engine\w*:", l)] + [len(lines)]
kern = [(lines[start[i]].split(":")[0], lines[start[i]:start[i + 1]]) for i in range(len(start) - 1)]
def dem(n):
    m = re.match(r"_Z(?:N4mm64)?8k_engineI((?:L[ib]\d+E)+)E", n)
    a = [(("true" if v == "1" else "false") if t == "b" else v) for t, v in re.findall(r"L([ib])(\d+)E", m.group(1))]
    return "k_engine<" + ", ".join(a) + ">"
ki = next(i for i, k in enumerate(kern) if pick in dem(k[0]))
body = kern[ki][1]
starts = H.stage_ranges(); sl = [s[0] for s in starts]
import bisect
src = open(H.BODY).read().split("\n")
blocks, cur, loc = [], [], None
for l in body:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        f = files.get(int(m.group(1)), "")
        loc = int(m.group(2)) if f.endswith("myosim_engine_body.inc") else loc
        continue
    if not t or t.startswith((";", "//")) or (t.startswith(".") and not t.startswith(".LBB")):
        continue
    if t.startswith(".LBB") or t.endswith(":"):
        if cur: blocks.append(cur); cur = []
        continue
    cur.append((t, loc))
    if t.startswith(("s_cbranch", "s_branch")):
        blocks.append(cur); cur = []
c = collections.Counter()
for b in blocks:
    ins = [x[0] for x in b]
    loads = [x for x in b if x[0].startswith(("ds_read", "global_load", "s_load", "s_buffer_load"))]
    if len(ins) <= 16 and loads and any(x.startswith("s_waitcnt") for x in ins):
        c[loads[0][1]] += 1
print(f"# {dem(kern[ki][0])}: {sum(c.values())} small load + wait blocks of {len(blocks)} basic blocks")
for ln, n in c.most_common(top):
    fn = starts[bisect.bisect_right(sl, ln) - 1][1] if ln else "?"
    print(f"{n:4d}  line {ln} [{fn}]: {src[ln - 1].strip()[:110] if ln else ''}")
