"""Authoring-time extraction of the object-size tables (DATA: 4 shape types x 250 rows each) of the reference's
InDistribution / OutofDistribution reorient test envs (myosuite/envs/myo/myobase/reorient_sar_v0.py:440-2590) into
myosuite_amd/envs/data/reorient_tables.npz.  Needs /root/reference; the product only reads the committed .npz."""
import os, re, sys
import numpy as np
SRC = "/root/reference/myosuite/envs/myo/myobase/reorient_sar_v0.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myosuite_amd", "envs", "data", "reorient_tables.npz")
s = open(SRC).read()
out = {}
for cls, tag in (("InDistribution", "ID"), ("OutofDistribution", "OOD")):
    i = s.index(f"class {cls}")
    m = re.search(r"\nclass ", s[i + 10:]); j = i + 10 + m.start() if m else len(s)
    blk = s[i:j]
    tabs = []
    for name in ("caps", "ellips", "cyl", "box"):          # geom types 3, 4, 5, 6
        a = blk.index(f"{name} = {{"); b = blk.index("\n        }", a)
        rows = re.findall(r"\d+: \[\[([^\]]*)\]", blk[a:b])
        tabs.append([[float(x) for x in r.split(",")] for r in rows])
    out[tag] = np.array(tabs, np.float32)
    print(tag, out[tag].shape)
np.savez_compressed(OUT, **out)
