import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.model import synth
case = sys.argv[1]
base = sys.argv[2] if len(sys.argv) > 2 else "hand_contact"
cm = synth.compile_spec(base, edit=lambda s: setattr(s, "nconmax", 2))
hm = E.HipModel(cm)
print("lanes", hm.launch_lanes(8), "njmax", cm.njmax, flush=True)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if case.startswith("lm0"):
    hm.set_option("lds_model", 0)
if case.startswith("lm2"):
    hm.set_option("lds_model", 2)
st = E.BatchState(hm, n)
a = torch.rand(n, cm.nu, device="cuda")
if "fwd" in case:
    E.forward(hm, st, a); torch.cuda.synchronize(); print(case, "forward ok", flush=True)
if "dump" in case:
    d = E.debug_dump(hm, st, a); torch.cuda.synchronize(); print(case, "dump ok", float(d.abs().max()), flush=True)
if "step" in case:
    for k in range(20):
        E.step(hm, st, a, 5); torch.cuda.synchronize()
    print(case, "step ok", int(st.status.max()), flush=True)
