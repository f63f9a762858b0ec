"""Single configuration run for rocprofv3: python tools/gpu_prof_one.py hand 32 4096 [nlaunch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
name, lanes, nenv = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cm = synth.get_model(name)
hm = E.HipModel(cm, lanes_per_env=lanes)
st = E.BatchState(hm, nenv)
a = torch.rand(nenv, cm.nu, device="cuda")
for _ in range(n):
    E.step(hm, st, a, 10)
torch.cuda.synchronize()
