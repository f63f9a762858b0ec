"""Throughput + in-kernel stage profile. python tools/gpu_sweep.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E

def bench(hm, cm, nenv, nsub=10, n=8):
    st = E.BatchState(hm, nenv)
    a = torch.rand(nenv, cm.nu, device="cuda")
    for _ in range(2):
        E.step(hm, st, a, nsub)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        E.step(hm, st, a, nsub)
    torch.cuda.synchronize()
    return (time.time() - t) / n

cfgs = {"elbow": [(4, 1, 0), (8, 1, 0)], "hand": [(32, 1, 0), (64, 1, 0)]}
for name, lst in cfgs.items():
    cm = synth.get_model(name)
    for lanes, lm, wpb in lst:
        try:
            hm = E.HipModel(cm, lanes_per_env=lanes)
            hm.set_option("lds_model", lm); hm.set_option("waves_per_block", wpb)
            for nenv in (4096, 32768):
                dt = bench(hm, cm, nenv)
                print(f"{name} lanes={lanes} ldsmodel={lm} wpb={wpb} nenv={nenv}: {dt*1e3:.3f} ms/10sub -> {nenv/dt/1e6:.3f} M env-steps/s", flush=True)
            st = E.BatchState(hm, 4096)
            a = torch.rand(4096, cm.nu, device="cuda")
            E.step(hm, st, a, 30)
            pr = E.profile_stages(lambda: E.step(hm, st, a, 10))
            print("   stage cycles/substep:", {k: v // 10 for k, v in pr.items()}, flush=True)
        except Exception as ex:
            print(name, lanes, lm, "ERR", ex)
