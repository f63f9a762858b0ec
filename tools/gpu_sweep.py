"""Throughput sweep over lanes-per-env (raw mm_step, 10 substeps). python tools/gpu_sweep.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
for name, lanes_list in (("elbow", (4, 8, 16)), ("hand", (8, 16, 32, 64))):
    cm = synth.get_model(name)
    for lanes in lanes_list:
        try:
            hm = E.HipModel(cm, lanes_per_env=lanes)
            for nenv in (4096, 16384):
                st = E.BatchState(hm, nenv)
                a = torch.rand(nenv, cm.nu, device="cuda")
                for _ in range(2):
                    E.step(hm, st, a, 10)
                torch.cuda.synchronize()
                t = time.time(); n = 10
                for _ in range(n):
                    E.step(hm, st, a, 10)
                torch.cuda.synchronize()
                dt = (time.time() - t) / n
                print(f"{name} lanes={lanes} nenv={nenv}: {dt*1e3:.3f} ms/10sub -> {nenv/dt/1e6:.3f} M env-steps/s", flush=True)
        except Exception as ex:
            print(name, lanes, "ERR", ex)
