#!/usr/bin/env python
"""Build an A/B variant of the HIP library with extra -D flags:  python tools/build_variant.py NAME -DMM_X=1 ...
-> myosuite_amd/csrc/_variants/NAME/libmyosim_hip.so   (run with MYOSIM_LIB=that path; travels to the GPU box with the snapshot)"""
import concurrent.futures, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_amd import engine as E
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(E.CSRC, "_variants", name); os.makedirs(out, exist_ok=True)
srcs = sorted(os.path.join(E.CSRC, f) for f in os.listdir(E.CSRC) if f.endswith(".hip"))
only = os.environ.get("VARIANT_ONLY")     # e.g. "inst_A,inst_B,engine": other objects are taken from the main build
def comp(src):
    base = os.path.basename(src)[:-4]
    obj = os.path.join(out, base + ".o")
    if only and not any(o in base for o in only.split(",")):
        return os.path.join(E.CSRC, "_build", base + ".o")
    sched = os.environ.get("VARIANT_SCHED") or E.SCHED_STRATEGY.get(os.path.basename(src), E.SCHED_STRATEGY["default"])   # VARIANT_SCHED=none: the compiler's default
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + E.EXTRA_FLAGS + E.FILE_FLAGS.get(os.path.basename(src), []) + flags + \
          (["-mllvm", f"-amdgpu-sched-strategy={sched}"] if "inst" in base and sched != "none" else []) + ["-c", "-o", obj, src]
    subprocess.check_call(cmd)
    return obj
with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count()) as ex:
    objs = list(ex.map(comp, srcs))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libmyosim_hip.so")] + objs)
print(os.path.join(out, "libmyosim_hip.so"))
