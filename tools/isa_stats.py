#!/usr/bin/env python
"""Static resource / instruction-mix report of the compiled k_engine instantiations (no GPU needed).

    python tools/isa_stats.py [inst_B] [--mix]      # default: every myosim_inst_*.hip

Compiles the translation unit device-only for gfx950 with the build's flags, reads the AMDGPU metadata notes (VGPRs, SGPRs,
spills, scratch, LDS) and -- with --mix -- the disassembly's instruction mix of each kernel.
"""
import os
import re
import subprocess
import sys
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_amd import engine as E   # flags only

LLVM = "/opt/rocm/lib/llvm/bin"


def device_object(src, out):
    base = os.path.basename(src)
    sched = E.SCHED_STRATEGY.get(base, E.SCHED_STRATEGY["default"])
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "--no-gpu-bundle-output"] + E.EXTRA_FLAGS + E.FILE_FLAGS.get(base, []) + \
          ["-mllvm", f"-amdgpu-sched-strategy={sched}", "-c", "-o", out, src]
    subprocess.check_call(cmd)


def notes(obj):
    txt = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", obj], text=True)
    kernels = []
    cur = None
    for line in txt.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":
            cur = {"agpr_count": v}; kernels.append(cur)
        elif cur is not None and k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                       "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v
    return kernels


def demangle(n):
    try:
        return subprocess.check_output([f"{LLVM}/llvm-cxxfilt", n], text=True).strip()
    except Exception:
        return n


def mix(obj):
    txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", obj], text=True)
    out = {}
    cur = None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = collections.Counter(); out[m.group(1)] = cur
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if m and cur is not None:
            cur[m.group(1)] += 1
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want_mix = "--mix" in sys.argv
    csrc = os.path.join(ROOT, "myosuite_amd", "csrc")
    srcs = sorted(f for f in os.listdir(csrc) if f.startswith("myosim_inst_") and f.endswith(".hip"))
    if args:
        srcs = [f for f in srcs if any(a in f for a in args)]
    os.makedirs("/tmp/isa", exist_ok=True)
    for f in srcs:
        obj = f"/tmp/isa/{f[:-4]}.co"
        device_object(os.path.join(csrc, f), obj)
        mx = mix(obj) if want_mix else {}
        for k in notes(obj):
            name = demangle(k.get("name", "?"))
            name = re.sub(r"void |\(KArgs\)", "", name)
            print(f"{f[:-4]:14s} {name:46s} vgpr {k.get('vgpr_count'):>4s} sgpr {k.get('sgpr_count'):>4s} "
                  f"vspill {k.get('vgpr_spill_count'):>4s} sspill {k.get('sgpr_spill_count'):>4s} scratch {k.get('private_segment_fixed_size'):>5s}")
            c = mx.get(k.get("name"))
            if c:
                tot = sum(c.values())
                grp = collections.Counter()
                for op, n in c.items():
                    key = ("v_readlane" if op.startswith("v_readlane") or op.startswith("v_readfirstlane") else
                           "v_fma/mac/pk" if re.match(r"v_(fma|fmac|mac|pk_)", op) else
                           "v_mov" if op.startswith("v_mov") or op.startswith("v_accvgpr") else
                           "v_cndmask" if op.startswith("v_cndmask") else
                           "ds_" if op.startswith("ds_") else
                           "s_nop" if op == "s_nop" else
                           "s_waitcnt" if op.startswith("s_waitcnt") else
                           "global/flat/scratch" if re.match(r"(global|flat|scratch|buffer)_", op) else
                           "s_load" if op.startswith("s_load") else
                           "salu" if op.startswith("s_") else
                           "valu other" if op.startswith("v_") else "other")
                    grp[key] += n
                print("    total %d: " % tot + ", ".join(f"{k} {100.0 * v / tot:.1f}%" for k, v in grp.most_common()))


if __name__ == "__main__":
    main()
