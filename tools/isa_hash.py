#!/usr/bin/env python
"""Per-kernel hash of the gfx950 machine code of every k_engine instantiation (no GPU needed).

    python tools/isa_hash.py OUT.json [inst_B ...]

Used to prove that a source refactoring left the shipped kernels bit-identical: run before and after, diff the JSON files.
"""
import concurrent.futures
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import isa_stats as S   # noqa: E402


def hash_unit(f):
    csrc = os.path.join(ROOT, "myosuite_amd", "csrc")
    obj = f"/tmp/isa/{f[:-4]}.hash.co"
    S.device_object(os.path.join(csrc, f), obj)
    txt = subprocess.check_output([f"{S.LLVM}/llvm-objdump", "-d", obj], text=True)
    out, cur, name = {}, None, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            if cur is not None:
                out[name] = cur.hexdigest()
            name, cur = m.group(1), hashlib.sha256()
            continue
        if cur is not None and line.strip():
            # drop the address column; keep mnemonic, operands and encoding
            cur.update(re.sub(r"^\s*", "", line).split("//")[0].encode())
    if cur is not None:
        out[name] = cur.hexdigest()
    return f, out


def main():
    out_path = sys.argv[1]
    args = sys.argv[2:]
    csrc = os.path.join(ROOT, "myosuite_amd", "csrc")
    srcs = sorted(f for f in os.listdir(csrc) if f.startswith("myosim_inst_") and f.endswith(".hip"))
    if args:
        srcs = [f for f in srcs if any(a in f for a in args)]
    os.makedirs("/tmp/isa", exist_ok=True)
    res = {}
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        for f, h in ex.map(hash_unit, srcs):
            res[f] = h
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    print(f"{sum(len(v) for v in res.values())} kernels hashed -> {out_path}")


if __name__ == "__main__":
    main()
