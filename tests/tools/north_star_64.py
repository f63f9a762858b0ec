"""North-star accuracy of the library at MYOSIM_LIB (default: the in-tree build), for A/B runs of kernel variants on the GPU box:
per-stage relative error of one forward pass (hand, 64 random states) and the 64-env 1000-substep divergence statistics next to
the fp32-state twin.   python tests/tools/north_star_64.py [tag] [lanes] [nenv]   ->  gpurun_out/north_star_64_<tag>.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np

from myosuite_amd import engine as E
from myosuite_amd.model import synth
import fp32_error_study as FS
import test_gpu_widths as TW

tag = sys.argv[1] if len(sys.argv) > 1 else "head"
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nenv = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cm = synth.get_model("hand")
hm = E.HipModel(cm, lanes_per_env=lanes)
out = {"lib": E.LIB_PATH, "lanes": lanes, "stage_rel_err": FS.stage_errors(cm, hm)}
rel, rel_tw, status = TW.north_star_run("hand", lanes, nenv=nenv)
pe, pt = rel.max(axis=0), rel_tw.max(axis=0)
out.update({"envs_below_1e-4": int((pe < 1e-4).sum()), "twin_envs_below_1e-4": int((pt < 1e-4).sum()), "nenv": int(pe.size),
            "median_env_max": float(np.median(pe)), "twin_median_env_max": float(np.median(pt)), "max_over_run": float(pe.max()),
            "per_env_max_sorted_top8": np.sort(pe)[-8:].tolist(), "twin_top8": np.sort(pt)[-8:].tolist(), "status": status})
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/north_star_64_{tag}.json", "w"), indent=1)
print(tag, json.dumps({k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in out.items() if k not in ("stage_rel_err", "per_env_max_sorted_top8", "twin_top8")}))
print(tag, "stages", {k: f"{v:.1e}" for k, v in out["stage_rel_err"].items()})
