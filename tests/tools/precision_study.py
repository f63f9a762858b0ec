"""Which stage's precision does the 1000-step state divergence hang on?  (CPU only; no GPU, no HIP engine.)

The fp64 oracle is compiled a second time with a scalar type that rounds the result of EVERY operation to fp32 inside the
pipeline stages selected by a bit mask (tests/tools/roundreal/): mask = all stages emulates a plain fp32 implementation of
the same algorithm, clearing one bit shows what computing that stage in higher precision would buy.  Next to it the two
floors: the fp64 oracle whose STATE is rounded to fp32 after every substep (what any engine that stores fp32 state can reach at
best), and the fp64 oracle from an initial qpos perturbed by one fp32 rounding.

    python tests/tools/precision_study.py [--model hand] [--nenv 64] [--out profiles/r03_precision_study.json]

The model is shifted so that its mean body position sits at the world origin, as the kernel's internal frame is
(Dims::ox/oy/oz); physics is translation invariant, fp32 rounding is not.
"""
import argparse
import copy
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from myosuite_amd.model import blob as B
from myosuite_amd.model import synth
from oracle import env_oracle as EO
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
STAGES = ["kin", "com", "tendon", "crb", "constr", "vel", "act", "acc", "solve", "integ"]
ALL = (1 << len(STAGES)) - 1
OMAP = {"tenlen": "ten_length", "tenvel": "ten_velocity", "actfrc": "actuator_force", "bias": "qfrc_bias", "smooth": "qfrc_smooth",
        "qaccsm": "qacc_smooth", "tenJ": "ten_J", "qacc": "qacc", "xpos": "xpos", "cdof": "cdof"}


def build_round_lib():
    out = os.path.join(tempfile.gettempdir(), "libmmo_round.so")
    src = os.path.join(HERE, "roundreal", "round_oracle.cpp")
    subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-std=c++17", "-fpermissive", "-w", "-DMMO_REAL_EXTERNAL", "-o", out, src])
    return out


def centred(cm):
    """the same model with its mean body position (reference configuration, 1/64 m grid) moved to the world origin"""
    om = O.OracleModel(cm); d = O.OracleData(om); d.forward()
    org = np.round(64.0 * d.xpos[1:].mean(axis=0)) / 64.0
    arrays = {k: np.array(v) for k, v in B.unpack(cm.blob).items()}
    par = arrays["BODY_PARENT"]
    bp = arrays["BODY_POS"].copy()
    for b in range(1, cm.nbody):
        if par[b] == 0:
            bp[b] -= org
    arrays["BODY_POS"] = bp
    c2 = copy.copy(cm)
    c2.blob = B.pack(arrays)
    del d, om
    return c2, org


def rollout(cm, q0, nsteps, nsub, twin=False, perturb=0.0):
    om = O.OracleModel(cm)
    ds = []
    for e in range(q0.shape[0]):
        d = O.OracleData(om)
        d.qpos[:] = q0[e]
        if perturb:
            d.qpos[:] = d.qpos * (1.0 + perturb)
        if twin:
            d.round_state_f32(True)
        ds.append(d)
    traj = np.zeros((nsteps, q0.shape[0], cm.nq))
    for s in range(nsteps):
        for e, d in enumerate(ds):
            a = EO.uniform_stream(q0.shape[0] * cm.nu, 0, s).reshape(q0.shape[0], cm.nu)[e].astype(np.float32).astype(np.float64)
            d.ctrl[:] = (1.0 / (1.0 + np.exp(-5.0 * (a - 0.5)))).astype(np.float32)
            d.step(nsub)
            traj[s, e] = d.qpos
    return traj


def stage_dump(cm, states):
    om = O.OracleModel(cm)
    out = []
    for (qp, qv, ac, ct) in states:
        d = O.OracleData(om)
        d.qpos[:] = qp; d.qvel[:] = qv; d.act[:] = ac; d.ctrl[:] = ct
        d.forward()
        out.append({k: np.array(getattr(d, v)).copy() for k, v in OMAP.items()})
    return out


def summarize(traj, ref):
    rel = np.abs(traj - ref).max(axis=2) / np.maximum(1.0, np.abs(ref).max(axis=2).max(axis=1, keepdims=True))   # [step, env]
    per_env = rel.max(axis=0)
    return {"median_env_max": float(np.median(per_env)), "envs_below_1e-4": int((per_env < 1e-4).sum()), "nenv": int(per_env.size),
            "max_over_run": float(per_env.max()), "end_of_run_max": float(rel[-1].max()),
            "per_env_max_sorted_top8": np.sort(per_env)[-8:].tolist()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hand")
    ap.add_argument("--nenv", type=int, default=64)
    ap.add_argument("--nsteps", type=int, default=100)
    ap.add_argument("--nsub", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--masks", default="all,floors,drop1", help="comma list: all | floors | drop1 | keep1 | hex masks (0x...)")
    args = ap.parse_args()
    cm0 = synth.get_model(args.model)
    cm, org = centred(cm0)
    lo, hi = cm.jnt_range[:, 0], cm.jnt_range[:, 1]
    q0 = np.stack([(lo + (hi - lo) * EO.pose_reset_draws(cm.nq, e, 0, 0)[0]).astype(np.float32) for e in range(args.nenv)]).astype(np.float64)
    rng = np.random.default_rng(0)
    states = []
    for e in range(32):
        states.append(((lo + (hi - lo) * rng.random(cm.nq)).astype(np.float32), (rng.standard_normal(cm.nv) * 2).astype(np.float32),
                       rng.random(cm.na).astype(np.float32), rng.random(cm.nu).astype(np.float32)))
    res = {"model": args.model, "origin_moved_to": org.tolist(), "nenv": args.nenv, "substeps": args.nsteps * args.nsub, "runs": {}}
    ref = rollout(cm, q0, args.nsteps, args.nsub)
    ref_st = stage_dump(cm, states)
    want = args.masks.split(",")
    if "floors" in want:
        res["runs"]["fp64_state_rounded_to_fp32_every_substep"] = summarize(rollout(cm, q0, args.nsteps, args.nsub, twin=True), ref)
        res["runs"]["fp64_qpos0_perturbed_1e-7"] = summarize(rollout(cm, q0, args.nsteps, args.nsub, perturb=1e-7), ref)
        print(json.dumps(res["runs"], indent=1), flush=True)
    # ---- the fp32-emulating build
    O._LIB_PATH = build_round_lib(); O._lib = None
    L = O.lib()
    L.mmo_round_mask.argtypes = [C.c_uint]
    masks = []
    for w in want:
        if w == "all":
            masks.append(("fp32_everywhere", ALL))
        elif w == "drop1":
            masks += [(f"fp32_except_{n}", ALL & ~(1 << k)) for k, n in enumerate(STAGES)]
        elif w == "keep1":
            masks += [(f"fp32_only_{n}", 1 << k) for k, n in enumerate(STAGES)]
        elif w.startswith("0x"):
            m = int(w, 16)
            masks.append(("fp32_in_" + "+".join(n for k, n in enumerate(STAGES) if m >> k & 1), m))
    for name, mask in masks:
        L.mmo_round_mask(mask)
        st = stage_dump(cm, states)
        serr = {}
        for k in OMAP:
            serr[k] = float(max(np.abs(a[k] - b[k]).max() / max(1e-9, np.abs(b[k]).max()) for a, b in zip(st, ref_st)))
        r = summarize(rollout(cm, q0, args.nsteps, args.nsub), ref)
        r["stage_rel_err"] = serr
        res["runs"][name] = r
        print(name, json.dumps(r), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
