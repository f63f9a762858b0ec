#!/usr/bin/env python
"""north_star's accuracy sentence, both readings of "1000 steps" (VERDICT r05 #7), on the GPU box:

    BASELINE.json: "state divergence vs CPU mj_step < 1e-4 rel over 1000 steps" (configs 2-3)

  * unit "substeps":  1000 mj_step calls = 100 env-steps x frame_skip 10 = ONE episode from the Pose task's random reset -- the
                      reading tests/test_gpu_widths.py gates (`test_north_star_*`);
  * unit "env-steps": 1000 env.step calls = 10 000 mj_step calls = TEN episodes; every 100 env-steps both sides are re-armed the
                      way the task's TimeLimit reset does it (qpos ~ U(jnt_range) from the env's Philox stream of that episode,
                      zero velocity / activation / warm start), SURVEY.md 8(d) "Parity run".

Error per env = max over the whole run of max|qpos_gpu - qpos_oracle| / max(1, max|qpos_oracle|) (free running,
identical control streams: actions U[0,1) from Philox, through the muscle ctrl map).  The HIP kernels run through the C ABI
(mm_step); the fp64 C oracle is the checker.  Writes gpurun_out/accuracy.json -> commit as profiles/r06_accuracy.json (bench.py
replays its `runs` into the contract line's `accuracy` block, marked replayed).

    python tests/tools/gpu_accuracy_run.py [--envs 256] [--quick]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from myosuite_amd import engine as E           # noqa: E402
from myosuite_amd.model import synth           # noqa: E402
from oracle import env_oracle as EO            # noqa: E402
from oracle import oracle as O                 # noqa: E402

PREC = {"f32": E.MM_PREC_F32, "f64_state": E.MM_PREC_F64_STATE}


def run(model, lanes, precision, nenv, env_steps, episode_len=100, nsub=10):
    cm = synth.get_model(model)
    om = O.OracleModel(cm)
    lo, hi = cm.jnt_range[:, 0], cm.jnt_range[:, 1]
    hm = E.HipModel(cm, lanes_per_env=lanes, precision=PREC[precision])
    assert hm.launch_lanes(nenv) == lanes
    st = E.BatchState(hm, nenv)
    ds = [O.OracleData(om) for _ in range(nenv)]
    a = torch.empty(nenv, cm.nu, device="cuda")
    per_env = np.zeros(nenv)
    status = 0
    t0 = time.time()
    for s in range(env_steps):
        if s % episode_len == 0:          # (re-)arm: the Pose task's random reset of episode s / episode_len
            ep = s // episode_len
            q0 = np.stack([(lo + (hi - lo) * EO.pose_reset_draws(cm.nq, e, ep, 0)[0]).astype(np.float32) for e in range(nenv)])
            st.qpos.copy_(torch.from_numpy(q0).to(st.qpos.dtype))
            st.qvel.zero_(); st.act.zero_(); st.qacc_warmstart.zero_(); st.time.zero_()
            for e in range(nenv):
                ds[e].reset()
                ds[e].qpos[:] = q0[e]
        E.uniform(a, 0, s)
        ctrl = (1.0 / (1.0 + torch.exp(-5.0 * (a - 0.5)))).contiguous()
        E.step(hm, st, ctrl, nsub)
        c = ctrl.cpu().numpy()
        for e in range(nenv):
            ds[e].ctrl[:] = c[e]
            ds[e].step(nsub)
        oq = np.stack([d.qpos for d in ds])
        scale = max(1.0, np.abs(oq).max())
        per_env = np.maximum(per_env, np.abs(st.qpos.cpu().numpy().astype(np.float64) - oq).max(axis=1) / scale)
        status |= int(st.status.max())
    return {"model": model, "kernel": f"{precision} G{lanes}", "envs": nenv, "envs_below_1e-4": int((per_env < 1e-4).sum()),
            "max_rel": float(per_env.max()), "median_rel": float(np.median(per_env)), "status_or": status,
            "seconds": round(time.time() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--quick", action="store_true", help="32 envs, 30 env-steps / 3 episodes of 10: a plumbing check")
    args = ap.parse_args()
    n = 32 if args.quick else args.envs
    sub, full, ep = (3, 30, 10) if args.quick else (100, 1000, 100)
    runs = []
    for model, lanes in (("hand", 32), ("elbow", 8)):
        for precision in ("f64_state", "f32"):
            for unit, steps in (("substeps", sub), ("env-steps", full)):
                r = run(model, lanes, precision, n, steps, episode_len=ep)
                r.update(steps=steps * 10 if unit == "substeps" else steps, unit=unit)
                print(json.dumps(r), flush=True)
                runs.append(r)
    out = {"what": "per-env max over the run of max|qpos_hip - qpos_oracle| / max(1, max|qpos|), free running, random actions; "
                   "unit substeps = mj_step calls (one 100-env-step episode), env-steps = env.step calls (ten episodes with the "
                   "Pose task's TimeLimit re-arm every 100)",
           "tool": "tests/tools/gpu_accuracy_run.py", "runs": [{k: r[k] for k in ("model", "kernel", "steps", "unit", "envs", "envs_below_1e-4", "max_rel")}
                                                               for r in runs],
           "detail": runs}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "accuracy.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
