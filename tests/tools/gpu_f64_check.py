#!/usr/bin/env python
"""GPU box: the precision-mode kernels (mm_model_set_option "precision") against the fp64 oracle after ONE launch of 10
substeps, per mode and group width, and their rollout throughput at the bench batch.   python tests/tools/gpu_f64_check.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myosuite_amd import engine as E          # noqa: E402
from myosuite_amd.envs import registry       # noqa: E402
from myosuite_amd.model import synth         # noqa: E402
from oracle import env_oracle as EO          # noqa: E402
from oracle import oracle as O               # noqa: E402

out = {}
for name, widths in (("elbow", (4, 8, 16)), ("hand", (32, 64))):
    cm = synth.get_model(name); om = O.OracleModel(cm)
    nenv = 16
    lo, hi = cm.jnt_range[:, 0], cm.jnt_range[:, 1]
    q0 = np.stack([(lo + (hi - lo) * EO.pose_reset_draws(cm.nq, e, 0, 0)[0]).astype(np.float32) for e in range(nenv)])
    a = torch.empty(nenv, cm.nu, device="cuda")
    for G in widths:
        for prec, pname in ((E.MM_PREC_F32, "f32"), (E.MM_PREC_F64, "f64"), (E.MM_PREC_F64_STATE, "f64_state")):
            hm = E.HipModel(cm, lanes_per_env=G, precision=prec)
            st = E.BatchState(hm, nenv); st.qpos.copy_(torch.from_numpy(q0))
            ds = []
            for e in range(nenv):
                d = O.OracleData(om); d.qpos[:] = q0[e]; ds.append(d)
            eq = ev = ea = 0.0
            for s in range(3):
                E.uniform(a, 0, s)
                ctrl = (1.0 / (1.0 + torch.exp(-5.0 * (a - 0.5)))).contiguous()
                E.step(hm, st, ctrl, 10)
                c = ctrl.cpu().numpy()
                for e in range(nenv):
                    ds[e].ctrl[:] = c[e]; ds[e].step(10)
                gq, gv, ga = (t.cpu().numpy().astype(np.float64) for t in (st.qpos, st.qvel, st.act))
                eq = max(eq, max(np.abs(gq[e] - ds[e].qpos).max() for e in range(nenv)))
                ev = max(ev, max(np.abs(gv[e] - ds[e].qvel).max() for e in range(nenv)))
                ea = max(ea, max(np.abs(ga[e] - ds[e].act).max() for e in range(nenv)))
            print(f"{name} G={G:2d} {pname:9s}: after 30 substeps max|dqpos| {eq:.2e} |dqvel| {ev:.2e} |dact| {ea:.2e}  status {int(st.status.max())}", flush=True)
            out[f"{name}_G{G}_{pname}"] = {"dqpos": eq, "dqvel": ev, "dact": ea}

# rollout throughput at the bench batch (one launch per env-step: mm_rollout_step)
for env_id, n in (("myoHandPoseRandom-v0", 4096), ("myoElbowPose1D6MRandom-v0", 4096)):
    for pname in ("f32", "f64", "f64_state"):
        env = registry.make(env_id, num_envs=n, seed=0, precision=pname)
        env.reset()
        env.rollout_setup(action_seed=0)
        for s in range(5):
            env.rollout_step(None, stream_id=s)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); K = 40
        for s in range(K):
            env.rollout_step(None, stream_id=5 + s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"{env_id} @ {n} {pname:9s}: {dt * 1e3:.3f} ms/step  {n / dt / 1e6:.3f} M env-steps/s  lanes {env.hm.launch_lanes(n)}", flush=True)
        out[f"rollout_{env_id}_{pname}"] = {"ms_per_step": dt * 1e3, "env_steps_per_s": n / dt}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/f64_check.json", "w"), indent=1)
