#!/usr/bin/env python
"""bench.py -- throughput of the fused batched env-step on MI355X.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one env.step over the whole per-GPU batch: sample actions U[0,1) on the device
(benchmarks/mjx_benchmark.py:29), ctrl map, frame_skip=10 physics substeps, the post-step
mj_forward, obs/reward, TimeLimit bookkeeping and the masked auto-reset -- exactly what the
reference's env.step does per environment, for 4096 envs per GPU.
Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

# algorithmic HBM bytes per env-step (fp32; state read+written once, ctrl, target, obs, 4 reward scalars):
# SURVEY.md 8(d)  B_alg = 4*[(nq+nv+na) + nu + n_task_in + (nq+nv+na) + obs_dim + 4]
def algorithmic_bytes(env) -> int:
    """144 B (elbow pose), 1 376 B (hand pose), 4 600 B (leg walk; +1 920 B with the fatigue state)."""
    cm = env.cm
    # per-step task inputs: pose targets [nq] | reach targets [3 ntip] | reorient des_rot 3 + axis_half 1 + geom_size 3
    n_task_in = {1: cm.nq, 2: 3 * getattr(env, "ntip", 0), 3: 7}.get(int(env._task.task), 0)
    b = 4 * (2 * (cm.nq + cm.nv + cm.na) + cm.nu + n_task_in + env.obs_dim + 4)
    if env.muscle_condition == "fatigue":
        b += 4 * 6 * cm.na          # MA/MR/MF read + written
    return b


HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(env_id: str, nenv: int, nsteps: int):
    """fp64 oracle ("port": CPU restatement, NOT libmujoco) on the host cores, bounded sample."""
    from myosuite_amd.envs import registry
    from myosuite_amd.model import synth
    from oracle import oracle as O
    from oracle import env_oracle as EO
    spec = registry.spec(env_id)
    cm = synth.get_model(spec["kwargs"]["model"])
    om = O.OracleModel(cm)
    cores = os.cpu_count() or 1
    lo, hi = cm.jnt_range[:, 0], cm.jnt_range[:, 1]
    ds = []
    for e in range(nenv):
        d = O.OracleData(om)
        if hasattr(cm, "key_qpos"):                 # walk: the "init" keyframe (walk_v0.py:362-363)
            d.qpos[:] = cm.key_qpos[2]; d.qvel[:] = cm.key_qvel[2]
        elif "Object" in cm.names["body"]:          # reorient: open hand, palm up, capsule size of the episode
            q = cm.qpos0.astype(np.float64).copy(); q[:-6] = 0; q[0] = -1.5
            d.qpos[:] = q
            gt, size, _, _ = EO.reorient_reset_draws(synth.reorient_tables("100"), e, 0, 0, 0.07)
            d.set_geom_size(cm.names["geom"]["obj"], size, gt)
        else:
            uq, _ = EO.pose_reset_draws(cm.nq, e, 0, 0)
            d.qpos[:] = (lo + (hi - lo) * uq).astype(np.float32)
        ds.append(d)
    acts = np.stack([EO.uniform_stream(nenv * cm.nu, 0, s).reshape(nenv, cm.nu) for s in range(nsteps)]).astype(np.float64)
    t0 = time.perf_counter()
    O.batch_rollout(om, ds, acts, nsub=spec["kwargs"].get("frame_skip", 10), nthreads=cores, normalize=True, do_forward=True)
    dt = time.perf_counter() - t0
    return {"value": nenv * nsteps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{nenv} envs x {nsteps} env-steps of {env_id} (fp64 C oracle, {cores} threads, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--env", default="myoHandPoseRandom-v0")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from myosuite_amd import dist as D
    from myosuite_amd import engine as E
    from myosuite_amd.envs import registry

    rank, world, local = D.init_from_env()
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local)
    n = args.envs_per_gpu
    start, _ = D.shard_envs(n * world, rank, world)
    # every rank owns its own shard of envs; Philox streams are keyed by the GLOBAL env index via the seed
    env = registry.make(args.env, num_envs=n, seed=1000 * rank, lanes_per_env=args.lanes)
    cm = env.cm
    act = torch.empty(n, cm.nu, device="cuda")
    ep_stats = torch.zeros(n, 3, device="cuda")        # (episode return, length, solved) per env
    need = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dense_col = env.rwd.shape[1] - 1     # reward rows end with ..., sparse, solved, done, dense (MM_RWD_* / MM_RWDW_*)

    def one_step(s, ev=None):
        E.uniform(act, seed=rank, stream_id=s)
        if ev is not None:
            ev[0].record()
        E.env_step(env.hm, env.state, act, env._task)
        if ev is not None:
            ev[1].record()
        # episode statistics + masked auto-reset (device side, no host sync)
        E.episode_stats(ep_stats, need, env.rwd, dense_col, dense_col - 2, env.done, env.truncated)
        env.reset(mask=need)

    for s in range(args.warmup):
        one_step(s)
    torch.cuda.synchronize()
    D.barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(args.warmup + s, evs[s])
    stats = D.gather_episode_stats(ep_stats)   # the one collective of a rollout
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, device="cuda" if world > 1 else None)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        b_alg = algorithmic_bytes(env)
        # HBM traffic per launch: PMC counters cannot be read from inside the process; the committed rocprofv3 summary
        # of this same command (tools/prof_round.sh -> profiles/*_pmc.json) is reported when it matches the workload
        traffic = None
        issue = None
        try:
            pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))[-1]
            pm = json.load(open(pmc_file)).get(f"{args.env}@{n}")
            if pm:
                traffic = (pm["fetch_kib"] + pm["write_kib"]) * 1024.0
                # the honest roof of this kernel: fp32 vector issue.  VALU-busy quad-cycles per SIMD over the quad-cycles
                # the kernel lasts (wave cycles / resident waves per SIMD); 1024 SIMDs on the chip
                waves_per_simd = pm["sq_waves"] / 1024.0
                issue = {"valu_busy_frac": pm["sq_active_inst_valu"] / (pm["sq_wave_quadcycles"] / waves_per_simd),
                         "wave_issue_frac": pm["sq_active_inst_any"] / pm["sq_wave_quadcycles"],
                         "wave_waitcnt_frac": pm["sq_wait_any"] / pm["sq_wave_quadcycles"],
                         "valu_insts_per_env_step": pm["sq_insts_valu"] * 64 / n / 64,
                         "source": f"profiles/{os.path.basename(pmc_file)} (rocprofv3 PMC of this command)"}
        except (OSError, ValueError, KeyError, IndexError):
            pass
        achieved = (b_alg * n / (kern_ms * 1e-3)) / 1e9 if b_alg else None
        out = {
            "metric": "env-steps/sec (whole node) at %d envs/GPU" % n,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env}, {n} envs/GPU, random actions U[0,1), frame_skip {env.frame_skip} + final forward + "
                                   f"obs/reward + auto-reset (synthetic model {cm.name}: nq={cm.nq} nv={cm.nv} nu={cm.nu})",
                       "envs_per_gpu": n, "lanes_per_env": env.hm.info(E.INFO_LANES), "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "k_engine (fused env-step)", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": (b_alg * n) if b_alg else None, "valu_issue": issue},
            "stats": {"mean_episode_return": float(stats[:, 0].mean()), "solved_frac": float(stats[:, 2].mean()),
                      "status_or": int(env.state.status.max())},
        }
        if world == 1 and not args.no_cpu_baseline:
            # a few seconds of wall time on the host cores
            nb, ns = (8192, 200) if cm.nv <= 4 else ((4096, 60) if cm.nv < 25 else (1024, 40))
            out["cpu_baseline"] = cpu_baseline(args.env, nb, ns)
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
