#!/usr/bin/env python
"""bench.py -- throughput of the fused batched env-step on MI355X.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python bench.py --gpus 8                       # no launcher: re-executes itself under torch.distributed.run, 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 2 --oversubscribe       # TEST ONLY: N ranks share the visible GPU(s) (gloo group), exercises the N > 1 path

One "step" = one env.step over the whole per-GPU batch, as ONE kernel launch (mm_rollout_step): draw actions U[0,1) on the
device (benchmarks/mjx_benchmark.py:29), muscle ctrl map, frame_skip physics substeps, the post-step mj_forward, obs /
reward, TimeLimit bookkeeping, episode statistics and the masked auto-reset -- exactly what the reference's env.step (+ a
gym autoreset / RecordEpisodeStatistics wrapper) does per environment, for 4096 envs per GPU.  Tasks whose
reset is not folded into the launch (everything but the Pose family) add their masked reset launch.
Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X fp32 vector peak (256 CUs x 4 SIMD x 64 lanes... per the microarch guide), dense, no MFMA

# BASELINE.json configs 2, 4, 5 (+ the self-colliding hand, docs/source/suite.rst:288, and the MuJoCo-default leg on the
# implicitfast integrator) reported next to the headline line
# the committed PMC session bench.py replays counters from (tools/prof_round.sh at the HEAD named in DESIGN.md section 5); pinned, not "the latest file"
PMC_PROFILE = os.path.join(ROOT, "profiles", "r06d_pmc.json")
ACCURACY_PROFILE = os.path.join(ROOT, "profiles", "r06_accuracy.json")     # tests/tools/gpu_accuracy_run.py (replayed into the line)
REPEATS = 3                 # timed regions of --steps steps each; the line reports the median region
EXTRA_MIN_TIMED_MS = 60.0   # an extra line times at least this much kernel work (a 1.5 ms timed region is launch-noise bound)
# Order matters: the driver's record keeps the TAIL of the printed line, so the BASELINE.json configs 2 / 4 / 5 (elbow, reorient,
# fati-leg) come last, and a compact `baseline_configs` digest of every line closes the JSON.
EXTRA_CONFIGS = [# the step of the reference's own GPU path: mjx_env.step = n_substeps x mjx.step, observation straight from the
                 # stepped data (envs/myo/mjx/mjx_base_env.py:74-91) -- no trailing mj_forward as in the CPU path
                 # (robot.py:595-607), whose outputs the Pose observation / reward do not read.  NOT the headline protocol.
                 ("myoHandPoseRandom-v0", 4096, {"do_forward": False}),
                 # precision modes (include/myosim.h): MM_PREC_F64_STATE = the same launch over real = double with fp64 state rows --
                 # the kernels that meet "state divergence < 1e-4 rel over 1000 steps" on every env (tests/test_gpu_widths.py)
                 ("myoElbowPose1D6MRandom-v0", 4096, {"precision": "f64_state"}),
                 ("myoHandPoseRandom-v0", 4096, {"precision": "f64_state"}),
                 # the one workload the reference publishes GPU numbers for (MjxHandReachRandom-v0)
                 ("myoHandReachRandom-v0", 4096, {}),
                 ("myoFatiLegWalk-v0", 1024, {"model": "leg_implicit"}),
                 ("myoHandPoseRandom-v0", 4096, {"model": "hand_contact"}),
                 # the reorient hand under MuJoCo's default collision filter: 189 candidate pairs (every skin capsule against every other
                 # non-adjacent one + the object against all twenty), swept in three chunks of one pair per lane (synth.make_hand_dense)
                 ("myoHandReorient100-v0", 2048, {"model": "hand_dense"}),
                 ("myoElbowPose1D6MRandom-v0", 4096, {}), ("myoHandReorient100-v0", 2048, {}), ("myoFatiLegWalk-v0", 1024, {})]


BASELINE_PRESETS = {2: ("myoElbowPose1D6MRandom-v0", 4096), 3: ("myoHandPoseRandom-v0", 4096), 4: ("myoHandReorient100-v0", 2048),
                    5: ("myoFatiLegWalk-v0", 1024)}      # --config N (config 5: 8192 envs over 8 GPUs = 1024 per GPU)


def algorithmic_bytes(env, include_carry: bool = False) -> int:
    """fp32, state read + written once per env-step (substeps are fused on chip), constant model excluded:
    B_alg = 4*[(nq+nv+na) + nu + n_task_in + n_aux + (nq+nv+na) + n_aux + obs_dim + 4], n_aux = the fatigue state (3 na) and,
    for models with a constraint solve that is warm started across steps (contacts / equalities), qacc_warmstart (nv).
    144 B (elbow pose), 1 376 B (hand pose), 2 012 B (reorient), 5 336 B (leg walk + fatigue) -- SURVEY's table; this is the figure
    `roofline.achieved / frac` are priced with.  `include_carry=True` adds the traffic this engine ADDED in round 4, the forward-carry
    row where it is on (8 nv + 4 bytes read and written: hand 1 752 B, reorient 2 484 B, leg 5 888 B): reported next to it as
    `*_incl_carry`, never as the contract figure."""
    cm = env.cm
    # per-step task inputs: pose targets [nq] | reach targets [3 ntip] | reorient geom type 1 + size 3 + axis_half 1 + des_rot 3 |
    # walk step counter 1
    n_task_in = {1: cm.nq, 2: 3 * getattr(env, "ntip", 0), 3: 8, 4: 1}.get(int(env._task.task), 0)
    n_aux = 0
    if env.muscle_condition == "fatigue":
        n_aux += 3 * cm.na          # MA / MR / MF
    if cm.npair > 0 or cm.neq > 0:
        n_aux += cm.nv              # qacc_warmstart
    if include_carry and getattr(env, "_fwd_carry", None) is not None:
        n_aux += 2 * cm.nv + 1      # forward-carry row (mm_task.fwd_carry): hash + qacc + Euler's damped acceleration, read and written
    sw = 8 if getattr(env, "precision", 0) == 2 else 4      # MM_PREC_F64_STATE: the state rows are fp64
    return sw * 2 * (cm.nq + cm.nv + cm.na) + 4 * (cm.nu + n_task_in + 2 * n_aux + env.obs_dim + 4)


def workload_key(env_id: str, n: int, overrides=None) -> str:
    """Key of one measured workload in the committed tables (profiles/r*_pmc.json, profiles/flops_per_env_step.json): the env id
    and batch PLUS every override that changes the kernel or the work (`model=`, `do_forward=`), e.g.
    "myoHandPoseRandom-v0@4096|model=hand_contact".  (Keyed by env id and batch alone, the self-contact hand would be
    priced with the contact-free hand's counters.)"""
    ov = "".join(f"|{k}={v}" for k, v in sorted((overrides or {}).items()))
    return f"{env_id}@{n}{ov}"


def algorithmic_flops(env_id: str, overrides=None):
    """fp operations of ONE env-step of the reference algorithm, counted by the instrumented oracle (tests/tools/count_flops.py
    -> profiles/flops_per_env_step.json); None when the table has no entry for this workload (env id + overrides)."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "flops_per_env_step.json")))
        ov = "".join(f"|{k}={v}" for k, v in sorted((overrides or {}).items()))
        return tab.get(env_id + ov)
    except (OSError, ValueError):
        return None


def host_topology():
    """CPUs this process may use: logical CPUs (affinity mask), physical cores behind them (/proc/cpuinfo: distinct
    (physical id, core id) pairs of the allowed CPUs), and the cgroup CPU quota when the container has one."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = set()
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in allowed:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except (OSError, ValueError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    phys = len(cores) or len(allowed)
    if quota is not None:
        phys = max(1, min(phys, int(quota)))
    return {"logical_cpus": len(allowed), "physical_cores": phys, "cgroup_cpu_quota": quota}


def _cpu_envs(env_id: str, nenv: int):
    """oracle model + `nenv` oracle envs in the task's reset state (same draws as the GPU batch) + frame_skip"""
    import numpy as np
    from myosuite_amd.envs import registry
    from myosuite_amd.model import synth
    from oracle import oracle as O
    from oracle import env_oracle as EO
    spec = registry.spec(env_id)
    cm = synth.get_model(spec["kwargs"]["model"])
    om = O.OracleModel(cm)
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
    return om, ds, cm, spec["kwargs"].get("frame_skip", 10)


def cpu_rollout_rate(env_id: str, nthreads: int, target_s: float = 4.0, per_thread_rate: float = None):
    """env-steps/s of the fp64 C oracle ("port": CPU restatement, NOT libmujoco) on `nthreads` host threads, one env at a time
    per thread (oracle/mmo_batch.c), random actions through the muscle ctrl map, frame_skip substeps + the final forward per
    env-step.  The sample is sized for ~`target_s` of wall time from a short calibration run (or `per_thread_rate`)."""
    import numpy as np
    from oracle import oracle as O
    from oracle import env_oracle as EO
    # the sample grows in ENVS (one env at a time per thread, ~40 env-steps each), sized for ~target_s of wall time from the
    # single-thread rate (given, or measured on a few envs first)
    nsteps = 40
    if per_thread_rate is None:
        om, ds, cm, nsub = _cpu_envs(env_id, 2)
        acts = np.stack([EO.uniform_stream(2 * cm.nu, 0, s_).reshape(2, cm.nu) for s_ in range(10)]).astype(np.float64)
        t0 = time.perf_counter()
        O.batch_rollout(om, ds, acts, nsub=nsub, nthreads=1, normalize=True, do_forward=True)
        per_thread_rate = 2 * 10 / (time.perf_counter() - t0)
    nenv = int(round(target_s * per_thread_rate * nthreads / nsteps))
    nenv = max(2 * nthreads, min(8192, ((nenv + nthreads - 1) // nthreads) * nthreads))
    om, ds, cm, nsub = _cpu_envs(env_id, nenv)
    acts = np.stack([EO.uniform_stream(nenv * cm.nu, 0, s_).reshape(nenv, cm.nu) for s_ in range(nsteps)]).astype(np.float64)
    t0 = time.perf_counter()
    O.batch_rollout(om, ds, acts, nsub=nsub, nthreads=nthreads, normalize=True, do_forward=True)
    dt = time.perf_counter() - t0
    return {"value": nenv * nsteps / dt, "threads": nthreads, "envs": nenv, "env_steps_each": nsteps, "seconds": dt}


def cpu_baseline(env_id: str):
    """The fp64 oracle on the host cores, two points: ONE thread -- the reference's own CPU protocol is one env on one core
    (benchmarks/mjx_benchmark_baseline.py:8-25) -- and one thread per PHYSICAL core (`value`, `cores`).  Bounded sample: ~4 s
    + ~8 s of wall time."""
    from oracle import oracle as O
    O.use_variant("fast")       # oracle/Makefile: the same sources -O3 -march=native + fma contraction, built on THIS host (never the checker)
    O.build(variant="fast")
    topo = host_topology()
    one = cpu_rollout_rate(env_id, 1, target_s=4.0)
    cores = topo["physical_cores"]
    allc = cpu_rollout_rate(env_id, cores, target_s=8.0, per_thread_rate=one["value"]) if cores > 1 else one
    return {"value": allc["value"], "unit": "env-steps/s", "cores": cores, "kind": "port",
            "build": "oracle/liboracle_fast.so: gcc -O3 -march=native -ffp-contract=fast, built on this host (the checker build "
                     "liboracle.so is -O2 without contraction and is not what is timed)",
            "sample": f"{allc['envs']} envs x {allc['env_steps_each']} env-steps of {env_id} (fp64 C oracle, one thread per physical "
                      f"core = {cores} threads, {allc['seconds']:.1f} s)",
            "sample_short": f"{allc['envs']} envs x {allc['env_steps_each']} env-steps, {cores} threads, {allc['seconds']:.1f} s",
            "single_thread": {"value": one["value"], "unit": "env-steps/s", "cores": 1,
                              "sample": f"{one['envs']} envs x {one['env_steps_each']} env-steps, one at a time on one thread ({one['seconds']:.1f} s)"},
            "parallel_efficiency": allc["value"] / (one["value"] * cores), "host": topo}


def respawn_under_launcher(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: run the N ranks ourselves (one process per GPU,
    torch.distributed.run, RCCL rendezvous on 127.0.0.1) and pass their output through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def status_or(status) -> int:
    """bitwise OR of the per-env status words (the max of bit flags would hide bits: 1 | 4 has max 4)"""
    import functools
    import operator
    return functools.reduce(operator.or_, (int(v) for v in status.unique().tolist()), 0)


def gpu_clocks():
    """current shader / memory clocks of the visible GPUs as rocm-smi reports them (MHz), or None: logged around the timed region
    because the general-row kernels were 5...18 % slower on some boxes of the pool with the same binary"""
    try:
        txt = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        js = json.loads(txt[txt.index("{"):])
        out = {}
        for card, rec in js.items():
            for k, v in rec.items():
                kl = k.lower()
                if "sclk" in kl or "mclk" in kl or "fclk" in kl:
                    import re
                    m = re.search(r"(\d+)\s*mhz", str(v).lower())
                    if m:
                        out[f"{card}.{'sclk' if 'sclk' in kl else ('mclk' if 'mclk' in kl else 'fclk')}_mhz"] = int(m.group(1))
        return out or None
    except Exception:
        return None


def measure(env_id, n, steps, warmup, rank, world, lanes=0, seed=0, overrides=None, repeats=1):
    """W untimed rollout steps, then `repeats` timed regions of exactly K steps each -- every region bracketed by barrier +
    synchronize on both sides, max over ranks -- of `env_id` with n envs on this rank.  Returns (MEDIAN elapsed_s of the regions,
    mean kernel ms of the median region, env, gathered stats, per-region elapsed list).  The reference protocol times its scan
    with timeit.repeat(..., repeat=3) and reports one of them (benchmarks/mjx_benchmark.py:46)."""
    import numpy as np
    import torch
    from myosuite_amd import dist as D
    from myosuite_amd.envs import registry
    # every rank owns the envs [rank*n, (rank+1)*n) of the global batch; all Philox streams (reset draws, actions) are keyed by
    # the GLOBAL env index (mm_state.env_index_base), so the rollout does not depend on how the envs are spread over ranks
    env = registry.make(env_id, num_envs=n, seed=seed, lanes_per_env=lanes, env_index_base=rank * n, **(overrides or {}))
    ep_stats = env.rollout_setup(action_seed=seed)     # (episode return, length, solved) per env, accumulated in the launch
    for s in range(warmup):
        env.rollout_step(None, stream_id=s)
    regions = []
    stats = None
    for r in range(repeats):
        torch.cuda.synchronize()
        D.barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            env.rollout_step(None, stream_id=warmup + r * steps + s, events=evs[s])   # events bracket the fused env-step kernel alone
        stats = D.gather_episode_stats(ep_stats)   # the one collective of a rollout
        torch.cuda.synchronize()
        D.barrier()
        own = time.perf_counter() - t0
        elapsed = D.max_over_ranks(own, device="cuda" if (world > 1 and D.backend() == "nccl") else None)
        regions.append((elapsed, float(np.mean([a.elapsed_time(b) for a, b in evs])), own))
    order = sorted(range(repeats), key=lambda i: regions[i][0])
    med = regions[order[(repeats - 1) // 2]]          # the median region (lower median for an even count)
    measure.ep_stats = ep_stats
    measure.own_elapsed_of_median_region = med[2]     # this rank's own clock around the region `value` is quoted on (N > 1 line: per-rank values)
    return med[0], med[1], env, stats, [e for e, _, _ in regions]


def collective_report(ep_stats, stats, n, world, own_elapsed, steps):
    """What the N > 1 line says about its one collective, so that an 8-GPU record is self-evidencing: backend, world
    size, RCCL version, the gathered row count (asserted = world x E), the all-gather's own latency (median of 20, synchronised),
    and every rank's OWN env-steps/s (its clock around the same timed region; `value` uses the slowest rank's)."""
    import torch
    import torch.distributed as dist
    from myosuite_amd import dist as D
    assert stats.shape[0] == world * n, f"gathered {stats.shape[0]} rows, expected world x E = {world * n}"
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); D.barrier()
        t0 = time.perf_counter()
        D.gather_episode_stats(ep_stats)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    own = torch.tensor([n * steps / own_elapsed], dtype=torch.float64)
    if world > 1:
        parts = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        if D.backend() == "nccl":
            dev = [p.cuda() for p in parts]
            dist.all_gather(dev, own.cuda())
            parts = [p.cpu() for p in dev]
        else:
            dist.all_gather(parts, own)
    else:
        parts = [own]
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())       # RCCL's version on ROCm
    except Exception:
        ver = None
    return {"backend": D.backend() or "none", "is_rccl": D.backend() == "nccl", "world_size": world, "rccl_version": ver,
            "what": "one all-gather of [E, 3] episode statistics per rollout; no collective on the step path",
            "gathered_rows": int(stats.shape[0]), "expected_rows": world * n, "bytes_per_rank": int(ep_stats.numel() * ep_stats.element_size()),
            "allgather_us": 1e6 * ts[len(ts) // 2], "per_rank_env_steps_per_s": [float(p.item()) for p in parts]}


def roofline(env, env_id, n, kern_ms, overrides=None):
    """HBM roofline of the fused kernel (the contract's definition) + the views that actually bound it: fp32 vector issue
    (PMC counters of the committed profile of this command) and algorithmic flops against the fp32 vector peak."""
    from myosuite_amd import engine as E
    b_alg = algorithmic_bytes(env)                              # the contract figure
    b_carry = algorithmic_bytes(env, include_carry=True)        # + the forward-carry row this engine added (equal when it is off)
    achieved = (b_alg * n / (kern_ms * 1e-3)) / 1e9
    achieved_c = (b_carry * n / (kern_ms * 1e-3)) / 1e9
    traffic = None
    profile = None
    launch = None
    try:
        launch = env.hm.launch_info(n)
    except Exception:      # (a library older than the binding: the line must still print)
        pass
    try:   # PMC counters cannot be read in-process: the COMMITTED rocprofv3 summary of this same command is replayed, and marked so
        pmc_file = PMC_PROFILE if os.path.exists(PMC_PROFILE) else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))[-1]
        pm = json.load(open(pmc_file)).get(workload_key(env_id, n, overrides))
        if pm:
            traffic = (pm["fetch_kib"] + pm["write_kib"]) * 1024.0
            # Waves that can be resident on one SIMD at a time: what the launch puts on a CU (occupancy of the kernel at its block
            # size and LDS footprint: LDS, VGPRs) and never more than the launch has.  A SIMD issues at most one VALU instruction
            # per quad-cycle, so SQ_ACTIVE_INST_VALU over (summed wave lifetime / resident waves) is a fraction of its issue slots.
            waves_total = pm["sq_waves"]
            if launch and launch["resident_blocks_per_cu"] > 0:
                resident = min(launch["resident_blocks_per_cu"] * launch["waves_per_block"] / 4.0, max(1.0, waves_total / 1024.0))
            else:
                resident = min(2.0, max(1.0, waves_total / 1024.0))
            simd_quadcycles = pm["sq_wave_quadcycles"] / resident
            profile = {"replayed_from": f"profiles/{os.path.basename(pmc_file)}",
                       "what": "rocprofv3 PMC means per dispatch of this command, collected in the session named by the file -- NOT measured in this run",
                       "hbm_bytes_per_launch": traffic,
                       "resident_waves_per_simd": resident,
                       "valu_busy_frac": min(1.0, pm["sq_active_inst_valu"] / simd_quadcycles),
                       "wave_issue_frac": pm["sq_active_inst_any"] / pm["sq_wave_quadcycles"],
                       "wave_waitcnt_frac": pm["sq_wait_any"] / pm["sq_wave_quadcycles"],
                       "lds_bank_conflict_frac": pm["sq_lds_bank_conflict"] / max(1.0, pm["sq_lds_idx_active"]),
                       "valu_insts_per_env_step": pm["sq_insts_valu"] / n,
                       "kernel_ms_in_that_session": pm.get("kernel_trace_avg_ns", 0.0) * 1e-6}
    except (OSError, ValueError, KeyError, IndexError):
        pass
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "bytes_definition": "SURVEY.md 8(d) B_alg x envs per launch (achieved / frac); *_incl_carry adds the forward-carry row this engine reads and writes per env-step",
           "achieved_incl_carry": achieved_c, "frac_incl_carry": achieved_c / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": (profile or {}).get("replayed_from"),
           "traffic_over_algorithmic": (traffic / (b_alg * n)) if traffic else None,
           "kernel": "k_engine (fused env-step)", "kernel_ms": kern_ms,
           "algorithmic_bytes_per_launch": b_alg * n, "algorithmic_bytes_per_launch_incl_carry": b_carry * n,
           "launch": launch, "profile": profile}
    fl = algorithmic_flops(env_id, overrides)
    if fl:
        tf = fl["flops"] * n / (kern_ms * 1e-3) / 1e12
        out["flops"] = {"algorithmic_flops_per_env_step": fl["flops"], "achieved_tflops": tf, "peak_tflops": FP32_PEAK_TFLOPS,
                        "frac": tf / FP32_PEAK_TFLOPS, "source": fl.get("source", "instrumented oracle")}
        # arithmetic intensity against the ridge point of the two roofs: far right of it the binding roof is fp32 issue, not HBM
        intensity = fl["flops"] / float(b_alg)
        ridge = FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        out["bound_by"] = "fp32-issue" if intensity > ridge else "hbm"
        out["bound_by_detail"] = {"flop_per_byte": intensity, "ridge_flop_per_byte": ridge, "x_ridge": intensity / ridge,
                                  "fp32_vector_peak_frac": tf / FP32_PEAK_TFLOPS,
                                  "note": "the HBM fraction above is the contract's definition; with the substeps fused on chip the kernel is "
                                          "bounded by dependent fp32 issue (valu_busy_frac / wave_waitcnt_frac in `profile`), not by bytes"}
    return out


def ppo_training_lines():
    """The consumer of the env-step on the reference's training path (benchmarks/mjx_benchmark_PPO.py:50-60, ppo_config of
    myosuite/envs/myo/mjx/__init__.py:43-67 -- (64, 64, 64) networks, 10-step unroll; here 8 minibatches x 4 passes per batch of
    10 x E steps, the setting of benchmarks/ppo_rollout.py; the reference's own 8192-env protocol is benchmarks/mjx_benchmark_PPO.py):
    whole PPO iterations on the device (myosuite_amd/ppo.py: rollout and update are two HIP graphs, the learner is the fused
    kernels of include/myosim_ppo.h).  Train env-steps/s end to end, and of the rollout graph alone; NOT the headline metric."""
    import torch
    from myosuite_amd.envs import registry
    from myosuite_amd.ppo import OnDevicePPO, PPOConfig
    lines = []
    for env_id, ne, iters in (("myoHandPoseRandom-v0", 4096, 12), ("myoFatiLegWalk-v0", 1024, 12)):
        try:
            env = registry.make(env_id, num_envs=ne, seed=0)
            cfg = PPOConfig(unroll_length=10, num_minibatches=8, num_updates_per_batch=4, entropy_cost=1e-2, policy_hidden=(64, 64, 64),
                            value_hidden=(64, 64, 64), squash="sigmoid", normalize_observations=True)
            ppo = OnDevicePPO(env, cfg, seed=0)
            ppo.iterate()                                  # warm-up + graph capture
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                ppo._g_roll.replay()
            torch.cuda.synchronize()
            t_roll = time.perf_counter() - t0
            r0 = float(ppo.mean_reward)
            t0 = time.perf_counter()
            for _ in range(iters):
                ppo.iterate()
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            steps = iters * ppo.steps_per_iteration
            lines.append({"workload": f"PPO on {env_id}, {ne} envs/GPU: ppo_config of the reference (64,64,64) MLPs, unroll 10, 8 minibatches x 4 passes",
                          "key": f"ppo|{env_id}@{ne}", "train_env_steps_per_s": steps / t_all, "rollout_env_steps_per_s": steps / t_roll,
                          "iterations": iters, "env_steps_per_iteration": ppo.steps_per_iteration, "minibatch_updates_per_iteration": 32,
                          "learner": "fused HIP kernels (include/myosim_ppo.h)" if ppo.kern is not None else "torch autograd",
                          "hip_graphs": ppo._g_roll is not None and ppo._g_upd is not None,
                          # 12 iterations are a throughput sample, NOT a learning curve: a hand episode is 10 iterations long and all envs
                          # start in phase, so first / last differ by episode phase.  Learning on these two workloads is shown by the
                          # committed curves (benchmarks/ppo_rollout.py --curve, 1000 iterations) and asserted on the hand by
                          # tests/test_gpu_parity.py::test_ppo_learns_on_the_baseline_hand_workload
                          "mean_reward_per_step_first_last_of_this_sample": [r0, float(ppo.mean_reward)],
                          "learning_curve": {"myoHandPoseRandom-v0": "profiles/r05_ppo_curve_hand4096.json: mean reward per step -3.65 -> -3.39 (200 it.) -> -2.97 (1000 it., 41 M env-steps)",
                                             "myoFatiLegWalk-v0": "profiles/r05_ppo_curve_fatileg1024.json: 2.27 -> 9.13 reward per step, mean episode length 38 -> 128 steps (1000 it., 10 M env-steps)"}[env_id]})
            del ppo, env
        except Exception as exc:          # never takes the headline line down
            lines.append({"key": f"ppo|{env_id}@{ne}", "error": repr(exc)})
    return lines


LINE_LIMIT = 4096        # the contract line (the LAST stdout line) stays below this; everything else goes to the extras file


def accuracy_block():
    """north_star's accuracy target ("state divergence vs CPU mj_step < 1e-4 rel over 1000 steps") as measured by
    tests/tools/gpu_accuracy_run.py on the GPU box (the HIP kernels against the fp64 oracle on the same action streams) and
    committed under profiles/: REPLAYED here, not measured in this run -- bench.py's timed legs never touch the checker."""
    try:
        a = json.load(open(ACCURACY_PROFILE))
        return {"replayed_from": "profiles/" + os.path.basename(ACCURACY_PROFILE), "runs": a["runs"]}
    except (OSError, ValueError, KeyError):
        return None


def compact_line(full):
    """The ONE line the driver parses: the contract keys, a compact roofline and cpu_baseline, the accuracy block and
    an id -> env-steps/s digest of the other measured workloads.  Every other field of `full` is in the extras file / earlier lines."""
    rf = full["roofline"]
    c = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                              "vs_baseline", "dtype", "data")}
    cfg = full["config"]
    c["config"] = {k: cfg[k] for k in ("workload", "envs_per_gpu", "lanes_per_env", "parallelism", "baseline_config", "oversubscribed", "overrides") if k in cfg}
    c["roofline"] = {"bound": rf["bound"], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"], "frac": rf["frac"],
                     "traffic": rf["traffic"], "traffic_source": rf["traffic_source"], "traffic_over_algorithmic": rf["traffic_over_algorithmic"],
                     "kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                     "bound_by": rf.get("bound_by"), "fp32_vector_peak_frac": (rf.get("flops") or {}).get("frac")}
    cb = full.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "build": "fp64 C oracle (not libmujoco), gcc -O3 -march=native", "sample": cb["sample_short"],
                             "single_thread": cb["single_thread"]["value"]}
    acc = full.get("accuracy")
    if acc:
        c["accuracy"] = acc
    if "collective" in full:
        co = full["collective"]
        c["collective"] = {k: co[k] for k in ("backend", "is_rccl", "world_size", "rccl_version", "gathered_rows", "expected_rows",
                                              "bytes_per_rank", "allgather_us", "per_rank_env_steps_per_s")}
    c["stats"] = full["stats"]
    c["baseline_configs"] = {k: (round(v["env_steps_per_s"]) if "env_steps_per_s" in v else
                                 (round(v["train_env_steps_per_s"]) if "train_env_steps_per_s" in v else "error"))
                             for k, v in full.get("baseline_configs", {}).items()}
    c["extras"] = full.get("extras_file")
    return c


def emit(full, extras_file):
    """Write everything to the extras file, print it as EARLIER stdout lines (prefixed so that no reader takes them for the
    contract line), then print the compact contract line LAST."""
    full["extras_file"] = None
    if extras_file:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(extras_file)), exist_ok=True)
            with open(extras_file, "w") as f:
                json.dump(full, f, indent=1)
            full["extras_file"] = os.path.relpath(extras_file, ROOT)
        except OSError:
            pass
    for k in ("extra_configs", "ppo_training"):
        for row in full.get(k, []):
            print("#extra " + json.dumps({k: row}))
    print("#extra " + json.dumps({k: v for k, v in full.items() if k not in ("extra_configs", "ppo_training")}))
    line = json.dumps(compact_line(full))
    if len(line) >= LINE_LIMIT:       # never outgrow the reader again: drop the digest first, then the optional blocks
        c = compact_line(full)
        for k in ("baseline_configs", "accuracy", "stats", "collective"):
            c.pop(k, None)
            line = json.dumps(c)
            if len(line) < LINE_LIMIT:
                break
    sys.stdout.flush()
    print(line)
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--env", default="myoHandPoseRandom-v0")
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="preset = BASELINE.json config number: 2 elbow@4096, 3 hand@4096 (the default headline), 4 reorient@2048, "
                         "5 myoFatiLegWalk-v0@1024 per GPU (8192 over 8: `--gpus 8 --config 5`); overrides --env / --envs-per-gpu")
    ap.add_argument("--repeats", type=int, default=REPEATS, help="timed regions of --steps steps each (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ppo", action="store_true", help="skip the ppo_training lines (whole PPO iterations on the device)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs lines (elbow / reorient / leg-walk / self-contact hand)")
    ap.add_argument("--model", default=None, help="model override of the headline env (e.g. hand_contact): profile collection")
    ap.add_argument("--no-forward", action="store_true", help="do_forward=False override of the headline env: profile collection")
    ap.add_argument("--precision", default=None, help="precision override of the headline env (f64 | f64_state): profile collection")
    ap.add_argument("--extras-file", default=os.path.join(ROOT, "gpurun_out", "bench_extras.json"),
                    help="everything beyond the compact contract line (extra_configs, ppo_training, replayed counters, launch geometry)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY: rank r runs on cuda:(r %% visible devices) and the process group is gloo, so that the N > 1 code "
                         "path (launcher respawn, barrier, max over ranks, stats gather, sharded Philox streams) can be exercised on a "
                         "one-GPU box.  The printed line is marked oversubscribed and is NOT a scaling measurement.")
    args = ap.parse_args()
    if args.config:
        args.env, args.envs_per_gpu = BASELINE_PRESETS[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_launcher(args))

    import torch
    from myosuite_amd import dist as D
    from myosuite_amd import engine

    engine.lib()        # load libmyosim_hip.so now: a missing extension fails here, before any process group or timing
    rank, world, local = D.init_from_env(backend="gloo" if args.oversubscribe else None)
    if args.oversubscribe:
        local = local % max(1, torch.cuda.device_count())
    if world != args.gpus:
        # never print a line whose n_gpus differs from the request
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} device(s) are visible")
    torch.cuda.set_device(local)
    n = args.envs_per_gpu
    head_ov = {}
    if args.model:
        head_ov["model"] = args.model
    if args.no_forward:
        head_ov["do_forward"] = False
    if args.precision:
        head_ov["precision"] = args.precision
    clocks_before = gpu_clocks() if rank == 0 else None
    elapsed, kern_ms, env, stats, regions = measure(args.env, n, args.steps, args.warmup, rank, world, args.lanes, overrides=head_ov,
                                                    repeats=max(1, args.repeats))
    clocks_after = gpu_clocks() if rank == 0 else None
    cm = env.cm
    # every rank takes part (one small all-gather + the per-rank values); only rank 0 prints
    coll = collective_report(measure.ep_stats, stats, n, world, measure.own_elapsed_of_median_region, args.steps) if world > 1 else None

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        prec_name = {0: "f32", 1: "f64 (fp32 state rows)", 2: "f64"}
        out = {
            "metric": "env-steps/sec (whole node) at %d envs/GPU" % n,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "repeats": len(regions),
            "region_ms_per_step": [1e3 * e / args.steps for e in regions],      # every timed region; `value` is the median one
            "gpu_clocks_mhz": {"before": clocks_before, "after": clocks_after},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": prec_name[int(getattr(env, "precision", 0))], "data": "synthetic",
            "config": {"workload": f"{args.env}, {n} envs/GPU, random actions, one fused launch per env-step (synthetic model {cm.name})",
                       "workload_detail": f"random actions U[0,1) drawn in the kernel, frame_skip {env.frame_skip} + final forward + obs/reward + "
                                          f"episode stats + auto-reset in one launch per step; nq={cm.nq} nv={cm.nv} nu={cm.nu}",
                       "envs_per_gpu": n, "lanes_per_env": env.hm.launch_lanes(n), "launches_per_step": 1 if env._ro.autoreset else "1 + the task's masked reset",
                       "parallelism": f"env-shard x{world}"},
            "roofline": roofline(env, args.env, n, kern_ms, head_ov),
            "accuracy": accuracy_block(),
            "stats": {"mean_episode_return": float(stats[:, 0].mean()), "solved_frac": float(stats[:, 2].mean()),
                      "envs_in_stats": int(stats.shape[0]), "status_or": status_or(env.state.status)},
        }
        if coll is not None:
            out["collective"] = coll
        if args.config:
            out["config"]["baseline_config"] = args.config
        if head_ov:
            out["config"]["overrides"] = {k: str(v) for k, v in head_ov.items()}
        if args.oversubscribe:
            out["config"]["oversubscribed"] = f"{world} ranks on {torch.cuda.device_count()} device(s), gloo group: a test of the N > 1 path, not a scaling point"
        digest = {workload_key(args.env, n, head_ov) + " [headline]": {
            "env_steps_per_s": value, "kernel_ms": kern_ms, "hbm_frac_8d": out["roofline"]["frac"],
            "fp32_peak_frac": (out["roofline"].get("flops") or {}).get("frac"), "traffic_over_algorithmic": out["roofline"]["traffic_over_algorithmic"]}}
        del env
        if not args.no_cpu_baseline:
            # ~12 s of wall time on rank 0's host cores, after the timed region (the other ranks wait in destroy_process_group)
            out["cpu_baseline"] = cpu_baseline(args.env)
        if world == 1 and not args.no_extra and not args.no_ppo:
            out["ppo_training"] = ppo_training_lines()
        for line in out.get("ppo_training", []):
            digest[line["key"]] = ({"train_env_steps_per_s": line["train_env_steps_per_s"], "rollout_env_steps_per_s": line["rollout_env_steps_per_s"]}
                                   if "error" not in line else {"error": line["error"][:120]})
        if world == 1 and not args.no_extra:
            # driver-visible numbers for the other BASELINE.json configs (same timed loop, shorter): not the headline value
            extra = []
            for env_id, ne, ov in EXTRA_CONFIGS:
                tag = f"{env_id}, {ne} envs/GPU" + (f", {ov}" if ov else "")
                try:
                    # a short probe sizes the timed region: at least EXTRA_MIN_TIMED_MS of kernel work (and >= steps // 2 steps)
                    _, km0, ev0, _, _ = measure(env_id, ne, 4, 2, 0, 1, overrides=ov)
                    del ev0
                    ks = int(max(8, args.steps // 2, min(2000, EXTRA_MIN_TIMED_MS / max(km0, 1e-3))))
                    el, km, ev, st, rg = measure(env_id, ne, ks, max(2, args.warmup // 2), 0, 1, overrides=ov, repeats=max(1, args.repeats))
                    rf = roofline(ev, env_id, ne, km, ov)
                    extra.append({"workload": tag, "key": workload_key(env_id, ne, ov), "value": ne * ks / el, "unit": "env-steps/s", "steps": ks,
                                  "dtype": prec_name[int(getattr(ev, "precision", 0))],
                                  "ms_per_step": 1e3 * el / ks, "repeats": len(rg), "region_ms_per_step": [1e3 * e / ks for e in rg],
                                  "lanes_per_env": ev.hm.launch_lanes(ne),
                                  "launches_per_step": 1 if ev._ro.autoreset else "1 + the task's masked reset", "roofline": rf,
                                  "status_or": status_or(ev.state.status)})
                    digest[workload_key(env_id, ne, ov)] = {
                        "env_steps_per_s": ne * ks / el, "kernel_ms": km, "hbm_frac_8d": rf["frac"],
                        "fp32_peak_frac": (rf.get("flops") or {}).get("frac"), "traffic_over_algorithmic": rf["traffic_over_algorithmic"]}
                    del ev
                except Exception as exc:      # an extra line must never take the headline line down
                    extra.append({"workload": tag, "error": repr(exc)})
                    digest[workload_key(env_id, ne, ov)] = {"error": repr(exc)[:120]}
            out["extra_configs"] = extra
        out["baseline_configs"] = digest
        emit(out, args.extras_file)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
