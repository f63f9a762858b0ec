"""bench.py output contract (driver-facing): checked on the JSON lines committed under profiles/ (captured on MI355X)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_under_rocprof.json")))
    assert files, "no committed bench lines"
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for f in files:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, (f, k)
        assert d["metric"].startswith("env-steps/sec") and base["metric"].startswith("env-steps/sec")
        assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
        assert "workload" in d["config"] and "model" not in d["config"] and d["n_gpus"] == 1
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert d["value"] > 0 and abs(d["ms_per_step"] - 1e3 * d["config"]["envs_per_gpu"] * d["n_gpus"] / d["value"]) < 1e-6 * d["ms_per_step"] + 1e-9


def test_bench_refuses_a_rank_count_that_differs_from_the_request(tmp_path):
    """`--gpus N` never prints a line for another N: under a launcher with a different WORLD_SIZE it exits with an error, and
    without a launcher it re-executes itself under torch.distributed.run (respawn_under_launcher) instead of running one rank."""
    import subprocess, sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (out.stderr + out.stdout) and "n_gpus" not in out.stdout
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "respawn_under_launcher" in src and "torch.distributed.run" in src and 'if args.gpus > 1 and "WORLD_SIZE" not in os.environ' in src


def test_bench_defaults_are_the_baseline_workload():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'default="myoHandPoseRandom-v0"' in src and "default=4096" in src            # BASELINE.json configs[2], 4096 envs/GPU
    assert "cpu_baseline" in src and "barrier" in src and "max_over_ranks" in src


COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline")


def _check_compact_line(d, cpu=True):
    """the contract line bench.py prints LAST: the driver's keys + compact `roofline` / `cpu_baseline`"""
    for k in COMPACT_KEYS:
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    for k in ("workload", "envs_per_gpu", "lanes_per_env"):
        assert k in d["config"], k
    assert "model" not in d["config"] and len(d["config"]["workload"]) <= 128        # (the driver's record keeps 128 characters of it)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "kernel_ms", "bound_by",
              "fp32_vector_peak_frac"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["value"] > 0 and abs(d["ms_per_step"] - 1e3 * d["config"]["envs_per_gpu"] * d["n_gpus"] / d["value"]) < 1e-6 * d["ms_per_step"] + 1e-9
    if cpu:
        cb = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "build", "sample", "single_thread"):
            assert k in cb, k
        assert cb["kind"] == "port" and cb["value"] > 0


def test_compact_line_of_a_full_record_is_short_and_complete():
    """bench.compact_line over the largest full record in the tree (the round-5 default run: 27 KB) stays below 4 KB and keeps the
    contract keys -- the size of the LAST stdout line is what the driver's reader depends on."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    assert len(json.dumps(full)) > 20000
    full["cpu_baseline"]["sample_short"] = "8192 envs x 40 env-steps, 16 threads, 5.1 s"
    full["config"]["workload"] = full["config"]["workload"][:120]
    full["accuracy"] = b.accuracy_block()
    full["collective"] = {"backend": "nccl", "is_rccl": True, "world_size": 8, "rccl_version": "2.26.6", "gathered_rows": 32768, "expected_rows": 32768,
                          "bytes_per_rank": 49152, "allgather_us": 55.2, "per_rank_env_steps_per_s": [6894156.123456] * 8, "what": "x"}
    c = b.compact_line(full)
    line = json.dumps(c)
    assert len(line) < b.LINE_LIMIT == 4096, len(line)
    _check_compact_line(c)
    assert len(c["baseline_configs"]) >= 10 and c["collective"]["world_size"] == 8
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "MM_PREC_MIXED" not in src and "mixed" not in src.lower().replace("mixed_study", "")


import pytest


@pytest.mark.gpu
def test_bench_n_gt_1_path_runs_oversubscribed_on_one_gpu():
    """`python bench.py --gpus 2 --oversubscribe`: the whole N > 1 path -- respawn_under_launcher (torch.distributed.run, two
    ranks), sharded Philox streams (env_index_base = rank * E), barrier + max-over-ranks timing, the episode-stats gather -- on
    ONE GPU with a gloo group (RCCL refuses two ranks on one device).  One JSON line, n_gpus = 2, marked oversubscribed; its
    gathered statistics are those of an UNSHARDED rollout of 2 E envs (rank r's envs are rows [r E, (r + 1) E))."""
    import subprocess, sys
    import torch
    from myosuite_amd.envs import registry
    E_, W, K = 256, 2, 6
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", str(K), "--warmup", str(W),
                          "--envs-per-gpu", str(E_), "--no-extra", "--repeats", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    assert out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096         # the contract line is the LAST line, and short
    d = json.loads(lines[0])
    _check_compact_line(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "oversubscribed" in d["config"] and d["config"]["parallelism"] == "env-shard x2"
    assert d["stats"]["envs_in_stats"] == 2 * E_                      # the gather returned both shards
    assert abs(d["ms_per_step"] - 1e3 * 2 * E_ / d["value"]) < 1e-6 * d["ms_per_step"] + 1e-9
    assert d["cpu_baseline"]["single_thread"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "-O3 -march=native" in d["cpu_baseline"]["build"]         # the timed CPU build is the optimised one, and says so
    # the N > 1 line is self-evidencing about its one collective
    c = d["collective"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and c["is_rccl"] is False          # (RCCL refuses two ranks on one device)
    assert c["gathered_rows"] == c["expected_rows"] == 2 * E_ and c["bytes_per_rank"] == E_ * 3 * 4
    assert c["allgather_us"] > 0 and len(c["per_rank_env_steps_per_s"]) == 2
    assert min(c["per_rank_env_steps_per_s"]) * 2 >= d["value"] * 0.999      # `value` is priced on the slowest rank's clock
    assert "rccl_version" in c
    # the same rollout unsharded, in this process
    full = registry.make("myoHandPoseRandom-v0", num_envs=2 * E_, seed=0)
    stats = full.rollout_setup(action_seed=0)
    for s in range(W + K):
        full.rollout_step(None, stream_id=s)
    torch.cuda.synchronize()
    assert abs(float(stats[:, 0].mean()) - d["stats"]["mean_episode_return"]) < 1e-4 * max(1.0, abs(float(stats[:, 0].mean())))
    assert abs(float(stats[:, 2].mean()) - d["stats"]["solved_frac"]) < 1e-6


@pytest.mark.gpu
def test_bench_config_5_preset_runs_the_fati_leg_share_on_two_ranks():
    """`--config 5` = BASELINE.json config 5's per-GPU share (myoFatiLegWalk-v0, 1024 envs per GPU: 8192 over 8): the preset through the
    N > 1 path on the one-GPU box (two oversubscribed ranks), with the collective block and both ranks' own rates in the line."""
    import subprocess, sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--config", "5", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--repeats", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["baseline_config"] == 5 and d["config"]["envs_per_gpu"] == 1024
    assert "myoFatiLegWalk-v0" in d["config"]["workload"] and d["metric"].endswith("1024 envs/GPU")
    c = d["collective"]
    assert c["gathered_rows"] == c["expected_rows"] == 2048 and len(c["per_rank_env_steps_per_s"]) == 2
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 5336 * 1024          # 5 336 B per env-step of the fatigue leg
    _check_compact_line(d, cpu=False)


def _fracs(node, path=""):
    """every (path, value) whose key ends in _frac / is `frac`, anywhere in the line"""
    if isinstance(node, dict):
        for k, v in node.items():
            if isinstance(v, (int, float)) and (k == "frac" or k.endswith("_frac")):
                yield path + "/" + k, float(v)
            else:
                yield from _fracs(v, path + "/" + k)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from _fracs(v, f"{path}[{i}]")


@pytest.mark.gpu
def test_bench_line_bookkeeping_repeats_fractions_and_replayed_counters(tmp_path):
    """The default run (short): the LAST stdout line is the compact contract line (< 4 KB: the round-5 line had grown to 27 KB and the
    driver could not read it back), everything else is in the extras file.  `repeats` timed regions with the median reported, no fraction above 1 anywhere (the valu-busy
    fraction is priced per RESIDENT wave: mm_model_launch_info), counters that were not measured in the run sit under
    roofline.profile and name the committed file they are replayed from, the GPU clocks are logged around the timed region."""
    import subprocess, sys
    xf = str(tmp_path / "bench_extras.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--extras-file", xf],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.rstrip().splitlines()[-1]
    assert len(last) < 4096 and [l for l in out.stdout.splitlines() if l.startswith("{")] == [last]      # ONE parseable line, the last, short
    c = json.loads(last)
    _check_compact_line(c)
    assert len(c["baseline_configs"]) >= 10 and all(isinstance(v, int) and v > 0 for v in c["baseline_configs"].values()), c["baseline_configs"]
    assert list(c["baseline_configs"].keys())[-3:] == ["myoElbowPose1D6MRandom-v0@4096", "myoHandReorient100-v0@2048", "myoFatiLegWalk-v0@1024"]
    assert c["roofline"]["algorithmic_bytes_per_launch"] == 1376 * 4096 and c["roofline"]["bound_by"] == "fp32-issue"
    # everything else: the extras file (and the '#extra' lines before the contract line)
    d = json.load(open(xf))
    assert any(l.startswith("#extra ") for l in out.stdout.splitlines())
    assert d["value"] == c["value"] and d["roofline"]["kernel_ms"] == c["roofline"]["kernel_ms"]
    assert d["repeats"] >= 3 and len(d["region_ms_per_step"]) == d["repeats"]
    assert sorted(d["region_ms_per_step"])[(d["repeats"] - 1) // 2] == pytest.approx(d["ms_per_step"], rel=1e-9)
    assert "gpu_clocks_mhz" in d
    fr = list(_fracs(d))
    assert len(fr) >= 10, fr
    for path, v in fr:
        assert 0.0 <= v <= 1.0, (path, v)
    lines = [d] + [x for x in d["extra_configs"] if "error" not in x]
    assert len(lines) == 1 + len(d["extra_configs"]), [x for x in d["extra_configs"] if "error" in x]
    for x in lines:
        r = x["roofline"]
        assert r["launch"]["resident_blocks_per_cu"] >= 1 and r["launch"]["vgprs"] > 0
        if r["profile"] is not None:
            assert r["profile"]["replayed_from"].startswith("profiles/") and os.path.exists(os.path.join(ROOT, r["profile"]["replayed_from"]))
            assert r["traffic_source"] == r["profile"]["replayed_from"]
        else:
            assert r["traffic"] is None
    # both byte figures, SURVEY 8(d)'s as the contract one
    r = d["roofline"]
    assert r["algorithmic_bytes_per_launch"] == 1376 * 4096 and r["algorithmic_bytes_per_launch_incl_carry"] == 1752 * 4096
    assert r["frac"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 / r["peak"], rel=1e-9)
    assert r["frac_incl_carry"] > r["frac"] and r["bound_by"] == "fp32-issue" and r["bound_by_detail"]["x_ridge"] > 10
    assert [x["key"] for x in d["extra_configs"]][-3:] == ["myoElbowPose1D6MRandom-v0@4096", "myoHandReorient100-v0@2048", "myoFatiLegWalk-v0@1024"]
    keys = {x["key"] for x in d["extra_configs"]}
    assert "myoHandReachRandom-v0@4096" in keys and "myoHandPoseRandom-v0@4096|precision=f64_state" in keys
    ppo = {x["key"]: x for x in d["ppo_training"]}
    assert not [x for x in d["ppo_training"] if "error" in x], d["ppo_training"]
    hand, leg = ppo["ppo|myoHandPoseRandom-v0@4096"], ppo["ppo|myoFatiLegWalk-v0@1024"]
    assert hand["hip_graphs"] and hand["learner"].startswith("fused") and hand["train_env_steps_per_s"] >= 1.5e6 and leg["train_env_steps_per_s"] >= 0.4e6
    assert hand["rollout_env_steps_per_s"] >= hand["train_env_steps_per_s"]
    prec = [x for x in d["extra_configs"] if x["key"] == "myoHandPoseRandom-v0@4096|precision=f64_state"][0]
    assert prec["dtype"] == "f64" and prec["value"] >= 1.0e6          # BASELINE.json's throughput target holds in precision mode too
