"""bench.py output contract (driver-facing): checked on the JSON lines committed under profiles/ (captured on MI355X)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01?_*_bench_under_rocprof.json")))
    assert files, "no committed bench lines"
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for f in files:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, (f, k)
        assert d["metric"].startswith("env-steps/sec") and base["metric"].startswith("env-steps/sec")
        assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert d["value"] > 0 and abs(d["ms_per_step"] - 1e3 * d["config"]["envs_per_gpu"] * d["n_gpus"] / d["value"]) < 1e-6 * d["ms_per_step"] + 1e-9


def test_bench_defaults_are_the_baseline_workload():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'default="myoHandPoseRandom-v0"' in src and "default=4096" in src            # BASELINE.json configs[2], 4096 envs/GPU
    assert "cpu_baseline" in src and "barrier" in src and "max_over_ranks" in src
