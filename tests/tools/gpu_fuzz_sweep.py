"""One-off wide sweep of the generated-model fuzz tests (tests/test_fuzz_models.py) over seeds the committed suite does not run:
bug hunting, not a gate.   python tests/tools/gpu_fuzz_sweep.py [--first 32] [--count 200] [--out gpurun_out/fuzz_sweep.json]
Each test function is called directly with its seed; failures are collected (seed, test, first line of the assertion), not raised."""
import argparse, json, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz_models as F          # noqa: E402
from oracle import oracle as O        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=32); ap.add_argument("--count", type=int, default=200)
ap.add_argument("--budget-s", type=float, default=480.0); ap.add_argument("--out", default=None)
a = ap.parse_args()
O.build()
runs = [("models", F.test_gpu_random_models_match_the_oracle, lambda s: (O, s)),
        ("rk4", F.test_gpu_random_models_on_rk4_and_implicitfast, lambda s: (O, s, 1)),
        ("implicitfast", F.test_gpu_random_models_on_rk4_and_implicitfast, lambda s: (O, s, 3)),
        ("precision", F.test_gpu_random_models_in_precision_mode_track_the_oracle_to_fp64_resolution, lambda s: (O, s)),
        ("contact_scenes", F.test_gpu_random_contact_scenes_match_the_oracle, lambda s: (O, s)),
        ("articulated", F.test_gpu_random_articulated_models_with_contacts_match_the_oracle, lambda s: (O, s))]
t0 = time.time()
res = {k: {"ran": 0, "failed": []} for k, _, _ in runs}
devnull = open(os.devnull, "w")
for seed in range(a.first, a.first + a.count):
    if time.time() - t0 > a.budget_s:
        break
    for name, fn, mk in runs:
        f = getattr(fn, "__wrapped__", fn)
        try:
            so = sys.stdout; sys.stdout = devnull
            try:
                f(*mk(seed))
            finally:
                sys.stdout = so
            res[name]["ran"] += 1
        except BaseException as e:      # noqa: BLE001  (pytest.skip raises a BaseException subclass)
            if type(e).__name__ in ("Skipped",):
                continue
            res[name]["ran"] += 1
            res[name]["failed"].append({"seed": seed, "error": (str(e).strip().splitlines() or [type(e).__name__])[0][:300],
                                        "where": traceback.format_exc().strip().splitlines()[-3][:200]})
out = {"seeds": [a.first, seed], "elapsed_s": round(time.time() - t0, 1), "results": res}
print(json.dumps(out, indent=1))
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
