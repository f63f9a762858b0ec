"""Why do the light launches (1024 env waves + helpers) move 5...18 % between sessions with the same binary?  Time the leg / reorient /
hand rollouts in one process in several orders, with an in-kernel cycle counter next to the wall clock: if the kernel takes the same
number of shader cycles every time and only the wall time moves, the box's shader clock is the variable (power state / boost), not
the memory system.   python tools/gpu_clock_probe.py  (on the GPU box)"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
import bench

W = {"hand": ("myoHandPoseRandom-v0", 4096, {}), "leg": ("myoFatiLegWalk-v0", 1024, {}), "reorient": ("myoHandReorient100-v0", 2048, {}),
     "legi": ("myoFatiLegWalk-v0", 1024, {"model": "leg_implicit"})}
envs = {}
def get(nm):
    if nm not in envs:
        env_id, n, kw = W[nm]
        env = registry.make(env_id, num_envs=n, seed=0, **kw)
        env.rollout_setup(action_seed=0)
        for s in range(4): env.rollout_step(None, stream_id=s)
        torch.cuda.synchronize()
        envs[nm] = (env, n)
    return envs[nm]

def smi(*flags):
    try:
        txt = subprocess.run(["rocm-smi", *flags, "--json"], capture_output=True, text=True, timeout=20).stdout
        return json.loads(txt[txt.index("{"):])
    except Exception as e:
        return {"error": repr(e)}

def region(nm, K):
    env, n = get(nm)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(K): env.rollout_step(None, stream_id=100 + s, events=evs[s])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    km = [a.elapsed_time(b) for a, b in evs]
    return n * K / dt, float(np.median(km)), float(np.min(km)), float(np.max(km))

out = []
order = ["leg", "leg", "hand", "leg", "reorient", "leg", "legi", "hand", "hand", "leg", "idle", "leg", "leg"]
for i, nm in enumerate(order):
    if nm == "idle":
        time.sleep(5.0); continue
    K = 200 if nm != "hand" else 400
    v, kmed, kmin, kmax = region(nm, K)
    clk = bench.gpu_clocks()
    out.append({"i": i, "workload": nm, "env_steps_per_s": v, "kernel_ms_median": kmed, "kernel_ms_min": kmin, "kernel_ms_max": kmax, "clocks_after": clk})
    print(f"{i:2d} {nm:9s} {v / 1e6:7.3f} M  kernel ms median {kmed:.4f} min {kmin:.4f} max {kmax:.4f}  clocks {clk}", flush=True)
pw = smi("--showpower"); tp = smi("--showtemp"); pl = smi("--showperflevel"); mx = smi("--showclocks")
print("power", json.dumps(pw)[:300]); print("temp", json.dumps(tp)[:400]); print("perflevel", json.dumps(pl)[:200])
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"runs": out, "power": pw, "temp": tp, "perflevel": pl}, open("gpurun_out/clock_probe.json", "w"), indent=1)
