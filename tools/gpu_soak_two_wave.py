"""Soak of the two-wave launches: long rollouts at the benchmark batch sizes and at small batches; no wave may ever give up waiting
for its partner (status bit 16), nothing may turn non-finite.   python tools/gpu_soak_two_wave.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd.envs import registry
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for env_id, n, kw in (("myoFatiLegWalk-v0", 1024, {}), ("myoElbowPose1D6MRandom-v0", 4096, {}), ("myoHandReorient100-v0", 96, {}),
                      ("myoHandPoseRandom-v0", 512, {}), ("myoFatiLegWalk-v0", 1024, {"model": "leg_implicit"}), ("myoHandKeyTurnRandom-v0", 200, {})):
    env = registry.make(env_id, num_envs=n, seed=0, **kw)
    env.rollout_setup(action_seed=0)
    t0 = time.perf_counter()
    for s in range(steps):
        env.rollout_step(None, stream_id=s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = int(env.state.status.max())
    fin = bool(torch.isfinite(env.state.qpos).all() and torch.isfinite(env.state.qvel).all())
    print(f"{env_id} {kw} n={n} steps={steps}: {n * steps / dt / 1e6:.3f} M env-steps/s, status_or={st} (bit16={'SET' if st & 16 else 'clear'}), finite={fin}")
    assert st & 16 == 0 and fin
print("SOAK OK")
