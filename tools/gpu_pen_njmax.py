"""PenTwirl with the pen's condim-4 contacts: throughput, launch geometry and dropped-contact frequency (status bit 8) for row bounds
56 / 60 / 64 at 2048 envs:  python tools/gpu_pen_njmax.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd import engine as E
from myosuite_amd.model import synth
from myosuite_amd.envs import registry

n = 2048
for nj, ncon in ((56, 12), (56, 12), (60, 12), (64, 12)):      # (the first line is the warm-up of the box)
    def edit(spec, nj=nj, ncon=ncon):
        spec.njmax = nj; spec.nconmax = ncon
    synth._CACHE["hand_pen"] = synth.compile_spec("hand_pen", edit)
    from myosuite_amd.envs import base_v0
    base_v0._MODEL_CACHE.clear()                     # the env layer keeps compiled models by name
    env = registry.make("myoHandPenTwirlRandom-v0", num_envs=n, seed=1)
    env.rollout_setup(action_seed=3)
    for s in range(30):
        env.rollout_step(None, stream_id=s)
    torch.cuda.synchronize()
    flagged = 0
    t0 = time.perf_counter()
    K = 200
    for s in range(K):
        env.rollout_step(None, stream_id=100 + s)
        if s % 20 == 19:
            flagged = max(flagged, int(((env.state.status & 8) != 0).sum()))
    assert env.cm.njmax == nj
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    info = env.hm.launch_info(n) if hasattr(env.hm, "launch_info") else {}
    print(f"njmax {nj} nconmax {ncon}: {n / dt / 1e6:.3f} M env-steps/s  {dt * 1e3:.3f} ms/step   envs with a dropped contact (max over samples) {flagged}/{n}   {info}")
    del env
