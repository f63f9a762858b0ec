"""Stage cycles of the hand kernel with 1 vs 2 waves per SIMD (2048 vs 4096 envs, G = 32): a stage whose wall time doubles with
the second wave is issue-bound, one whose time stays is latency-bound (the other wave fits into its stalls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
res = {}
for n in (1024, 2048, 4096, 8192):
    env = registry.make("myoHandPoseRandom-v0", num_envs=n, seed=0, lanes_per_env=32)
    a = torch.rand(n, env.cm.nu, device="cuda")
    for _ in range(3): env.step(a)
    tot = None
    for rep in range(3):
        pf = E.profile_stages(lambda: E.env_step(env.hm, env.state, a, env._task))
        tot = pf if tot is None else {k: min(tot[k], pf[k]) for k in pf}
    res[n] = {k: v // 11 for k, v in tot.items() if v}
    print(n, res[n])
print("ratio 4096/2048:", {k: round(res[4096][k] / max(1, res[2048][k]), 2) for k in res[4096]})
print("ratio 8192/4096:", {k: round(res[8192][k] / max(1, res[4096][k]), 2) for k in res[4096]})
