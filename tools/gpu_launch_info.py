import sys; sys.path.insert(0, '/root/repo')
from myosuite_amd.envs import registry
for env_id, n, kw in (("myoFatiLegWalk-v0", 1024, {}), ("myoFatiLegWalk-v0", 1024, {"model": "leg_implicit"}), ("myoElbowPose1D6MRandom-v0", 4096, {}), ("myoHandPoseRandom-v0", 4096, {})):
    env = registry.make(env_id, num_envs=n, seed=0, **kw)
    cm = env.cm
    print(env_id, kw, env.hm.launch_info(n), dict(nq=cm.nq, nv=cm.nv, nbody=cm.nbody, nu=cm.nu, ntendon=cm.ntendon, npair=cm.npair, neq=cm.neq, njmax=cm.njmax))
