"""Time the REFERENCE's CPU path (MyoSuite + libmujoco) with the protocol of its own benchmarks/mjx_benchmark_baseline.py:8-25,
for anyone who has `mujoco`, `gymnasium` and the `myo_sim` models installed (none of them exist in the authoring container or
on the GPU box, which is why bench.py's `cpu_baseline` is the fp64 oracle, kind "port").

    python tools/time_reference_cpu.py [--num_steps 131072] [--envs myoElbowPose1D6MRandom-v0 myoHandPoseRandom-v0 ...]

Prints env-steps/s per env id, directly comparable with bench.py's `value` (same unit: one env-step = frame_skip physics
substeps + the sensor forward + obs / reward).
"""
import argparse
import timeit

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_steps", type=int, default=8192 * 16)      # mjx_benchmark_baseline.py:8
    ap.add_argument("--envs", nargs="*", default=["myoElbowPose1D6MRandom-v0", "myoFingerPoseRandom-v0", "myoHandReachRandom-v0",
                                                  "myoHandPoseRandom-v0", "myoHandReorient100-v0", "myoLegWalk-v0"])
    args = ap.parse_args()
    try:
        import myosuite  # noqa: F401  (registers the envs)
        from myosuite.utils import gym
    except Exception as e:   # pragma: no cover
        raise SystemExit(f"needs an installed MyoSuite (mujoco + gymnasium + myo_sim): {e}")
    out = {}
    for env_name in args.envs:
        env = gym.make(env_name)
        env.reset()

        def single_step():
            a = np.random.uniform(low=0.0, high=1.0, size=env.action_space.shape)
            env.step(a)

        res = timeit.repeat(single_step, number=args.num_steps, repeat=3)
        out[env_name] = args.num_steps / float(np.mean(res))
        print(f"{env_name}: {out[env_name]:.1f} env-steps/s (single CPU env, {args.num_steps} steps x 3)")
        env.close()
    print(out)


if __name__ == "__main__":
    main()
