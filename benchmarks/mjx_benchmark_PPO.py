"""PPO training-time benchmark -- the protocol of the reference's benchmarks/mjx_benchmark_PPO.py:18-89 on the batched HIP
engine: same CLI (``--env_name --impl --num_envs``), same PPO hyper-parameters (``ppo_config`` of
myosuite/envs/myo/mjx/__init__.py:43-67 with ``num_timesteps = 5_000_000`` and ``num_evals = 2``), ``timeit.repeat(number=1,
repeat=3)`` around a full training run, same ``.npy`` result file.  brax's PPO is restated in torch: per training step
``batch_size * num_minibatches`` trajectories of ``unroll_length`` steps are collected from ``num_envs`` envs, then
``num_updates_per_batch`` passes over ``num_minibatches`` minibatches (clipped surrogate + value + entropy losses, GAE,
running observation normalisation, global grad-norm clipping, Adam) -- with the unroll and the minibatch passes captured into
HIP graphs and GAE as one kernel (myosuite_amd/ppo.py).

    python benchmarks/mjx_benchmark_PPO.py --env_name MjxElbowPoseRandom-v0 --impl hip --num_envs 8192
"""
import argparse
import os
import sys
import timeit

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from myosuite_amd import mjx_api
from myosuite_amd.ppo import OnDevicePPO, PPOConfig


def train(env, num_envs, num_timesteps, seed, p, log=print, graphs=True):
    """brax ppo.train restated on the device (myosuite_amd/ppo.py): per training step `batch_size * num_minibatches` trajectories
    of `unroll_length` steps from `num_envs` envs, then `num_updates_per_batch` passes over `num_minibatches` minibatches."""
    nf = p["network_factory"]
    ntraj = p["batch_size"] * p["num_minibatches"]
    assert ntraj % num_envs == 0 or num_envs % ntraj == 0
    cfg = PPOConfig(unroll_length=p["unroll_length"], num_minibatches=p["num_minibatches"], num_updates_per_batch=p["num_updates_per_batch"],
                    learning_rate=p["learning_rate"], discounting=p["discounting"], gae_lambda=p["gae_lambda"], entropy_cost=p["entropy_cost"],
                    clipping_epsilon=p["clipping_epsilon"], max_grad_norm=p.get("max_grad_norm"), reward_scaling=p["reward_scaling"],
                    normalize_observations=p["normalize_observations"], policy_hidden=tuple(nf["policy_hidden_layer_sizes"]),
                    value_hidden=tuple(nf["value_hidden_layer_sizes"]), squash="tanh",
                    unrolls=max(1, ntraj // num_envs))           # brax: batch_size * num_minibatches // num_envs unrolls per step
    ppo = OnDevicePPO(env, cfg, seed=int(seed), use_graphs=graphs)
    niter = max(1, int(np.ceil(num_timesteps / ppo.steps_per_iteration)))
    done_steps = 0
    for it in range(niter):
        ppo.iterate()
        done_steps += ppo.steps_per_iteration
        if it in (0, niter // 2, niter - 1):                  # num_evals = 2 (+ the initial one): report the training reward
            log(f"  step {done_steps}: mean reward/step {float(ppo.mean_reward):.4f}")
    torch.cuda.synchronize()
    return ppo


def measure_num_env_training_steps(meta_seed=0, env_name="MjxElbowPoseRandom-v0", impl="hip", num_timesteps=5_000_000,
                                   num_envs=8192, repeat=3):
    """Measure how the number of envs influences execution time (total number of steps) -- mjx_benchmark_PPO.py:18-66."""
    rng = np.random.default_rng(seed=meta_seed)
    env = mjx_api.TrainingWrapper(mjx_api.make(env_name, config_overrides={"impl": impl}, num_envs=num_envs))
    ppo_params = dict(mjx_api.PPO_CONFIG)
    ppo_params["num_timesteps"] = num_timesteps
    ppo_params["num_evals"] = 2
    print(f"Training on environment:\n{env_name}")
    print(f"Using backend:\n{impl}")
    print(f"Environment Config:\n{mjx_api.get_default_config(env_name)}")
    print(f"PPO Training Parameters:\n{mjx_api.PPO_CONFIG}")
    print(f"Testing env num: {num_envs}")

    def run_benchmark():
        train(env, num_envs, num_timesteps, rng.integers(0, 10000), ppo_params)

    results = timeit.repeat(run_benchmark, number=1, repeat=repeat)
    print(f"Results for {num_envs} envs: PPO training for {num_timesteps} total steps take {results} seconds.")
    print(np.mean(results))
    return results


def main():
    parser = argparse.ArgumentParser(description="Benchmark PPO training for the MJX-style MyoSuite environments on the HIP engine.")
    parser.add_argument("--env_name", type=str, default="MjxElbowPoseRandom-v0", help="Environment name")
    parser.add_argument("--impl", type=str, default="hip", help="Backend implementation (only this engine)")
    parser.add_argument("--num_envs", type=int, default=8192, help="Number of environments")
    parser.add_argument("--num_timesteps", type=int, default=5_000_000, help="total env steps per training run (reference: 5e6)")
    parser.add_argument("--repeat", type=int, default=3, help="timeit repeats (reference: 3)")
    args = parser.parse_args()
    results = {}
    print(f"MyoSuite env {args.env_name} -- Testing implementation: {args.impl} ({args.num_envs} envs)")
    results[f"{args.env_name}_{args.impl}_{args.num_envs}"] = measure_num_env_training_steps(
        meta_seed=0, env_name=args.env_name, impl=args.impl, num_envs=args.num_envs, num_timesteps=args.num_timesteps,
        repeat=args.repeat)
    print(results)
    np.save(f"mjx_benchmark_PPO_results_{args.env_name}_{args.impl}_{args.num_envs}.npy", results, allow_pickle=True)


if __name__ == "__main__":
    main()
