"""PPO on the batched HIP envs: rollout collection + clipped-surrogate updates, the workload of the reference's
benchmarks/mjx_benchmark_PPO.py:18-66 (brax PPO on 8192 MJX envs), with the whole iteration on the device
(myosuite_amd/ppo.py: the unroll and the minibatch passes are HIP graphs, GAE is one kernel).

One process per GPU (torch.distributed over RCCL when launched with torch.distributed.run): every rank owns its own env shard;
the physics never communicates; gradients are all-reduced as ONE flat buffer per minibatch; episode statistics use one all-gather
per run (myosuite_amd/dist.py).

    python benchmarks/ppo_rollout.py --env myoFatiLegWalk-v0 --num-envs 1024 --iters 3
Prints one JSON line with rollout-only and end-to-end (train) env-steps/s.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from myosuite_amd import dist as D
from myosuite_amd.envs import registry
from myosuite_amd.ppo import OnDevicePPO, PPOConfig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="myoFatiLegWalk-v0")
    ap.add_argument("--num-envs", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--unroll", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--minibatches", type=int, default=8)
    ap.add_argument("--gamma", type=float, default=0.97)
    ap.add_argument("--lam", type=float, default=0.95)
    ap.add_argument("--clip", type=float, default=0.3)
    ap.add_argument("--eager", action="store_true", help="no HIP graphs (round 3's form): every op its own launch")
    ap.add_argument("--nets", choices=["reference", "brax"], default="reference",
                    help="reference: policy / value MLPs of (64, 64, 64), the reference's ppo_config (myosuite/envs/myo/mjx/__init__.py:43-67); "
                         "brax: brax PPO's own defaults, policy (32,)*4 and value (256,)*5 -- wider than the fused learner kernels take, "
                         "runs on the torch-autograd learner")
    ap.add_argument("--torch-learner", action="store_true", help="torch autograd + torch Adam instead of the fused learner kernels")
    ap.add_argument("--oversubscribe", action="store_true", help="TEST ONLY: ranks share the visible GPU(s), gloo group")
    ap.add_argument("--curve", default=None, help="write the learning curve (JSON: per iteration mean reward per step, terminations, truncations; "
                                                  "per window of --window iterations their means) to this file; reads three scalars back per "
                                                  "iteration, so the throughput of such a run is NOT the benchmark figure")
    ap.add_argument("--window", type=int, default=10, help="iterations per curve window (hand pose: 10 iterations = one 100-step episode)")
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--entropy", type=float, default=1e-2)
    ap.add_argument("--reward-scaling", type=float, default=1.0)
    ap.add_argument("--skip-rollout-only", action="store_true", help="no rollout-only timing pass before training (curve runs)")
    args = ap.parse_args()

    rank, world, local = D.init_from_env(backend="gloo" if args.oversubscribe else None)
    if args.oversubscribe:
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # rank r owns the global envs [r n, (r+1) n): Philox streams are keyed by the global env index (mm_state.env_index_base)
    env = registry.make(args.env, num_envs=args.num_envs, seed=rank, device=dev, env_index_base=rank * args.num_envs)
    cfg = PPOConfig(unroll_length=args.unroll, num_minibatches=args.minibatches, num_updates_per_batch=args.epochs,
                    discounting=args.gamma, gae_lambda=args.lam, clipping_epsilon=args.clip, entropy_cost=args.entropy, value_cost=0.25,
                    learning_rate=args.lr, reward_scaling=args.reward_scaling,
                    policy_hidden=(64, 64, 64) if args.nets == "reference" else (32, 32, 32, 32),
                    value_hidden=(64, 64, 64) if args.nets == "reference" else (256, 256, 256, 256, 256),
                    squash="sigmoid", normalize_observations=True)
    ppo = OnDevicePPO(env, cfg, seed=0, world=world, use_graphs=not args.eager, fused=False if args.torch_learner else None)
    ppo.iterate()                                   # warm-up + graph capture (not timed)
    torch.cuda.synchronize(); D.barrier()
    # rollout alone (the graph of the unroll), then full iterations
    t0 = time.perf_counter()
    for _ in range(0 if args.skip_rollout_only else args.iters):
        if ppo._g_roll is not None:
            ppo._g_roll.replay()
        else:
            ppo._rollout()
    torch.cuda.synchronize(); D.barrier(); t_roll = max(time.perf_counter() - t0, 1e-9)
    r0 = float(ppo.mean_reward)
    curve = []
    t0 = time.perf_counter()
    for _ in range(args.iters):
        ppo.iterate()
        if args.curve:
            curve.append((float(ppo.mean_reward) / cfg.reward_scaling, float(ppo.term_b.sum()), float(ppo.trunc_b.sum())))
    torch.cuda.synchronize(); D.barrier(); t_all = time.perf_counter() - t0
    stats = D.gather_episode_stats(ppo.ep_stats)
    # data-parallel ranks must hold the same parameters after every update (one all-reduce of the flat gradient per minibatch)
    in_sync, norm_in_sync, noise_differs = True, True, True
    if world > 1:
        def gathered(vec):
            vec = vec.cpu() if D.backend() == "gloo" else vec.to(dev)
            parts = [torch.zeros_like(vec) for _ in range(world)]
            torch.distributed.all_gather(parts, vec)
            return [q.cpu() for q in parts]
        parts = gathered(torch.stack([ppo.flat_p.double().sum(), ppo.flat_p.double().abs().sum(), ppo.flat_p[::97].double().sum()]))
        in_sync = all(bool(torch.allclose(parts[0], q, rtol=1e-6, atol=0)) for q in parts)
        if ppo.norm is not None:       # the observation normaliser is part of the policy / value function: merged over all ranks' rows
            parts = gathered(torch.cat([ppo.norm.mean.double(), ppo.norm.std.double(), ppo.norm.n.double().reshape(1)]))
            norm_in_sync = all(bool(torch.equal(parts[0], q)) for q in parts)
        if ppo.noise is not None:      # ... while the exploration noise must NOT be the same draw on every rank
            parts = gathered(ppo.noise[0, 0, :4].double())
            noise_differs = not any(bool(torch.equal(parts[0], q)) for q in parts[1:])
    t_roll = D.max_over_ranks(t_roll, device="cuda" if (world > 1 and D.backend() == "nccl") else None)
    t_all = D.max_over_ranks(t_all, device="cuda" if (world > 1 and D.backend() == "nccl") else None)
    if rank == 0 and args.curve:
        W = max(1, args.window)
        wins = []
        for k in range(0, len(curve) - W + 1, W):
            c = curve[k:k + W]
            ended = sum(x[1] + x[2] for x in c)
            wins.append({"iterations": [k, k + W], "env_steps_so_far": (k + W) * ppo.steps_per_iteration,
                         "mean_reward_per_step": sum(x[0] for x in c) / W,
                         "mean_episode_length": (W * ppo.T * ppo.n / ended) if ended else None,
                         "terminated_frac_of_ended": (sum(x[1] for x in c) / ended) if ended else None})
        json.dump({"env": args.env, "envs": args.num_envs, "unroll": args.unroll, "epochs": args.epochs, "minibatches": args.minibatches,
                   "lr": args.lr, "entropy_cost": args.entropy, "gamma": args.gamma, "lam": args.lam, "reward_scaling": args.reward_scaling,
                   "learner": "fused HIP kernels" if ppo.kern is not None else "torch autograd", "window_iterations": W,
                   "windows": wins, "per_iteration_mean_reward_per_step": [x[0] for x in curve]}, open(args.curve, "w"))
    if rank == 0:
        steps = args.iters * ppo.steps_per_iteration
        print(json.dumps({"env": args.env, "n_gpus": world, "envs_per_gpu": args.num_envs, "unroll": args.unroll, "iters": args.iters,
                          "graphs": ppo._g_roll is not None, "update_graph": ppo._g_upd is not None, "nets": args.nets,
                          "fused_learner_kernels": ppo.kern is not None, "epochs": args.epochs, "minibatches": args.minibatches,
                          "rollout_env_steps_per_s": steps / t_roll, "train_env_steps_per_s": steps / t_all,
                          "mean_reward_per_step_first_last": [r0, float(ppo.mean_reward)],
                          "params_in_sync_across_ranks": in_sync, "normaliser_in_sync_across_ranks": norm_in_sync,
                          "action_noise_differs_across_ranks": noise_differs, "mean_return_per_env": float(stats[:, 0].mean())}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
