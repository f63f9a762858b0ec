"""PPO on the batched HIP envs: rollout collection + clipped-surrogate updates, the workload of the reference's
benchmarks/mjx_benchmark_PPO.py:18-66 (brax PPO on 8192 MJX envs) restated in plain torch.

One process per GPU (torch.distributed over RCCL when launched with torch.distributed.run): every rank owns its own env
shard; the physics never communicates; gradients are all-reduced (one flattened buffer per minibatch) during the update;
episode statistics
use one all-gather per iteration (myosuite_amd/dist.py).

    python benchmarks/ppo_rollout.py --env myoFatiLegWalk-v0 --num-envs 1024 --iters 3
Prints one JSON line with rollout-only and end-to-end env-steps/s.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from myosuite_amd import dist as D
from myosuite_amd.envs import registry


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, act_dim, pi_hidden=(32, 32, 32, 32), v_hidden=(256, 256, 256, 256, 256)):   # brax PPO defaults
        super().__init__()
        def mlp(sizes):
            layers = []
            for a, b in zip(sizes[:-1], sizes[1:]):
                layers += [nn.Linear(a, b), nn.SiLU()]
            return nn.Sequential(*layers[:-1])
        self.pi = mlp((obs_dim,) + tuple(pi_hidden) + (act_dim,))
        self.v = mlp((obs_dim,) + tuple(v_hidden) + (1,))
        self.log_std = nn.Parameter(torch.full((act_dim,), -0.5))

    def dist(self, obs):
        return torch.distributions.Normal(self.pi(obs), self.log_std.exp())


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
    args = ap.parse_args()

    rank, world, local = D.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # rank r owns the global envs [r n, (r+1) n): Philox streams are keyed by the global env index (mm_state.env_index_base)
    env = registry.make(args.env, num_envs=args.num_envs, seed=0, device=dev, env_index_base=rank * args.num_envs)
    env.rollout_setup()                              # env.step + auto-reset + episode stats as one launch (mm_rollout_step)
    dense_col = env.rwd.shape[1] - 1
    n, T = args.num_envs, args.unroll
    obs_dim, act_dim = env.obs_dim, env.cm.nu
    torch.manual_seed(0)
    net = ActorCritic(obs_dim, act_dim).to(dev)
    if world > 1:                                   # identical initial weights on every rank
        for p_ in net.parameters():
            torch.distributed.broadcast(p_.data, src=0)
    opt = torch.optim.Adam(net.parameters(), lr=3e-4)
    obs, _ = env.reset(seed=rank)
    obs = obs.clone()
    buf = dict(obs=torch.zeros(T, n, obs_dim, device=dev), act=torch.zeros(T, n, act_dim, device=dev),
               logp=torch.zeros(T, n, device=dev), rew=torch.zeros(T, n, device=dev), done=torch.zeros(T, n, device=dev),
               val=torch.zeros(T + 1, n, device=dev))
    t_roll = t_all = 0.0
    ret_sum = torch.zeros(n, device=dev)
    for it in range(args.iters + 1):                   # iteration 0 is the warm-up (not timed)
        torch.cuda.synchronize(); D.barrier(); t0 = time.perf_counter()
        with torch.no_grad():
            for t in range(T):
                d = net.dist(obs)
                a = d.sample()
                buf["obs"][t] = obs; buf["act"][t] = a; buf["logp"][t] = d.log_prob(a).sum(-1); buf["val"][t] = net.v(obs).squeeze(-1)
                o, rw, ended = env.rollout_step(torch.sigmoid(a).contiguous())       # policy output -> [0,1] excitations
                r = rw[:, dense_col]
                buf["rew"][t] = r; buf["done"][t] = ended.float()
                ret_sum += r
                obs = o.clone()
            buf["val"][T] = net.v(obs).squeeze(-1)
            adv = torch.zeros(T, n, device=dev); last = torch.zeros(n, device=dev)
            for t in reversed(range(T)):                                   # GAE
                nd = 1.0 - buf["done"][t]
                delta = buf["rew"][t] + args.gamma * buf["val"][t + 1] * nd - buf["val"][t]
                last = delta + args.gamma * args.lam * nd * last
                adv[t] = last
            ret = adv + buf["val"][:T]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        B = T * n
        fo, fa, fl = buf["obs"].reshape(B, -1), buf["act"].reshape(B, -1), buf["logp"].reshape(B)
        fadv = ((adv - adv.mean()) / (adv.std() + 1e-8)).reshape(B); fret = ret.reshape(B)
        for _ in range(args.epochs):
            perm = torch.randperm(B, device=dev)
            for mb in perm.chunk(args.minibatches):
                mean = net.pi(fo[mb])
                dist = torch.distributions.Normal(mean, net.log_std.exp())
                ratio = (dist.log_prob(fa[mb]).sum(-1) - fl[mb]).exp()
                pg = -torch.min(ratio * fadv[mb], ratio.clamp(1 - args.clip, 1 + args.clip) * fadv[mb]).mean()
                vl = 0.5 * ((net.v(fo[mb]).squeeze(-1) - fret[mb]) ** 2).mean()
                loss = pg + 0.5 * vl - 1e-2 * dist.entropy().sum(-1).mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                if world > 1:                      # data-parallel PPO: one flattened gradient all-reduce per minibatch (RCCL)
                    flat = torch.cat([p_.grad.reshape(-1) for p_ in net.parameters()])
                    torch.distributed.all_reduce(flat); flat /= world
                    o_ = 0
                    for p_ in net.parameters():
                        n_ = p_.numel(); p_.grad.copy_(flat[o_:o_ + n_].view_as(p_.grad)); o_ += n_
                opt.step()
        torch.cuda.synchronize(); D.barrier(); t2 = time.perf_counter()
        if it > 0:
            t_roll += t1 - t0; t_all += t2 - t0
    stats = D.gather_episode_stats(torch.stack([ret_sum, torch.ones_like(ret_sum), torch.zeros_like(ret_sum)], dim=1))
    t_roll = D.max_over_ranks(t_roll, device="cuda" if world > 1 else None)
    t_all = D.max_over_ranks(t_all, device="cuda" if world > 1 else None)
    if rank == 0:
        steps = args.iters * T * n * world
        print(json.dumps({"env": args.env, "n_gpus": world, "envs_per_gpu": n, "unroll": T, "iters": args.iters,
                          "rollout_env_steps_per_s": steps / t_roll, "train_env_steps_per_s": steps / t_all,
                          "mean_return_per_env": float(stats[:, 0].mean())}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
