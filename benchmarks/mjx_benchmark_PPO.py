"""PPO training-time benchmark -- the protocol of the reference's benchmarks/mjx_benchmark_PPO.py:18-89 on the batched HIP
engine: same CLI (``--env_name --impl --num_envs``), same PPO hyper-parameters (``ppo_config`` of
myosuite/envs/myo/mjx/__init__.py:43-67 with ``num_timesteps = 5_000_000`` and ``num_evals = 2``), ``timeit.repeat(number=1,
repeat=3)`` around a full training run, same ``.npy`` result file.  brax's PPO is restated in torch: per training step
``batch_size * num_minibatches`` trajectories of ``unroll_length`` steps are collected from ``num_envs`` envs, then
``num_updates_per_batch`` passes over ``num_minibatches`` minibatches (clipped surrogate + value + entropy losses, GAE,
running observation normalisation, global grad-norm clipping, Adam).

    python benchmarks/mjx_benchmark_PPO.py --env_name MjxElbowPoseRandom-v0 --impl hip --num_envs 8192
"""
import argparse
import os
import sys
import timeit

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn

from myosuite_amd import mjx_api


def mlp(sizes):
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers += [nn.Linear(a, b), nn.SiLU()]          # brax networks: swish activations
    return nn.Sequential(*layers[:-1])


class RunningNorm:
    """brax running_statistics: per-feature mean / std over everything seen so far."""

    def __init__(self, dim, device):
        self.n = torch.zeros((), device=device); self.mean = torch.zeros(dim, device=device); self.m2 = torch.zeros(dim, device=device)

    def update(self, x):
        x = x.reshape(-1, x.shape[-1])
        b = x.shape[0]
        tot = self.n + b
        d = x.mean(0) - self.mean
        self.m2 += ((x - x.mean(0)) ** 2).sum(0) + d * d * self.n * b / tot
        self.mean += d * b / tot
        self.n = tot

    def __call__(self, x):
        std = torch.sqrt(self.m2 / torch.clamp(self.n, min=1.0)).clamp(1e-6, 1e6)
        return ((x - self.mean) / std).clamp(-5.0, 5.0)


def train(env, num_envs, num_timesteps, seed, p, log=print):
    dev = env.env._env.device
    obs_dim, act_dim = env.observation_size, env.action_size
    nf = p["network_factory"]
    torch.manual_seed(int(seed))
    pi = mlp((obs_dim,) + tuple(nf["policy_hidden_layer_sizes"]) + (2 * act_dim,)).to(dev)      # mean and scale parameters
    vf = mlp((obs_dim,) + tuple(nf["value_hidden_layer_sizes"]) + (1,)).to(dev)
    params = list(pi.parameters()) + list(vf.parameters())
    opt = torch.optim.Adam(params, lr=p["learning_rate"])
    norm = RunningNorm(obs_dim, dev) if p["normalize_observations"] else None
    T = p["unroll_length"]
    ntraj = p["batch_size"] * p["num_minibatches"]
    assert ntraj % num_envs == 0 or num_envs % ntraj == 0
    unrolls = max(1, ntraj // num_envs)                      # brax: batch_size * num_minibatches // num_envs unrolls per step
    steps_per_iter = unrolls * T * num_envs
    niter = max(1, int(np.ceil(num_timesteps / steps_per_iter)))

    def dist(o):
        out = pi(norm(o) if norm else o)
        mean, raw = out[..., :act_dim], out[..., act_dim:]
        return torch.distributions.Normal(mean, torch.nn.functional.softplus(raw) + 1e-3)     # brax NormalTanh parametric dist

    st = env.reset(int(seed))
    done_steps = 0
    for it in range(niter):
        O, A, LP, R, Dn, Tr, V = [], [], [], [], [], [], []
        with torch.no_grad():
            for u in range(unrolls * T):
                o = st.obs["state"].clone()
                d = dist(o)
                raw = d.sample()
                lp = (d.log_prob(raw) - 2.0 * (np.log(2.0) - raw - torch.nn.functional.softplus(-2.0 * raw))).sum(-1)   # tanh squash
                st = env.step(st, torch.tanh(raw))
                O.append(o); A.append(raw); LP.append(lp); R.append(st.reward * p["reward_scaling"]); Dn.append(st.done)
                Tr.append(st.info["truncation"]); V.append(vf(norm(o) if norm else o).squeeze(-1))
            last_o = st.obs["state"].clone()
            O = torch.stack(O); A = torch.stack(A); LP = torch.stack(LP); R = torch.stack(R); Dn = torch.stack(Dn); Tr = torch.stack(Tr)
            V = torch.stack(V + [vf(norm(last_o) if norm else last_o).squeeze(-1)])
            if norm:
                norm.update(O)
            # GAE with truncation (brax compute_gae): bootstrap through time-limit ends, cut at terminations
            term = Dn * (1.0 - Tr)
            adv = torch.zeros_like(R); last = torch.zeros_like(R[0])
            for t in reversed(range(R.shape[0])):
                delta = R[t] + p["discounting"] * (1.0 - term[t]) * V[t + 1] - V[t]
                last = delta + p["discounting"] * p["gae_lambda"] * (1.0 - term[t]) * (1.0 - Tr[t]) * last
                adv[t] = last
            ret = adv + V[:-1]
        B = O.shape[0] * O.shape[1]
        fo, fa, fl = O.reshape(B, -1), A.reshape(B, -1), LP.reshape(B)
        fadv = ((adv - adv.mean()) / (adv.std() + 1e-8)).reshape(B); fret = ret.reshape(B)
        for _ in range(p["num_updates_per_batch"]):
            perm = torch.randperm(B, device=dev)
            for mb in perm.chunk(p["num_minibatches"]):
                d = dist(fo[mb])
                raw = fa[mb]
                lp = (d.log_prob(raw) - 2.0 * (np.log(2.0) - raw - torch.nn.functional.softplus(-2.0 * raw))).sum(-1)
                ratio = (lp - fl[mb]).exp()
                eps = p["clipping_epsilon"]
                pg = -torch.min(ratio * fadv[mb], ratio.clamp(1 - eps, 1 + eps) * fadv[mb]).mean()
                v = vf(norm(fo[mb]) if norm else fo[mb]).squeeze(-1)
                vl = 0.5 * 0.5 * ((v - fret[mb]) ** 2).mean()
                ent = d.entropy().sum(-1).mean()
                loss = pg + vl - p["entropy_cost"] * ent
                opt.zero_grad(set_to_none=True)
                loss.backward()
                if p.get("max_grad_norm"):
                    torch.nn.utils.clip_grad_norm_(params, p["max_grad_norm"])
                opt.step()
        done_steps += steps_per_iter
        if it in (0, niter // 2, niter - 1):                  # num_evals = 2 (+ the initial one): report the training reward
            log(f"  step {done_steps}: mean reward/step {float(R.mean()):.4f}  episode-done rate {float(Dn.mean()):.4f}")
    torch.cuda.synchronize()
    return pi, vf


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
