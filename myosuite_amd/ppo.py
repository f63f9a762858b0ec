"""PPO on the batched HIP envs with the whole training iteration resident on the device.

The learner of the reference's accelerated path is brax PPO driven by ``benchmarks/mjx_benchmark_PPO.py:50-60`` with the
hyper-parameters of ``myosuite/envs/myo/mjx/__init__.py:43-67``.  Restated in torch, an eager loop issues ~30 small launches per
env-step and ~100 per minibatch update, and the learner -- not the physics -- sets the pace (round 3: 0.45 M train env-steps/s on
a 6.4 M env-steps/s hand).  Here one training iteration is TWO HIP graphs:

* ``rollout``: for every step of the unroll -- observation normalisation, policy forward, sampling, the fused env-step launch
  (``mm_rollout_step``: physics + obs / reward + episode statistics + masked auto-reset), value forward, buffer writes -- then
  the bootstrap value, GAE as one kernel (``mm_gae``), advantage normalisation and the running-statistics update;
* ``update``: one pass over the batch -- a device-side permutation, then for every minibatch gather, forward, clipped-surrogate /
  value / entropy losses, backward, global-norm clipping and a capturable Adam step.

What follows brax's PPO term by term: GAE (mm_gae = compute_gae with its truncation masks), the clipped surrogate, the value loss
0.5 * 0.5 * mse, global-norm clipping, Adam, and -- since the end of round 5 -- the entropy bonus of NormalTanhDistribution: the
pre-squash normal's entropy PLUS the squashing log-det-Jacobian at a reparametrised sample x = mean + std e, which also feeds
gradient into the mean (`PPOConfig.entropy_squash_term`; e is drawn once per iteration per sample, brax draws per loss call; both
learners, torch and fused, implement it and check each other), and -- round 6 -- the advantage normalisation: per MINIBATCH and per
device with the population standard deviation, as `compute_ppo_loss` does on the data it is handed (`PPOConfig.normalize_advantage =
"minibatch"`; "batch" = rounds 4-5: once per iteration over the rank's whole batch).

Data parallel (one process per GPU): the running observation statistics are merged over ALL ranks' rows (one small all-reduce per
iteration: `_Norm.update(world=...)`), so the normaliser -- part of the policy and value function -- is identical on every rank; parameters and gradients live in ONE flat buffer each, so the exchange is a single
all-reduce of the flat gradient per minibatch (RCCL over xGMI; < 100 KB) with no flatten / copy-back; the update then runs eagerly
between the collectives (a gloo group cannot be captured).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine as E


@dataclasses.dataclass
class PPOConfig:
    unroll_length: int = 10
    num_minibatches: int = 32
    num_updates_per_batch: int = 8
    learning_rate: float = 3e-4
    discounting: float = 0.97
    gae_lambda: float = 0.95
    entropy_cost: float = 1e-3
    clipping_epsilon: float = 0.3
    max_grad_norm: Optional[float] = 1.0
    reward_scaling: float = 1.0
    value_cost: float = 0.25                      # brax: 0.5 * 0.5 * mse
    normalize_observations: bool = True
    policy_hidden: Tuple[int, ...] = (64, 64, 64)
    value_hidden: Tuple[int, ...] = (64, 64, 64)
    squash: str = "tanh"                          # "tanh": brax NormalTanhDistribution, actions in [-1, 1]; "sigmoid": excitations in [0, 1]
    unrolls: int = 1                              # unrolls of `unroll_length` per iteration (brax: batch_size * num_minibatches // num_envs)
    entropy_squash_term: bool = True              # brax NormalTanhDistribution.entropy: + log|d squash / d x| at a reparametrised sample
    normalize_advantage: str = "minibatch"        # "minibatch": brax (inside the loss, per minibatch and device, population std) | "batch" | "none"


def _mlp(sizes):
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers += [nn.Linear(a, b), nn.SiLU()]          # brax networks: swish
    return nn.Sequential(*layers[:-1])


class _Norm:
    """brax running_statistics over everything seen so far; all state in device tensors (graph safe)."""

    def __init__(self, dim, device):
        self.n = torch.zeros((), device=device); self.mean = torch.zeros(dim, device=device); self.m2 = torch.zeros(dim, device=device)
        self.std = torch.ones(dim, device=device)

    def update(self, x, world: int = 1):
        """Chan / Welford merge of one batch into the running statistics.  world > 1 (data-parallel ranks): the batch is the UNION of
        every rank's rows -- count, sum and sum of squares about the (rank-identical) current mean go through ONE all-reduce, so the
        normaliser, which is part of the policy and the value function, stays bit-identical on every rank (brax pmean-reduces
        running_statistics the same way).  The world > 1 form runs outside the captured graph (a gloo group cannot be captured)."""
        x = x.reshape(-1, x.shape[-1])
        if world > 1:
            c = x - self.mean
            pack = torch.cat([c.sum(0), (c * c).sum(0), torch.full((1,), float(x.shape[0]), device=x.device)]).double()
            if pack.is_cuda and torch.distributed.get_backend() == "gloo":
                h = pack.cpu(); torch.distributed.all_reduce(h); pack = h.to(x.device)
            else:
                torch.distributed.all_reduce(pack)
            dim = x.shape[1]
            b = pack[2 * dim]
            d = pack[:dim] / b                                    # batch mean - running mean
            bm2 = pack[dim:2 * dim] - b * d * d                   # sum of squares about the batch mean
            tot = self.n.double() + b
            self.m2.add_((bm2 + d * d * self.n.double() * b / tot).float())
            self.mean.add_((d * b / tot).float())
            self.n.copy_(tot.float())
        else:
            b = float(x.shape[0])
            tot = self.n + b
            bm = x.mean(0)
            d = bm - self.mean
            self.m2.add_(((x - bm) ** 2).sum(0) + d * d * self.n * b / tot)
            self.mean.add_(d * b / tot)
            self.n.copy_(tot)
        self.std.copy_(torch.sqrt(self.m2 / torch.clamp(self.n, min=1.0)).clamp(1e-6, 1e6))

    def __call__(self, x):
        return ((x - self.mean) / self.std).clamp(-5.0, 5.0)


class OnDevicePPO:
    """PPO learner + rollout over an env of ``myosuite_amd.envs`` (anything with ``rollout_setup / rollout_step / obs / rwd /
    truncated``), captured into HIP graphs.  ``world`` > 1: data-parallel ranks with one flat-gradient all-reduce per minibatch."""

    def __init__(self, env, cfg: PPOConfig, seed: int = 0, world: int = 1, use_graphs: bool = True, fused: Optional[bool] = None):
        """fused: None = the fused HIP learner kernels (include/myosim_ppo.h: one launch per rollout step for policy + sampling +
        value, three per minibatch update) whenever the networks fit them (hidden widths <= 128), else the torch-autograd form;
        True = require them; False = torch autograd (the checker of the fused kernels, and the only form for wider networks)."""
        self.env, self.cfg, self.world = env, cfg, world
        dev = env.device
        self.dev = dev
        n, T = env.num_envs, cfg.unroll_length * cfg.unrolls
        self.n, self.T = n, T
        od, ad = env.obs_dim, env.cm.nu
        torch.manual_seed(int(seed))
        self.pi = _mlp((od,) + tuple(cfg.policy_hidden) + (2 * ad,)).to(dev)         # mean and raw scale
        self.vf = _mlp((od,) + tuple(cfg.value_hidden) + (1,)).to(dev)
        self.params = list(self.pi.parameters()) + list(self.vf.parameters())
        # one flat buffer for the parameters and one for the gradients: the data-parallel exchange is ONE all-reduce, no copies
        sizes = [p.numel() for p in self.params]
        self.flat_p = torch.cat([p.detach().reshape(-1) for p in self.params]).contiguous()
        self.flat_g = torch.zeros_like(self.flat_p)
        o = 0
        for p, k in zip(self.params, sizes):
            p.data = self.flat_p[o:o + k].view_as(p)
            p.grad = self.flat_g[o:o + k].view_as(p)
            o += k
        if world > 1:
            torch.distributed.broadcast(self.flat_p, src=0)
            # parameters are rank 0's; everything drawn from here on (action noise, minibatch permutations) must DIFFER per rank, or the
            # data-parallel ranks explore with identical noise (brax folds the process index into its keys)
            torch.manual_seed(int(seed) + 1000003 * (1 + torch.distributed.get_rank()))
        self.kern = None
        if fused is not False and torch.cuda.is_available():
            try:
                self.kern = E.FusedPPO(od, ad, cfg.policy_hidden, cfg.value_hidden, cfg.squash, max_minibatch=max(1, (T * n) // cfg.num_minibatches),
                                       learning_rate=cfg.learning_rate, clipping_epsilon=cfg.clipping_epsilon, entropy_cost=cfg.entropy_cost,
                                       value_cost=cfg.value_cost, max_grad_norm=cfg.max_grad_norm, device=dev)
                assert self.kern.param_count == self.flat_p.numel()
            except E.EngineError:
                if fused:
                    raise
                self.kern = None
        elif fused:
            raise E.EngineError("fused PPO kernels need a HIP device")
        self.opt = None if self.kern else torch.optim.Adam(self.params, lr=cfg.learning_rate, capturable=True, foreach=True)
        self.noise = torch.zeros(T, n, ad, dtype=torch.float32, device=dev) if self.kern else None
        # draws of the entropy term's reparametrised sample (refreshed once per iteration inside the rollout graph)
        self.ent_noise = torch.zeros(T, n, ad, dtype=torch.float32, device=dev) if cfg.entropy_squash_term else None
        if self.kern and self.ent_noise is not None:
            self.kern.set_entropy_noise(self.ent_noise.view(T * n, ad))
        self.norm = _Norm(od, dev) if cfg.normalize_observations else None
        f = dict(dtype=torch.float32, device=dev)
        self.obs_b = torch.zeros(T, n, od, **f); self.act_b = torch.zeros(T, n, ad, **f); self.logp_b = torch.zeros(T, n, **f)
        self.rew_b = torch.zeros(T, n, **f); self.term_b = torch.zeros(T, n, **f); self.trunc_b = torch.zeros(T, n, **f)
        self.val_b = torch.zeros(T + 1, n, **f); self.adv_b = torch.zeros(T, n, **f); self.ret_b = torch.zeros(T, n, **f)
        self.nadv_b = torch.zeros(T, n, **f)
        self.action = torch.zeros(n, ad, **f)
        self.mean_reward = torch.zeros((), **f)
        env.reset(int(seed))
        self.ep_stats = env.rollout_setup()
        self.dense_col = env.rwd.shape[1] - 1
        self._g_roll = self._g_upd = None
        self._side = torch.cuda.Stream(device=dev) if torch.cuda.is_available() else None
        self.use_graphs = use_graphs and torch.cuda.is_available()
        self._captured = False

    # ------------------------------------------------------------------ policy
    def _dist(self, obs):
        ad = self.action.shape[1]
        out = self.pi(self.norm(obs) if self.norm else obs)
        return out[..., :ad], F.softplus(out[..., ad:]) + 1e-3

    def _logp(self, mean, std, raw):
        lp = -0.5 * ((raw - mean) / std) ** 2 - torch.log(std) - 0.5 * math.log(2 * math.pi)
        if self.cfg.squash == "tanh":          # log |d tanh / d raw|
            lp = lp - 2.0 * (math.log(2.0) - raw - F.softplus(-2.0 * raw))
        else:                                  # sigmoid squashing: log sigma(raw) + log(1 - sigma(raw))
            lp = lp - (-F.softplus(-raw) - F.softplus(raw))
        return lp.sum(-1)

    def _value(self, obs):
        return self.vf(self.norm(obs) if self.norm else obs).squeeze(-1)

    # ------------------------------------------------------------------ one iteration, eager form (also what gets captured)
    def _rollout(self):
        env, cfg = self.env, self.cfg
        with torch.no_grad():
            if self.kern:                          # fused: ONE launch per step next to the env-step launch (+ one for the buffer rows)
                K = self.kern
                mean, std = (self.norm.mean, self.norm.std) if self.norm else (None, None)
                self.noise.normal_()
                for t in range(self.T):
                    K.act(self.flat_p, env.obs, mean, std, self.noise[t], self.obs_b[t], self.act_b[t], self.logp_b[t], self.val_b[t], self.action)
                    _, rw, ended = env.rollout_step(self.action)
                    K.store(rw, self.dense_col, cfg.reward_scaling, ended, env.truncated, self.rew_b[t], self.trunc_b[t], self.term_b[t])
                K.act(self.flat_p, env.obs, mean, std, None, None, None, None, self.val_b[self.T], None)
            for t in range(0 if self.kern else self.T):
                obs = env.obs
                self.obs_b[t].copy_(obs)
                mean, std = self._dist(obs)
                raw = mean + std * torch.randn_like(mean)
                self.act_b[t].copy_(raw)
                self.logp_b[t].copy_(self._logp(mean, std, raw))
                self.val_b[t].copy_(self._value(obs))
                self.action.copy_(torch.tanh(raw) if cfg.squash == "tanh" else torch.sigmoid(raw))
                _, rw, ended = env.rollout_step(self.action)
                self.rew_b[t].copy_(rw[:, self.dense_col] * cfg.reward_scaling)
                tr = env.truncated.to(torch.float32)
                en = ended.to(torch.float32)
                self.trunc_b[t].copy_(tr * en)
                self.term_b[t].copy_(en * (1.0 - tr))             # ended without a time-limit: a true termination
            if not self.kern:
                self.val_b[self.T].copy_(self._value(env.obs))
            E.gae(self.rew_b, self.term_b, self.trunc_b, self.val_b, self.adv_b, self.ret_b, cfg.discounting, cfg.gae_lambda)
            # (whole-batch form; with normalize_advantage = "minibatch" _epoch() overwrites each minibatch's entries before its gradient)
            if cfg.normalize_advantage == "none":
                self.nadv_b.copy_(self.adv_b)
            else:
                self.nadv_b.copy_((self.adv_b - self.adv_b.mean()) / (self.adv_b.std(unbiased=False) + 1e-8))
            if self.norm and self.world == 1:
                self.norm.update(self.obs_b)       # (world > 1: iterate() merges every rank's batch, outside the graph)
            self.mean_reward.copy_(self.rew_b.mean())
            if self.ent_noise is not None:
                self.ent_noise.normal_()

    def _minibatch_backward(self, idx):
        """gradients of one minibatch into the flat gradient buffer.  The policy and the value network share nothing but the
        gathered rows, so their forward / backward chains run on two streams (two parallel branches of the captured graph): each
        chain is ~50 small kernels that leave most of the chip idle."""
        cfg = self.cfg
        B = self.T * self.n
        fo = self.obs_b.reshape(B, -1).index_select(0, idx)
        if self.norm:
            fo = self.norm(fo)
        raw = self.act_b.reshape(B, -1).index_select(0, idx)
        old = self.logp_b.reshape(B).index_select(0, idx)
        adv = self.nadv_b.reshape(B).index_select(0, idx)
        ret = self.ret_b.reshape(B).index_select(0, idx)
        ad = self.action.shape[1]
        self.flat_g.zero_()
        cur = torch.cuda.current_stream(self.dev)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):                 # value branch
            vl = cfg.value_cost * ((self.vf(fo).squeeze(-1) - ret) ** 2).mean()
            vl.backward()
        out = self.pi(fo)                                   # policy branch
        mean, std = out[..., :ad], F.softplus(out[..., ad:]) + 1e-3
        ratio = (self._logp(mean, std, raw) - old).exp()
        eps = cfg.clipping_epsilon
        pg = -torch.min(ratio * adv, ratio.clamp(1 - eps, 1 + eps) * adv).mean()
        ent = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).sum(-1)            # entropy of the pre-squash normal ...
        if self.ent_noise is not None:      # ... + log|d squash / d x| at x = mean + std e (brax NormalTanhDistribution.entropy)
            x = mean + std * self.ent_noise.reshape(B, -1).index_select(0, idx)
            ldj = 2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x)) if cfg.squash == "tanh" else (-F.softplus(-x) - F.softplus(x))
            ent = ent + ldj.sum(-1)
        ent = ent.mean()
        (pg - cfg.entropy_cost * ent).backward()
        cur.wait_stream(self._side)

    def _minibatch_fused(self, idx):
        """the same gradient from the fused kernels: two launches (forward + losses + backward into partials; ordered reduction)"""
        B = self.T * self.n
        mean, std = (self.norm.mean, self.norm.std) if self.norm else (None, None)
        self.kern.grad(self.flat_p, self.obs_b.view(B, -1), mean, std, idx, self.act_b.view(B, -1), self.logp_b.view(B), self.nadv_b.view(B),
                       self.ret_b.view(B), self.flat_g)

    def _step_opt(self):
        if self.cfg.max_grad_norm:
            torch.nn.utils.clip_grad_norm_(self.params, self.cfg.max_grad_norm, foreach=True)
        self.opt.step()

    def _epoch(self):
        B = self.T * self.n
        perm = torch.argsort(torch.rand(B, device=self.dev))          # device-side permutation (graph safe)
        mb = B // self.cfg.num_minibatches
        if self.cfg.normalize_advantage == "minibatch":
            # brax compute_ppo_loss: (advantages - mean) / (std + 1e-8) over the minibatch it is handed (jnp.std: population) -- every
            # minibatch of this pass at once (a handful of launches per pass over the batch, not per minibatch)
            nmb = self.cfg.num_minibatches
            sel = perm[:nmb * mb]
            a_ = self.adv_b.view(B).index_select(0, sel).view(nmb, mb)
            a_ = (a_ - a_.mean(1, keepdim=True)) / (a_.std(1, unbiased=False, keepdim=True) + 1e-8)
            self.nadv_b.view(B).index_copy_(0, sel, a_.reshape(-1))
        for k in range(self.cfg.num_minibatches):
            if self.kern:
                self._minibatch_fused(perm[k * mb:(k + 1) * mb])
            else:
                self._minibatch_backward(perm[k * mb:(k + 1) * mb])
            if self.world > 1:                 # ONE collective per minibatch on the flat gradient buffer
                if self.flat_g.is_cuda and torch.distributed.get_backend() == "gloo":
                    h = self.flat_g.cpu(); torch.distributed.all_reduce(h); self.flat_g.copy_(h)
                else:
                    torch.distributed.all_reduce(self.flat_g)
                if self.kern:
                    self.kern.adam(self.flat_p, self.flat_g, 1.0 / self.world, recompute_norm=True)
                    continue
                self.flat_g.div_(self.world)
            if self.kern:
                self.kern.adam(self.flat_p, self.flat_g)
            else:
                self._step_opt()

    # ------------------------------------------------------------------ graphs
    def _capture(self):
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):                      # warm-up outside the capture: kernel attributes, autograd buffers, Adam state
            for _ in range(3):
                self._rollout()
                self._epoch()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self._g_roll = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_roll):
            self._rollout()
        if self.world == 1:
            self._g_upd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_upd):
                self._epoch()
        self._captured = True

    def iterate(self):
        """one training iteration: T * unrolls env-steps on every env, then num_updates_per_batch passes over the batch"""
        if self.use_graphs and not self._captured:
            self._capture()
        if self._g_roll is not None:
            self._g_roll.replay()
        else:
            self._rollout()
        if self.norm and self.world > 1:
            with torch.no_grad():
                self.norm.update(self.obs_b, world=self.world)
        for _ in range(self.cfg.num_updates_per_batch):
            if self._g_upd is not None:
                self._g_upd.replay()
            else:
                self._epoch()

    @property
    def steps_per_iteration(self) -> int:
        return self.T * self.n * self.world
