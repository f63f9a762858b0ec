"""Fused PPO learner kernels (include/myosim_ppo.h, myosuite_amd/csrc/myosim_ppo.hip) against the torch-autograd restatement of
the same learner (myosuite_amd/ppo.py with fused=False) -- the reference's learner is brax PPO driven by
benchmarks/mjx_benchmark_PPO.py:50-60 with the hyper-parameters of myosuite/envs/myo/mjx/__init__.py:43-67.  fp32 on both sides:
the bounds are summation-order bounds."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fused_ppo_header_symbols_are_exported_and_the_config_struct_matches():
    from myosuite_amd import engine as E
    lib = ctypes.CDLL(E.build())
    hdr = open(os.path.join(ROOT, "include", "myosim_ppo.h")).read()
    names = sorted(set(re.findall(r"\b(mm_ppo_[a-z_]+)\s*\(", hdr)))
    assert len(names) >= 9, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/myosim_ppo.h but not exported"
    body = hdr[:hdr.index("} mm_ppo_config;")]
    body = re.sub(r"/\*.*?\*/", "", body[body.rindex("typedef struct {") + len("typedef struct {"):], flags=re.S)
    fields = []
    for stmt in body.split(";"):
        for decl in stmt.strip().split(","):
            if decl.strip():
                fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", re.sub(r"\[.*?\]", "", decl))[-1])
    assert fields == [f[0] for f in E.mm_ppo_config._fields_], fields
    for k in ("MM_PPO_MAX_LAYERS", "MM_PPO_MAX_WIDTH", "MM_PPO_MAX_OBS", "MM_PPO_MAX_OUT"):
        assert int(re.search(rf"#define {k}\s+(\d+)", hdr).group(1)) == getattr(E, k)


CONFIGS = [  # env id, envs, policy hidden, value hidden, squash, normalise, minibatches, samples per workgroup (None: the launcher's choice)
    ("myoHandPoseRandom-v0", 96, (64, 64, 64), (64, 64, 64), "tanh", True, 4, None),            # the reference's ppo_config networks
    ("myoHandPoseRandom-v0", 96, (64, 64, 64), (64, 64, 64), "tanh", True, 1, "32"),            # 960 rows: 30 workgroups of 32
    ("myoElbowPose1D6MRandom-v0", 50, (32, 48), (16,), "sigmoid", False, 3, "16"),              # odd widths, ragged last workgroup, no normalisation
    ("myoElbowPose1D6MRandom-v0", 50, (32, 32, 32, 32), (128, 128), "sigmoid", True, 2, "32"),  # four hidden layers; the widest hidden the kernels take
    ("myoFatiLegWalk-v0", 40, (64, 64, 64), (64, 64, 64), "tanh", True, 2, None),               # obs 403 (not a multiple of 4: scalar weight loads), 160 outputs
    ("myoElbowPose1D6MRandom-v0", 50, (32, 48), (16,), "tanh", True, 2, "16", False),           # entropy of the pre-squash normal only (no squash log-det term)
]


def _make(env_id, n, ph, vh, squash, norm, nmb, samples, ent_squash=True, fused=None):
    from myosuite_amd.envs import registry
    from myosuite_amd.ppo import OnDevicePPO, PPOConfig
    if samples:
        os.environ["MYOSIM_PPO_SAMPLES"] = samples
    else:
        os.environ.pop("MYOSIM_PPO_SAMPLES", None)
    try:
        env = registry.make(env_id, num_envs=n, seed=5)
        cfg = PPOConfig(unroll_length=10, num_minibatches=nmb, num_updates_per_batch=2, policy_hidden=ph, value_hidden=vh, squash=squash,
                        normalize_observations=norm, entropy_cost=1e-2, clipping_epsilon=0.2, max_grad_norm=0.5, entropy_squash_term=ent_squash)
        return OnDevicePPO(env, cfg, seed=3, use_graphs=False, fused=fused)
    finally:
        os.environ.pop("MYOSIM_PPO_SAMPLES", None)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CONFIGS, ids=[f"{c[0]}-{'x'.join(map(str, c[2]))}-{'x'.join(map(str, c[3]))}-S{c[7]}" for c in CONFIGS])
def test_fused_minibatch_gradient_matches_torch_autograd(cfg):
    """mm_ppo_grad (gather + normalise + policy / value forward + clipped surrogate / entropy / value losses + backward, MFMA tiles
    over LDS-resident activations) against loss.backward() of the torch restatement on the same minibatch of a real unroll, with the
    parameters perturbed after the unroll so that ratios leave the clipping range on both sides.  The entropy is brax's
    NormalTanhDistribution.entropy (pre-squash normal + squash log-det-Jacobian at a reparametrised sample, both squashings) in all
    configurations but the last, which keeps the pre-squash form."""
    ppo = _make(*cfg)
    assert ppo.kern is not None and (ppo.ent_noise is not None) == (len(cfg) < 9 or cfg[8])

    ppo._rollout()                                                     # fills the unroll buffers through mm_ppo_act / mm_ppo_store
    torch.manual_seed(1)
    ppo.flat_p.add_(0.03 * torch.randn_like(ppo.flat_p))               # "after a few updates": ratios spread around 1
    B = ppo.T * ppo.n
    perm = torch.argsort(torch.rand(B, device=ppo.dev))
    mb = B // ppo.cfg.num_minibatches
    for k in range(ppo.cfg.num_minibatches):
        idx = perm[k * mb:(k + 1) * mb]
        ppo._minibatch_backward(idx)
        torch.cuda.synchronize()
        ref = ppo.flat_g.clone()
        ppo.flat_g.fill_(7.0)                                          # the kernels overwrite every entry
        ppo._minibatch_fused(idx)
        torch.cuda.synchronize()
        got = ppo.flat_g.clone()
        vo = ppo.kern.value_offset
        for name, a, b in (("policy", got[:vo], ref[:vo]), ("value", got[vo:], ref[vo:])):
            scale = float(b.abs().max())
            assert scale > 0 and float((a - b).abs().max()) < 2e-4 * scale, (name, k, float((a - b).abs().max()), scale)
    # both clipping branches occurred in the last minibatch
    with torch.no_grad():
        fo = ppo.obs_b.reshape(B, -1)[idx]
        fo = ppo.norm(fo) if ppo.norm else fo
        out = ppo.pi(fo)
        ad = ppo.action.shape[1]
        mean, std = out[:, :ad], torch.nn.functional.softplus(out[:, ad:]) + 1e-3
        ratio = (ppo._logp(mean, std, ppo.act_b.reshape(B, -1)[idx]) - ppo.logp_b.reshape(B)[idx]).exp()
    eps = ppo.cfg.clipping_epsilon
    assert bool((ratio > 1 + eps).any()) and bool((ratio < 1 - eps).any()) and bool(((ratio > 1 - eps) & (ratio < 1 + eps)).any())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [CONFIGS[0], CONFIGS[2], CONFIGS[4]], ids=["hand", "elbow", "leg"])
def test_fused_act_matches_the_torch_policy_and_value(cfg):
    """mm_ppo_act: raw action = mean + std * noise, log-density of the squashed action, squashed action, value, observation copy"""
    ppo = _make(*cfg)
    env, K = ppo.env, ppo.kern
    for s in range(3):
        env.rollout_step(None, stream_id=s)
    if ppo.norm:
        ppo.norm.update(env.obs.clone()[None])
    n, ad = ppo.n, ppo.action.shape[1]
    noise = torch.randn(n, ad, device=ppo.dev)
    obs_o, raw_o, lp_o, v_o, act_o = (torch.zeros(n, env.obs_dim, device=ppo.dev), torch.zeros(n, ad, device=ppo.dev), torch.zeros(n, device=ppo.dev),
                                     torch.zeros(n, device=ppo.dev), torch.zeros(n, ad, device=ppo.dev))
    mean_, std_ = (ppo.norm.mean, ppo.norm.std) if ppo.norm else (None, None)
    K.act(ppo.flat_p, env.obs, mean_, std_, noise, obs_o, raw_o, lp_o, v_o, act_o)
    v_only = torch.zeros(n, device=ppo.dev)
    K.act(ppo.flat_p, env.obs, mean_, std_, None, None, None, None, v_only, None)
    torch.cuda.synchronize()
    with torch.no_grad():
        mean, std = ppo._dist(env.obs)
        raw = mean + std * noise
        lp = ppo._logp(mean, std, raw)
        val = ppo._value(env.obs)
        act = torch.tanh(raw) if ppo.cfg.squash == "tanh" else torch.sigmoid(raw)
    assert torch.equal(obs_o, env.obs)
    assert float((raw_o - raw).abs().max()) < 2e-5 * max(1.0, float(raw.abs().max()))
    assert float((act_o - act).abs().max()) < 2e-5
    assert float((lp_o - lp).abs().max()) < 2e-4 * max(1.0, float(lp.abs().max()))
    assert float((v_o - val).abs().max()) < 2e-5 * max(1.0, float(val.abs().max())) and torch.equal(v_o, v_only)


@pytest.mark.gpu
def test_fused_adam_matches_torch_clip_and_adam():
    """mm_ppo_adam: clip_grad_norm_ + Adam(capturable) over five steps, with and without the clip active, and the data-parallel form
    (gradient scaled by 1 / world, norm recomputed)"""
    from myosuite_amd import engine as E
    torch.manual_seed(0)
    for max_norm, gscale in ((0.5, 1.0), (None, 1.0), (1e3, 0.5)):
        K = E.FusedPPO(9, 6, (32, 48), (16,), "tanh", max_minibatch=64, learning_rate=3e-3, clipping_epsilon=0.3, entropy_cost=1e-2, value_cost=0.25,
                       max_grad_norm=max_norm)
        P = K.param_count
        p = torch.randn(P, device="cuda")
        ref = p.clone().requires_grad_(True)
        opt = torch.optim.Adam([ref], lr=3e-3)
        for step in range(5):
            g = torch.randn(P, device="cuda") * (0.1 if step % 2 else 1.0)
            K.adam(p, g, gscale, recompute_norm=True)          # the gradient comes from outside mm_ppo_grad: norm recomputed
            ref.grad = (g * gscale).clone()
            if max_norm:
                torch.nn.utils.clip_grad_norm_([ref], max_norm)
            opt.step()
        torch.cuda.synchronize()
        d = float((p - ref.detach()).abs().max())
        assert d < 1e-5, (max_norm, gscale, d)           # five steps of 3e-3
        K.reset_optimizer()
        q = p.clone()
        K.adam(q, g, gscale, recompute_norm=True)        # first step after a reset: |update| = lr (bias correction of step 1)
        torch.cuda.synchronize()
        assert float(((q - p).abs() - 3e-3).abs().max()) < 1e-4
        del K
    # with the full sequence (grad -> adam), against the torch learner
    cfg = CONFIGS[2]
    a = _make(*cfg, fused=True)
    b = _make(*cfg, fused=False)
    b.flat_p.copy_(a.flat_p)
    a._rollout()
    for name in ("obs_b", "act_b", "logp_b", "nadv_b", "ret_b", "ent_noise"):
        getattr(b, name).copy_(getattr(a, name))
    B = a.T * a.n
    torch.manual_seed(2)
    for it in range(6):
        idx = torch.argsort(torch.rand(B, device=a.dev))[:B // 3]
        a._minibatch_fused(idx); a.kern.adam(a.flat_p, a.flat_g)
        b._minibatch_backward(idx); b._step_opt()
    torch.cuda.synchronize()
    d = float((a.flat_p - b.flat_p).abs().max())
    assert d < 2e-5, d                       # six steps of lr 3e-4: parameters move ~2e-3; agreement to summation order


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "torch"])
def test_advantages_are_normalised_per_minibatch_like_brax(fused):
    """brax's compute_ppo_loss normalises the advantages of the minibatch it is handed -- (A - mean) / (std + 1e-8), jnp.std = the
    population form -- not the batch's (benchmarks/mjx_benchmark_PPO.py:50-60 trains with brax's defaults: normalize_advantage on).
    After one pass over the batch every minibatch's slice of the normalised buffer has mean 0 and population std 1 although the
    raw advantages of the minibatches differ; "batch" keeps the round-5 behaviour (the whole batch normalised once)."""
    cfg = ("myoElbowPose1D6MRandom-v0", 64, (32, 32), (32, 32), "tanh", True, 4, None)
    for mode in ("minibatch", "batch"):
        ppo = _make(*cfg, fused=fused)
        assert (ppo.kern is not None) == fused and ppo.cfg.normalize_advantage == "minibatch"
        ppo.cfg.normalize_advantage = mode
        ppo._rollout()
        B = ppo.T * ppo.n
        # make the minibatches differ in location and scale beyond sampling noise (the rollout's advantages are nearly homogeneous)
        ppo.adv_b.view(B).mul_(torch.linspace(0.5, 3.0, B, device=ppo.dev)).add_(torch.linspace(-1.0, 1.0, B, device=ppo.dev))
        torch.manual_seed(7); torch.cuda.manual_seed_all(7)
        ppo._epoch()
        torch.manual_seed(7); torch.cuda.manual_seed_all(7)
        perm = torch.argsort(torch.rand(B, device=ppo.dev))
        mb = B // ppo.cfg.num_minibatches
        means, stds = [], []
        for k in range(ppo.cfg.num_minibatches):
            v = ppo.nadv_b.view(B).index_select(0, perm[k * mb:(k + 1) * mb])
            means.append(float(v.mean())); stds.append(float(v.std(unbiased=False)))
        if mode == "minibatch":
            assert max(abs(m) for m in means) < 1e-5 and max(abs(s_ - 1.0) for s_ in stds) < 1e-4, (means, stds)
        else:
            assert max(abs(m) for m in means) > 1e-3, means          # normalised over the batch, not per minibatch


@pytest.mark.gpu
def test_fused_learner_rejects_networks_it_cannot_take():
    from myosuite_amd import engine as E
    with pytest.raises(E.EngineError, match="width 256"):
        E.FusedPPO(108, 39, (32, 32), (256, 256), "tanh", max_minibatch=64, learning_rate=3e-4, clipping_epsilon=0.3, entropy_cost=1e-2,
                   value_cost=0.25, max_grad_norm=1.0)
    ppo = _make("myoElbowPose1D6MRandom-v0", 32, (32, 32), (256, 256), "tanh", True, 2, None)          # auto: falls to the torch learner
    assert ppo.kern is None and ppo.opt is not None
    with pytest.raises(E.EngineError):
        _make("myoElbowPose1D6MRandom-v0", 32, (32, 32), (256, 256), "tanh", True, 2, None, fused=True)


@pytest.mark.gpu
def test_fused_learner_is_deterministic_and_graph_replay_equals_eager_launches():
    """No float atomics anywhere in the learner (partial gradients are added in a fixed order): two runs from the same seed end with
    bit-identical parameters, and so do the HIP-graph form and the eager form of the same iterations."""
    from myosuite_amd.envs import registry
    from myosuite_amd.ppo import OnDevicePPO, PPOConfig
    outs = []
    for graphs in (False, False, True):
        env = registry.make("myoElbowPose1D6MRandom-v0", num_envs=256, seed=1)
        cfg = PPOConfig(unroll_length=5, num_minibatches=4, num_updates_per_batch=2, policy_hidden=(64, 64, 64), value_hidden=(64, 64, 64), squash="sigmoid")
        torch.manual_seed(11); torch.cuda.manual_seed_all(11)
        ppo = OnDevicePPO(env, cfg, seed=2, use_graphs=graphs)
        assert ppo.kern is not None
        if graphs:
            ppo._capture()               # three warm-up iterations + capture: replay from a known state below
        # same starting point for all three: parameters, optimiser, env, RNG
        ppo.flat_p.copy_(torch.linspace(-0.05, 0.05, ppo.flat_p.numel(), device=ppo.dev))
        ppo.kern.reset_optimizer()
        if ppo.norm:
            ppo.norm.n.zero_(); ppo.norm.mean.zero_(); ppo.norm.m2.zero_(); ppo.norm.std.fill_(1.0)
        env.reset(2); ppo.ep_stats = env.rollout_setup()
        torch.manual_seed(5); torch.cuda.manual_seed_all(5)
        for _ in range(3):
            ppo.iterate()
        torch.cuda.synchronize()
        outs.append(ppo.flat_p.clone())
    assert torch.equal(outs[0], outs[1])
    # graph replay draws its random numbers from the same generator through torch's graph-safe Philox offsets: same stream, same values
    assert torch.equal(outs[0], outs[2]) or float((outs[0] - outs[2]).abs().max()) < 1e-6
