"""The one collective of the rollout path on the real backend: a one-rank RCCL ("nccl" on ROCm) process group on the GPU box
(8-GPU runs are the driver's; this checks that the RCCL call path works with device tensors).  The N > 1 logic is covered on CPU
by tests/test_dist_gloo.py (world size 2, gloo)."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def test_gather_episode_stats_over_rccl_world_1():
    import torch
    import torch.distributed as dist
    from myosuite_amd import dist as D
    from myosuite_amd.envs import registry
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        env = registry.make("myoElbowPose1D6MRandom-v0", num_envs=256, seed=0)
        stats = env.rollout_setup(action_seed=0)
        for s_ in range(5):
            env.rollout_step(None, stream_id=s_)
        out = D.gather_episode_stats(stats, always_collective=True)       # all_gather_into_tensor on device tensors
        torch.cuda.synchronize()
        assert out.shape == (256, 3) and torch.equal(out, stats) and float(out[:, 1].min()) == 5.0
        t = D.max_over_ranks(1.25, device="cuda")
        assert t == 1.25
        D.barrier()
    finally:
        dist.destroy_process_group()
