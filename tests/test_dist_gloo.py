"""N>1 path on CPU: world_size-2 gloo processes exercise env sharding, the episode-stat all-gather
and the max-over-ranks timing reduction used by bench.py."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from myosuite_amd import dist as D
    r, w, _ = D.init_from_env(backend="gloo")
    start, count = D.shard_envs(10, r, w)
    stats = torch.stack([torch.arange(start, start + 5, dtype=torch.float32), torch.full((5,), float(r)),
                         torch.ones(5)], dim=1)
    allst = D.gather_episode_stats(stats)
    mx = D.max_over_ranks(1.0 + r)
    D.barrier()
    q.put((r, start, count, allst.tolist(), mx))


def test_shard_and_allgather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert [(r[1], r[2]) for r in res] == [(0, 5), (5, 5)]
    for r in res:
        assert len(r[3]) == 10 and [row[1] for row in r[3]] == [0.0] * 5 + [1.0] * 5
        assert [row[0] for row in r[3]] == [float(i) for i in range(10)]
        assert r[4] == 2.0


def test_shard_remainder():
    from myosuite_amd.dist import shard_envs
    parts = [shard_envs(8192 + 3, r, 8) for r in range(8)]
    assert sum(c for _, c in parts) == 8195 and parts[0] == (0, 1025) and parts[7][0] + parts[7][1] == 8195
    for (s0, c0), (s1, _) in zip(parts, parts[1:]):
        assert s0 + c0 == s1


def _norm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from myosuite_amd.ppo import _Norm
    g = torch.Generator().manual_seed(7)
    allx = [torch.randn(world, 40 + 13 * k, 5, generator=g) * (1.0 + k) + 3.0 * k for k in range(3)]      # the same draws on every rank
    nm = _Norm(5, "cpu")
    for x in allx:
        nm.update(x[rank], world)                     # each rank contributes ITS rows; the statistics are those of the union
    td.barrier()
    q.put((rank, nm.n.item(), nm.mean.tolist(), nm.std.tolist()))
    td.destroy_process_group()


def test_observation_normaliser_is_merged_over_data_parallel_ranks():
    """the running observation normaliser is part of the policy, so data-parallel ranks must hold the SAME statistics --
    those of the union of all ranks' rows (brax pmean-reduces running_statistics).  Two gloo ranks feed different rows through three
    updates: both end bit-identical, equal to one process that saw everything."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_norm_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(world))
    [p.join(60) for p in ps]
    assert res[0][1:] == res[1][1:]                                        # bit-identical on both ranks
    sys.path.insert(0, ROOT)
    from myosuite_amd.ppo import _Norm
    g = torch.Generator().manual_seed(7)
    allx = [torch.randn(world, 40 + 13 * k, 5, generator=g) * (1.0 + k) + 3.0 * k for k in range(3)]
    one = _Norm(5, "cpu")
    for x in allx:
        one.update(x.reshape(-1, 5))
    import numpy as np
    assert res[0][1] == one.n.item() == sum(2 * (40 + 13 * k) for k in range(3))
    np.testing.assert_allclose(res[0][2], one.mean.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(res[0][3], one.std.numpy(), rtol=2e-6)
    everything = torch.cat([x.reshape(-1, 5) for x in allx])
    np.testing.assert_allclose(res[0][2], everything.mean(0).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(res[0][3], everything.std(0, unbiased=False).numpy(), rtol=1e-5)
