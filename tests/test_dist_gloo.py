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
