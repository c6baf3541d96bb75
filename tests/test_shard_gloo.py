"""Multi-process (world_size 2, gloo, CPU) coverage of the sharding layer: scene-first partition with no data-path
collective, the flat-bucket gradient all-reduce for a view-split scene, and the rank-0 gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spfsplatv2_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_render(means, colors, cam):
    """A differentiable stand-in for one (scene, view) render on CPU: any pure function of (scene, camera)."""
    return (means @ cam[:3, :3] + cam[3, :3]).tanh().sum(-1, keepdim=True) * colors


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gen = torch.Generator().manual_seed(0)
        S, V, G = 5, 4, 7
        means = torch.randn(S, G, 3, generator=gen)
        colors = torch.randn(S, G, 3, generator=gen)
        cams = torch.randn(S, V, 4, 4, generator=gen)
        # 1) scene-first sharding: no collective in the data path
        mine = shard.scene_shard(S, rank, world)
        local = torch.stack([torch.stack([_toy_render(means[s], colors[s], cams[s, v]) for v in range(V)])
                             for s in mine])
        full = shard.gather_rendered(local, S)
        if rank == 0:
            ref = torch.stack([torch.stack([_toy_render(means[s], colors[s], cams[s, v]) for v in range(V)])
                               for s in range(S)])
            q.put(("gather", bool(torch.allclose(full, ref))))
        # 2) one scene, views split across ranks -> per-Gaussian grads need the all-reduce
        m = means[0].clone().requires_grad_(True)
        c = colors[0].clone().requires_grad_(True)
        loss = sum(_toy_render(m, c, cams[0, v]).pow(2).sum() for v in shard.view_shard(V, rank, world))
        loss.backward()
        shard.allreduce_gaussian_grads([m.grad, None, c.grad])
        m2 = means[0].clone().requires_grad_(True)
        c2 = colors[0].clone().requires_grad_(True)
        sum(_toy_render(m2, c2, cams[0, v]).pow(2).sum() for v in range(V)).backward()
        ok = torch.allclose(m.grad, m2.grad, atol=1e-5) and torch.allclose(c.grad, c2.grad, atol=1e-5)
        q.put(("allreduce", rank, bool(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = [q.get(timeout=5) for _ in range(3)]
    assert ("gather", True) in got
    assert ("allreduce", 0, True) in got and ("allreduce", 1, True) in got


def test_partitions_cover_everything_once():
    for n, world in ((64, 8), (5, 2), (3, 8), (1, 1)):
        seen = sorted(i for r in range(world) for i in shard.scene_shard(n, r, world))
        assert seen == list(range(n))
    assert shard.scene_shard(64, 3, 8) == list(range(3, 64, 8))       # BASELINE config 4: 8 scenes per GPU
    with pytest.raises(ValueError):
        shard.scene_shard(4, 2, 2)


def test_single_process_is_a_no_op():
    g = torch.ones(3)
    shard.allreduce_gaussian_grads([g])
    assert torch.equal(g, torch.ones(3))
