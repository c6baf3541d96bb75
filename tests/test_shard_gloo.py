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
        # 3) the same exchange through a GradBucket: gradients written into views of ONE flat buffer by the backward,
        # summed by one collective (asynchronously), no cat / copy-back -- bit-equal to the list form above
        m3 = means[0].clone().requires_grad_(True)
        c3 = colors[0].clone().requires_grad_(True)
        z = torch.zeros(G, 4)
        bucket = shard.GradBucket(m3, c3, z, z[:, 0], z[:, :0])

        class _Into(torch.autograd.Function):          # stands in for the rasterizer's backward (which asks the bucket)
            @staticmethod
            def forward(ctx, a, b):
                ctx.save_for_backward(a, b)
                return a.clone(), b.clone()

            @staticmethod
            def backward(ctx, ga, gb):
                bk = shard.active_bucket()
                oa, ob = bk.take("means", ga), bk.take("scales", gb)
                oa.copy_(ga); ob.copy_(gb)
                return oa, ob

        a3, b3 = _Into.apply(m3, c3)
        loss3 = sum(_toy_render(a3, b3, cams[0, v]).pow(2).sum() for v in shard.view_shard(V, rank, world))
        with bucket:
            loss3.backward()
        aliased = (m3.grad.data_ptr() == bucket.views["means"].data_ptr()
                   and c3.grad.data_ptr() == bucket.views["scales"].data_ptr())
        work = bucket.all_reduce(async_op=True)
        if work is not None:
            work.wait()
        same = torch.equal(m3.grad, m.grad) and torch.equal(c3.grad, c.grad)
        q.put(("bucket", rank, bool(aliased), bool(same)))
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
    got = [q.get(timeout=5) for _ in range(5)]
    assert ("gather", True) in got
    assert ("allreduce", 0, True) in got and ("allreduce", 1, True) in got
    # the leaves' .grad ALIAS the bucket (no copy) and the one-collective result equals the list form bit for bit
    assert ("bucket", 0, True, True) in got and ("bucket", 1, True, True) in got


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


def test_grad_bucket_layout():
    """Five contiguous, 16-byte aligned views of one flat buffer, in the documented order; `take` hands out aliases of
    them -- each ONCE per `with` block, and only for the tensor the bucket was built for (a second backward
    under one block used to overwrite the first one's gradient silently, a mismatched shape used to fall back to a
    fresh buffer that the all-reduce never saw); nothing is active outside a `with` block."""
    S, G, K = 2, 5, 4
    t = [torch.zeros(S, G, 3), torch.zeros(S, G, 3), torch.zeros(S, G, 4), torch.zeros(S, G), torch.zeros(S, G, 3, K)]
    b = shard.GradBucket(*t)
    assert list(b.views) == ["means", "scales", "rotations", "opacities", "harmonics"]
    assert b.flat.numel() >= S * G * (11 + 3 * K)
    base = b.flat.data_ptr()
    for name, like in zip(b.NAMES, t):
        v = b.views[name]
        assert v.shape == like.shape and v.is_contiguous() and (v.data_ptr() - base) % 16 == 0
        a = b.take(name, like)
        assert a is not v and a.data_ptr() == v.data_ptr()
    import pytest
    with pytest.raises(RuntimeError, match="already written"):
        b.take("means", t[0])                                   # a second backward under the same block
    assert b.take("colors", torch.zeros(S, G, 3)) is None       # not a bucket tensor: the caller allocates
    assert shard.active_bucket() is None
    with b:
        assert shard.active_bucket() is b and not b.taken       # entering the block starts a new backward
        with pytest.raises(RuntimeError, match="built for"):
            b.take("harmonics", torch.zeros(S, G, K, 3))        # the other SH layout: loud, not a silent fresh buffer
        with pytest.raises(RuntimeError, match="no backward wrote"):
            b.all_reduce()
        b.flat.fill_(7.0)
        assert b.take("means", t[0]) is not None
        b.all_reduce()                                          # (no process group here: returns after the zero fill)
        assert float(b.views["means"].min()) == 7.0 and float(b.views["scales"].abs().max()) == 0.0
        with pytest.raises(RuntimeError, match="already active"):
            b.__enter__()
    assert shard.active_bucket() is None
