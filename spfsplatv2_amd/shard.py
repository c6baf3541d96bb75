"""Multi-GPU sharding of the render path (one process per GPU, torch.distributed; backend "nccl" = RCCL).

A render is a pure function of one scene's Gaussians and one camera, and the reference already treats ``(b v)`` as a
flat list of independent calls (decoder_splatting_cuda.py:53-64, cuda_splatting.py:96), so:

* ``scene_shard``      scene-first partition: rank r renders scenes r, r+N, ...; all views of a scene stay on one GPU,
                       so the sum over views of every per-Gaussian gradient is local.  No data-path collective.
* ``allreduce_gaussian_grads``  the only exchange the path ever needs: when ONE scene's views are split across ranks
                       (BASELINE config 5, the outer training step), the per-Gaussian parameter gradients are summed
                       with a single flat-bucket all-reduce (RCCL over xGMI picks its own multi-link schedule; one
                       bucket of G*(11+3K)*4 bytes keeps it bandwidth- rather than latency-bound).
"""
from __future__ import annotations

import threading
from typing import Iterable, Optional, Sequence

import torch
import torch.distributed as dist

# The active GradBucket is PROCESS-wide, not thread-local: autograd runs the backward of device tensors on its own
# worker thread, which would never see a thread-local of the thread that called `loss.backward()`.  One process drives
# one GPU with one training loop (src/main.py:141-145), so there is one backward at a time; nesting is a stack.
_active: list = []
_active_lock = threading.Lock()


class GradBucket:
    """ONE flat float32 buffer that holds the gradients of a decoder call's Gaussian tensors -- means [S,G,3], scales
    [S,G,3], rotations [S,G,4], opacities [S,G], harmonics (any shape) -- as contiguous views, in this order.

    With ``with bucket:`` around the backward pass the rasterizer's backward writes its results straight into those
    views (rasterizer._backward_impl takes its output buffers from the active bucket instead of allocating them), the
    leaves' ``.grad`` alias the bucket, and ``bucket.all_reduce()`` sums the whole parameter set across the ranks with
    ONE collective on the flat buffer: no ``torch.cat`` into a fresh bucket, no copy back (what
    ``allreduce_gaussian_grads`` costs: two extra passes over G*(11+3K)*4 bytes).  ``async_op=True`` returns the
    work handle: issue it after one micro-batch's backward and wait after the next one's -- the collective then runs on
    RCCL's own stream under the next micro-batch's kernels (the reference gets the same overlap from DDP's bucketed
    hooks, src/main.py:141-145).
    """

    NAMES = ("means", "scales", "rotations", "opacities", "harmonics")

    def __init__(self, means: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor, opacities: torch.Tensor,
                 harmonics: torch.Tensor):
        shapes = [tuple(t.shape) for t in (means, scales, rotations, opacities, harmonics)]
        sizes = [int(torch.Size(sh).numel()) for sh in shapes]
        # every view starts at a multiple of 4 floats: the kernels' 16-byte stores stay aligned
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 3) // 4 * 4
        self.flat = torch.empty((total,), dtype=torch.float32, device=means.device)
        self.views = {name: self.flat[o:o + n].view(sh) for name, o, n, sh in zip(self.NAMES, offs, sizes, shapes)}
        self.taken: set = set()      # names handed out since the bucket was last entered

    def take(self, name: str, like: torch.Tensor) -> Optional[torch.Tensor]:
        """The bucket's view for gradient `name` of a backward that runs inside ``with bucket:``.

        One backward per ``with`` block: the views are the ONLY storage, so a second rasterizer backward under the same
        block (a per-view loop, a context render next to the target render, a ``retain_graph`` second backward) would
        overwrite what the first one's ``.grad`` still aliases and autograd would then add two aliases of one buffer --
        twice the last gradient, silently.  That raises here instead.  A tensor that does not match the view the bucket
        was built with (shape, device, dtype -- e.g. harmonics handed over as [S,G,K,3] where the bucket was built from
        the [S,G,3,K] leaf) raises too: falling back to a fresh buffer would leave that leaf out of the all-reduce and
        put uninitialised bucket memory into it.  Names the bucket does not know (`colors`, ...) return None: the caller
        allocates as usual."""
        v = self.views.get(name)
        if v is None:
            return None
        if tuple(v.shape) != tuple(like.shape) or v.device != like.device or like.dtype != torch.float32:
            raise RuntimeError(f"GradBucket: the gradient of `{name}` is {tuple(like.shape)} {like.dtype} on {like.device}, "
                               f"the bucket was built for {tuple(v.shape)} float32 on {v.device}")
        if name in self.taken:
            raise RuntimeError(f"GradBucket: `{name}` was already written by a backward inside this `with` block; a second "
                               "rasterizer backward would overwrite the gradient the first one's .grad aliases "
                               "(use one bucket per backward, or run the second backward outside the block)")
        self.taken.add(name)
        # (a fresh alias of the view: autograd keeps a gradient it is handed without copying only when nobody else holds
        #  that tensor object -- the storage is the bucket's either way)
        return v.view(v.shape)

    def __enter__(self):
        with _active_lock:
            if self in _active:
                raise RuntimeError("GradBucket: already active (nested `with` on the same bucket)")
            self.taken = set()
            _active.append(self)
        return self

    def __exit__(self, *exc):
        with _active_lock:
            _active.remove(self)
        return False

    def all_reduce(self, group=None, async_op: bool = False, skip_single: bool = True):
        """In-place SUM over the ranks of everything in the bucket; returns the work handle with ``async_op=True``
        (None when there is nothing to do).  Views no backward wrote since the bucket was last entered (a gradient that
        was switched off, e.g. scales / rotations with ``enable_cov_grad=False``) are zero-filled first, so that the
        collective never sums uninitialised memory; a bucket NO backward wrote into raises."""
        if not self.taken:
            raise RuntimeError("GradBucket.all_reduce: no backward wrote into this bucket since it was entered "
                               "(run loss.backward() inside `with bucket:` first)")
        for name, v in self.views.items():
            if name not in self.taken:
                v.zero_()
        if not dist.is_available() or not dist.is_initialized():
            return None
        if skip_single and dist.get_world_size(group) == 1:
            return None
        if self.flat.is_cuda and dist.get_backend(group) == "gloo":
            # testing configuration only (a CPU backend under device tensors): stage through the host, synchronously
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            self.flat.copy_(host)
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def active_bucket() -> Optional[GradBucket]:
    """The GradBucket of the innermost enclosing ``with`` block of this process (what the rasterizer's backward asks
    for, from autograd's worker thread)."""
    with _active_lock:
        return _active[-1] if _active else None


def scene_shard(n_scenes: int, rank: int, world: int) -> list[int]:
    """Scene indices rendered by `rank` (round-robin, scene-first)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_scenes, world))


def view_shard(n_views: int, rank: int, world: int) -> list[int]:
    """View indices of ONE scene rendered by `rank` when a single scene is split across ranks (config 5)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_views, world))


def allreduce_gaussian_grads(grads: Sequence[torch.Tensor | None], group=None, skip_single: bool = True) -> None:
    """In-place SUM all-reduce of a list of gradient tensors through one flat bucket.  `skip_single=False` runs the
    bucket through the backend even in a world of one (a pre-flight of the RCCL path on a single-GPU box)."""
    if not dist.is_available() or not dist.is_initialized():
        return
    if skip_single and dist.get_world_size(group) == 1:
        return
    live = [g for g in grads if g is not None]
    if not live:
        return
    flat = torch.cat([g.reshape(-1) for g in live])
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # testing configuration only (a CPU backend under device tensors): stage the bucket through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in live:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def gather_rendered(local: torch.Tensor, n_scenes: int, group=None) -> torch.Tensor | None:
    """Collect scene-sharded outputs [n_local, ...] on rank 0 in scene order (evaluation / logging only; the
    training path never needs it).  Returns the full [n_scenes, ...] tensor on rank 0, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [len(scene_shard(n_scenes, r, world)) for r in range(world)]
    pad = max(counts)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0, group=group)
    if rank != 0:
        return None
    full = torch.empty((n_scenes,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = scene_shard(n_scenes, r, world)
        full[idx] = out[r][:len(idx)]
    return full
