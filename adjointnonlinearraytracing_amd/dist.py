"""Ray-sharded multi-GPU execution.  Not present in the reference, which is
single-process / single-GPU (grep for nccl|mpi|distributed in /root/reference: no hits).

Partition: rays are independent units of the forward march -> contiguous equal shards, one per
rank (one process per GPU); the refractive-index grid is replicated.  The adjoint has exactly ONE
exchange step: every rank accumulates its rays' dL/dn into a private fp32 grid, then a single
all-reduce(sum) over the grid (RCCL over xGMI with backend "nccl"; gloo on CPU for tests) gives
every rank the full gradient, so replicated optimisers stay in lock-step.

The march is the HIP path (``drrt.TracerC``); the CPU tests (gloo, world 2) replace the two module-level
functions that call it with stand-ins, so the sharding / reduction logic is testable without a GPU.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple, Union

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun env).
    Returns (rank, world, local_rank).  backend: "nccl" (= RCCL on ROCm) when CUDA is visible,
    else "gloo"."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous near-equal shard [lo, hi) of n rays for `rank` of `world`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_views(rays_per_view: Sequence[int], rank: int, world: int) -> List[Tuple[int, int]]:
    """Global [lo, hi) of this rank's share of EVERY view of a multi-view ray set (``rays_per_view`` = the ``nrays`` list
    the generators return, e.g. ``rand_rays_cube``): each view is split into `world` contiguous near-equal shards, so a
    rank marches a strip of every view.  Views differ in cost (rays of an oblique view run up to 1.5x longer per
    ray-step than those of an axis-aligned one), so a contiguous split of the concatenated set --
    whole views to single ranks -- is unbalanced by construction; this one is balanced whatever the views are."""
    out, base = [], 0
    for n in rays_per_view:
        lo, hi = shard_bounds(int(n), rank, world)
        out.append((base + lo, base + hi))
        base += int(n)
    return out


def shard_rays(rank: int, world: int, *tensors: torch.Tensor, views: Union[None, int, Sequence[int]] = None):
    """Slice every (N, ...) tensor to this rank's shard.

    views=None          one contiguous near-equal shard of the whole set (single-view sets);
    views=k (int)       the set is k equal views back to back: a strip of every view, concatenated in view order;
    views=[n0, n1, ..]  the same for views of different sizes (the generators' ``nrays`` list).
    With views the rank's own per-view counts are ``[hi - lo for lo, hi in shard_views(...)]`` (``local_views``)."""
    n = tensors[0].shape[0]
    if views is None:
        lo, hi = shard_bounds(n, rank, world)
        return tuple(t[lo:hi] for t in tensors)
    per = _per_view(n, views)
    spans = shard_views(per, rank, world)
    return tuple(torch.cat([t[lo:hi] for lo, hi in spans]) for t in tensors)


def _per_view(n: int, views: Union[int, Sequence[int]]) -> List[int]:
    if isinstance(views, int):
        if views < 1 or n % views:
            raise ValueError(f"{n} rays do not split into {views} equal views; pass the per-view counts")
        return [n // views] * views
    per = [int(v) for v in views]
    if sum(per) != n:
        raise ValueError(f"per-view counts sum to {sum(per)}, the set has {n} rays")
    return per


def local_views(n: int, views: Union[int, Sequence[int]], rank: int, world: int) -> List[int]:
    """This rank's ray count per view after ``shard_rays(..., views=views)`` -- what a per-view sensor loop splits by
    (``xmp.split(rpv)``, core/image_opt.py:102)."""
    return [hi - lo for lo, hi in shard_views(_per_view(n, views), rank, world)]


def allreduce_grad(grad: torch.Tensor, group=None) -> torch.Tensor:
    """The path's single collective: sum the per-rank dL/dn grids in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return grad


class _AllReduceImage(torch.autograd.Function):
    """forward: all-reduce(sum) of the per-rank partial images; backward: identity.  Every rank goes on to evaluate
    the SAME loss on the SAME summed image, so dL/d(total) is already identical everywhere and
    d(total)/d(partial of this rank) = I: no collective in the backward pass."""

    @staticmethod
    def forward(ctx, img, group):
        out = img.detach().clone()
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def allreduce_image(img: torch.Tensor, group=None) -> torch.Tensor:
    """Differentiable all-reduce(sum) of a sensor image for image losses on ray shards: the reference
    normalises and compares WHOLE images (``sum_norm`` + MSE, core/image_opt.py:99-112), which are sums over rays, so
    each rank splats its own rays (``sensor.generate_sensor``), the partial images (512^2 x 4 B = 1 MiB) are summed
    here BEFORE the normalisation, every rank computes the same loss, and each rank back-propagates through its own
    rays only; ``ShardedBackTracerC.backward`` then sums the per-rank dL/dn grids.  One rank: the identity."""
    return _AllReduceImage.apply(img, group)


def _hip_trace(rif_flat, shape, x, v, h, ds):
    """-> (xt, vt, order): the forward's visit order rides along so the sharded adjoint can reuse it."""
    from . import drrt
    xt, vt = drrt.TracerC().trace(rif_flat, shape, x, v, h, ds)
    return xt, vt, drrt.keep_order(drrt.last_order)


def _hip_backtrace(rif_flat, shape, xt, vt, gx, gv, h, ds, order=None):
    from . import drrt
    return drrt.TracerC().backtrace(rif_flat, shape, xt, vt, gx, gv, h, ds, order=order)


class ShardedBackTracerC(torch.autograd.Function):
    """``BackTracerC`` (core/tracer.py:294-335) over this rank's ray shard; backward all-reduces
    dL/dn so every rank returns the gradient of the GLOBAL ray set.

    ``apply(rif, x_local, v_local, h, ds, group=None)``

    The forward's visit order of the shard (and, riding on it, the per-ray iteration counts) is kept on ``ctx`` and
    handed to the adjoint exactly as ``tracer.BackTracerC`` does on one GPU, so the sharded adjoint runs the same fast
    path (no re-sort by exit rays).  The march is the HIP path (module functions ``_hip_trace`` / ``_hip_backtrace``;
    the CPU tests replace those two names with stand-ins -- there is no injection hook in the product signature)."""

    @staticmethod
    def forward(ctx, rif, x, v, h, ds, group=None):
        ctx.shape = rif.shape
        ctx.h, ctx.ds, ctx.group = h, ds, group
        outx, outv, ctx.order = _hip_trace(rif.detach().flatten(), ctx.shape, x.detach(), v.detach(), h, ds)
        ctx.save_for_backward(rif, outx, outv)          # version-checked: no silent use of a modified grid
        return outx, outv

    @staticmethod
    def backward(ctx, grad_x, grad_v):
        rif, outx, outv = ctx.saved_tensors
        drif = _hip_backtrace(rif.detach().flatten(), ctx.shape, outx, outv, grad_x, grad_v, ctx.h, ctx.ds, order=ctx.order)
        drif = allreduce_grad(drif.reshape(*ctx.shape).contiguous(), ctx.group)
        return drif, None, None, None, None, None
