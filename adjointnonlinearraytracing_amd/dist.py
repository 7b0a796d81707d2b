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

    ``apply(rif, x_local, v_local, h, ds, group=None, overlap_chunks=0)``

    The forward's visit order of the shard (and, riding on it, the per-ray iteration counts) is kept on ``ctx`` and
    handed to the adjoint exactly as ``tracer.BackTracerC`` does on one GPU, so the sharded adjoint runs the same fast
    path (no re-sort by exit rays).  The march is the HIP path (module functions ``_hip_trace`` / ``_hip_backtrace``;
    the CPU tests replace those names with stand-ins -- there is no injection hook in the product signature).
    ``overlap_chunks`` = K > 1: the adjoint runs in K depth chunks and final slabs of the grid are reduced while the later
    chunks march (``backtrace_allreduce_overlapped``); for plane-source sets on small shards."""

    @staticmethod
    def forward(ctx, rif, x, v, h, ds, group=None, overlap_chunks=0):
        ctx.shape = rif.shape
        ctx.h, ctx.ds, ctx.group, ctx.chunks = h, ds, group, int(overlap_chunks or 0)
        outx, outv, ctx.order = _hip_trace(rif.detach().flatten(), ctx.shape, x.detach(), v.detach(), h, ds)
        ctx.save_for_backward(rif, outx, outv)          # version-checked: no silent use of a modified grid
        return outx, outv

    @staticmethod
    def backward(ctx, grad_x, grad_v):
        rif, outx, outv = ctx.saved_tensors
        if ctx.chunks > 1:
            drif = backtrace_allreduce_overlapped(rif.detach().flatten(), ctx.shape, outx, outv, grad_x, grad_v, ctx.h, ctx.ds,
                                                  order=ctx.order, chunks=ctx.chunks, group=ctx.group)
            return drif.reshape(*ctx.shape), None, None, None, None, None, None
        drif = _hip_backtrace(rif.detach().flatten(), ctx.shape, outx, outv, grad_x, grad_v, ctx.h, ctx.ds, order=ctx.order)
        drif = allreduce_grad(drif.reshape(*ctx.shape).contiguous(), ctx.group)
        return drif, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------------
# Slab-wise all-reduce of dL/dn under a depth-chunked adjoint (the path's one exchange step).
#
# The gradient of any RAY subset is a full grid, so splitting the rays cannot hide the reduce.  Splitting the STEPS can:
# the rays of a plane-source set march on one clock, so after the iterations [0, k) of all rays every grid plane the rays
# have left behind is FINAL on this rank -- no later chunk adds to it -- and can be summed across ranks while the next chunk
# marches.  `drrt.TracerC.backtrace_chunked` runs the march in chunks and reports, per chunk, where the still-marching rays
# stand and head (bounding boxes) and which samples the chunk contributed at; `SlabReducer` turns that into reduces:
#   * after chunk k the ranks AGREE (one tiny all-reduce(max)) on an axis, a direction and the first plane that is final on
#     every rank; the new slab of planes is packed into a contiguous buffer and its all-reduce is started on a side stream;
#   * whether "final" held is checked after the fact: the sample box of every later chunk must stay clear of the planes
#     already handed in (rays of a lens can turn around; the check is exact, the prediction is not).  If it did not hold on
#     any rank, the slab results are dropped and the whole grid is reduced at the end -- slower, never wrong;
#   * what is left after the last chunk is reduced then -- the only part of the exchange that is exposed.
# The packed buffers are copied back into the grid when everything is done, so the grid is never read while it is written.
# ---------------------------------------------------------------------------------------------------------------------
class SlabReducer:
    def __init__(self, grad: torch.Tensor, shape: Sequence[int], h: float, group=None):
        """grad: flat fp32 dL/dn of this rank (the grid the chunks accumulate into); shape = rif.shape = (D, H, W)."""
        self.g3 = grad.view(*shape)                      # [z, y, x]
        self.shape, self.h, self.group = tuple(int(v) for v in shape), float(h), group
        self.on = dist.is_initialized() and dist.get_world_size(group) > 1
        self.cuda = grad.is_cuda
        self.side = torch.cuda.Stream(device=grad.device) if (self.cuda and self.on) else None
        self.choice = None                               # (axis, down): grid axis 0=x 1=y 2=z and the direction the rays move
        self.edge = None                                 # down: first final plane so far; up: last final plane so far
        self.parts = []                                  # (slice tuple, packed buffer, work handle)
        self.violated = False
        self.stopped = False

    # grid axis a (0 = x, 1 = y, 2 = z) -> dimension of the [z, y, x] view
    def _slice(self, axis: int, lo: int, hi: int):
        idx = [slice(None)] * 3
        idx[2 - axis] = slice(lo, hi)
        return tuple(idx)

    def _agree(self, vec: List[float]) -> List[float]:
        """Element-wise max over the ranks (one small collective; every rank gets the same list)."""
        if not self.on:
            return vec
        t = torch.tensor(vec, dtype=torch.float32, device=self.g3.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t.cpu().tolist()

    def _start(self, axis: int, lo: int, hi: int, ready_event) -> None:
        """Pack planes [lo, hi) of `axis` and start their all-reduce (side stream on the GPU)."""
        if hi <= lo:
            return
        sl = self._slice(axis, lo, hi)
        if self.side is not None:
            if ready_event is not None:
                self.side.wait_event(ready_event)
            with torch.cuda.stream(self.side):
                buf = self.g3[sl].contiguous()
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.on else None
        else:
            buf = self.g3[sl].contiguous()
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.on else None
        self.parts.append((sl, buf, work))

    def after_chunk(self, progress: dict, ready_event=None) -> None:
        """progress: drrt.decode_chunk_progress() of the chunk that has just finished on this rank (ready_event: recorded
        on the march's stream behind it).  Every rank must call this once per chunk, in the same order."""
        n = {0: self.shape[2], 1: self.shape[1], 2: self.shape[0]}
        h = self.h
        # (i) did this chunk's samples stay clear of what has been handed in?  (ii) candidates for what is final now
        touched = 0.0
        if self.choice is not None and progress["sample_min"] is not None:
            a, down = self.choice
            if down:
                touched = 1.0 if int(progress["sample_max"][a] / h) + 1 >= self.edge else 0.0
            else:
                touched = 1.0 if int(progress["sample_min"][a] / h) <= self.edge else 0.0
        vec = [touched]
        for a in range(3):
            for down in (True, False):
                if progress["active"] == 0:
                    ok, bound = 1.0, (0.0 if down else float(n[a] - 1))          # nothing marches: everything is final
                elif down and progress["vel_min"][a] > 0.0:                        # x -= ds * v: moving towards smaller a
                    ok, bound = 1.0, float(min(n[a], max(0, int(progress["pos_max"][a] / h) + 2)))
                elif (not down) and progress["vel_max"][a] < 0.0:
                    ok, bound = 1.0, float(max(-1, min(n[a] - 1, int(progress["pos_min"][a] / h) - 1)))
                else:
                    ok, bound = 0.0, (float(n[a]) if down else -1.0)
                # max-reduced: -ok -> -(min ok); down: first final plane -> max; up: last final plane -> min via negation
                vec += [-ok, bound if down else -bound]
        vec = self._agree(vec)
        if vec[0] > 0.0:
            self.violated = True
        if self.violated or self.stopped:
            return
        cands = {}
        k = 1
        for a in range(3):
            for down in (True, False):
                ok, b = -vec[k], vec[k + 1]
                k += 2
                if ok >= 1.0:
                    edge = int(b) if down else int(-b)
                    cands[(a, down)] = (edge, (n[a] - edge) if down else (edge + 1))   # (edge, planes final)
        if self.choice is None:
            if not cands:
                return
            self.choice = max(cands, key=lambda c: (cands[c][1], -c[0], c[1]))         # the same on every rank
            a, down = self.choice
            self.edge = n[a] if down else -1
        a, down = self.choice
        if self.choice not in cands:
            self.stopped = True                              # some ray turned: hand in nothing more before the end
            return
        edge = cands[self.choice][0]
        if down and edge < self.edge:
            self._start(a, edge, self.edge, ready_event)
            self.edge = edge
        elif (not down) and edge > self.edge:
            self._start(a, self.edge + 1, edge + 1, ready_event)
            self.edge = edge

    def finish(self, ready_event=None) -> torch.Tensor:
        """After the last chunk: reduce what is left (or, after a violation, the whole grid), copy the packed slabs back.
        Returns the flat grid.  On the GPU the copies are queued on the CURRENT stream behind the side stream's work."""
        n = {0: self.shape[2], 1: self.shape[1], 2: self.shape[0]}
        if self.violated:
            for _, _, w in self.parts:                       # let the collectives in flight finish, then ignore them
                if w is not None:
                    w.wait()
            self.parts = []
            if self.on:
                dist.all_reduce(self.g3, op=dist.ReduceOp.SUM, group=self.group)
            return self.g3.reshape(-1)
        if self.choice is None:
            if self.on:
                dist.all_reduce(self.g3, op=dist.ReduceOp.SUM, group=self.group)
            return self.g3.reshape(-1)
        a, down = self.choice
        if down:
            self._start(a, 0, self.edge, ready_event)
        else:
            self._start(a, self.edge + 1, n[a], ready_event)
        for sl, buf, w in self.parts:
            if w is not None:
                w.wait()                                     # GPU: the current stream waits for the collective
            elif self.side is not None:
                torch.cuda.current_stream(self.g3.device).wait_stream(self.side)
            self.g3[sl].copy_(buf)
        self.parts = []
        return self.g3.reshape(-1)


def _hip_backtrace_chunked(rif_flat, shape, xt, vt, gx, gv, h, ds, order, chunks, on_chunk):
    from . import drrt
    return drrt.TracerC().backtrace_chunked(rif_flat, shape, xt, vt, gx, gv, h, ds, order=order, chunks=chunks, on_chunk=on_chunk)


def _decode_progress(progress):
    from . import drrt
    return drrt.decode_chunk_progress(progress)


def backtrace_allreduce_overlapped(rif_flat, shape, xt, vt, gx, gv, h, ds, order=None, chunks: int = 4, group=None):
    """dL/dn of the GLOBAL ray set from this rank's shard, with the all-reduce started slab by slab under the march
    (``SlabReducer``): the adjoint runs in `chunks` depth chunks, all queued at once; as each finishes the ranks agree on
    the planes that are final everywhere and reduce them on a side stream.  Same result as ``allreduce_grad`` of the
    one-launch adjoint up to fp32 summation order.  Pays off for plane-source sets (rays on one clock) at small shards,
    where the whole-grid reduce is a large part of the step; for multi-view sets nothing is final early and it
    degenerates to the plain reduce."""
    done = []                                                # (progress tensor, event) per chunk, in order

    def on_chunk(k, grad, progress):
        ev = None
        if grad.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(grad.device))
        done.append((grad, progress, ev))

    grad = _hip_backtrace_chunked(rif_flat, shape, xt, vt, gx, gv, h, ds, order, chunks, on_chunk)
    red = SlabReducer(grad, shape, h, group)
    for _, progress, ev in done:                             # every chunk is queued; now follow them as they finish
        if ev is not None:
            ev.synchronize()
        red.after_chunk(_decode_progress(progress), ev)
    return red.finish(done[-1][2] if done else None)
