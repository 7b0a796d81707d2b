"""Counterpart of the reference's ``core/tracer.py``: the ``torch.autograd.Function`` classes the
``*_opt.py`` scripts call, with identical names, argument order and return arity
(``/root/reference/core/tracer.py:294-526``), backed by ``drrt.TracerC`` (HIP kernels).

Contract reproduced from the reference:
  * forward inputs are detached, flattened C-order and narrowed to fp32 (``:299-301``);
  * forward keeps ``rif``, the exit rays and the scalars for backward and returns fresh tensors.  The reference
    snapshots the grid (``FloatC(rif.flatten())`` copies, ``:299``); here ``rif`` and the exit rays are kept with
    ``ctx.save_for_backward`` -- no copies, and autograd's version check turns an in-place update of ``rif`` (or of
    the returned exit rays) between forward and backward into a RuntimeError instead of a silently wrong gradient;
  * backward returns ``drif`` reshaped to ``rif.shape`` and ``None`` for every other input
    (no gradient w.r.t. ``x``, ``v``: ``:335,386,432,479,526``);
  * ``BackPlaneTracerC`` / ``BackTargetTracerC`` backward run the GENERIC ``backtrace`` from the
    recorded state (``:376,422``); ``BackPlaneTracerC.backward`` zeroes ``grad_x`` on
    rays whose ``outmask`` gradient is set, as written (``:366-367``).
The enoki pool trim (``enoki.cuda_malloc_trim()``, ``:314``) has no counterpart: outputs are
torch allocations.
"""
from __future__ import annotations

import torch

from . import drrt


class BackTracerC(torch.autograd.Function):
    """core/tracer.py:294-335 -- ``apply(rif, x, v, h, ds) -> (xt, vt)``."""

    @staticmethod
    def forward(ctx, rif, x, v, h, ds):
        ctx.shape = rif.shape
        ctx.h, ctx.ds = h, ds
        outx, outv = drrt.TracerC().trace(rif.detach().flatten(), ctx.shape, x.detach(), v.detach(), h, ds)
        ctx.order = drrt.keep_order(drrt.last_order)     # the adjoint visits rays in the forward's bundle order (a private
        #                                                  copy: other tracer calls may come before backward)
        ctx.save_for_backward(rif, outx, outv)
        return outx, outv

    @staticmethod
    def backward(ctx, grad_x, grad_v):
        rif, outx, outv = ctx.saved_tensors
        drif = drrt.TracerC().backtrace(rif.detach().flatten(), ctx.shape, outx, outv, grad_x, grad_v,
                                        ctx.h, ctx.ds, order=ctx.order).reshape(*ctx.shape)
        return drif, None, None, None, None


class BackPlaneTracerC(torch.autograd.Function):
    """core/tracer.py:338-386 -- ``apply(rif, x, v, sp, sn, h, ds) -> (xt, vt, failmask bool)``."""

    @staticmethod
    def forward(ctx, rif, x, v, sp, sn, h, ds):
        ctx.shape = rif.shape
        ctx.h, ctx.ds = h, ds
        outx, outv, outmask = drrt.TracerC().trace_pln(
            rif.detach().flatten(), ctx.shape, x.detach(), v.detach(), sp.detach(), sn.detach(), h, ds)
        ctx.order = drrt.keep_order(drrt.last_order)
        outmask = outmask.to(torch.bool)
        ctx.mark_non_differentiable(outmask)
        ctx.save_for_backward(rif, outx, outv)
        return outx, outv, outmask

    @staticmethod
    def backward(ctx, grad_x, grad_v, outmask):
        rif, outx, outv = ctx.saved_tensors
        if outmask is not None and outmask.dtype == torch.bool:     # as written, :366-367
            grad_x = grad_x.clone()
            grad_x[outmask] = 0
        drif = drrt.TracerC().backtrace(rif.detach().flatten(), ctx.shape, outx, outv, grad_x, grad_v,
                                        ctx.h, ctx.ds, order=ctx.order).reshape(*ctx.shape)
        return drif, None, None, None, None, None, None


class BackTargetTracerC(torch.autograd.Function):
    """core/tracer.py:389-432 -- ``apply(rif, x, v, sp, h, ds) -> (xt, vt, dist2)``."""

    @staticmethod
    def forward(ctx, rif, x, v, sp, h, ds):
        ctx.shape = rif.shape
        ctx.h, ctx.ds = h, ds
        outx, outv, dist2 = drrt.TracerC().trace_target(
            rif.detach().flatten(), ctx.shape, x.detach(), v.detach(), sp.detach(), h, ds)
        ctx.order = drrt.keep_order(drrt.last_order)
        ctx.save_for_backward(rif, outx, outv)
        return outx, outv, dist2

    @staticmethod
    def backward(ctx, grad_x, grad_v, outdist):
        rif, outx, outv = ctx.saved_tensors
        drif = drrt.TracerC().backtrace(rif.detach().flatten(), ctx.shape, outx, outv, grad_x, grad_v,
                                        ctx.h, ctx.ds, order=ctx.order).reshape(*ctx.shape)
        return drif, None, None, None, None, None


class BackSDFTracerC(torch.autograd.Function):
    """core/tracer.py:435-479 -- ``apply(rif, sdf, x, v, h, ds) -> (xt, vt)``."""

    @staticmethod
    def forward(ctx, rif, sdf, x, v, h, ds):
        ctx.shape = rif.shape
        ctx.h, ctx.ds = h, ds
        outx, outv = drrt.TracerC().trace_sdf(rif.detach().flatten(), sdf.detach().flatten(), ctx.shape, x.detach(),
                                              v.detach(), h, ds)
        ctx.order = drrt.keep_order(drrt.last_order)
        ctx.save_for_backward(rif, sdf, outx, outv)
        return outx, outv

    @staticmethod
    def backward(ctx, grad_x, grad_v):
        rif, sdf, outx, outv = ctx.saved_tensors
        drif = drrt.TracerC().backtrace_sdf(rif.detach().flatten(), sdf.detach().flatten(), ctx.shape, outx, outv,
                                            grad_x, grad_v, ctx.h, ctx.ds, order=ctx.order).reshape(*ctx.shape)
        return drif, None, None, None, None, None


class BackCableTracerC(torch.autograd.Function):
    """core/tracer.py:482-526 -- ``apply(rif (Rr,), radius, length, x, v, sp, ds) -> (xt, vt, dist2)``."""

    @staticmethod
    def forward(ctx, rif, radius, length, x, v, sp, ds):
        ctx.radius, ctx.length, ctx.ds = radius, length, ds
        outx, outv, dist2 = drrt.TracerC().trace_cable(
            rif.detach().flatten(), radius, length, x.detach(), v.detach(), sp.detach(), ds)
        ctx.save_for_backward(rif, outx, outv)
        return outx, outv, dist2

    @staticmethod
    def backward(ctx, grad_x, grad_v, outdist):
        rif, outx, outv = ctx.saved_tensors
        drif = drrt.TracerC().backtrace_cable(rif.detach().flatten(), ctx.radius, ctx.length, outx, outv,
                                              grad_x, grad_v, ctx.ds).reshape(rif.shape)
        return drif, None, None, None, None, None, None


# The enoki-autodiff classes of the reference (core/tracer.py:16-291) are out of scope (two of
# them are broken upstream).  Scripts select them with `autodiff=True`
# (core/luneburg_opt.py:80-83); the names resolve to the adjoint classes so that flag keeps
# working, with the documented difference that no gradient flows to x, v.
ADTracerC = BackTracerC
ADPlaneTracerC = BackPlaneTracerC
ADSDFTracerC = BackSDFTracerC
ADCableTracerC = BackCableTracerC
