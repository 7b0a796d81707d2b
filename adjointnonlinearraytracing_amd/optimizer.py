"""Counterpart of the reference's ``core/optimizer.py``: ``upres_scene`` (``:7-10``, via ``core/grid.py:318-330``
``upres_volume``), ``reload_opto`` (``:13-41``, Adam-state transfer to the next resolution level) and ``multires_opt``
(``:44-84``), the coarse-to-fine optimisation loop every experiment of the reference runs (``core/luneburg_opt.py:147``,
``core/image_opt.py``, ``core/fiber_opt.py`` ...).

The loop's per-iteration tail -- ``n.grad[mask] = 0`` on the boundary layer, ``opto.step()`` (``torch.optim.Adam``) and
``n.clamp_(min=1)`` -- is one fused HIP pass here (``MaskedAdam``, ``drrt_adam_step_f32``): in torch it is a boolean-mask
``index_put`` (a ``nonzero`` with a host sync), about ten element-wise launches and three extra passes over the volume
and its two moments, which at 256^3 costs about as much as the forward march.  ``MaskedAdam`` keeps its state under
torch.optim.Adam's keys (``step``, ``exp_avg``, ``exp_avg_sq``), so ``state_dict()`` / ``reload_opto`` / the saved
checkpoints look like the reference's.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.optim as optim

from . import _lib


def upres_scene(n: torch.Tensor, res: int) -> torch.Tensor:
    """core/optimizer.py:7-10: resample the cubic volume ``n`` to ``(res,)*3`` (trilinear, evaluated in
    float64 like the reference, returned in ``n.dtype``).  HIP kernel; cuda tensors only."""
    if not n.is_cuda:
        raise RuntimeError("upres_scene expects a tensor on the cuda (ROCm) device (no CPU path)")
    src = n.detach().to(torch.float32).contiguous()
    dst = torch.empty((int(res),) * n.dim(), dtype=torch.float32, device=n.device)
    if n.dim() != 3:
        raise RuntimeError("upres_scene: only 3-D volumes are supported")
    with torch.cuda.device(n.device):
        _lib.check(_lib.load().drrt_upres_volume_f32(
            C.c_void_p(src.data_ptr()), (C.c_int * 3)(*src.shape), C.c_void_p(dst.data_ptr()),
            (C.c_int * 3)(*dst.shape), C.c_void_p(torch.cuda.current_stream(n.device).cuda_stream)))
    return dst.to(n.dtype)


class MaskedAdam(optim.Optimizer):
    """``torch.optim.Adam`` (amsgrad = maximize = False) for 3-D fp32 volumes on the cuda (ROCm) device, fused with the
    two statements around ``opto.step()`` in core/optimizer.py:61-66: with ``mask_boundary`` the gradient of the
    outermost voxel layer is treated as (and set to) zero before the update (:54-55, :61), with ``clamp_min`` the
    updated parameter is clamped from below (:66).  Same update formula, same state keys and param-group keys as
    torch's Adam; one kernel launch per parameter per step, no host sync.  There is no CPU path."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, mask_boundary=True,
                 clamp_min=1.0):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameter")            # torch/optim/adam.py raises the same way
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      mask_boundary=mask_boundary, clamp_min=clamp_min))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            flags = (_lib.ADAM_MASK_BOUNDARY if group["mask_boundary"] else 0) | \
                    (_lib.ADAM_CLAMP_MIN if group["clamp_min"] is not None else 0)
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.dim() != 3 or not p.is_contiguous():
                    raise RuntimeError("MaskedAdam expects contiguous 3-D float32 parameters on the cuda (ROCm) device")
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32 or not g.is_contiguous():
                    raise RuntimeError("MaskedAdam expects dense contiguous float32 gradients")
                state = self.state[p]
                if len(state) == 0:                                     # like torch: lazily, zeros
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if m.dtype != torch.float32 or v.dtype != torch.float32 or not m.is_contiguous() or not v.is_contiguous():
                    m = state["exp_avg"] = m.to(torch.float32).contiguous()
                    v = state["exp_avg_sq"] = v.to(torch.float32).contiguous()
                with torch.cuda.device(p.device):
                    _lib.check(lib.drrt_adam_step_f32(
                        C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                        C.c_void_p(v.data_ptr()), (C.c_int * 3)(*p.shape), float(state["step"]), float(group["lr"]),
                        float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]),
                        float(group["clamp_min"] if group["clamp_min"] is not None else 0.0), flags,
                        C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)))
                # The kernel wrote p and p.grad through raw pointers: tell autograd, as any in-place torch op would.  The
                # version counters are what tracer.Back*TracerC's save_for_backward check and the pair-copy reuse token
                # of drrt._march_workspace key on -- without the bump a retained-graph backward after step() would
                # silently pair the NEW grid with the stale pair copy of the old one.
                torch.autograd.graph.increment_version(p)
                torch.autograd.graph.increment_version(g)
        return loss


_ADAM_HYPER = ("betas", "lr", "weight_decay", "eps")
_MASKED_HYPER = ("mask_boundary", "clamp_min")


def _stepped_moments(optimizer: optim.Optimizer):
    """-> (hyper-parameter group, Adam state) of the last parameter of ``optimizer`` that has taken a step, or
    (last group, None) when none has."""
    hyper, moments = None, None
    for hyper in optimizer.param_groups:
        for p in hyper["params"]:
            st = optimizer.state[p]
            if len(st):
                moments = st
    return hyper, moments


def reload_opto(old_o: optim.Optimizer, n: torch.Tensor, lr: float) -> optim.Optimizer:
    """Same contract as core/optimizer.py:13-41: an Adam for the up-sampled parameter ``n`` that continues the previous
    level's run -- its step count, its first and second moments trilinearly up-sampled to ``n``'s resolution
    (``upres_scene``), its hyper-parameters (``lr`` included: like the reference's, the ``lr`` argument only seeds the
    constructor).  A ``MaskedAdam`` yields a ``MaskedAdam`` (+ its two extra hyper-parameters)."""
    masked = isinstance(old_o, MaskedAdam)
    hyper, moments = _stepped_moments(old_o)
    fresh = (MaskedAdam if masked else optim.Adam)([n], lr=lr)
    carried = None
    if moments is not None:
        side = n.shape[0]
        carried = {"step": moments["step"],
                   "exp_avg": upres_scene(moments["exp_avg"], side),
                   "exp_avg_sq": upres_scene(moments["exp_avg_sq"], side)}
    for group in fresh.param_groups:
        if hyper is not None:
            group.update({k: hyper[k] for k in _ADAM_HYPER + (_MASKED_HYPER if masked else ())})
        if carried is not None:
            for p in group["params"]:
                fresh.state[p] = carried
    return fresh


def _boundary_mask(n: torch.Tensor) -> torch.Tensor:
    """True on the outermost voxel layer (core/optimizer.py:54-55)."""
    shell = torch.ones_like(n, dtype=torch.bool)
    shell[1:-1, 1:-1, 1:-1] = False
    return shell


def _run_level(func, n, opto, steps, first_iteration, log_func, fused):
    """``steps`` optimisation steps on one resolution level -> the level's losses (device scalars).  ``fused``: mask +
    Adam + clamp are MaskedAdam's one HIP pass; otherwise the reference's three statements (:61, :63, :66)."""
    shell = None if fused else _boundary_mask(n)
    losses = []
    for k in range(steps):
        opto.zero_grad()
        loss = func(n)
        loss.backward()
        with torch.no_grad():
            if log_func is not None:
                log_func(first_iteration + k, n)                                            # :60
            if shell is not None:
                n.grad[shell] = 0
        opto.step()
        with torch.no_grad():
            if shell is not None:
                n.clamp_(min=1)
            losses.append(loss.detach().reshape(()))
    return losses


def multires_opt(func, eta, iterations, res_list, log_func=None, lr=1e-3, statename="result", fused=True):
    """Same contract as core/optimizer.py:44-84: coarse-to-fine Adam optimisation of the volume ``eta`` --
    ``iterations * (level + 1)`` steps on level ``level`` of ``res_list``, boundary gradients masked, values clamped at 1,
    a checkpoint ``statename`` (keys ``rif``, ``opto_state_dict``, ``loss_hist``) after every level, then the volume and the
    Adam moments up-sampled to the next entry of ``res_list`` with the learning-rate seed halved per level.  Returns
    ``(n, loss_hist)``.

    ``fused=True`` (default) runs mask + Adam + clamp as one HIP pass (``MaskedAdam``); ``fused=False`` runs the
    reference's statements literally (A/B and parity testing).  Differences from the reference as written: ``log_func``
    may be None; no tqdm bars; the losses stay on the device and are read back once per level instead of ``loss.item()``
    every iteration (a host sync per step)."""
    n = eta.clone().requires_grad_(True)
    opto = MaskedAdam([n], lr=lr) if fused else optim.Adam([n], lr=lr)
    done, loss_hist = 0, []
    levels = len(res_list)
    for level in range(levels):
        steps = iterations * (level + 1)                                                    # :56
        losses = _run_level(func, n, opto, steps, done, log_func, fused)
        done += steps
        with torch.no_grad():
            if losses:
                loss_hist += torch.stack(losses).cpu().tolist()                             # :67, one read-back per level
            torch.save({"rif": n, "opto_state_dict": opto.state_dict(), "loss_hist": torch.tensor(loss_hist)},
                       statename)                                                           # :72-76
            if level + 1 < levels:
                n = upres_scene(n, res_list[level + 1]).requires_grad_(True)                # :78
                opto = reload_opto(opto, n, lr * 0.5 ** level)                              # :80
    return n, loss_hist
