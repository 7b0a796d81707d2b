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


def reload_opto(old_o: optim.Optimizer, n: torch.Tensor, lr: float) -> optim.Optimizer:
    """core/optimizer.py:13-41: new Adam for the up-sampled parameter ``n`` whose moments are the
    up-sampled moments of the previous level (hyper-parameters carried over).  A ``MaskedAdam`` yields a ``MaskedAdam``."""
    ogroup, state = None, None
    for group in old_o.param_groups:
        ogroup = group
        for p in group["params"]:
            if len(old_o.state[p]) == 0:
                continue
            ostate = old_o.state[p]
            state = dict(step=ostate["step"],
                         exp_avg=upres_scene(ostate["exp_avg"], n.shape[0]),
                         exp_avg_sq=upres_scene(ostate["exp_avg_sq"], n.shape[0]))
    fused = isinstance(old_o, MaskedAdam)
    opto = MaskedAdam([n], lr=lr) if fused else optim.Adam([n], lr=lr)
    for group in opto.param_groups:
        if ogroup is not None:
            for key in ("betas", "lr", "weight_decay", "eps") + (("mask_boundary", "clamp_min") if fused else ()):
                group[key] = ogroup[key]
        for p in group["params"]:
            if state is not None:
                opto.state[p] = state
    return opto


def multires_opt(func, eta, iterations, res_list, log_func=None, lr=1e-3, statename="result", fused=True):
    """core/optimizer.py:44-84: coarse-to-fine Adam optimisation of the volume ``eta`` -- ``iterations * (level + 1)``
    steps per entry of ``res_list``, boundary gradients masked, values clamped at 1, the volume and the Adam moments
    up-sampled between levels, a checkpoint saved per level.  Returns ``(n, loss_hist)`` like the reference.

    ``fused=True`` (default) runs mask + Adam + clamp as one HIP pass (``MaskedAdam``); ``fused=False`` runs the
    reference's statements literally (A/B and parity testing).  Differences from the reference as written: ``log_func``
    may be None; no tqdm bars; the loss history is collected on the device and read back once per level instead of
    ``loss.item()`` every iteration (a host sync per step)."""
    n = eta.clone()
    n.requires_grad = True
    opto = MaskedAdam([n], lr=lr) if fused else optim.Adam([n], lr=lr)
    iteration_count = 0
    loss_hist = []
    for res_iter in range(len(res_list)):
        if not fused:
            mask = torch.ones_like(n, dtype=torch.bool, requires_grad=False)               # :54-55
            mask[1:-1, 1:-1, 1:-1] = 0
        level_losses = []
        for _ in range(iterations * (res_iter + 1)):                                        # :56
            opto.zero_grad()
            loss = func(n)
            loss.backward()
            with torch.no_grad():
                if log_func is not None:
                    log_func(iteration_count, n)                                            # :60
                if not fused:
                    n.grad[mask] = 0                                                        # :61
            opto.step()                                                                     # :63 (fused: + :61, :66)
            with torch.no_grad():
                if not fused:
                    n.clamp_(min=1)                                                         # :66
                level_losses.append(loss.detach())
            iteration_count += 1
        with torch.no_grad():
            if level_losses:
                loss_hist.extend(torch.stack([l.reshape(()) for l in level_losses]).cpu().tolist())   # :67
            torch.save({"rif": n, "opto_state_dict": opto.state_dict(), "loss_hist": torch.tensor(loss_hist)},
                       statename)                                                           # :72-76
            if res_iter < len(res_list) - 1:
                n = upres_scene(n, res_list[res_iter + 1])                                  # :78
                n.requires_grad = True
                opto = reload_opto(opto, n, (0.5 ** res_iter) * lr)                         # :80
    return n, loss_hist
