"""Differentiable pure-torch restatement of ``Tracer::trace`` -- TEST INFRASTRUCTURE ONLY.

Plays the role the reference gives to ``drrt.TracerD`` (enoki autodiff,
``/root/reference/core/tracer.py:16-66``): an automatic-differentiation comparator for
the hand-written adjoint ``Tracer::backtrace`` (``src/tracer.cpp:384-440``).  The forward
recurrences follow ``src/tracer.cpp:35-100`` and ``src/volume.cpp:101-181`` (axis order Q2:
``p.x`` indexes the LAST torch axis).  float64 by default.
"""
from __future__ import annotations

import torch


def eval_grad(rif: torch.Tensor, p: torch.Tensor, h: float, mask: torch.Tensor):
    """volume::eval_grad on a (D,H,W) tensor at points p (N,3); masked lanes read 0."""
    D, H, W = rif.shape
    res = torch.tensor([W, H, D], device=p.device)            # (W,H,D) = reversed torch shape
    pm = p * (1.0 / h)
    ip = torch.floor(pm).long()
    w0 = pm - ip.to(p.dtype)
    w1 = 1.0 - w0
    i0 = torch.minimum(torch.clamp(ip, min=0), res - 1)
    i1 = torch.minimum(torch.clamp(ip + 1, min=0), res - 1)
    flat = rif.reshape(-1)

    def tap(bx, by, bz):
        x = i1[:, 0] if bx else i0[:, 0]
        y = i1[:, 1] if by else i0[:, 1]
        z = i1[:, 2] if bz else i0[:, 2]
        v = flat[(z * H + y) * W + x]
        return torch.where(mask, v, torch.zeros_like(v))

    v = {(a, b, c): tap(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)}
    wx = (w1[:, 0], w0[:, 0])
    wy = (w1[:, 1], w0[:, 1])
    wz = (w1[:, 2], w0[:, 2])
    n = sum(wx[a] * wy[b] * wz[c] * v[(a, b, c)] for a in (0, 1) for b in (0, 1) for c in (0, 1))
    nx = sum((1 if a else -1) * wy[b] * wz[c] * v[(a, b, c)] for a in (0, 1) for b in (0, 1) for c in (0, 1))
    ny = sum((1 if b else -1) * wx[a] * wz[c] * v[(a, b, c)] for a in (0, 1) for b in (0, 1) for c in (0, 1))
    nz = sum((1 if c else -1) * wx[a] * wy[b] * v[(a, b, c)] for a in (0, 1) for b in (0, 1) for c in (0, 1))
    return n, torch.stack([nx, ny, nz], dim=-1) * (1.0 / h)


def inbounds(shape, h, p):
    D, H, W = shape
    hi = torch.tensor([(W - 1) * h, (H - 1) * h, (D - 1) * h], dtype=p.dtype, device=p.device)
    return ((p >= 0) & (p < hi)).all(dim=-1)


def escaped(shape, h, p, v):
    D, H, W = shape
    hi = torch.tensor([(W - 1) * h, (H - 1) * h, (D - 1) * h], dtype=p.dtype, device=p.device)
    return (((p < 0) & (v < 0)) | ((p >= hi) & (v > 0))).any(dim=-1)


def trace(rif: torch.Tensor, pos: torch.Tensor, vel: torch.Tensor, h: float, ds: float):
    """Differentiable w.r.t. ``rif``.  Returns (xt, vt) like Tracer::trace."""
    shape = rif.shape
    max_steps = int(4 * h * max(shape) / ds)
    x, v = pos.clone(), vel.clone()
    xt, vt = pos.clone(), vel.clone()
    inside = inbounds(shape, h, x)
    esc = torch.zeros_like(inside)
    for _ in range(max_steps):
        n, g = eval_grad(rif, x, h, inside)
        v = v + (ds * n)[:, None] * g
        x = x + ds * v
        cur_inside = inbounds(shape, h, x)
        cross = inside & ~cur_inside
        esc = esc | cross | escaped(shape, h, x, v)
        xt = torch.where(cross[:, None], x, xt)
        vt = torch.where(cross[:, None], v, vt)
        if bool(esc.all()):
            break
        inside = cur_inside
    if not bool(esc.all()):
        xt = torch.where(esc[:, None], xt, x)
    return xt, vt
