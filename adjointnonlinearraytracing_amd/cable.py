"""Counterpart of the reference's ``core/cable.py``: the radial refractive-index profile of the fibre experiment as a
torch object -- ``Cable(rif, radius, length)`` with ``bounds`` (``:20-24``) and ``GetLinear`` (``:94-119``: n and grad n at
points, linear in the radius).  ``core/fiber_opt.py`` uses ``GetLinear`` once per pass to scale the start directions by
the boundary index (``:159-161``); the march itself is ``tracer.BackCableTracerC`` (HIP).  (``upres_volume`` ``:122-134``
constructs ``Cable`` with two arguments and cannot run as written; the experiment up-samples with its own midpoint rule,
``core/fiber_opt.py:60-68``, which ``examples/fiber_demo.py`` restates.)  These are a handful of element-wise torch operations and one
two-tap gather of the profile, evaluated on whatever device the tensors live on, differentiable w.r.t. the profile like
the reference's (pinned against the reference's own ``Cable.GetLinear`` by ``tests/golden/getlinear_cable.npz``)."""
from __future__ import annotations

import torch


class Cable:

    def __init__(self, rif: torch.Tensor, radius, length):
        self.rif = rif
        self.res = rif.size()
        self.radius = radius
        self.length = length
        self.h = radius / (rif.shape[0] - 1)                       # :11
        self.device = rif.device

    def check_input(self, x: torch.Tensor) -> None:
        if x.device != self.device:                                  # :14-17
            raise ValueError("input on device: {}, grid on device: {}".format(x.device, self.device))

    def bounds(self, x: torch.Tensor) -> torch.Tensor:
        """:20-24 as written (the radial norm is taken over ALL points at once, not per point)."""
        r = torch.norm(x[:, [0, 2]])
        half = torch.abs(x[:, 1] - (self.length / 2))
        return (r < self.radius) & (half < (self.length / 2))

    def GetLinear(self, x: torch.Tensor):
        """:94-119 -> (n, grad n): the profile sampled at the distance from the axis (x = z = radius), clamped two-tap
        linear interpolation with the weight taken against the UNclamped lower index, gradient along the radial unit
        vector (zero on the axis), scaled by 1/h."""
        self.check_input(x)
        off = x - self.radius
        off = torch.stack([off[:, 0], torch.zeros_like(off[:, 1]), off[:, 2]], dim=-1)
        r = torch.norm(off, dim=-1)
        rn = r / self.h
        i0 = torch.floor(rn).long()
        w1 = torch.clip(rn - i0, 0, 1)
        top = self.res[0] - 1
        f0 = self.rif[torch.clip(i0, 0, top)]
        f1 = self.rif[torch.clip(i0 + 1, 0, top)]
        f = f0 * (1 - w1) + f1 * w1
        unit = off / r[:, None]
        unit = torch.where((r < 1e-6)[:, None], torch.zeros_like(unit), unit)
        return f, (f1 - f0)[:, None] * unit / self.h

