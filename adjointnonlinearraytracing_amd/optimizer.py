"""Counterparts of the multires helpers of the reference's ``core/optimizer.py`` that touch the
volume: ``upres_scene`` (``:7-10``, via ``core/grid.py:318-330`` ``upres_volume``) and ``reload_opto``
(``:13-41``, Adam-state transfer to the next resolution level).  The optimisation loop itself
(``multires_opt`` ``:44-84``) is plain ``torch.optim`` control flow and is not re-implemented.
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


def reload_opto(old_o: optim.Optimizer, n: torch.Tensor, lr: float) -> optim.Adam:
    """core/optimizer.py:13-41: new Adam for the up-sampled parameter ``n`` whose moments are the
    up-sampled moments of the previous level (hyper-parameters carried over)."""
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
    opto = optim.Adam([n], lr=lr)
    for group in opto.param_groups:
        if ogroup is not None:
            for key in ("betas", "lr", "weight_decay", "eps"):
                group[key] = ogroup[key]
        for p in group["params"]:
            if state is not None:
                opto.state[p] = state
    return opto
