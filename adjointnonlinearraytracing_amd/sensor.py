"""Counterpart of the part of the reference's ``core/sensor.py`` that directly follows the march:
``trace_rays_to_plane`` (``:195-202``), ``get_tan_vecs`` (``:219-231``), ``generate_sensor``
(``:5-28``), the differentiable 2-D image splat used by the image / Luneburg experiments
(``core/image_opt.py:99-101``, ``core/luneburg_opt.py:121-123``), and ``generate_inf_sensor`` (``:31-53``),
the far-field (direction histogram) sensor of the image experiments (``core/image_opt.py:116``).

``get_sdf_vals_near`` / ``get_sdf_vals_far`` (``:102-138``) sample a texture at the same sensor coordinates (fused kernels).

``generate_sensor`` keeps the reference's signature; on ``cuda`` (ROCm) tensors it runs the fused
HIP kernels (``csrc/drrt_sensor.hip``: ray -> plane -> sensor frame -> 16 tent taps -> atomics, and
the analytic backward that yields ``(grad_x, grad_v)`` directly) instead of ~20 (N,16)-sized torch
temporaries + ``index_put_``.  There is no CPU compute path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class _RaysToPlane(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, v, p, n):
        dev = x.device
        with torch.cuda.device(dev):
            x_ = x.detach().to(torch.float32).contiguous()
            v_ = v.detach().to(torch.float32).contiguous()
            p_ = p.detach().to(device=dev, dtype=torch.float32).contiguous()
            n_ = n.detach().to(device=dev, dtype=torch.float32).contiguous()
            stride = 3 if p_.shape[0] > 1 else 0
            xo = torch.empty_like(x_)
            _lib.check(_lib.load().drrt_rays_to_plane_f32(
                x_.shape[0], C.c_void_p(x_.data_ptr()), C.c_void_p(v_.data_ptr()), C.c_void_p(p_.data_ptr()),
                C.c_void_p(n_.data_ptr()), stride, C.c_void_p(xo.data_ptr()),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        ctx.save_for_backward(x_, v_, p_, n_)
        ctx.stride = stride
        return xo

    @staticmethod
    def backward(ctx, g):
        x_, v_, p_, n_ = ctx.saved_tensors
        dev = x_.device
        with torch.cuda.device(dev):
            g_ = g.detach().to(torch.float32).contiguous()
            gx, gv = torch.empty_like(x_), torch.empty_like(v_)
            _lib.check(_lib.load().drrt_rays_to_plane_bwd_f32(
                x_.shape[0], C.c_void_p(x_.data_ptr()), C.c_void_p(v_.data_ptr()), C.c_void_p(p_.data_ptr()),
                C.c_void_p(n_.data_ptr()), ctx.stride, C.c_void_p(g_.data_ptr()), C.c_void_p(gx.data_ptr()),
                C.c_void_p(gv.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return gx, gv, None, None


def trace_rays_to_plane(rays, plane):
    """core/sensor.py:195-202: intersect the rays with their planes, ``(x + t v, v)`` with ``t = n.(p - x) / n.v``.

    The reference writes the two dot products as ``torch.matmul`` on (N,1,3) x (N,3,1) operands -- a batched matmul of
    N one-by-three products, ~25 ms forward + ~55 ms backward for 1M rays on this GPU, 15x the march.  For fp32 (N,3)
    rays on the cuda (ROCm) device with (N,3) or (1,3) planes that are constants (they are in every experiment of the
    reference) this runs one fused HIP kernel each way; any other input takes the reference's torch expressions."""
    x, v = rays
    p, n = plane
    fused = (x.is_cuda and x.dtype == torch.float32 and v.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 3
             and v.shape == x.shape and p.dim() == 2 and n.dim() == 2 and p.shape == n.shape and p.shape[1] == 3
             and p.shape[0] in (1, x.shape[0]) and not p.requires_grad and not n.requires_grad)
    if fused:
        return _RaysToPlane.apply(x, v, p, n), v
    t = torch.matmul(n[:, None, :], (p - x)[:, :, None]).squeeze(2)
    t = t / torch.matmul(n[:, None, :], v[:, :, None]).squeeze(2)
    return (x + t * v), v


def get_tan_vecs(n, t=None):
    """core/sensor.py:219-231."""
    if t is None:
        t2 = torch.zeros_like(n)
        if torch.abs(n)[0, -1] > 0.001:
            t2[0, 0] = 1
        else:
            t2[0, -1] = 1
    else:
        t2 = t
    t1 = torch.cross(n, t2, dim=1)
    return t1, t2


def _vec3(t: torch.Tensor):
    v = t.detach().reshape(-1)[:3].to(torch.float32).cpu().tolist()
    return (C.c_float * 3)(*v)


def _single_plane(*vecs) -> None:
    """The fused sensor operators take ONE plane / frame per call (the reference calls them once per view with (1, 3)
    tensors, core/image_opt.py:99-119).  A (N, 3) tensor whose rows differ -- per-ray planes -- would silently use row 0:
    refuse it (``trace_rays_to_plane`` is the operator that takes per-ray planes)."""
    from .drrt import _capturing
    if _capturing():               # the row comparison reads a device value on the host: not capturable; a captured call
        return                     # has been checked when it ran eagerly before the capture
    for t in vecs:
        if isinstance(t, torch.Tensor) and t.dim() == 2 and t.shape[0] > 1 and not bool((t == t[:1]).all()):
            raise RuntimeError("this sensor operator takes one plane / frame per call (rows differ); split the rays by "
                               "view as the reference does (core/image_opt.py:99-119)")


def _frame(dev, p, n, t1, t2):
    """The sensor frame of a call -> (frame12, host): when every given vector is a tensor on `dev` (the reference keeps
    its planes there, core/image_opt.py:88-119) they are packed into ONE 12-float device tensor (p, n, t1, t2; zeros for
    p / n of the far field) for the *_dframe entries -- no copy back to the host, hence no device sync per call;
    otherwise host float[3] arrays for the plain entries."""
    given = [t for t in (p, n, t1, t2) if t is not None]
    if all(isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev for t in given):
        zero = None
        parts = []
        for t in (p, n, t1, t2):
            if t is None:
                zero = torch.zeros(3, dtype=torch.float32, device=dev) if zero is None else zero
                parts.append(zero)
            else:
                parts.append(t.detach().reshape(-1)[:3].to(torch.float32))
        return torch.cat(parts).contiguous(), None
    return None, tuple(_vec3(t) for t in given)


class _SensorSplat(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, v, e, p, n, t1, t2, res, span):
        if not x.is_cuda:
            raise RuntimeError("generate_sensor expects tensors on the cuda (ROCm) device (no CPU path)")
        dev = x.device
        with torch.cuda.device(dev):
            x_ = x.detach().to(torch.float32).contiguous()
            v_ = v.detach().to(torch.float32).contiguous()
            nr = x_.shape[0]
            if isinstance(e, torch.Tensor) and e.numel() > 1:
                e_ = e.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
                if e_.numel() != nr:
                    raise RuntimeError("e must be a scalar or have one entry per ray")
                e_s = 0.0
            else:
                e_, e_s = None, float(e)
            img = torch.empty(int(res), int(res), dtype=torch.float32, device=dev)
            fdev, frame = _frame(dev, p, n, t1, t2)
            head = (nr, C.c_void_p(x_.data_ptr()), C.c_void_p(v_.data_ptr()), C.c_void_p(0 if e_ is None else e_.data_ptr()), e_s)
            tail = (int(res), float(span), C.c_void_p(img.data_ptr()), 0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if fdev is not None:
                _lib.check(_lib.load().drrt_sensor_splat_dframe_f32(*head, C.c_void_p(fdev.data_ptr()), *tail))
            else:
                _lib.check(_lib.load().drrt_sensor_splat_f32(*head, *frame, *tail))
        ctx.saved = (x_, v_, e_, e_s, fdev, frame, int(res), float(span))
        return img

    @staticmethod
    def backward(ctx, grad_img):
        x_, v_, e_, e_s, fdev, frame, res, span = ctx.saved
        dev = x_.device
        with torch.cuda.device(dev):
            g = grad_img.detach().to(torch.float32).contiguous()
            gx, gv = torch.empty_like(x_), torch.empty_like(v_)
            head = (x_.shape[0], C.c_void_p(x_.data_ptr()), C.c_void_p(v_.data_ptr()), C.c_void_p(0 if e_ is None else e_.data_ptr()), e_s)
            tail = (res, span, C.c_void_p(g.data_ptr()), C.c_void_p(gx.data_ptr()), C.c_void_p(gv.data_ptr()),
                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if fdev is not None:
                _lib.check(_lib.load().drrt_sensor_splat_dframe_bwd_f32(*head, C.c_void_p(fdev.data_ptr()), *tail))
            else:
                _lib.check(_lib.load().drrt_sensor_splat_bwd_f32(*head, *frame, *tail))
        return gx, gv, None, None, None, None, None, None, None


def generate_sensor(rays, e, plane, res, span, tangent=None):
    """core/sensor.py:5-28: image (res, res) of the rays splatted onto the sensor plane
    (p, n given as (1,3) tensors, one plane per call as in the reference); differentiable w.r.t.
    the rays."""
    x, v = rays
    p, n = plane
    t1, t2 = get_tan_vecs(n, tangent)
    return _SensorSplat.apply(x, v, e, p, n, t1, t2, res, span)


class _FarSensorSplat(torch.autograd.Function):

    @staticmethod
    def forward(ctx, v, e, t1, t2, res, ang_cut):
        if not v.is_cuda:
            raise RuntimeError("generate_inf_sensor expects tensors on the cuda (ROCm) device (no CPU path)")
        dev = v.device
        with torch.cuda.device(dev):
            v_ = v.detach().to(torch.float32).contiguous()
            nr = v_.shape[0]
            if isinstance(e, torch.Tensor) and e.numel() > 1:
                e_ = e.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
                if e_.numel() != nr:
                    raise RuntimeError("e must be a scalar or have one entry per ray")
                e_s = 0.0
            else:
                e_, e_s = None, float(e)
            img = torch.empty(int(res), int(res), dtype=torch.float32, device=dev)
            fdev, frame = _frame(dev, None, None, t1, t2)
            head = (nr, C.c_void_p(v_.data_ptr()), C.c_void_p(0 if e_ is None else e_.data_ptr()), e_s)
            tail = (int(res), float(ang_cut), C.c_void_p(img.data_ptr()), 0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if fdev is not None:
                _lib.check(_lib.load().drrt_sensor_far_splat_dframe_f32(*head, C.c_void_p(fdev.data_ptr()), *tail))
            else:
                _lib.check(_lib.load().drrt_sensor_far_splat_f32(*head, *frame, *tail))
        ctx.saved = (v_, e_, e_s, fdev, frame, int(res), float(ang_cut))
        return img

    @staticmethod
    def backward(ctx, grad_img):
        v_, e_, e_s, fdev, frame, res, ang_cut = ctx.saved
        dev = v_.device
        with torch.cuda.device(dev):
            g = grad_img.detach().to(torch.float32).contiguous()
            gx, gv = torch.empty_like(v_), torch.empty_like(v_)
            head = (v_.shape[0], C.c_void_p(v_.data_ptr()), C.c_void_p(0 if e_ is None else e_.data_ptr()), e_s)
            tail = (res, ang_cut, C.c_void_p(g.data_ptr()), C.c_void_p(gx.data_ptr()), C.c_void_p(gv.data_ptr()),
                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if fdev is not None:
                _lib.check(_lib.load().drrt_sensor_far_splat_dframe_bwd_f32(*head, C.c_void_p(fdev.data_ptr()), *tail))
            else:
                _lib.check(_lib.load().drrt_sensor_far_splat_bwd_f32(*head, *frame, *tail))
        return gv, None, None, None, None, None


def generate_inf_sensor(rays, e, plane, res, angle_span=120, tangent=None):
    """core/sensor.py:31-53: far-field image (res, res) -- the normalised ray directions splatted in the sensor
    frame over [-ang_cut, ang_cut]^2, ang_cut = sin(angle_span / 2); differentiable w.r.t. the directions (the
    positions do not enter, as in the reference)."""
    x, v = rays
    p, n = plane
    ang_cut = float(torch.sin(0.5 * torch.deg2rad(torch.tensor(float(angle_span), dtype=torch.float32))))   # :38
    t1, t2 = get_tan_vecs(n, tangent)
    return _FarSensorSplat.apply(v, e, t1, t2, res, ang_cut)


class _TexGet(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, v, tex, p, n, t1, t2, span, mode):
        if not x.is_cuda:
            raise RuntimeError("get_sdf_vals_* expect tensors on the cuda (ROCm) device (no CPU path)")
        if isinstance(tex, torch.Tensor) and tex.requires_grad:
            # the reference's Grid.Get path is differentiable w.r.t. the texture; this fused operator only returns the
            # gradients of the rays -- say so instead of handing back a silent zero
            raise RuntimeError("get_sdf_vals_*: the fused operator is not differentiable w.r.t. the texture; pass "
                               "d_tex.detach() (the experiments optimise the volume, not the measured texture)")
        _single_plane(p, n)            # (t1, t2 of get_tan_vecs are set in row 0 only, as in the reference)
        dev = x.device
        with torch.cuda.device(dev):
            x_ = x.detach().to(torch.float32).contiguous()
            v_ = v.detach().to(torch.float32).contiguous()
            tex_ = tex.detach().to(device=dev, dtype=torch.float32).contiguous()
            if tex_.dim() != 2 or tex_.shape[0] != tex_.shape[1]:
                raise RuntimeError("the texture must be square (the reference clips both axes with res[0])")
            fdev, frame = _frame(dev, p, n, t1, t2)
            f = torch.empty(x_.shape[0], dtype=torch.float32, device=dev)
            head = (x_.shape[0], C.c_void_p(x_.data_ptr()), C.c_void_p(v_.data_ptr()))
            tail = (C.c_void_p(tex_.data_ptr()), int(tex_.shape[0]), float(span), int(mode), C.c_void_p(f.data_ptr()),
                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if fdev is not None:
                _lib.check(_lib.load().drrt_sensor_tex_get_dframe_f32(*head, C.c_void_p(fdev.data_ptr()), *tail))
            else:
                _lib.check(_lib.load().drrt_sensor_tex_get_f32(*head, *frame, *tail))
        ctx.saved = (x_, v_, tex_, fdev, frame, float(span), int(mode))
        return f

    @staticmethod
    def backward(ctx, grad_f):
        x_, v_, tex_, fdev, frame, span, mode = ctx.saved
        dev = x_.device
        with torch.cuda.device(dev):
            g = grad_f.detach().to(torch.float32).contiguous()
            gx, gv = torch.empty_like(x_), torch.empty_like(v_)
            head = (x_.shape[0], C.c_void_p(x_.data_ptr()), C.c_void_p(v_.data_ptr()))
            tail = (C.c_void_p(tex_.data_ptr()), int(tex_.shape[0]), span, mode, C.c_void_p(g.data_ptr()),
                    C.c_void_p(gx.data_ptr()), C.c_void_p(gv.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if fdev is not None:
                _lib.check(_lib.load().drrt_sensor_tex_get_dframe_bwd_f32(*head, C.c_void_p(fdev.data_ptr()), *tail))
            else:
                _lib.check(_lib.load().drrt_sensor_tex_get_bwd_f32(*head, *frame, *tail))
        return gx, gv, None, None, None, None, None, None, None


def get_sdf_vals_near(rays, d_tex, plane, span, tangent=None):
    """core/sensor.py:102-119: the (res, res) texture ``d_tex`` (e.g. the signed distance to a measured caustic)
    sampled at the points where the rays meet the sensor plane (Grid.Get's radial-tent interpolant); differentiable
    w.r.t. the rays.  One fused HIP kernel each way (plane intersection, sensor frame, 16 taps)."""
    x, v = rays
    p, n = plane
    t1, t2 = get_tan_vecs(n, tangent)
    return _TexGet.apply(x, v, d_tex, p, n, t1, t2, span, 0)


def get_sdf_vals_far(rays, d_tex, plane, ang_span, tangent=None):
    """core/sensor.py:122-138: the texture sampled at the rays' directions projected on the sensor frame,
    ``v . T + ang_cut`` with ``ang_cut = sin(ang_span / 2)`` (the direction as it is, not normalised -- as written)."""
    x, v = rays
    p, n = plane
    ang_cut = float(torch.sin(0.5 * torch.deg2rad(torch.tensor(float(ang_span), dtype=torch.float32))))
    t1, t2 = get_tan_vecs(n, tangent)
    return _TexGet.apply(x, v, d_tex, p, n, t1, t2, 2.0 * ang_cut, 1)

