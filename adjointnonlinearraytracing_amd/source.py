"""Counterpart of the ray generators of the reference's ``core/source.py`` that feed the march in the
3-D scripts: ``plane_source3_rand`` (``:54-69``, with ``rotate_pts_to_source`` ``:275-293`` and
``rotate_ray3`` ``:303-312``), ``point_source3_rand`` (``:72-104``), ``rand_rays_in_sphere`` (``:352-357``),
``rand_ptrays_in_sphere`` (``:360-365``), ``cone_source3_rand`` (``:186-203``, with ``hatbox_sample`` ``:531-545``),
``rand_ptcone_in_sphere`` (``:386-395``), ``rand_rays_cube`` (``:398-412``, plane and cone sources) and
``random_rotate_ic`` (``:555-563``).

Same names, argument order and return structure as the reference; the rays are produced ON the
device by ``csrc/drrt_source.hip`` (all views of a call in three launches, order-preserving disc
compaction included) instead of on the host followed by an upload.  Keyword-only extras:

* ``device``  -- where the rays are generated (default ``cuda``; there is no CPU path),
* ``offset``  -- the uniform [0,1) jitter draws, shape ``(nviews, 2*spp, P0, P1)`` (``(2*spp, P0, P1)``
  for a single view); default ``torch.rand`` on the device.  The reference draws them with the host
  generator (``:56``), so passing its draws reproduces its rays,
* ``rotmat``  -- a 3x3 matrix: fuses ``random_rotate_ic`` into the generation (saves a second pass
  over the 15 floats per ray).

The area sources of the image / focal-stack / fuel-injection experiments -- ``area_source3_rand_bias`` (``:107-150``),
``area_source3_cone`` (``:152-183``), ``rand_area_in_sphere`` (``:368-376``), ``rand_cone_in_sphere`` (``:379-385``) --
and ``sum_norm`` (``:415-420``) are the reference's element-wise torch expressions evaluated ON the device (they contain
no gather and no batched matmul, so there is nothing for a hand-written kernel to win); their keyword-only ``offset`` /
``tosense`` / ``hatbox`` arguments take the uniform draws, so the reference's draws reproduce its rays.

The deterministic point source ``point_source3`` (``:29-51``, also ``rand_rays_cube(src_type='point')``) is likewise a
handful of torch expressions on the chosen device; it draws nothing, so a run of the reference pins it outright
(``tests/golden/point_rays.npz``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _view_matrix(angle, vert=False) -> np.ndarray:
    """rotate_ray3's matrix (core/source.py:303-312), rounded to fp32 exactly as there: the angle keeps
    its dtype through np.radians / np.cos / np.sin (fp32 for a float32 tensor, fp64 for ints and python
    floats), the matrix passes through float64 and lands in the rays' dtype (fp32)."""
    if isinstance(angle, torch.Tensor):
        angle = angle.detach().cpu().numpy()
    theta = np.radians(angle)
    c, s = np.cos(theta), np.sin(theta)
    c, s = np.asarray(c).reshape(-1)[0], np.asarray(s).reshape(-1)[0]
    if vert:
        rn = np.array(((1, 0, 0), (0, c, -s), (0, s, c))).astype(float)
    else:
        rn = np.array(((c, -s, 0), (s, c, 0), (0, 0, 1))).astype(float)
    return rn.astype(np.float32)


def rotate_ray3(x, angle, vert=False):
    """core/source.py:303-312 (plain torch on the tensor's device)."""
    R = torch.from_numpy(_view_matrix(angle, vert)).to(device=x.device, dtype=x.dtype)
    return torch.matmul(x, R.T)


def _cone_cos(cone_angle) -> float:
    """cos(cone_angle / 2) evaluated as hatbox_sample does (core/source.py:533-534): a float32 tensor throughout."""
    return float(torch.cos(torch.deg2rad(torch.tensor(float(cone_angle))) / 2))


def _generate(view_mats, pixels, spp, width, circle, sensor_dist, independent, offset, device, rotmat, span, kind=0,
              cone_angle=100.0):
    """All views of one call through drrt_gen_rays_f32; returns (x, v, planes), nrays."""
    dev = torch.device("cuda" if device is None else device)
    if dev.type != "cuda":
        raise RuntimeError("ray generation runs on the cuda (ROCm) device only (no CPU path)")
    nv, p0, p1, spp = len(view_mats), int(pixels[0]), int(pixels[1]), int(spp)
    with torch.cuda.device(dev):
        if offset is None:
            u = torch.rand(nv, 2 * spp, p0, p1, device=dev, dtype=torch.float32)
        else:
            u = offset.detach().to(device=dev, dtype=torch.float32).reshape(nv, 2 * spp, p0, p1).contiguous()
        rots = torch.from_numpy(np.ascontiguousarray(np.stack(view_mats), dtype=np.float32)).to(dev)
        cap = nv * spp * p0 * p1
        x = torch.empty(cap, 3, dtype=torch.float32, device=dev)
        v = torch.empty(cap, 3, dtype=torch.float32, device=dev)
        planes = torch.empty(cap, 3, 3, dtype=torch.float32, device=dev)
        counts = torch.empty(nv + 1, dtype=torch.int32, device=dev)
        lib = _lib.load()
        ws_bytes = lib.drrt_gen_workspace_bytes(nv, spp, p0, p1)
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        if rotmat is None:
            ic = None
        else:
            m = rotmat.detach().cpu().numpy() if isinstance(rotmat, torch.Tensor) else np.asarray(rotmat)
            ic = (C.c_float * 9)(*np.asarray(m, dtype=np.float64).astype(np.float32).reshape(9).tolist())
        tail = (ic, float(span if span is not None else width),
                C.c_void_p(x.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(planes.data_ptr()),
                C.c_void_p(counts.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if kind == 2:
            _lib.check(lib.drrt_gen_cone_rays_f32(
                C.c_void_p(u.data_ptr()), C.c_void_p(rots.data_ptr()), nv, spp, p0, p1, float(width), float(sensor_dist),
                _cone_cos(cone_angle), *tail))
        else:
            _lib.check(lib.drrt_gen_rays_f32(
                int(kind), C.c_void_p(u.data_ptr()), C.c_void_p(rots.data_ptr()), nv, spp, p0, p1, float(width),
                float(sensor_dist), int(bool(circle)), int(bool(independent)), *tail))
        if kind == 2 or not circle:
            # no rejection (only the disc mask drops samples, csrc/drrt_source.hip): the counts are known on the host,
            # so the call returns without waiting for the device
            per = spp * p0 * p1
            return (x, v, planes), [per] * nv
        pre = counts.cpu().tolist()               # disc mask: the one host sync (the reference syncs on its boolean mask too)
    total = pre[-1]
    nrays = [pre[i + 1] - pre[i] for i in range(nv)]
    return (x[:total], v[:total], planes[:total]), nrays


def plane_source3_rand(angle, pixels, spp, width, circle=False, xaxis=False, sensor_dist=1.0, independent=False,
                       *, offset=None, device=None, rotmat=None, span=None):
    """core/source.py:54-69 -> (x, v, planes)."""
    iv, _ = _generate([_view_matrix(angle, xaxis)], pixels, spp, width, circle, sensor_dist, independent,
                      None if offset is None else offset[None], device, rotmat, span)
    return iv


def rand_rays_in_sphere(nviews, im_res, spp, width, angle_span=360, circle=False, xaxis=False, sensor_dist=1.0,
                        indep=False, *, offset=None, device=None, rotmat=None, span=None):
    """core/source.py:352-357 -> ((x, v, planes), nrays)."""
    angles = torch.linspace(0, angle_span, nviews + 1)
    mats = [_view_matrix(angles[i], xaxis) for i in range(nviews)]
    return _generate(mats, im_res, spp, width, circle, sensor_dist, indep, offset, device, rotmat, span)


def point_source3_rand(angle, pixels, spp, width, circle=False, xaxis=False, sensor_dist=1.0,
                       *, offset=None, device=None, rotmat=None, span=None):
    """core/source.py:72-104 -> (x, v, planes): rays from the point (0, -width/2, 0) (rotated) through jittered
    pixel centres."""
    iv, _ = _generate([_view_matrix(angle, xaxis)], pixels, spp, width, circle, sensor_dist, False,
                      None if offset is None else offset[None], device, rotmat, span, kind=1)
    return iv


def rand_ptrays_in_sphere(nviews, im_res, spp, width, angle_span=360, circle=False, xaxis=False, sensor_dist=0.0,
                          *, offset=None, device=None, rotmat=None, span=None):
    """core/source.py:360-365 -> ((x, v, planes), nrays)."""
    angles = torch.linspace(0, angle_span, nviews + 1)
    mats = [_view_matrix(angles[i], xaxis) for i in range(nviews)]
    return _generate(mats, im_res, spp, width, circle, sensor_dist, False, offset, device, rotmat, span, kind=1)


def cone_source3_rand(angle, pixels, spp, width, circle=False, xaxis=False, sensor_dist=1.0, cone_angle=100.0,
                      *, offset=None, device=None, rotmat=None, span=None):
    """core/source.py:186-203 -> (x, v, planes): pixels[0]*pixels[1]*spp rays from the point (0, -width/2, 0)
    (rotated) with directions drawn by hatbox_sample (:531-545) in a cone of full angle ``cone_angle`` -- the source
    of the fibre experiment (core/fiber_opt.py:131).  ``circle`` is accepted and ignored, as in the reference.
    ``offset``: the two uniform draws of hatbox_sample, shape (2, N) (z first, then theta)."""
    iv, _ = _generate([_view_matrix(angle, xaxis)], pixels, spp, width, False, sensor_dist, False,
                      None if offset is None else offset.reshape(1, -1), device, rotmat, span, kind=2, cone_angle=cone_angle)
    return iv


def rand_ptcone_in_sphere(nviews, im_res, spp, width, angle_span=360, circle=False, xaxis=False, sensor_dist=1.0,
                          cone_angle=90.0, *, offset=None, device=None, rotmat=None, span=None):
    """core/source.py:386-395 -> ((x, v, planes), dists, nrays)."""
    angles = torch.linspace(0, angle_span, nviews + 1)
    mats = [_view_matrix(angles[i], xaxis) for i in range(nviews)]
    iv, nrays = _generate(mats, im_res, spp, width, False, sensor_dist, False, offset, device, rotmat, span, kind=2,
                          cone_angle=cone_angle)
    return iv, torch.zeros(nviews), nrays


def point_source3(angle, pixels, spp, width, cone_angle=90, xaxis=False, sensor_dist=0.0, circle=False, *, device=None):
    """core/source.py:29-51 -> (x, v, planes): the deterministic point source. One ray per node of a
    (pixels[0]*s) x (pixels[1]*s) lattice of (theta, phi) in [-cone_angle/2, cone_angle/2], s = max(floor(sqrt(spp)), 1),
    all leaving (0, -width/2, 0) rotated into the view; `circle` is accepted and unused, as there.
    Plain torch on `device` (default: cpu, like the reference) -- a few elementwise ops, not worth a kernel."""
    dev = torch.device("cpu" if device is None else device)
    half = float(np.radians(cone_angle / 2))
    s = max(int(np.floor(np.sqrt(spp))), 1)
    axes = [torch.linspace(-half, half, int(p) * s, device=dev) for p in pixels]
    theta, phi = torch.meshgrid(axes, indexing="ij")
    theta, phi = theta.flatten(), phi.flatten()
    vel = torch.stack([torch.cos(theta) * torch.sin(phi), torch.cos(theta) * torch.cos(phi), torch.sin(theta)], dim=-1)
    vel = vel / torch.norm(vel, dim=-1, keepdim=True)
    n = theta.shape[0]
    pos = torch.tensor([[0.0, -width / 2, 0.0]], device=dev).repeat(n, 1)
    x = rotate_ray3(pos, angle, vert=xaxis) + width / 2
    v = rotate_ray3(vel, angle, vert=xaxis)
    return x, v, _planes_of_view(n, angle, xaxis, width, sensor_dist, dev)


def rand_rays_cube(im_res, spp, width, circle=False, src_type='plane', cone_ang=90,
                   *, offset=None, device=None, rotmat=None, span=None):
    """core/source.py:398-412 -> ((x, v, planes), nrays): four views about z, two about x, sensor_dist = 0.
    src_type 'plane' (plane_source3_rand), 'point' (the deterministic point_source3 with cone_angle = cone_ang, :401) or
    anything else (cone_source3_rand with cone_angle = cone_ang, :402-404)."""
    angles = torch.linspace(0, 360, 5)
    vangles = torch.tensor([90, -90])
    if src_type == 'point':
        dev = torch.device("cuda" if device is None else device)
        views = [point_source3(angles[i], im_res, spp, width, cone_angle=cone_ang, xaxis=False, device=dev)
                 for i in range(len(angles) - 1)]
        views += [point_source3(va, im_res, spp, width, cone_angle=cone_ang, xaxis=True, device=dev) for va in vangles]
        nrays = [vw[0].shape[0] for vw in views]
        return tuple(map(torch.cat, zip(*views))), nrays
    mats = [_view_matrix(angles[i], False) for i in range(len(angles) - 1)]
    mats += [_view_matrix(va, True) for va in vangles]
    if src_type != 'plane':
        return _generate(mats, im_res, spp, width, False, 0.0, False, offset, device, rotmat, span, kind=2, cone_angle=cone_ang)
    return _generate(mats, im_res, spp, width, circle, 0.0, False, offset, device, rotmat, span)


def random_rotmat():
    """core/source.py:548-552."""
    from scipy.spatial.transform import Rotation as R
    return torch.from_numpy(R.random().as_matrix())


def random_rotate_ic(x, v, planes, span, rotmat=None):
    """core/source.py:555-563 on existing rays (plain torch on their device).  Prefer ``rotmat=`` of the
    generators above, which applies the same rotation while the rays are being written."""
    rotmat = (random_rotmat() if rotmat is None else torch.as_tensor(rotmat)).to(device=x.device, dtype=x.dtype)
    xn = torch.matmul(rotmat, x[..., None] - (span / 2)) + (span / 2)
    vn = torch.matmul(rotmat, v[..., None])
    sp = torch.matmul(rotmat, planes[:, 0, :, None] - (span / 2)) + (span / 2)
    sn = torch.matmul(rotmat, planes[:, 1, :, None])
    st = torch.matmul(rotmat, planes[:, 2, :, None])
    return xn.squeeze(-1), vn.squeeze(-1), torch.stack([sp.squeeze(-1), sn.squeeze(-1), st.squeeze(-1)], dim=1)


# ---- area sources and image normalisation: the reference's torch expressions, on the device -------------------------
def sum_norm(im, scale=False):
    """core/source.py:415-420: scale the image to unit mean."""
    scalar = torch.numel(im) / im.sum()
    if scale:
        return scalar * im, scalar
    return scalar * im


def _area_points(pixels, spp, width, circle, y_value, offset, dev):
    """Jittered pixel-centre samples of the source plane (:108-120 / :153-165): (N,3) points and the disc mask."""
    p0, p1, spp = int(pixels[0]), int(pixels[1]), int(spp)
    u = torch.rand(2 * spp, p0, p1, device=dev) if offset is None else \
        offset.detach().to(device=dev, dtype=torch.float32).reshape(2 * spp, p0, p1)
    off = (u - 0.5) * (width / p0)
    rng = [width * ((torch.arange(p, device=dev) + 0.5) / p - 0.5) for p in (p0, p1)]
    g0, g1 = torch.meshgrid(*rng, indexing="ij")
    pts = [g0 + off[:spp], torch.full((p0, p1, spp), float(y_value), device=dev), g1 + off[spp:]]
    pos = torch.stack([q.flatten() for q in pts], dim=-1)
    if circle:
        pos = pos[torch.norm(pos, dim=-1) < (width / 2)]
    return pos


def _planes_of_view(n, angle, xaxis, width, sensor_dist, dev):
    e_y = torch.tensor([0.0, 1.0, 0.0], device=dev).repeat(n, 1)
    e_z = torch.tensor([0.0, 0.0, 1.0], device=dev).repeat(n, 1)
    plane_v = rotate_ray3(e_y, angle, vert=xaxis)
    plane_t = rotate_ray3(e_z, angle, vert=xaxis)
    plane_x = (sensor_dist + width / 2) * plane_v + width / 2
    return torch.stack([plane_x, plane_v, plane_t], dim=1)


def area_source3_rand_bias(angle, pixels, spp, width, circle=False, xaxis=False, sensor_dist=1.0, *, offset=None,
                           tosense=None, device=None):
    """core/source.py:107-150 -> ((x, v, planes), xt, tpv): every source sample aims at a random point of the far face."""
    dev = torch.device("cuda" if device is None else device)
    pos = _area_points(pixels, spp, width, circle, 0.0, offset, dev)
    n = pos.shape[0]
    up = torch.tensor([[0.0, 1.0, 0.0]], device=dev)
    pt = -pos + (sensor_dist + width / 2) * up                                     # :122, :124
    pos = pos - (sensor_dist + width / 2) * up                                     # :123
    ts = torch.rand(2, n, device=dev) if tosense is None else tosense.detach().to(device=dev, dtype=torch.float32).reshape(2, n)
    ts = (ts - 0.5) * (1.0 * width)                                                # :126-127
    target = torch.stack([ts[0], width * torch.ones(n, device=dev) / 2, ts[1]], dim=-1)
    vel = target - pos
    vel = vel / torch.norm(vel, dim=-1, keepdim=True)
    tpv = sensor_dist / vel[..., 1]
    npos = pos + tpv[:, None] * vel
    xt = rotate_ray3(pt, angle, vert=xaxis) + width / 2
    x = rotate_ray3(npos, angle, vert=xaxis) + width / 2
    v = rotate_ray3(vel, angle, vert=xaxis)
    return (x, v, _planes_of_view(n, angle, xaxis, width, sensor_dist, dev)), xt, tpv


def hatbox_sample(v, angle, *, draws=None):
    """core/source.py:531-545: directions uniformly distributed in the cone of full angle ``angle`` around ``v``."""
    dev = v.device
    basis = torch.tensor([[0.0, 0.0, 1.0]], device=dev)
    dist = torch.cos(torch.deg2rad(torch.tensor(float(angle))) / 2).to(dev)
    u = torch.rand(2, v.shape[0], device=dev) if draws is None else draws.detach().to(device=dev, dtype=torch.float32).reshape(2, -1)
    z = u[0] * (1 - dist) + dist
    theta = 2 * np.pi * u[1]
    scale = torch.sqrt(1 - z ** 2)
    cx, cy = torch.cos(theta) * scale, torch.sin(theta) * scale
    t1 = torch.cross(basis.expand_as(v), v, dim=-1)
    t2 = torch.cross(t1, v, dim=-1)
    return cx[:, None] * t1 + cy[:, None] * t2 + z[:, None] * v


def area_source3_cone(angle, pixels, spp, width, circle=False, xaxis=False, sensor_dist=1.0, cone_angle=90, *,
                      offset=None, hatbox=None, device=None):
    """core/source.py:152-183 -> ((x, v, planes), tpv): an area source on the near face emitting into a cone."""
    dev = torch.device("cuda" if device is None else device)
    pos = _area_points(pixels, spp, width, circle, -width / 2, offset, dev)
    forward = torch.zeros_like(pos)
    forward[:, 1] = 1
    vel = hatbox_sample(forward, cone_angle, draws=hatbox)
    tpv = sensor_dist / vel[..., 1]
    x = rotate_ray3(pos, angle, vert=xaxis) + width / 2
    v = rotate_ray3(vel, angle, vert=xaxis)
    return (x, v, _planes_of_view(pos.shape[0], angle, xaxis, width, sensor_dist, dev)), tpv


def rand_area_in_sphere(nviews, im_res, spp, width, angle_span=360, circle=False, xaxis=False, sensor_dist=1.0, *,
                        device=None):
    """core/source.py:368-376 -> ((x, v, planes), targets, dists, nrays)."""
    angles = torch.linspace(0, angle_span, nviews + 1)
    view_list = [area_source3_rand_bias(angles[i], im_res, spp, width, circle=circle, xaxis=xaxis,
                                        sensor_dist=sensor_dist, device=device) for i in range(nviews)]
    views, targets, dists = zip(*view_list)
    nrays = [v[0].shape[0] for v in views]
    return tuple(map(torch.cat, zip(*views))), torch.cat(targets), torch.cat(dists), nrays


def rand_cone_in_sphere(nviews, im_res, spp, width, angle_span=360, circle=False, xaxis=False, sensor_dist=1.0,
                        cone_angle=90.0, *, device=None):
    """core/source.py:379-385 -> ((x, v, planes), dists, nrays)."""
    angles = torch.linspace(0, angle_span, nviews + 1)
    view_list = [area_source3_cone(angles[i], im_res, spp, width, circle=circle, xaxis=xaxis, sensor_dist=sensor_dist,
                                   cone_angle=cone_angle, device=device) for i in range(nviews)]
    views, dists = zip(*view_list)
    nrays = [v[0].shape[0] for v in views]
    return tuple(map(torch.cat, zip(*views))), torch.cat(dists), nrays

