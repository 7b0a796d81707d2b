"""CPU restatement of the reference's sensor image splat and its backward -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/core/sensor.py:5-28 (generate_sensor), :195-202 (trace_rays_to_plane),
:219-231 (get_tan_vecs) and core/grid.py:37-64 (Grid.index_values), :77-81 (rbf_tent),
:133-151 (Grid.Splat, average=False).  Pinned by tests/golden/sensor_splat.npz, which is produced by
RUNNING the reference's own torch code (and torch.autograd for the backward) in the build
container -- for this row the oracle is pinned by the reference itself.
"""
from __future__ import annotations

import numpy as np

SQRT2 = np.sqrt(2.0)


def tan_vecs(n, t=None):
    """core/sensor.py:219-231: t2 = given tangent (or a default axis), t1 = n x t2."""
    n = np.asarray(n, dtype=np.float64).reshape(3)
    if t is None:
        t2 = np.zeros(3)
        if abs(n[2]) > 0.001:
            t2[0] = 1.0
        else:
            t2[2] = 1.0
    else:
        t2 = np.asarray(t, dtype=np.float64).reshape(3)
    return np.cross(n, t2), t2


def _taps(xn, res, span):
    """Grid.index_values + rbf_tent for a 2-D grid: 16 taps per point."""
    hs = span / res
    u = xn / hs - 0.5                                   # grid.py:38
    i1 = np.floor(u).astype(np.int64)                   # :40
    offs = np.array([-1, 0, 1, 2])
    ia = (i1[:, 0, None] + offs[None, :])[:, :, None] + np.zeros((1, 1, 4), np.int64)   # (N,4,4) dim-0 index
    ib = (i1[:, 1, None] + offs[None, :])[:, None, :] + np.zeros((1, 4, 1), np.int64)   # (N,4,4) dim-1 index
    da = u[:, 0, None, None] - ia
    db = u[:, 1, None, None] - ib
    r = np.sqrt(da * da + db * db)                      # :55
    w = np.maximum(SQRT2 - r, 0.0)                      # :79
    valid = (ia >= 0) & (ia < res) & (ib >= 0) & (ib < res)   # :140
    return hs, ia, ib, da, db, r, w, valid


def generate_sensor(x, v, e, p, n, res, span, tangent=None, dtype=np.float64):
    """-> image (res,res).  e: scalar or (N,)."""
    x = np.asarray(x, dtype=dtype); v = np.asarray(v, dtype=dtype)
    p = np.asarray(p, dtype=dtype).reshape(3); n = np.asarray(n, dtype=dtype).reshape(3)
    t1, t2 = (a.astype(dtype) for a in tan_vecs(n, tangent))
    den = v @ n
    t = ((p - x) @ n) / den                             # sensor.py:199-200
    xp = x + t[:, None] * v
    fs = np.abs(den)                                    # :18-19
    q = xp - p
    xn = np.stack([q @ t1, q @ t2], -1) + span / 2      # :22-23
    hs, ia, ib, da, db, r, w, valid = _taps(xn, res, span)
    we = w / w.sum(axis=(1, 2), keepdims=True)          # grid.py:145 (normalised over ALL 16 taps)
    f = fs * np.asarray(e, dtype=dtype)
    img = np.zeros((res, res), dtype=dtype)
    np.add.at(img, (ia[valid], ib[valid]), (we * f[:, None, None])[valid])   # :150
    return img


def generate_sensor_backward(x, v, e, p, n, res, span, grad_img, tangent=None, dtype=np.float64):
    """Analytic gradient of sum(grad_img * image) w.r.t. (x, v) -- what torch.autograd produces
    through the reference's generate_sensor."""
    x = np.asarray(x, dtype=dtype); v = np.asarray(v, dtype=dtype)
    p = np.asarray(p, dtype=dtype).reshape(3); n = np.asarray(n, dtype=dtype).reshape(3)
    gI = np.asarray(grad_img, dtype=dtype)
    t1, t2 = (a.astype(dtype) for a in tan_vecs(n, tangent))
    den = v @ n
    t = ((p - x) @ n) / den
    xp = x + t[:, None] * v
    q = xp - p
    xn = np.stack([q @ t1, q @ t2], -1) + span / 2
    hs, ia, ib, da, db, r, w, valid = _taps(xn, res, span)
    W = w.sum(axis=(1, 2))
    F = np.abs(den) * np.asarray(e, dtype=dtype)
    g = np.where(valid, gI[np.clip(ia, 0, res - 1), np.clip(ib, 0, res - 1)], 0.0)
    G = (g * w).sum(axis=(1, 2)) / W
    live = (w > 0) & (r > 0)
    rs = np.where(r > 0, r, 1.0)
    dwa = np.where(live, -da / rs, 0.0)                 # d w / d u_a
    dwb = np.where(live, -db / rs, 0.0)
    ga = (F / W) * ((g * dwa).sum(axis=(1, 2)) - G * dwa.sum(axis=(1, 2))) / hs     # dL/d xn_a
    gb = (F / W) * ((g * dwb).sum(axis=(1, 2)) - G * dwb.sum(axis=(1, 2))) / hs
    gxp = ga[:, None] * t1[None, :] + gb[:, None] * t2[None, :]
    gx = gxp - n[None, :] * ((v * gxp).sum(-1) / den)[:, None]      # (I - v n^T/den)^T gxp
    gv = t[:, None] * gx + (G * np.asarray(e, dtype=dtype) * np.sign(den))[:, None] * n[None, :]
    return gx, gv


def _ang_cut(angle_span):
    return np.sin(0.5 * np.deg2rad(float(angle_span)))               # sensor.py:38


def generate_inf_sensor(v, e, n, res, angle_span=120.0, tangent=None, dtype=np.float64):
    """core/sensor.py:31-53 generate_inf_sensor -> image (res,res): the normalised directions in the sensor frame
    + ang_cut, Grid(zeros, 2*ang_cut/res).Splat(vn, e, average=False)."""
    v = np.asarray(v, dtype=dtype)
    t1, t2 = (a.astype(dtype) for a in tan_vecs(n, tangent))
    ac = _ang_cut(angle_span)
    vh = v / np.linalg.norm(v, axis=-1, keepdims=True)                 # :36
    vn = np.stack([vh @ t1, vh @ t2], -1) + ac                         # :46-47
    hs, ia, ib, da, db, r, w, valid = _taps(vn, res, 2 * ac)
    we = w / w.sum(axis=(1, 2), keepdims=True)
    f = np.asarray(e, dtype=dtype) * np.ones(len(v), dtype=dtype)      # :49
    img = np.zeros((res, res), dtype=dtype)
    np.add.at(img, (ia[valid], ib[valid]), (we * f[:, None, None])[valid])
    return img


def generate_inf_sensor_backward(v, e, n, res, grad_img, angle_span=120.0, tangent=None, dtype=np.float64):
    """Analytic gradient of sum(grad_img * image) w.r.t. v through generate_inf_sensor."""
    v = np.asarray(v, dtype=dtype)
    gI = np.asarray(grad_img, dtype=dtype)
    t1, t2 = (a.astype(dtype) for a in tan_vecs(n, tangent))
    ac = _ang_cut(angle_span)
    nv = np.linalg.norm(v, axis=-1, keepdims=True)
    vh = v / nv
    vn = np.stack([vh @ t1, vh @ t2], -1) + ac
    hs, ia, ib, da, db, r, w, valid = _taps(vn, res, 2 * ac)
    W = w.sum(axis=(1, 2))
    F = np.asarray(e, dtype=dtype) * np.ones(len(v), dtype=dtype)
    g = np.where(valid, gI[np.clip(ia, 0, res - 1), np.clip(ib, 0, res - 1)], 0.0)
    G = (g * w).sum(axis=(1, 2)) / W
    live = (w > 0) & (r > 0)
    rs = np.where(r > 0, r, 1.0)
    dwa = np.where(live, -da / rs, 0.0)
    dwb = np.where(live, -db / rs, 0.0)
    ga = (F / W) * ((g * dwa).sum(axis=(1, 2)) - G * dwa.sum(axis=(1, 2))) / hs
    gb = (F / W) * ((g * dwb).sum(axis=(1, 2)) - G * dwb.sum(axis=(1, 2))) / hs
    gp = ga[:, None] * t1[None, :] + gb[:, None] * t2[None, :]        # dL/d vhat
    return (gp - vh * (vh * gp).sum(-1, keepdims=True)) / nv          # (I - vh vh^T)/|v|
