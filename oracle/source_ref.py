"""CPU restatement of the reference's plane-source ray generators -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/core/source.py:54-69 (plane_source3_rand), :275-293 (rotate_pts_to_source),
:303-312 (rotate_ray3), :352-357 (rand_rays_in_sphere), :398-412 (rand_rays_cube, plane source) and
:555-563 (random_rotate_ic), in numpy float32 with the reference's operation order.  The uniform
draws are an INPUT (`u`, shape (V, 2*spp, P0, P1)); the reference takes them from torch's host
generator.  Pinned by tests/golden/source_rays.npz, produced by RUNNING the reference's own functions
on the same draws in the build container.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def view_matrix(angle, vert=False):
    """rotate_ray3's R (source.py:304-311); `angle` keeps its dtype through radians/cos/sin."""
    theta = np.radians(angle)
    c, s = np.cos(theta), np.sin(theta)
    c, s = np.asarray(c).reshape(-1)[0], np.asarray(s).reshape(-1)[0]
    if vert:
        rn = np.array(((1, 0, 0), (0, c, -s), (0, s, c))).astype(float)
    else:
        rn = np.array(((c, -s, 0), (s, c, 0), (0, 0, 1))).astype(float)
    return rn.astype(F)


def _rot(x, R):
    """x @ R.T accumulated left to right in float32."""
    out = np.empty_like(x)
    for k in range(3):
        out[:, k] = (x[:, 0] * R[k, 0] + x[:, 1] * R[k, 1]) + x[:, 2] * R[k, 2]
    return out


def plane_view(u, R, pixels, spp, width, circle=False, sensor_dist=1.0, independent=False):
    """One view: u (2*spp, P0, P1) uniforms -> x (n,3), v (n,3), planes (n,3,3), float32."""
    p0, p1 = int(pixels[0]), int(pixels[1])
    u = np.asarray(u, F).reshape(2 * spp, p0, p1)
    w, hw = F(width), F(width / 2)
    off = u * w                                                           # :56
    if independent:                                                       # :61-63
        px = off[:spp] - hw
        pz = off[spp:] - hw
    else:                                                                 # :57-58, 65-68
        r0 = w * (np.arange(p0).astype(F) / F(p0) - F(0.5))
        r1 = w * (np.arange(p1).astype(F) / F(p1) - F(0.5))
        px = r0[None, :, None] + off[:spp] / F(p0)
        pz = r1[None, None, :] + off[spp:] / F(p1)
    pts = np.stack([px.reshape(-1), np.zeros(px.size, F), pz.reshape(-1)], axis=-1).astype(F)
    if circle:                                                            # :277-280
        r = np.sqrt(pts[:, 0] * pts[:, 0] + pts[:, 2] * pts[:, 2])
        pts = pts[r < hw]
    n = len(pts)
    vdir, tdir = R[:, 1].copy(), R[:, 2].copy()                           # R e_y, R e_z
    x = _rot(pts, R) + hw                                                 # :284
    x = x - (w * vdir)[None, :] / F(2)                                    # :287
    plane_x = F(sensor_dist + width / 2) * vdir + hw                      # :290
    planes = np.broadcast_to(np.stack([plane_x, vdir, tdir])[None], (n, 3, 3)).astype(F).copy()
    v = np.broadcast_to(vdir[None], (n, 3)).astype(F).copy()
    return x.astype(F), v, planes


def point_view(u, R, pixels, spp, width, circle=False, sensor_dist=1.0):
    """One view of point_source3_rand (:72-104): u (2*spp, P0, P1) uniforms -> x, v, planes (float32)."""
    p0, p1 = int(pixels[0]), int(pixels[1])
    u = np.asarray(u, F).reshape(2 * spp, p0, p1)
    w, hw = F(width), F(width / 2)
    off = u - F(0.5)                                                       # :73
    r0 = w * ((np.arange(p0).astype(F) + F(0.5)) / F(p0) - F(0.5))         # :75
    r1 = w * ((np.arange(p1).astype(F) + F(0.5)) / F(p1) - F(0.5))
    px = (r0[None, :, None] + off[:spp]).reshape(-1)
    pz = (r1[None, None, :] + off[spp:]).reshape(-1)
    if circle:                                                             # :81-83, 91-92
        keep = np.sqrt(px * px + pz * pz) < hw
        px, pz = px[keep], pz[keep]
    n = len(px)
    nrm = np.sqrt((px * px + w * w) + pz * pz)                             # :88-89
    vel = np.stack([px / nrm, np.full(n, w, F) / nrm, pz / nrm], axis=-1).astype(F)
    vdir, tdir = R[:, 1].copy(), R[:, 2].copy()
    pos = np.tile(np.array([[0.0, -hw, 0.0]], F), (n, 1))
    x = _rot(pos, R) + hw                                                  # :94-95
    v = _rot(vel, R)                                                       # :96
    plane_x = (F(sensor_dist * width) * vdir) / F(2) + hw                  # :102
    planes = np.broadcast_to(np.stack([plane_x, vdir, tdir])[None], (n, 3, 3)).astype(F).copy()
    return x.astype(F), v.astype(F), planes


def cone_cos(cone_angle):
    """cos(cone_angle / 2) as hatbox_sample evaluates it (:533-534): float32 throughout."""
    import torch
    return float(torch.cos(torch.deg2rad(torch.tensor(float(cone_angle))) / 2))


def cone_view(u, R, pixels, spp, width, sensor_dist=1.0, cone_angle=100.0):
    """One view of cone_source3_rand (:186-203) with hatbox_sample (:531-545): u (2, N) = its two uniform draws
    (z, then theta), N = spp*P0*P1 -> x, v, planes (float32)."""
    n = int(pixels[0]) * int(pixels[1]) * int(spp)
    u = np.asarray(u, F).reshape(2, n)
    w, hw = F(width), F(width / 2)
    dist = F(cone_cos(cone_angle))
    z = u[0] * (F(1) - dist) + dist                                        # :535
    theta = F(2 * np.pi) * u[1]                                            # :536
    scale = np.sqrt(F(1) - z * z)                                          # :537
    cx, cy = np.cos(theta) * scale, np.sin(theta) * scale                  # :539-540
    # t1 = e_z x e_y = -e_x, t2 = t1 x e_y = -e_z (:542-543): vel = cx*t1 + cy*t2 + z*e_y
    vel = np.stack([-cx, z, -cy], axis=-1).astype(F)
    vdir, tdir = R[:, 1].copy(), R[:, 2].copy()
    pos = np.tile(np.array([[0.0, -hw, 0.0]], F), (n, 1))
    x = _rot(pos, R) + hw                                                  # :191
    v = _rot(vel, R)                                                       # :192
    plane_x = F(sensor_dist + width / 2) * vdir + hw                       # :198
    planes = np.broadcast_to(np.stack([plane_x, vdir, tdir])[None], (n, 3, 3)).astype(F).copy()
    return x.astype(F), v.astype(F), planes


def views(u, mats, pixels, spp, width, circle=False, sensor_dist=1.0, independent=False, kind="plane", cone_angle=100.0):
    """Concatenation over views (:352-357 / :360-365 / :398-412): ((x, v, planes), nrays)."""
    if kind == "cone":
        parts = [cone_view(u[i], mats[i], pixels, spp, width, sensor_dist, cone_angle) for i in range(len(mats))]
    elif kind == "point":
        parts = [point_view(u[i], mats[i], pixels, spp, width, circle, sensor_dist) for i in range(len(mats))]
    else:
        parts = [plane_view(u[i], mats[i], pixels, spp, width, circle, sensor_dist, independent) for i in range(len(mats))]
    nrays = [len(p[0]) for p in parts]
    return tuple(np.concatenate(q) for q in zip(*parts)), nrays


def sphere_mats(nviews, angle_span=360, xaxis=False):
    import torch  # torch.linspace's fp32 values are part of the reference's behaviour (:353)
    angles = torch.linspace(0, angle_span, nviews + 1)
    return [view_matrix(angles[i].numpy(), xaxis) for i in range(nviews)]


def cube_mats():
    import torch
    angles = torch.linspace(0, 360, 5)
    return [view_matrix(angles[i].numpy(), False) for i in range(4)] + \
           [view_matrix(np.int64(a), True) for a in (90, -90)]


def rotate_ic(x, v, planes, span, M):
    """random_rotate_ic (:555-563) with a given matrix M (float64 -> float32 as in the reference)."""
    M = np.asarray(M, np.float64).astype(F)
    hs = F(span / 2)
    xn = _rot(x - hs, M) + hs
    vn = _rot(v, M)
    sp = _rot(planes[:, 0] - hs, M) + hs
    sn = _rot(planes[:, 1], M)
    st = _rot(planes[:, 2], M)
    return xn, vn, np.stack([sp, sn, st], axis=1)
