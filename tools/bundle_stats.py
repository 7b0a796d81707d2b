#!/usr/bin/env python3
"""Offline (numpy, no GPU) look at the 64-ray bundles a sort key produces: how large is the box of grid cells a bundle
occupies while it crosses the volume (straight chords: the shape question does not need the refraction)?

  python tools/bundle_stats.py [cube6|plane|shifted] [--rays N]

Key variants: chord6 = the 60-bit Morton interleave of the chord end points (drrt_sort.hip, rounds 1-2);
uvdir = 4-D Morton of the ray's offset from the box centre in ITS OWN transverse plane (u, v) and of its direction
(octahedral map); uvdir_h = the same with a Hilbert curve on (u, v) inside a direction cell."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rays_plane(n, shift=0.0, seed=0):
    side = int(round(n ** 0.5))
    rng = np.random.default_rng(seed)
    off = rng.random((2, side, side))
    i = np.arange(side)[:, None] + np.zeros((1, side)); j = np.arange(side)[None, :] + np.zeros((side, 1))
    x = (i + off[0]) / side + shift / side; z = (j + off[1]) / side + shift / side
    pos = np.stack([x.ravel(), np.zeros(n), z.ravel()], -1).clip(0, 1 - 1e-6); pos[:, 1] = 0
    vel = np.zeros((n, 3)); vel[:, 1] = 1
    return pos, vel


def rays_cube6(n, seed=0):
    from scipy.spatial.transform import Rotation
    from oracle import source_ref as S
    rot = Rotation.random(random_state=seed).as_matrix()
    P = int(np.ceil(np.sqrt(n * 1.004 / (6.0 * np.pi / 4.0))))
    rng = np.random.default_rng(seed)
    xs, vs = [], []
    mats = [S.view_matrix(np.float32(a), False) for a in (0., 90., 180., 270.)] + [S.view_matrix(np.float32(a), True) for a in (90., -90.)]
    for R in mats:
        u = rng.random((2, P, P)).astype(np.float32)
        x, v, _ = S.plane_view(u, R, (P, P), 1, 1.0, circle=True, sensor_dist=0.0)
        xs.append(x); vs.append(v)
    x = np.concatenate(xs).astype(np.float64); v = np.concatenate(vs).astype(np.float64)
    x = (x - 0.5) @ rot.T + 0.5; v = v @ rot.T
    return x[:n], v[:n]


def chords(p, d):
    """entry / exit parameters of the straight line through the unit box"""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t1 = (0.0 - p) * inv; t2 = (1.0 - p) * inv
    lo = np.where(np.abs(d) > 1e-20, np.minimum(t1, t2), -np.inf)
    hi = np.where(np.abs(d) > 1e-20, np.maximum(t1, t2), np.inf)
    t0 = np.maximum(lo.max(1), 0.0); t1_ = hi.min(1)
    hit = t1_ >= t0
    return np.where(hit, t0, 0.0), np.where(hit, t1_, 0.0), hit


def interleave(qs, bits):
    key = np.zeros(len(qs[0]), dtype=np.uint64)
    k = len(qs)
    for b in range(bits):
        for j, q in enumerate(qs):
            key |= ((q.astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(k * b + (k - 1 - j))
    return key


def key_chord6(p, d):
    t0, t1, _ = chords(p, d)
    e0 = np.clip(p + t0[:, None] * d, 0, 0.99999); e1 = np.clip(p + t1[:, None] * d, 0, 0.99999)
    q = [(e0[:, a] * 1024).astype(np.uint32) for a in range(3)] + [(e1[:, a] * 1024).astype(np.uint32) for a in range(3)]
    return interleave(q, 10)


def frame(d):
    """orthonormal (t1, t2) for each unit direction: the coordinate axis least aligned with d, Gram-Schmidt"""
    a = np.argmin(np.abs(d), axis=1)
    e = np.zeros_like(d); e[np.arange(len(d)), a] = 1.0
    t1 = e - (e * d).sum(1, keepdims=True) * d
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(d, t1)
    return t1, t2


def octa(d):
    s = np.abs(d).sum(1, keepdims=True)
    o = d[:, :2] / s
    neg = d[:, 2] < 0
    ox = np.where(neg, (1 - np.abs(o[:, 1])) * np.sign(o[:, 0] + 1e-30), o[:, 0])
    oy = np.where(neg, (1 - np.abs(o[:, 0])) * np.sign(o[:, 1] + 1e-30), o[:, 1])
    return ox * 0.5 + 0.5, oy * 0.5 + 0.5


def hilbert2(x, y, bits):
    """2-D Hilbert index (x, y in [0, 2^bits))"""
    x = x.astype(np.int64).copy(); y = y.astype(np.int64).copy()
    d = np.zeros_like(x)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64); ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - 1 - x, x); y = np.where(flip, s - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        x &= (s - 1); y &= (s - 1)          # keep the low bits only (equivalent to the classic rotate on the sub-square)
        s >>= 1
    return d.astype(np.uint64)


def key_uvdir(p, d, hilbert=False, dir_bits=6, pos_bits=11):
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    t1, t2 = frame(dn)
    c = p - 0.5
    u = (c * t1).sum(1) * 0.5 + 0.5; v = (c * t2).sum(1) * 0.5 + 0.5     # power-of-two scale: an axis-aligned pixel grid stays aligned with the key cells
    qu = np.clip(u * (1 << pos_bits), 0, (1 << pos_bits) - 1).astype(np.uint32)
    qv = np.clip(v * (1 << pos_bits), 0, (1 << pos_bits) - 1).astype(np.uint32)
    ox, oy = octa(dn)
    qa = np.clip(ox * (1 << dir_bits), 0, (1 << dir_bits) - 1).astype(np.uint32)
    qb = np.clip(oy * (1 << dir_bits), 0, (1 << dir_bits) - 1).astype(np.uint32)
    dirkey = interleave([qa, qb], dir_bits)
    poskey = hilbert2(qu, qv, pos_bits) if hilbert else interleave([qu, qv], pos_bits)
    return (dirkey << np.uint64(2 * pos_bits)) | poskey


def bundle_report(name, p, d, key, R=256, nsamp=12, win=9, mode="fwd"):
    """mode fwd: all rays of a bundle have made the same number of steps from their START (the wavefront of a plane
    source); adj: the same number of steps backwards from their EXIT point (how the adjoint march starts them)."""
    order = np.argsort(key, kind="stable")
    p, d = p[order], d[order]
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    t0, t1, hit = chords(p, dn)
    n = len(p) // 64 * 64
    ext = np.zeros((n // 64, 3))
    smax = float(np.max(np.where(hit, t1, 0)))
    for s_ in np.linspace(0.02, smax, nsamp * 4):
        s_ray = (np.full(len(p), s_) if mode == "fwd" else t1 - s_ * (t1 > 0))
        act = hit & (s_ray > t0) & (s_ray < t1)
        q = (p + s_ray[:, None] * dn)[:n] * (R - 1)
        c = np.floor(q).reshape(-1, 64, 3)
        ok = act[:n].reshape(-1, 64)
        big = 1e9
        lo = np.where(ok[..., None], c, big).min(1); hi = np.where(ok[..., None], c, -big).max(1)
        e = np.where(hi >= lo, hi - lo + 2, 0)          # slots per axis a window needs
        ext = np.maximum(ext, e)
    es = np.sort(ext, axis=1)
    mx = es[:, 2]
    vol = (ext[:, 0] + 1) * ext[:, 1] * ext[:, 2]
    pr = lambda a: " ".join(f"{np.percentile(a, q):5.1f}" for q in (50, 90, 99))
    print(f"{name:8s} {mode}: slots per axis (sorted) p50/p90/p99: min {pr(es[:, 0])} | mid {pr(es[:, 1])} | max {pr(mx)} | "
          f"> {win}: {100 * np.mean(mx > win):5.1f} % | box p50/p90: {np.percentile(vol, 50):6.0f} {np.percentile(vol, 90):6.0f} | "
          f"fits 1000: {100 * np.mean(vol <= 1000):5.1f} %")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("set", nargs="?", default="cube6")
    ap.add_argument("--rays", type=int, default=1 << 20)
    a = ap.parse_args()
    if a.set == "cube6":
        p, d = rays_cube6(a.rays)
    elif a.set == "shifted":
        p, d = rays_plane(a.rays, 1.0 / 3.0)
    else:
        p, d = rays_plane(a.rays)
    print(f"{a.set}: {len(p)} rays")
    for mode in ("fwd", "adj"):
        bundle_report("chord6", p, d, key_chord6(p, d), mode=mode)
        bundle_report("uvdir", p, d, key_uvdir(p, d), mode=mode)
        bundle_report("uvdir_h", p, d, key_uvdir(p, d, hilbert=True), mode=mode)
