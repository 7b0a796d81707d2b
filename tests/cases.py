"""Deterministic synthetic scenes and ray sets shared by the parity tests (numpy only).

Scene shapes follow the reference's workloads: Luneburg ball n(r)=sqrt(2-r^2)
(/root/reference/core/fiber_opt.py:165-166), smooth tomography-like field in [1, 1.0003]
(value range of data/fuel_injection_64.npy), uniform medium (src/test.cpp:117-146)."""
import numpy as np


def luneburg(R, span=1.0):
    g = np.linspace(0.0, span, R)
    Z, Y, X = np.meshgrid(g, g, g, indexing="ij")          # torch (D,H,W) order: rif[z,y,x]
    r = np.sqrt((X - span / 2) ** 2 + (Y - span / 2) ** 2 + (Z - span / 2) ** 2) / (span / 2)
    return np.sqrt(2.0 - np.minimum(r, 1.0) ** 2).astype(np.float32)


def smooth_field(R, seed=0, amp=0.2):
    """Band-limited random field 1 + amp*U, asymmetric in x,y,z (catches axis-order bugs)."""
    rng = np.random.default_rng(seed)
    g = np.linspace(0.0, 1.0, R)
    Z, Y, X = np.meshgrid(g, g, g, indexing="ij")
    f = np.zeros((R, R, R))
    for _ in range(6):
        k = rng.uniform(0.5, 3.0, 3) * np.pi
        ph = rng.uniform(0, 2 * np.pi, 3)
        f += rng.uniform(0.3, 1.0) * np.sin(k[0] * X + ph[0]) * np.sin(k[1] * Y + ph[1]) * np.sin(k[2] * Z + ph[2])
    f = (f - f.min()) / (f.max() - f.min())
    return (1.0 + amp * f).astype(np.float32)


def sphere_sdf(R, span=1.0, radius=0.4):
    g = np.linspace(0.0, span, R)
    Z, Y, X = np.meshgrid(g, g, g, indexing="ij")
    return (np.sqrt((X - span / 2) ** 2 + (Y - span / 2) ** 2 + (Z - span / 2) ** 2) - radius).astype(np.float32)


def plane_rays(n, span, ds, seed=0, axis=1, offset=-0.3, tilt=0.05, lo=0.05, hi=0.95):
    """Jittered plane source just OUTSIDE the face `axis`=0 (offset*ds before it), travelling along
    +axis with a small tilt.  Starting off-face by a fraction of a step keeps the adjoint's
    termination test (escaped(x,-v), src/tracer.cpp:425) away from floating-point ties."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(lo * span, hi * span, (n, 3))
    pos[:, axis] = offset * ds
    vel = rng.normal(0, tilt, (n, 3))
    vel[:, axis] = 1.0
    vel /= np.linalg.norm(vel, axis=1, keepdims=True)
    return pos.astype(np.float32), vel.astype(np.float32)


def cube_rays(n_per_face, span, ds, seed=0, tilt=0.1):
    """Six views (one per face), like source.rand_rays_cube (/root/reference/core/source.py:398-412)."""
    ps, vs = [], []
    for f in range(6):
        axis, sign = f // 2, 1 - 2 * (f % 2)
        p, v = plane_rays(n_per_face, span, ds, seed=seed + f, axis=axis, tilt=tilt)
        if sign < 0:
            p[:, axis] = span - p[:, axis]
            v[:, axis] = -v[:, axis]
        ps.append(p); vs.append(v)
    return np.concatenate(ps), np.concatenate(vs)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def step_flip_report(xt, vt, xt_ref, vt_ref, ds, tol):
    """Fraction of rays whose exit sample differs from the reference by more than `tol` and
    how many of those are explained by exiting one march step earlier/later (xt differs by
    ~ +-ds*v): the boundary test `p < (res-1)*h` is discontinuous, so an ulp of difference in
    the trajectory legitimately flips the exit step."""
    d = np.linalg.norm(xt.astype(np.float64) - xt_ref.astype(np.float64), axis=1)
    bad = d > tol
    if not bad.any():
        return 0.0, 0.0
    step = np.linalg.norm(ds * vt_ref[bad].astype(np.float64), axis=1)
    explained = np.abs(d[bad] - step) < 10 * tol + 1e-3 * step
    return float(bad.mean()), float((bad.sum() - explained.sum()) / len(d))


def fuzz_config(seed):
    """Seeded nasty configuration for the differential fuzz tests (tests/test_gpu_fuzz.py on the GPU,
    tests/test_hostcheck.py::test_fuzz_hostcheck on the host build of the per-ray code)."""
    rng = np.random.default_rng(1000 + seed)
    W, H, D = (int(v) for v in rng.integers(2, 25, 3))
    if seed % 7 == 0:
        D = 2
    h = float(np.float32(rng.uniform(0.05, 2.0)))
    ds = float(np.float32(h * rng.uniform(0.2, 1.7)))
    if seed % 3 == 0:
        rif = 1.0 + 0.5 * rng.random((D, H, W))
    else:
        z, y, x = np.meshgrid(np.linspace(0, 1, D), np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
        rif = 1.0 + 0.3 * np.sin(3 * x + 1) * np.cos(2 * y) + 0.2 * z * z
    rif = rif.astype(np.float32)
    sdf = (rng.random((D, H, W)) - 0.6).astype(np.float32)
    n = 600
    ext = np.array([(W - 1) * h, (H - 1) * h, (D - 1) * h])
    pos = rng.uniform(-0.15, 1.15, (n, 3)) * ext
    face = rng.integers(0, 6, n)
    onface = rng.random(n) < 0.3                       # exactly on a face
    for i in np.nonzero(onface)[0]:
        a = face[i] // 2
        pos[i, a] = 0.0 if face[i] % 2 == 0 else ext[a]
    vel = rng.normal(size=(n, 3))
    vel /= np.linalg.norm(vel, axis=1, keepdims=True)
    vel *= rng.uniform(0.5, 1.5, (n, 1))
    vel[rng.random(n) < 0.02] = 0.0                    # rays that never move
    return dict(res=(W, H, D), h=h, ds=ds, rif=rif, sdf=sdf, pos=pos.astype(np.float32), vel=vel.astype(np.float32),
                dx=rng.normal(size=(n, 3)).astype(np.float32), dv=rng.normal(size=(n, 3)).astype(np.float32),
                po=(rng.uniform(0.2, 0.8, (n, 3)) * ext).astype(np.float32),
                pd=np.tile(np.array([[0.3, 0.9, 0.1]], np.float32), (n, 1)),
                tg=(rng.uniform(0.0, 1.0, (n, 3)) * ext).astype(np.float32))


def fuzz_cable_config(seed):
    """Seeded nasty configuration for the cable (radial profile) variants: random profile length / radius /
    cable length, rays starting inside, outside and on the axis of the cylinder, before and after its ends,
    with arbitrary directions; steps from a fraction of a radial sample to several."""
    rng = np.random.default_rng(2000 + seed)
    rres = int(rng.integers(2, 200))
    radius = float(np.float32(rng.uniform(0.1, 3.0)))
    length = float(np.float32(rng.uniform(0.5, 8.0) * radius))
    ds = float(np.float32(radius / max(rres - 1, 1) * rng.uniform(0.3, 3.0)))
    prof = (1.0 + 0.5 * rng.random(rres)).astype(np.float32) if seed % 2 else \
        np.sqrt(2.0 - np.linspace(0, 1, rres) ** 2).astype(np.float32)
    n = 400
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = radius * rng.uniform(0, 1.2, n)
    pos = np.stack([radius + rad * np.cos(ang), rng.uniform(-0.1, 1.1, n) * length, radius + rad * np.sin(ang)], -1)
    pos[rng.random(n) < 0.05, 0] = radius
    pos[rng.random(n) < 0.05, 2] = radius
    pos[:3] = [[radius, 0.0, radius], [radius, 0.3 * ds, radius], [2 * radius, 0.5 * length, radius]]
    vel = rng.normal(0, 0.4, (n, 3)); vel[:, 1] = rng.choice([1.0, 1.0, 1.0, -1.0], n)
    vel /= np.linalg.norm(vel, axis=1, keepdims=True)
    vel[1] = [0, 1, 0]
    tg = np.stack([radius + rng.normal(0, 0.3 * radius, n), rng.uniform(0, 1.2, n) * length,
                   radius + rng.normal(0, 0.3 * radius, n)], -1)
    return dict(prof=prof, radius=radius, length=length, ds=ds, pos=pos.astype(np.float32), vel=vel.astype(np.float32),
                tg=tg.astype(np.float32), dx=rng.normal(size=(n, 3)).astype(np.float32),
                dv=rng.normal(size=(n, 3)).astype(np.float32))


def grads_agree(a, b, tol=2e-5):
    """Adjoint grids equal up to summation order; non-finite entries (the adjoint recurrences grow exponentially
    on long marches of arbitrary rays) must sit in the same places."""
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return False
    if not fa.any():
        return True
    scale = np.abs(b[fb]).max()
    return scale < 1e-20 or float(np.linalg.norm(a[fa] - b[fb]) / max(np.linalg.norm(b[fb]), 1e-300)) < tol
