#!/usr/bin/env python3
"""Generates the committed golden fixtures (tests/golden/*.npz).

Run ONCE in the build container, where /root/reference exists (it does not exist on the GPU
box; nothing at test time reads it).  The reference's native path cannot run (enoki missing), so
the fixtures are produced from the pieces of the reference that DO import here -- its pure-torch
helpers -- plus the CPU oracle:

  getlinear_grid.npz   core/grid.py  Grid.GetLinear   (:227-273) on the axis-permuted scene
                       -> pins volume::eval_grad values, gradients, clamping, axis order (Q2)
  getlinear_cable.npz  core/cable.py Cable.GetLinear  (:92-119)  -> pins cylinder_volume::eval_grad
  luneburg_cube.npz    rays from core/source.py rand_rays_cube + random_rotate_ic (:398-412,:555-563),
                       loss and (grad_x, grad_v) from core/sensor.py trace_rays_to_plane (:195-202)
                       with the Luneburg loss of core/luneburg_opt.py:93-102; exit rays and dL/dn
                       from the oracle (literal arithmetic, float64)
  ad_vs_adjoint.npz    torch.autograd (float64) through oracle/torch_ad.py vs the oracle adjoint
                       -> pins Tracer::backtrace as the exact discrete adjoint (h = 1 and h != 1, Q3)
  fuel_injection.npz   data/fuel_injection_64.npy (float64, F-order) cast to fp32 and padded to 65^3
                       as core/fuel_injection_opt.py:40-43 does; forward exit rays from the oracle

  splat_linear.npz     core/grid.py  Grid.SplatLinear (:275-315) RUN AS IS -> pins volume::splat (src/volume.cpp:182-244):
                       value weights, signed gradient weights, index pattern, clamping
  hessians.npz         torch.autograd (float64) through Grid.GetLinear / Cable.GetLinear RUN AS IS (Jacobian of their
                       gradient output) -> pins volume::eval_hess (:40-99) and cylinder_volume::eval_hess (:61-111)

  sensor_splat.npz     core/sensor.py generate_sensor (:5-28) and torch.autograd through it, RUN AS IS in
                       float64 -> pins the sensor image splat and its backward

  sensor_far.npz       core/sensor.py generate_inf_sensor (:31-53) and torch.autograd through it, RUN AS IS in float64
                       -> pins the far-field sensor (core/image_opt.py:116)

  cone_rays.npz        core/source.py cone_source3_rand (:186-203) and rand_rays_cube(src_type='cone') RUN AS IS on replayed
                       draws -> pins the device cone source (core/fiber_opt.py:131)

  upres.npz            core/optimizer.py upres_scene (:7-10) RUN AS IS -> pins the multires up-sampling

  source_rays.npz      core/source.py rand_rays_cube (:398-412), rand_rays_in_sphere (:352-357, circle and
                       independent variants) and random_rotate_ic (:555-563) RUN AS IS; the torch.rand draws
                       they consume are recorded by replaying the same host-generator seed
                       -> pins the device ray generation

  march_getlinear.npz  the march loop (src/tracer.cpp:66-87) in float64 with (n, grad n) of every sample taken from the
                       reference's OWN Grid.GetLinear (:227-273, axis-permuted scene) and torch.autograd of a random linear
                       functional of the exit rays with respect to the scene -> pins Tracer::trace's exit state and
                       Tracer::backtrace's dL/dn (h = 1 as written; h != 1 with the 1/h correction, Q3) with reference
                       code inside the loop body

Only DATA is stored (inputs and expected outputs); no reference source text.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/core")
sys.dont_write_bytecode = True      # the reference tree is read-only for us: importing from it must not drop .pyc files there

import cable as ref_cable      # noqa: E402
import grid as ref_grid        # noqa: E402
import sensor as ref_sensor    # noqa: E402
import source as ref_source    # noqa: E402

from oracle import oracle as O          # noqa: E402
from oracle import torch_ad as TA       # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB", {k: np.asarray(v).shape for k, v in arrs.items()})


def getlinear_grid():
    torch.manual_seed(0)
    R, h = 9, 0.25
    scene = torch.rand(R, R, R, dtype=torch.float64) + 1.0           # our layout: scene[z,y,x]
    G = ref_grid.Grid(scene.permute(2, 1, 0).contiguous(), h)        # reference layout: scene[x,y,z]
    pts = torch.rand(400, 3, dtype=torch.float64) * (R + 1) * h - h  # includes out-of-range points
    f, fx = G.GetLinear(pts)
    save("getlinear_grid.npz", scene=scene.numpy(), h=h, pts=pts.numpy(), f=f.numpy(), fx=fx.numpy())


def getlinear_cable():
    torch.manual_seed(1)
    rres, radius = 17, 2.0
    prof = torch.rand(rres, dtype=torch.float64) + 1.0
    Cb = ref_cable.Cable(prof, radius, 10.0)
    pts = torch.rand(300, 3, dtype=torch.float64) * 2.4 * radius - 0.2 * radius
    pts[0, 0] = radius; pts[0, 2] = radius                            # r = 0 -> zero gradient branch
    f, fx = Cb.GetLinear(pts)
    save("getlinear_cable.npz", prof=prof.numpy(), radius=radius, pts=pts.numpy(), f=f.numpy(), fx=fx.numpy())


def splat_linear():
    """core/grid.py Grid.SplatLinear (:275-315) RUN AS IS: scene += wp*f + h*dot(fx, wi) at the 8 corners.  That is
    volume::splat(p, val = f, grad = h*fx) (src/volume.cpp:182-244) on the axis-permuted scene for every point
    SplatLinear keeps (0 <= p/h < res per axis) -- including the band res-1 <= p/h < res where the +1 neighbour is
    clamped onto the same voxel.  Pins the 8 value weights, the sign pattern and index pattern of the 8 gradient
    weights, and the clamping of the adjoint's scatter."""
    torch.manual_seed(5)
    R, h = 7, 0.5
    n = 500
    pts = torch.rand(n, 3, dtype=torch.float64) * R * h              # 0 <= p/h < R: includes the clamped band
    pts[:40] = torch.rand(40, 3, dtype=torch.float64) * h + (R - 1) * h   # all three axes in the clamped band
    pts[40] = torch.tensor([0.0, 0.0, 0.0], dtype=torch.float64)     # exactly on a corner
    pts[41] = torch.tensor([1.0, 2.0, 1.5], dtype=torch.float64)     # exactly on voxel planes (w0 = 0)
    f = torch.randn(n, dtype=torch.float64)
    fx = torch.randn(n, 3, dtype=torch.float64)
    G = ref_grid.Grid(torch.zeros(R, R, R, dtype=torch.float64), h)  # reference layout: scene[x,y,z]
    G.weights = torch.zeros(R, R, R, dtype=torch.float64)
    G.SplatLinear(pts, f, fx)
    # stored in OUR layout scene[z,y,x]
    save("splat_linear.npz", R=R, h=h, pts=pts.numpy(), f=f.numpy(), fx=fx.numpy(),
         scene=G.scene.permute(2, 1, 0).contiguous().numpy(), weights=G.weights.permute(2, 1, 0).contiguous().numpy())


def hessians_by_autograd():
    """Second derivatives of the REFERENCE's own interpolants: torch.autograd (float64) through Grid.GetLinear
    (:227-273) and Cable.GetLinear (:92-119) RUN AS IS -- the Jacobian of their gradient output with respect to
    the sample position.  For the trilinear cell that is exactly what volume::eval_hess (src/volume.cpp:40-99)
    writes out by hand: zero diagonal, the three mixed partials, / h^2 (Q10); for the radial profile it is
    cylinder_volume::eval_hess (src/cylinder_volume.cpp:61-111): (I - rhat rhat^T)_{xz} n'/r with zero y row
    and column.  Points strictly inside the grid / the profile and away from cell faces (where the interpolant's
    gradient is discontinuous and autograd's one-sided value is a convention)."""
    torch.manual_seed(7)
    R, h = 8, 0.2
    scene = torch.rand(R, R, R, dtype=torch.float64) + 1.0           # our layout scene[z,y,x]
    G = ref_grid.Grid(scene.permute(2, 1, 0).contiguous(), h)
    cells = torch.randint(0, R - 1, (200, 3)).to(torch.float64)
    pts = (cells + 0.05 + 0.9 * torch.rand(200, 3, dtype=torch.float64)) * h
    H = torch.stack([torch.autograd.functional.jacobian(lambda q: G.GetLinear(q[None, :])[1][0], p) for p in pts])
    rres, radius = 13, 1.5
    prof = torch.rand(rres, dtype=torch.float64) + 1.0
    Cb = ref_cable.Cable(prof, radius, 10.0)
    hc = radius / (rres - 1)
    rr = (torch.randint(0, rres - 1, (150,)).to(torch.float64) + 0.05 + 0.9 * torch.rand(150, dtype=torch.float64)) * hc
    ang = torch.rand(150, dtype=torch.float64) * 2 * np.pi
    cp = torch.stack([radius + rr * torch.cos(ang), torch.rand(150, dtype=torch.float64) * 10, radius + rr * torch.sin(ang)], dim=-1)
    Hc = torch.stack([torch.autograd.functional.jacobian(lambda q: Cb.GetLinear(q[None, :])[1][0], p) for p in cp])
    save("hessians.npz", scene=scene.numpy(), h=h, pts=pts.numpy(), H=H.numpy(),
         prof=prof.numpy(), radius=radius, cpts=cp.numpy(), Hc=Hc.numpy())


def luneburg_cube():
    torch.manual_seed(0)
    np.random.seed(0)                                                 # scipy Rotation.random()
    R, span, nb = 17, 20.0, 12
    h = span / (R - 1)
    ds = h / 2
    g = np.linspace(0, span, R)
    Z, Y, X = np.meshgrid(g, g, g, indexing="ij")
    r = np.sqrt((X - span / 2) ** 2 + (Y - span / 2) ** 2 + (Z - span / 2) ** 2) / (span / 2)
    rif = np.sqrt(2 - np.minimum(r, 1.0) ** 2).astype(np.float32)
    (x, v, planes), rpv = ref_source.rand_rays_cube((nb, nb), 1, span, circle=True, src_type="plane")
    x, v, planes = ref_source.random_rotate_ic(x, v, planes, span)
    x, v, planes = x.float(), v.float(), planes.float()
    o = O.trace(rif, rif.shape, x.numpy(), v.numpy(), h, ds, dtype=np.float64)
    xm = torch.tensor(o["xt"], requires_grad=True)
    vm = torch.tensor(o["vt"], requires_grad=True)
    sn = planes[:, 1, :].double()
    sp = planes[:, 0, :].double()
    xmp, _ = ref_sensor.trace_rays_to_plane((xm, vm), (sp, sn))
    loss = torch.sum((xmp - sp) ** 2) / x.shape[0] / span             # core/luneburg_opt.py:100-102
    loss.backward()
    b = O.backtrace(rif, rif.shape, o["xt"], o["vt"], xm.grad.numpy(), vm.grad.numpy(), h, ds, dtype=np.float64)
    bc = O.backtrace(rif, rif.shape, o["xt"], o["vt"], xm.grad.numpy(), vm.grad.numpy(), h, ds, dtype=np.float64,
                     corrected_h=True)
    save("luneburg_cube.npz", rif=rif, h=h, ds=ds, x=x.numpy(), v=v.numpy(), planes=planes.numpy(),
         rpv=np.asarray(rpv), xt=o["xt"], vt=o["vt"], steps=o["steps"], loss=float(loss),
         grad_x=xm.grad.numpy(), grad_v=vm.grad.numpy(), drif=b["grad"], drif_corrected=bc["grad"])


def ad_vs_adjoint():
    torch.manual_seed(1)
    R, N = 9, 96
    rif = (1 + 0.3 * torch.rand(R, R, R, dtype=torch.float64))
    pos0 = torch.rand(N, 3, dtype=torch.float64) * (R - 1)
    vel = torch.tensor([[0.1, 1.0, 0.05]], dtype=torch.float64).repeat(N, 1)
    vel = vel / vel.norm(dim=-1, keepdim=True)
    out = dict(rif=rif.numpy(), vel=vel.numpy())
    for tag, h in (("h1", 1.0), ("h05", 0.5)):
        ds = h / 2
        p = pos0 * h
        p[:, 1] = -0.3 * ds                                           # off-face start (no termination ties)
        rr = rif.clone().requires_grad_(True)
        xt, vt = TA.trace(rr, p, vel, h, ds)
        gx, gv = torch.randn_like(xt), torch.randn_like(vt)
        ((xt * gx).sum() + (vt * gv).sum()).backward()
        out.update({f"{tag}_h": h, f"{tag}_ds": ds, f"{tag}_pos": p.numpy(), f"{tag}_xt": xt.detach().numpy(),
                    f"{tag}_vt": vt.detach().numpy(), f"{tag}_gx": gx.numpy(), f"{tag}_gv": gv.numpy(),
                    f"{tag}_ad_grad": rr.grad.numpy()})
    save("ad_vs_adjoint.npz", **out)


def march_getlinear():
    """Symplectic-Euler march (src/tracer.cpp:66-87: masked sample, v += ds n grad n, x += ds v, box tests, exit record,
    global loop until every ray has escaped) in float64 where the sample (n, grad n) of every step comes from the
    reference's own trilinear interpolant Grid.GetLinear RUN AS IS on the axis-permuted scene (Q2), followed by
    torch.autograd of L = sum(cx . xt) + sum(cv . vt) with respect to the scene.  Rays start a fraction of a step in
    front of the y = 0 face (the reference's sources start on or outside the box; Tracer::backtrace re-marches until the
    ray leaves the box, so it is the adjoint of rays that ENTER it) and are only ever sampled in bounds (Q4), where
    GetLinear's extra weight clip is the identity."""
    out = {}
    for tag, R, h, seed in (("h1", 13, 1.0, 21), ("h04", 17, 0.4, 22), ("h1b", 9, 1.0, 23)):
        torch.manual_seed(seed)
        ds = 0.37 * h                                                  # incommensurate with the cell size: no face ties
        g1 = torch.linspace(-1.0, 1.0, R, dtype=torch.float64)
        Z, Y, X = torch.meshgrid(g1, g1, g1, indexing="ij")           # OUR layout scene[z, y, x]
        scene = 1.2 + 0.25 * torch.exp(-2.5 * ((X - 0.1) ** 2 + (Y + 0.2) ** 2 + Z ** 2)) \
            + 0.04 * torch.rand(R, R, R, dtype=torch.float64)
        N = 160
        pos = torch.rand(N, 3, dtype=torch.float64) * (R - 3) * h + h  # at least one cell away from every face
        pos[:, 1] = -(0.1 + 0.8 * torch.rand(N, dtype=torch.float64)) * ds * 1.2 * 0.9   # enters the box with its first step
        vel = torch.randn(N, 3, dtype=torch.float64) * 0.25
        vel[:, 1] = 1.0
        vel = vel / vel.norm(dim=1, keepdim=True) * 1.2
        rr = scene.clone().requires_grad_(True)
        G = ref_grid.Grid(rr.permute(2, 1, 0), h)                      # reference layout scene[x, y, z]; a VIEW of rr
        shape = (R, R, R)
        max_steps = int(4 * h * R / ds)                                # src/tracer.cpp:51
        x, v = pos.clone(), vel.clone()
        xt, vt = pos.clone(), vel.clone()                              # :56-57
        inside = TA.inbounds(shape, h, x)                              # :61
        esc = torch.zeros_like(inside)
        iters = 0
        for _ in range(max_steps):
            n = torch.zeros(N, dtype=torch.float64); g = torch.zeros(N, 3, dtype=torch.float64)
            if bool(inside.any()):
                f_, fx_ = G.GetLinear(x[inside])                       # the reference's interpolant, as is
                n = n.index_put((inside.nonzero().squeeze(1),), f_)
                g = g.index_put((inside.nonzero().squeeze(1),), fx_)
            v = v + (ds * n)[:, None] * g                              # :70
            x = x + ds * v                                             # :71
            cur = TA.inbounds(shape, h, x)                             # :73
            cross = inside & ~cur                                      # :74
            esc = esc | cross | TA.escaped(shape, h, x, v)             # :75-76
            xt = torch.where(cross[:, None], x, xt); vt = torch.where(cross[:, None], v, vt)   # :79-80
            iters += 1
            if bool(esc.all()):                                        # :82
                break
            inside = cur                                               # :86
        assert bool(esc.all()), "golden march: a ray did not escape"
        cx, cv = torch.randn(N, 3, dtype=torch.float64), torch.randn(N, 3, dtype=torch.float64)
        ((xt * cx).sum() + (vt * cv).sum()).backward()
        out.update({f"{tag}_scene": scene.numpy(), f"{tag}_h": h, f"{tag}_ds": ds, f"{tag}_pos": pos.numpy(),
                    f"{tag}_vel": vel.numpy(), f"{tag}_xt": xt.detach().numpy(), f"{tag}_vt": vt.detach().numpy(),
                    f"{tag}_cx": cx.numpy(), f"{tag}_cv": cv.numpy(), f"{tag}_grad": rr.grad.numpy(),
                    f"{tag}_iters": iters})
    save("march_getlinear.npz", **out)


def fuel_injection():
    torch.manual_seed(0)
    data = np.load("/root/reference/data/fuel_injection_64.npy")      # (64,64,64) float64, F-order
    fuel_val = 0.0003
    vol = np.full((65, 65, 65), 1 + fuel_val, dtype=np.float32)       # core/fuel_injection_opt.py:40-42
    vol[:-1, :-1, :-1] = np.ascontiguousarray(data).astype(np.float32)
    span = 1.0
    h = span / 64
    ds = h / 2
    (x, v, planes), nrays = ref_source.rand_rays_in_sphere(3, (24, 24), 1, span, angle_span=180, circle=False,
                                                           xaxis=False, sensor_dist=1.0)
    x, v = x.float().numpy(), v.float().numpy()
    o = O.trace(vol, vol.shape, x, v, h, ds, dtype=np.float64)
    save("fuel_injection.npz", vol=vol, h=h, ds=ds, x=x, v=v, xt=o["xt"], vt=o["vt"], steps=o["steps"])


def upres():
    """core/optimizer.py upres_scene (-> core/grid.py upres_volume) RUN AS IS."""
    import optimizer as ref_optimizer
    torch.manual_seed(5)
    out = {}
    for tag, R, S in (("a", 5, 9), ("b", 9, 17), ("c", 17, 24)):
        n = (1 + 0.5 * torch.rand(R, R, R)).float()
        up = ref_optimizer.upres_scene(n, S)
        out.update({f"{tag}_src": n.numpy(), f"{tag}_dst": up.numpy()})
    save("upres.npz", **out)


def sensor_splat():
    """core/sensor.py generate_sensor (+ torch.autograd) RUN AS IS, float64, on CPU."""
    torch.manual_seed(3)
    out = {}
    for tag, res, with_t, with_e in (("a", 16, True, True), ("b", 33, False, False)):
        N, span = 1200, 2.0
        x = torch.rand(N, 3, dtype=torch.float64) * span
        x[:, 1] = span * 1.01
        v = torch.randn(N, 3, dtype=torch.float64) * 0.3
        v[:, 1] = 1.0
        e = (torch.rand(N, dtype=torch.float64) + 0.5) if with_e else 1.0
        p = torch.tensor([[span / 2, span * 1.2, span / 2]], dtype=torch.float64)
        n = torch.tensor([[0.1, 1.0, 0.05]], dtype=torch.float64)
        n = n / n.norm()
        t = torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float64) if with_t else None
        x.requires_grad_(True); v.requires_grad_(True)
        img = ref_sensor.generate_sensor((x, v), e, (p, n), res, span, t)
        gI = torch.randn_like(img)
        (img * gI).sum().backward()
        out.update({f"{tag}_x": x.detach().numpy(), f"{tag}_v": v.detach().numpy(),
                    f"{tag}_e": np.asarray(e.numpy() if with_e else e), f"{tag}_p": p.numpy(), f"{tag}_n": n.numpy(),
                    f"{tag}_t": (t.numpy() if with_t else np.zeros((0, 3))), f"{tag}_res": res, f"{tag}_span": span,
                    f"{tag}_img": img.detach().numpy(), f"{tag}_gI": gI.numpy(),
                    f"{tag}_gx": x.grad.numpy(), f"{tag}_gv": v.grad.numpy()})
    save("sensor_splat.npz", **out)


def sensor_far():
    """core/sensor.py generate_inf_sensor (:31-53) and torch.autograd through it RUN AS IS (float64, CPU): the
    far-field sensor of core/image_opt.py:116."""
    torch.manual_seed(4)
    out = {}
    for tag, res, with_t, span_deg in (("a", 24, True, 120), ("b", 17, False, 60)):
        N = 1500
        x = torch.rand(N, 3, dtype=torch.float64)
        v = torch.randn(N, 3, dtype=torch.float64) * (0.35 if tag == "a" else 0.2)
        v[:, 1] = 1.0
        v = v * (0.8 + 0.7 * torch.rand(N, 1, dtype=torch.float64))          # |v| = n != 1 behind a lens
        p = torch.tensor([[0.5, 1.2, 0.5]], dtype=torch.float64)
        n = torch.tensor([[0.08, 1.0, -0.04]], dtype=torch.float64)
        n = n / n.norm()
        t = torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float64) if with_t else None
        v.requires_grad_(True)
        e = 1 if tag == "a" else 0.7
        img = ref_sensor.generate_inf_sensor((x, v), e, (p, n), res, span_deg, t)
        gI = torch.randn_like(img)
        (img * gI).sum().backward()
        out.update({f"{tag}_x": x.numpy(), f"{tag}_v": v.detach().numpy(), f"{tag}_e": np.asarray(float(e)),
                    f"{tag}_p": p.numpy(), f"{tag}_n": n.numpy(), f"{tag}_t": (t.numpy() if with_t else np.zeros((0, 3))),
                    f"{tag}_res": res, f"{tag}_angle_span": span_deg, f"{tag}_img": img.detach().numpy(),
                    f"{tag}_gI": gI.numpy(), f"{tag}_gv": v.grad.numpy()})
    save("sensor_far.npz", **out)


def rays_to_plane():
    """core/sensor.py trace_rays_to_plane (:195-202) and torch.autograd through it RUN AS IS (float64, CPU): the
    statement right after the march in every experiment (core/image_opt.py:95, core/luneburg_opt.py:97), with per-ray
    planes (a) and one broadcast plane (b)."""
    torch.manual_seed(6)
    out = {}
    for tag, per_ray in (("a", True), ("b", False)):
        N = 900
        x = torch.rand(N, 3, dtype=torch.float64)
        v = torch.randn(N, 3, dtype=torch.float64) * 0.3
        v[:, 1] = 1.0
        v = v * (0.8 + 0.6 * torch.rand(N, 1, dtype=torch.float64))
        M = N if per_ray else 1
        p = torch.tensor([[0.5, 1.3, 0.5]], dtype=torch.float64) + 0.05 * torch.randn(M, 3, dtype=torch.float64)
        n = torch.tensor([[0.05, 1.0, -0.1]], dtype=torch.float64) + 0.1 * torch.randn(M, 3, dtype=torch.float64)
        n = n / n.norm(dim=1, keepdim=True)
        x.requires_grad_(True); v.requires_grad_(True)
        xo, vo = ref_sensor.trace_rays_to_plane((x, v), (p, n))
        gx, gv = torch.randn_like(xo), torch.randn_like(vo)
        ((xo * gx).sum() + (vo * gv).sum()).backward()
        out.update({f"{tag}_x": x.detach().numpy(), f"{tag}_v": v.detach().numpy(), f"{tag}_p": p.numpy(),
                    f"{tag}_n": n.numpy(), f"{tag}_xo": xo.detach().numpy(), f"{tag}_vo": vo.detach().numpy(),
                    f"{tag}_gxo": gx.numpy(), f"{tag}_gvo": gv.numpy(), f"{tag}_gx": x.grad.numpy(),
                    f"{tag}_gv": v.grad.numpy()})
    save("rays_to_plane.npz", **out)


def sdf_vals():
    """core/sensor.py get_sdf_vals_near (:102-119) / get_sdf_vals_far (:122-138) and torch.autograd through them RUN AS IS
    (float64, CPU); rays partly off the texture (edge extrapolation through the clipped tap indices)."""
    torch.manual_seed(8)
    out = {}
    N, res, span = 700, 24, 2.0
    tex = torch.randn(res, res, dtype=torch.float64).cumsum(0).cumsum(1) * 0.05
    x = torch.rand(N, 3, dtype=torch.float64) * span * 1.3 - 0.15 * span
    x[:, 1] = span * 0.9
    v = torch.randn(N, 3, dtype=torch.float64) * 0.25
    v[:, 1] = 1.0
    p = torch.tensor([[span / 2, span * 1.1, span / 2]], dtype=torch.float64)
    n = torch.tensor([[0.06, 1.0, -0.03]], dtype=torch.float64); n = n / n.norm()
    t = torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float64)
    for tag, fn, arg in (("near", ref_sensor.get_sdf_vals_near, span), ("far", ref_sensor.get_sdf_vals_far, 70.0)):
        xx = x.clone().requires_grad_(True); vv = v.clone().requires_grad_(True)
        f = fn((xx, vv), tex, (p, n), arg, t)
        gf = torch.randn_like(f)
        (f * gf).sum().backward()
        gx = xx.grad if xx.grad is not None else torch.zeros_like(xx)        # far: the positions do not enter
        out.update({f"{tag}_f": f.detach().numpy(), f"{tag}_gf": gf.numpy(), f"{tag}_gx": gx.numpy(),
                    f"{tag}_gv": vv.grad.numpy(), f"{tag}_arg": np.asarray(arg)})
    out.update({"x": x.numpy(), "v": v.numpy(), "p": p.numpy(), "n": n.numpy(), "t": t.numpy(), "tex": tex.numpy(),
                "span": np.asarray(span)})
    save("sdf_vals.npz", **out)


def area_rays():
    """core/source.py area_source3_rand_bias (:107-150) and area_source3_cone (:152-183) RUN AS IS (float32, CPU); `*_u*`
    are the uniforms they drew: the host generator is re-seeded and the same torch.rand calls replayed."""
    out = {}
    for tag, circle, xaxis, ang in (("a", False, False, 30.0), ("b", True, True, 115.0)):
        pix, spp, width, sd = (9, 7), 3, 2.0, 0.6
        torch.manual_seed(11)
        (x, v, pl), xt, tpv = ref_source.area_source3_rand_bias(torch.tensor(ang), pix, spp, width, circle=circle,
                                                                 xaxis=xaxis, sensor_dist=sd)
        torch.manual_seed(11)
        u_off = torch.rand(2 * spp, pix[0], pix[1]); u_ts = torch.rand(2, x.shape[0])
        out.update({f"bias_{tag}_x": x.numpy(), f"bias_{tag}_v": v.numpy(), f"bias_{tag}_planes": pl.numpy(),
                    f"bias_{tag}_xt": xt.numpy(), f"bias_{tag}_tpv": tpv.numpy(), f"bias_{tag}_uoff": u_off.numpy(),
                    f"bias_{tag}_uts": u_ts.numpy(), f"bias_{tag}_args": np.array([ang, pix[0], pix[1], spp, width, sd,
                                                                                     float(circle), float(xaxis)])})
        torch.manual_seed(12)
        (x, v, pl), tpv = ref_source.area_source3_cone(torch.tensor(ang), pix, spp, width, circle=circle, xaxis=xaxis,
                                                        sensor_dist=sd, cone_angle=70.0)
        torch.manual_seed(12)
        u_off = torch.rand(2 * spp, pix[0], pix[1]); u_z = torch.rand(x.shape[0]); u_th = torch.rand(x.shape[0])
        out.update({f"cone_{tag}_x": x.numpy(), f"cone_{tag}_v": v.numpy(), f"cone_{tag}_planes": pl.numpy(),
                    f"cone_{tag}_tpv": tpv.numpy(), f"cone_{tag}_uoff": u_off.numpy(),
                    f"cone_{tag}_uhat": torch.stack([u_z, u_th]).numpy()})
    save("area_rays.npz", **out)


def point_rays():
    """core/source.py point_source3 (:29-51) and rand_rays_cube(src_type='point') (:398-412) RUN AS IS (float32, CPU);
    deterministic, so the arrays are the whole fixture."""
    out = {}
    for tag, ang, pix, spp, width, cone, xaxis, sd in (("a", 40.0, (5, 4), 4, 2.0, 60.0, False, 0.0),
                                                       ("b", -115.0, (3, 6), 3, 1.5, 90.0, True, 0.25)):
        x, v, pl = ref_source.point_source3(torch.tensor(ang), pix, spp, width, cone_angle=cone, xaxis=xaxis, sensor_dist=sd)
        out.update({f"{tag}_x": x.numpy(), f"{tag}_v": v.numpy(), f"{tag}_planes": pl.numpy(),
                    f"{tag}_args": np.array([ang, pix[0], pix[1], spp, width, cone, float(xaxis), sd])})
    (x, v, pl), nrays = ref_source.rand_rays_cube((4, 3), 4, 1.0, src_type='point', cone_ang=50)
    out.update({"cube_x": x.numpy(), "cube_v": v.numpy(), "cube_planes": pl.numpy(), "cube_nrays": np.array(nrays)})
    save("point_rays.npz", **out)


def source_rays():
    """core/source.py generators RUN AS IS; `u_*` are the uniforms they drew (same seed replayed)."""
    out = {}
    pix, spp, width = (12, 10), 2, 20.0

    def draws(seed, nviews):
        torch.manual_seed(seed)
        return torch.stack([torch.rand(2 * spp, *pix) for _ in range(nviews)])

    # rand_rays_cube (circle=True as luneburg_opt.py:62) then random_rotate_ic
    out["cube_u"] = draws(11, 6).numpy()
    torch.manual_seed(11)
    (x, v, pl), nr = ref_source.rand_rays_cube(pix, spp, width, circle=True)
    out.update(cube_x=x.numpy(), cube_v=v.numpy(), cube_planes=pl.numpy(), cube_nrays=np.array(nr))
    np.random.seed(4)
    M = ref_source.random_rotmat()
    np.random.seed(4)
    xr, vr, plr = ref_source.random_rotate_ic(x, v, pl, width)
    out.update(cube_M=M.numpy(), cube_xr=xr.numpy(), cube_vr=vr.numpy(), cube_planes_r=plr.numpy())
    # rand_rays_in_sphere as image_opt.py:46 (no circle), 5 views over 300 degrees, sensor_dist 0.7
    out["sph_u"] = draws(12, 5).numpy()
    torch.manual_seed(12)
    (x, v, pl), nr = ref_source.rand_rays_in_sphere(5, pix, spp, width, angle_span=300, circle=False, xaxis=False,
                                                    sensor_dist=0.7)
    out.update(sph_x=x.numpy(), sph_v=v.numpy(), sph_planes=pl.numpy(), sph_nrays=np.array(nr))
    # independent samples, about the x axis, disc mask
    out["ind_u"] = draws(13, 3).numpy()
    torch.manual_seed(13)
    (x, v, pl), nr = ref_source.rand_rays_in_sphere(3, pix, spp, 0.3, angle_span=360, circle=True, xaxis=True,
                                                    sensor_dist=1.0, indep=True)
    out.update(ind_x=x.numpy(), ind_v=v.numpy(), ind_planes=pl.numpy(), ind_nrays=np.array(nr))
    # point sources (fuel_injection_opt.py:51 / image_opt.py:49): 4 views, no mask; and 3 views with the disc mask
    out["pt_u"] = draws(14, 4).numpy()
    torch.manual_seed(14)
    (x, v, pl), nr = ref_source.rand_ptrays_in_sphere(4, pix, spp, width, angle_span=360, circle=False, xaxis=False,
                                                      sensor_dist=0.5)
    out.update(pt_x=x.numpy(), pt_v=v.numpy(), pt_planes=pl.numpy(), pt_nrays=np.array(nr))
    out["ptc_u"] = draws(15, 3).numpy()
    torch.manual_seed(15)
    (x, v, pl), nr = ref_source.rand_ptrays_in_sphere(3, pix, spp, 3.0, angle_span=270, circle=True, xaxis=True,
                                                      sensor_dist=0.0)
    out.update(ptc_x=x.numpy(), ptc_v=v.numpy(), ptc_planes=pl.numpy(), ptc_nrays=np.array(nr))
    save("source_rays.npz", **out)


def cone_rays():
    """core/source.py cone_source3_rand (:186-203, hatbox_sample :531-545) and rand_rays_cube(src_type='cone') (:398-412)
    RUN AS IS; `u_*` are the uniforms hatbox_sample drew (same host-generator seed replayed: two torch.rand(N) per view)."""
    out = {}
    pix, spp, width = (9, 7), 3, 4.0
    n = pix[0] * pix[1] * spp
    torch.manual_seed(21)
    out["single_u"] = torch.stack([torch.rand(n), torch.rand(n)]).numpy()
    torch.manual_seed(21)
    x, v, pl = ref_source.cone_source3_rand(torch.tensor(35.0), pix, spp, width, sensor_dist=0.7, cone_angle=100.0)
    out.update(single_x=x.numpy(), single_v=v.numpy(), single_planes=pl.numpy())
    torch.manual_seed(22)
    out["cube_u"] = torch.stack([torch.stack([torch.rand(n), torch.rand(n)]) for _ in range(6)]).numpy()
    torch.manual_seed(22)
    (x, v, pl), nr = ref_source.rand_rays_cube(pix, spp, width, src_type='cone', cone_ang=60)
    out.update(cube_x=x.numpy(), cube_v=v.numpy(), cube_planes=pl.numpy(), cube_nrays=np.array(nr),
               pix=np.array(pix), spp=spp, width=width)
    save("cone_rays.npz", **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    if only:                                   # e.g. `make_golden.py splat_linear hessians_by_autograd`
        for name in only:
            globals()[name]()
        sys.exit(0)
    splat_linear()
    hessians_by_autograd()
    march_getlinear()
    getlinear_grid()
    getlinear_cable()
    luneburg_cube()
    ad_vs_adjoint()
    fuel_injection()
    sensor_splat()
    sensor_far()
    rays_to_plane()
    sdf_vals()
    area_rays()
    point_rays()
    cone_rays()
    upres()
    source_rays()
