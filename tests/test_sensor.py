"""Sensor image splat (SURVEY 8.8 "next" row 1).  The fixture tests/golden/sensor_splat.npz was produced
by RUNNING the reference's own core/sensor.py generate_sensor and torch.autograd through it (float64),
so for this row parity is pinned by the reference itself.
  CPU tier: the numpy restatement (oracle/sensor_ref.py) reproduces the fixture to rounding.
  GPU tier: the fused HIP kernels (through the C ABI / the autograd Function) reproduce the fixture
            and, at BASELINE size (1M rays, 512^2 sensor), the restatement and conservation laws."""
import os

import numpy as np
import pytest
import torch

import cases

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sensor_splat.npz")


def _case(z, tag):
    t = z[f"{tag}_t"]
    e = z[f"{tag}_e"]
    return dict(x=z[f"{tag}_x"], v=z[f"{tag}_v"], e=(e if e.ndim else float(e)), p=z[f"{tag}_p"], n=z[f"{tag}_n"],
                t=(t if t.size else None), res=int(z[f"{tag}_res"]), span=float(z[f"{tag}_span"]),
                img=z[f"{tag}_img"], gI=z[f"{tag}_gI"], gx=z[f"{tag}_gx"], gv=z[f"{tag}_gv"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_restatement_matches_reference_run(tag):
    from oracle import sensor_ref as S
    c = _case(np.load(G), tag)
    img = S.generate_sensor(c["x"], c["v"], c["e"], c["p"], c["n"], c["res"], c["span"], c["t"])
    assert np.abs(img - c["img"]).max() < 1e-12
    gx, gv = S.generate_sensor_backward(c["x"], c["v"], c["e"], c["p"], c["n"], c["res"], c["span"], c["gI"], c["t"])
    assert np.abs(gx - c["gx"]).max() < 1e-10 * np.abs(c["gx"]).max()
    assert np.abs(gv - c["gv"]).max() < 1e-10 * np.abs(c["gv"]).max()
    # every ray lands inside the image here, so energy is conserved up to taps falling off the border
    assert img.sum() <= (np.abs(c["v"] @ c["n"].reshape(3)) * c["e"]).sum() * (1 + 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_sensor_matches_reference_run(gpu, tag):
    from adjointnonlinearraytracing_amd import sensor
    c = _case(np.load(G), tag)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu, torch.float32)
    x, v = f(c["x"]).requires_grad_(True), f(c["v"]).requires_grad_(True)
    e = f(c["e"]) if isinstance(c["e"], np.ndarray) else c["e"]
    t = None if c["t"] is None else f(c["t"])
    img = sensor.generate_sensor((x, v), e, (f(c["p"]), f(c["n"])), c["res"], c["span"], t)
    assert img.shape == (c["res"], c["res"])
    assert cases.rel_l2(img.detach().cpu().numpy(), c["img"]) <= 2e-6
    (img * f(c["gI"])).sum().backward()
    assert cases.rel_l2(x.grad.cpu().numpy(), c["gx"]) <= 2e-4       # fp32 tent-weight derivatives
    assert cases.rel_l2(v.grad.cpu().numpy(), c["gv"]) <= 2e-4


@pytest.mark.gpu
def test_hip_sensor_full_size(gpu):
    """1M rays onto a 512^2 sensor (BASELINE config 5 shape): vs the float64 restatement on a sub-sample,
    energy conservation on the full set, gradient check by a directional finite difference."""
    from adjointnonlinearraytracing_amd import sensor
    from oracle import sensor_ref as S
    torch.manual_seed(0)
    n, res, span = 1 << 20, 512, 1.0
    x = torch.rand(n, 3, device=gpu) * span * 0.8 + 0.1 * span
    x[:, 1] = span
    v = torch.randn(n, 3, device=gpu) * 0.1
    v[:, 1] = 1.0
    p = torch.tensor([[span / 2, 1.1 * span, span / 2]], device=gpu)
    nn = torch.tensor([[0.0, 1.0, 0.0]], device=gpu)
    tt = torch.tensor([[0.0, 0.0, 1.0]], device=gpu)
    img = sensor.generate_sensor((x, v), 1.0, (p, nn), res, span, tt)
    assert abs(float(img.double().sum()) - float(v[:, 1].abs().double().sum())) <= 1e-4 * n   # all taps inside
    sub = slice(0, 20000)
    img_s = sensor.generate_sensor((x[sub], v[sub]), 1.0, (p, nn), res, span, tt)
    ref = S.generate_sensor(x[sub].cpu().numpy(), v[sub].cpu().numpy(), 1.0, p.cpu().numpy(), nn.cpu().numpy(), res, span,
                            tt.cpu().numpy())
    assert cases.rel_l2(img_s.cpu().numpy(), ref) <= 1e-4
    gI = torch.randn(res, res, device=gpu)
    xs, vs = x[sub].clone().requires_grad_(True), v[sub].clone().requires_grad_(True)
    (sensor.generate_sensor((xs, vs), 1.0, (p, nn), res, span, tt) * gI).sum().backward()
    gx, gv = S.generate_sensor_backward(x[sub].cpu().numpy(), v[sub].cpu().numpy(), 1.0, p.cpu().numpy(), nn.cpu().numpy(),
                                        res, span, gI.cpu().numpy(), tt.cpu().numpy())
    assert cases.rel_l2(xs.grad.cpu().numpy(), gx) <= 5e-3 and cases.rel_l2(vs.grad.cpu().numpy(), gv) <= 5e-3
    with pytest.raises(RuntimeError):
        sensor.generate_sensor((x.cpu(), v.cpu()), 1.0, (p.cpu(), nn.cpu()), res, span, tt.cpu())


@pytest.mark.gpu
def test_upres_matches_reference_run(gpu):
    """Multires up-sampling: fixture produced by running core/optimizer.py upres_scene."""
    from adjointnonlinearraytracing_amd import optimizer
    z = np.load(os.path.join(os.path.dirname(G), "upres.npz"))
    for tag in ("a", "b", "c"):
        src, ref = z[f"{tag}_src"], z[f"{tag}_dst"]
        out = optimizer.upres_scene(torch.from_numpy(src).to(gpu), ref.shape[0])
        assert out.shape == ref.shape and out.dtype == torch.float32
        assert np.abs(out.cpu().numpy() - ref).max() <= 2e-7 * np.abs(ref).max()
    # Adam-state transfer keeps hyper-parameters and up-samples both moments (core/optimizer.py:13-41)
    n0 = torch.rand(5, 5, 5, device=gpu, requires_grad=True)
    o0 = torch.optim.Adam([n0], lr=3e-3, betas=(0.8, 0.95))
    (n0 ** 2).sum().backward(); o0.step()
    n1 = optimizer.upres_scene(n0.detach(), 9).requires_grad_(True)
    o1 = optimizer.reload_opto(o0, n1, 1e-3)
    st = o1.state[n1]
    assert st["exp_avg"].shape == (9, 9, 9) and st["exp_avg_sq"].shape == (9, 9, 9)
    assert o1.param_groups[0]["betas"] == (0.8, 0.95) and o1.param_groups[0]["lr"] == 3e-3
    (n1 ** 2).sum().backward(); o1.step()
    with pytest.raises(RuntimeError):
        optimizer.upres_scene(torch.rand(5, 5, 5), 9)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_hip_sensor_fuzz(gpu, seed):
    """Seeded random sensors: oblique planes and tangents (not axis aligned, not unit length), image sizes
    from 3 to 97 pixels, rays that partly or entirely miss the image, scalar and per-ray e -- the fused HIP
    forward/backward against the float64 restatement (which tests above pin to the reference's own run)."""
    from adjointnonlinearraytracing_amd import sensor
    from oracle import sensor_ref as S
    rng = np.random.default_rng(300 + seed)
    n = 5000
    res = int(rng.integers(3, 98))
    span = float(rng.uniform(0.3, 20.0))
    nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
    tan = np.cross(nrm, rng.normal(size=3)); tan *= rng.uniform(0.5, 2.0) / np.linalg.norm(tan)
    p = rng.uniform(0.3, 0.7, 3) * span + nrm * span
    t1, t2 = S.tan_vecs(nrm, tan)
    # ray origins behind the plane, aimed at points spread over 1.6x the image footprint
    aim = p + (rng.uniform(-0.8, 0.8, (n, 1)) * span) * t1 / np.linalg.norm(t1) + \
          (rng.uniform(-0.8, 0.8, (n, 1)) * span) * t2 / np.linalg.norm(t2)
    x = aim - nrm * rng.uniform(0.2, 2.0, (n, 1)) * span + rng.normal(0, 0.1 * span, (n, 3))
    v = aim - x + rng.normal(0, 0.05 * span, (n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    if seed % 3 == 2:
        v = -v                                           # den < 0: foreshortening uses |v.n|
    e = float(rng.uniform(0.5, 2.0)) if seed % 2 else rng.uniform(0.1, 2.0, n)
    gI = rng.normal(size=(res, res))
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu, torch.float32)
    xs, vs = f(x).requires_grad_(True), f(v).requires_grad_(True)
    eg = f(e) if isinstance(e, np.ndarray) else e
    img = sensor.generate_sensor((xs, vs), eg, (f(p[None]), f(nrm[None])), res, span, f(tan[None]))
    x32, v32 = xs.detach().cpu().numpy().astype(np.float64), vs.detach().cpu().numpy().astype(np.float64)
    e32 = e if not isinstance(e, np.ndarray) else e.astype(np.float32).astype(np.float64)
    p32, n32, t32 = (a.astype(np.float32).astype(np.float64) for a in (p, nrm, tan))
    ref = S.generate_sensor(x32, v32, e32, p32, n32, res, span, t32)
    assert cases.rel_l2(img.detach().cpu().numpy(), ref) <= 2e-4
    (img * f(gI)).sum().backward()
    gx, gv = S.generate_sensor_backward(x32, v32, e32, p32, n32, res, span, gI.astype(np.float32).astype(np.float64), t32)
    assert cases.rel_l2(xs.grad.cpu().numpy(), gx) <= 1e-2 and cases.rel_l2(vs.grad.cpu().numpy(), gv) <= 1e-2


# ---------------------------------------------------------------------------------------- far-field sensor
GF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sensor_far.npz")


def _far_case(z, tag):
    t = z[f"{tag}_t"]
    return dict(x=z[f"{tag}_x"], v=z[f"{tag}_v"], e=float(z[f"{tag}_e"]), p=z[f"{tag}_p"], n=z[f"{tag}_n"],
                t=(t if t.size else None), res=int(z[f"{tag}_res"]), angle_span=float(z[f"{tag}_angle_span"]),
                img=z[f"{tag}_img"], gI=z[f"{tag}_gI"], gv=z[f"{tag}_gv"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_far_restatement_matches_reference_run(tag):
    """core/sensor.py:31-53 generate_inf_sensor RUN AS IS (fixture) vs the numpy restatement, forward and backward."""
    from oracle import sensor_ref as S
    c = _far_case(np.load(GF), tag)
    img = S.generate_inf_sensor(c["v"], c["e"], c["n"], c["res"], c["angle_span"], c["t"])
    assert np.abs(img - c["img"]).max() < 1e-6 * max(1.0, np.abs(c["img"]).max())    # the reference mixes f32 `fe` into f64
    gv = S.generate_inf_sensor_backward(c["v"], c["e"], c["n"], c["res"], c["gI"], c["angle_span"], c["t"])
    assert cases.rel_l2(gv, c["gv"]) < 1e-6
    assert c["img"].sum() > 0.5 * c["e"] * len(c["v"])               # most directions fall inside the angular window


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_far_sensor_matches_reference_run(gpu, tag):
    from adjointnonlinearraytracing_amd import sensor
    c = _far_case(np.load(GF), tag)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu, torch.float32)
    x, v = f(c["x"]).requires_grad_(True), f(c["v"]).requires_grad_(True)
    t = None if c["t"] is None else f(c["t"])
    img = sensor.generate_inf_sensor((x, v), c["e"], (f(c["p"]), f(c["n"])), c["res"], c["angle_span"], t)
    assert img.shape == (c["res"], c["res"])
    assert cases.rel_l2(img.detach().cpu().numpy(), c["img"]) <= 5e-6
    (img * f(c["gI"])).sum().backward()
    assert x.grad is None or float(x.grad.abs().sum()) == 0.0        # positions do not enter (sensor.py:33)
    assert cases.rel_l2(v.grad.cpu().numpy(), c["gv"]) <= 3e-4       # fp32 tent-weight derivatives


@pytest.mark.gpu
def test_hip_far_sensor_full_size(gpu):
    """1M directions onto a 512^2 far-field image (config 5 shape): conservation + the restatement on a sub-sample."""
    from adjointnonlinearraytracing_amd import sensor
    from oracle import sensor_ref as S
    torch.manual_seed(1)
    n, res = 1 << 20, 512
    x = torch.rand(n, 3, device=gpu)
    v = torch.randn(n, 3, device=gpu) * 0.15
    v[:, 1] = 1.0
    p = torch.tensor([[0.5, 1.1, 0.5]], device=gpu); nn = torch.tensor([[0.0, 1.0, 0.0]], device=gpu)
    tt = torch.tensor([[0.0, 0.0, 1.0]], device=gpu)
    img = sensor.generate_inf_sensor((x, v), 1, (p, nn), res, 120, tt)
    assert abs(float(img.double().sum()) - n) <= 1e-4 * n            # every direction is inside +-60 degrees here
    sub = slice(0, 20000)
    img_s = sensor.generate_inf_sensor((x[sub], v[sub]), 1, (p, nn), res, 120, tt)
    ref = S.generate_inf_sensor(v[sub].cpu().numpy(), 1.0, nn.cpu().numpy(), res, 120, tt.cpu().numpy())
    assert cases.rel_l2(img_s.cpu().numpy(), ref) <= 1e-4


# ---- trace_rays_to_plane (core/sensor.py:195-202): the statement right after the march -------------------------------
GP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rays_to_plane.npz")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_rays_to_plane_torch_path_matches_reference_run(tag):
    """CPU tensors (and any input the fused kernel does not take) run the reference's own torch expressions: they
    reproduce the fixture made by RUNNING core/sensor.py trace_rays_to_plane + autograd (float64)."""
    from adjointnonlinearraytracing_amd import sensor
    z = np.load(GP)
    x = torch.from_numpy(z[f"{tag}_x"]).requires_grad_(True)
    v = torch.from_numpy(z[f"{tag}_v"]).requires_grad_(True)
    xo, vo = sensor.trace_rays_to_plane((x, v), (torch.from_numpy(z[f"{tag}_p"]), torch.from_numpy(z[f"{tag}_n"])))
    ((xo * torch.from_numpy(z[f"{tag}_gxo"])).sum() + (vo * torch.from_numpy(z[f"{tag}_gvo"])).sum()).backward()
    assert np.abs(xo.detach().numpy() - z[f"{tag}_xo"]).max() < 1e-13
    assert np.abs(x.grad.numpy() - z[f"{tag}_gx"]).max() < 1e-12 and np.abs(v.grad.numpy() - z[f"{tag}_gv"]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_rays_to_plane_fused_matches_reference_run(gpu, tag):
    """fp32 rays on the device take the fused HIP kernels (forward + analytic backward): 4e-6 relative to the fixture
    (fp32 rounding of t = a / b and of the products), per-ray planes and one broadcast plane; v passes through."""
    from adjointnonlinearraytracing_amd import sensor
    z = np.load(GP)
    f = lambda k: torch.from_numpy(z[f"{tag}_{k}"]).to(torch.float32).to(gpu)
    x, v = f("x").requires_grad_(True), f("v").requires_grad_(True)
    xo, vo = sensor.trace_rays_to_plane((x, v), (f("p"), f("n")))
    assert vo is v and xo.grad_fn is not None and type(xo.grad_fn).__name__.startswith("_RaysToPlane")
    ((xo * f("gxo")).sum() + (vo * f("gvo")).sum()).backward()
    for got, want in ((xo.detach(), z[f"{tag}_xo"]), (x.grad, z[f"{tag}_gx"]), (v.grad, z[f"{tag}_gv"])):
        assert np.abs(got.cpu().numpy() - want).max() <= 4e-6 * np.abs(want).max(), tag
    # planes that require grad fall back to the torch expressions (their gradients exist, as in the reference)
    p = f("p").requires_grad_(True)
    xo2, _ = sensor.trace_rays_to_plane((x.detach(), v.detach()), (p, f("n")))
    xo2.sum().backward()
    assert p.grad is not None and torch.allclose(xo2.detach(), xo.detach(), rtol=1e-5, atol=1e-6)


# ---- get_sdf_vals_near / get_sdf_vals_far (core/sensor.py:102-138): texture lookups at the sensor --------------------
GS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sdf_vals.npz")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["near", "far"])
def test_sdf_vals_match_reference_run(gpu, tag):
    """Fixture made by RUNNING the reference's functions and autograd through them (float64); a third of the rays land
    off the texture (edge extrapolation).  The fused kernels reproduce values and ray gradients to fp32 rounding."""
    from adjointnonlinearraytracing_amd import sensor
    z = np.load(GS)
    f32 = lambda k: torch.from_numpy(z[k]).to(torch.float32).to(gpu)
    x, v = f32("x").requires_grad_(True), f32("v").requires_grad_(True)
    fn = sensor.get_sdf_vals_near if tag == "near" else sensor.get_sdf_vals_far
    f = fn((x, v), f32("tex"), (f32("p"), f32("n")), float(z[f"{tag}_arg"]), f32("t"))
    (f * f32(f"{tag}_gf")).sum().backward()
    want = z[f"{tag}_f"]
    assert np.abs(f.detach().cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max()
    for got, key in ((x.grad, "gx"), (v.grad, "gv")):
        w = z[f"{tag}_{key}"]
        assert np.abs(got.cpu().numpy() - w).max() <= 2e-4 * max(np.abs(w).max(), 1e-30), key


@pytest.mark.gpu
def test_sensor_frame_on_host_and_on_device_agree(gpu):
    """The sensor operators take their frame (plane point, normal, tangents) either from host float[3] arrays
    (drrt_sensor_*_f32) or from 12 floats in device memory (drrt_sensor_*_dframe_*: planes that already live on the
    device, as in core/image_opt.py, cost no copy back / sync).  Same kernels: images, samples and gradients are equal."""
    from adjointnonlinearraytracing_amd import sensor
    torch.manual_seed(3)
    n, res, span = 20000, 96, 1.0
    x0 = torch.rand(n, 3, device=gpu) * 0.8 + 0.1
    v0 = torch.randn(n, 3, device=gpu) * 0.15
    v0[:, 1] = 1.0
    p = torch.tensor([[0.5, 1.1, 0.5]])
    nn = torch.tensor([[0.0, 1.0, 0.0]])
    tt = torch.tensor([[0.0, 0.0, 1.0]])
    tex = torch.rand(64, 64, device=gpu)
    gI = torch.rand(res, res, device=gpu)

    def run(dev):
        P, N, T = p.to(dev), nn.to(dev), tt.to(dev)
        out = []
        for fn in (lambda r: (sensor.generate_sensor(r, 1.0, (P, N), res, span, T) * gI).sum(),
                   lambda r: (sensor.generate_inf_sensor(r, 1, (P, N), res, 120, T) * gI).sum(),
                   lambda r: (sensor.get_sdf_vals_near(r, tex, (P, N), span, T) ** 2).sum(),
                   lambda r: (sensor.get_sdf_vals_far(r, tex, (P, N), 100, T) ** 2).sum()):
            x, v = x0.clone().requires_grad_(True), v0.clone().requires_grad_(True)
            val = fn((x, v))
            val.backward()
            out += [val.detach(), torch.zeros_like(x) if x.grad is None else x.grad.clone(), v.grad.clone()]   # far field: no x
        return out

    host, dev = run("cpu"), run(gpu)
    for k, (a, b) in enumerate(zip(host, dev)):
        if k % 3 == 0:      # the scalar: a sum of atomically accumulated pixels (order-dependent in the last place)
            assert abs(float(a) - float(b)) <= 1e-5 * max(1.0, abs(float(a))), k
        else:
            assert torch.equal(a, b), k


@pytest.mark.gpu
def test_sdf_vals_refuse_what_they_cannot_differentiate(gpu):
    """The fused texture lookups return gradients of the rays only and take ONE plane per call: a texture that requires
    grad, or per-ray planes that differ, must raise instead of returning a silent zero / using row 0."""
    from adjointnonlinearraytracing_amd import sensor
    N = 64
    x = torch.rand(N, 3, device=gpu); v = torch.zeros(N, 3, device=gpu); v[:, 1] = 1.0
    tex = torch.rand(16, 16, device=gpu)
    p = torch.tensor([[0.5, 1.2, 0.5]], device=gpu); n = torch.tensor([[0.0, 1.0, 0.0]], device=gpu)
    f = sensor.get_sdf_vals_near((x, v), tex, (p, n), 1.0)
    assert f.shape == (N,)
    with pytest.raises(RuntimeError, match="not differentiable w.r.t. the texture"):
        sensor.get_sdf_vals_near((x, v), tex.clone().requires_grad_(True), (p, n), 1.0)
    pp = p.repeat(N, 1); pp[5, 1] = 1.5
    with pytest.raises(RuntimeError, match="one plane"):
        sensor.get_sdf_vals_near((x, v), tex, (pp, n.repeat(N, 1)), 1.0)
    assert torch.equal(sensor.get_sdf_vals_near((x, v), tex, (p.repeat(N, 1), n.repeat(N, 1)), 1.0), f)   # equal rows are fine
