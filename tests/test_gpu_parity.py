"""GPU parity tests proper: the HIP path (through the C ABI, via drrt.TracerC) against the CPU
oracle on identical seeded inputs.

Bar (fp32 path):
  * vs the oracle's FACTORED float32 arithmetic (the explicit IEEE op sequence the kernels
    implement, validated against the literal restatement of the reference in float64 by
    tests/test_oracle.py): exit rays, exit steps, closest-approach records, fail masks and step
    statistics are BIT-EXACT; adjoint grids agree to rel-L2 <= 2e-5 (only the order of the fp32
    atomic sums differs).
  * vs the LITERAL float64 oracle: >= 99 % of rays within 2e-5*span (the trilinear gradient is
    discontinuous across cell faces and the box test is discontinuous, so any two float
    implementations diverge on the few rays that land within an ulp of a face);
    adjoint rel-L2 <= 2e-2 (fp32 second differences / h^2 make the Hessian term noisy; measured
    fp32-vs-fp64 spread of the ORACLE ITSELF on these scenes is 5e-6 .. 7e-3).
"""
import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def drrt_mod(gpu):
    from adjointnonlinearraytracing_amd import drrt
    drrt.options.check_failed = False
    return drrt


def _scene(kind, R):
    if kind == "luneburg":
        return cases.luneburg(R)
    if kind == "smooth":
        return cases.smooth_field(R, seed=3)
    if kind == "uniform":
        return np.ones((R, R, R), np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind,R,n", [("luneburg", 33, 4096), ("smooth", 33, 4096), ("uniform", 17, 1024),
                                      ("luneburg", 65, 20000)])
@pytest.mark.parametrize("sort", [False, True])
def test_trace_matches_oracle(gpu, oracle, drrt_mod, kind, R, n, sort):
    span = 1.0
    h = span / (R - 1); ds = h / 2
    rif = _scene(kind, R)
    pos, vel = cases.cube_rays(n // 6 + 1, span, ds, seed=1)
    drrt_mod.options.sort_rays = sort
    xt, vt = drrt_mod.TracerC().trace(_t(rif, gpu), rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    st = drrt_mod.read_stats()
    xt, vt = xt.cpu().numpy(), vt.cpu().numpy()
    with oracle.arith("factored"):
        ref32 = oracle.trace(rif, rif.shape, pos, vel, h, ds, dtype=np.float32)
    assert np.array_equal(xt, ref32["xt"]) and np.array_equal(vt, ref32["vt"])          # bit-exact
    assert st["ray_steps"] == int(ref32["steps"].sum()) and st["iters"] == ref32["iters"]
    assert st["n_failed"] == ref32["n_failed"] == 0
    ref = oracle.trace(rif, rif.shape, pos, vel, h, ds, dtype=np.float64)             # literal, float64
    d = np.linalg.norm(xt - ref["xt"], axis=1)
    assert np.mean(d <= 2e-5 * span) >= 0.99
    assert np.median(d) <= 2e-6 * span


@pytest.mark.parametrize("kind,R,n", [("luneburg", 33, 6000), ("smooth", 33, 6000), ("uniform", 33, 1024)])
@pytest.mark.parametrize("sort", [False, True])
@pytest.mark.parametrize("corrected", [False, True])
def test_backtrace_matches_oracle(gpu, oracle, drrt_mod, kind, R, n, sort, corrected):
    span = 1.0
    h = span / (R - 1); ds = h / 2
    rif = _scene(kind, R)
    pos, vel = cases.cube_rays(n // 6, span, ds, seed=5)
    drrt_mod.options.sort_rays = sort
    drrt_mod.options.corrected_h = corrected
    try:
        T = drrt_mod.TracerC()
        xt, vt = T.trace(_t(rif, gpu), rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
        rng = np.random.default_rng(7)
        dx = rng.normal(size=pos.shape).astype(np.float32)
        dv = rng.normal(size=pos.shape).astype(np.float32)
        g = T.backtrace(_t(rif, gpu), rif.shape, xt, vt, _t(dx, gpu), _t(dv, gpu), h, ds)
        st = drrt_mod.read_stats()
    finally:
        drrt_mod.options.corrected_h = False
    xt_np, vt_np, g_np = xt.cpu().numpy(), vt.cpu().numpy(), g.cpu().numpy()
    with oracle.arith("factored"):
        ref32 = oracle.backtrace(rif, rif.shape, xt_np, vt_np, dx, dv, h, ds, dtype=np.float32, corrected_h=corrected)
    assert st["ray_steps"] == ref32["steps_total"]                       # identical reverse trajectories
    err32 = cases.rel_l2(g_np, ref32["grad"])
    assert err32 <= 2e-5, f"rel-L2 vs factored fp32 oracle {err32}"
    ref64 = oracle.backtrace(rif, rif.shape, xt_np, vt_np, dx, dv, h, ds, dtype=np.float64, corrected_h=corrected)
    err64 = cases.rel_l2(g_np, ref64["grad"])
    assert err64 <= 2e-2, f"rel-L2 vs literal fp64 oracle {err64}"
    if kind == "uniform":
        # KAT: grad n = 0 => value splat 0 and the gradient-splat weights sum to 0
        assert abs(float(g.sum())) <= 1e-3 * float(g.abs().sum() + 1e-30)


@pytest.mark.parametrize("sort", [True, False])
def test_quad_grid_copy_is_bit_identical(gpu, oracle, drrt_mod, sort):
    """DRRT_FLAG_QUAD_GRID (two 16-byte loads per interior cell from the quad copy) vs the plain grid: the
    taps are the same floats, so forward results are identical bit for bit and the adjoint differs only by
    atomic summation order.  Covers cubic and non-cubic grids, the plane / sdf variants, a grid too small to
    have interior cells, and the paired-adjoint reuse of the copy -- including after ANOTHER grid has been
    marched in between (the stale copy must not be used)."""
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = sort
    try:
        for shape, n in (((65, 65, 65), 6000), ((40, 33, 48), 3000), ((3, 3, 3), 300)):
            D_, H_, W_ = shape
            span = 1.0
            h = span / (max(shape) - 1); ds = h / 2
            rng = np.random.default_rng(21)
            rif_np = (1.0 + 0.3 * rng.random(shape, dtype=np.float32)).astype(np.float32)
            rif = _t(rif_np, gpu)
            other = _t((1.0 + 0.3 * rng.random(shape, dtype=np.float32)).astype(np.float32), gpu)
            sdf = _t((rng.random(shape, dtype=np.float32) - 0.7).astype(np.float32), gpu)
            res = (W_, H_, D_)
            ext = np.array([(W_ - 1) * h, (H_ - 1) * h, (D_ - 1) * h], np.float32)
            pos = (rng.uniform(0.02, 0.98, (n, 3)) * ext).astype(np.float32); pos[:, 1] = 0.0
            vel = rng.normal(0, 0.3, (n, 3)).astype(np.float32); vel[:, 1] = 1.0
            vel /= np.linalg.norm(vel, axis=1, keepdims=True)
            po = np.tile((np.array([[0.5, 0.6, 0.5]], np.float32) * ext), (n, 1)).astype(np.float32)
            pd = np.tile(np.array([[0, 1, 0]], np.float32), (n, 1))
            dx = rng.normal(size=(n, 3)).astype(np.float32); dv = rng.normal(size=(n, 3)).astype(np.float32)
            P, V_, PO, PD, DX, DV = (_t(a, gpu) for a in (pos, vel, po, pd, dx, dv))
            out = {}
            for quad in (True, False):
                drrt_mod.options.quad_grid = quad
                xt, vt = T.trace(rif, res, P, V_, h, ds)
                st, order = drrt_mod.read_stats(), drrt_mod.keep_order(drrt_mod.last_order)   # held across other trace calls
                g_paired = T.backtrace(rif, res, xt, vt, DX, DV, h, ds, order=order)          # reuses the copy
                xo, vo = T.trace(other, res, P, V_, h, ds)                                    # another grid in between
                g_after = T.backtrace(rif, res, xt, vt, DX, DV, h, ds, order=order)           # must rebuild
                g_plain = T.backtrace(rif, res, xt, vt, DX, DV, h, ds)                        # unpaired: rebuilds
                xp, vp, fm = T.trace_pln(rif, res, P, V_, PO, PD, h, ds)
                xs, vs = T.trace_sdf(rif, sdf, res, P, V_, h, ds)
                gs = T.backtrace_sdf(rif, sdf, res, xs, vs, DX, DV, h, ds)
                out[quad] = [t.cpu() for t in (xt, vt, xo, vo, xp, vp, fm, xs, vs)] + [st] + \
                            [t.cpu().numpy() for t in (g_paired, g_after, g_plain, gs)]
            a, b = out[True], out[False]
            for k in range(9):
                assert torch.equal(a[k], b[k]), (shape, k)
            assert a[9] == b[9]
            for k in range(10, 14):
                assert cases.rel_l2(a[k], b[k]) <= 2e-6, (shape, k)
            with oracle.arith("factored"):
                o = oracle.trace(rif_np, res, pos, vel, h, ds, dtype=np.float32)
            assert np.array_equal(a[0].numpy(), o["xt"]) and np.array_equal(a[1].numpy(), o["vt"])
    finally:
        drrt_mod.options.quad_grid = False
        drrt_mod.options.sort_rays = True


@pytest.mark.parametrize("kind", ["luneburg", "smooth"])
def test_window_kernel_equals_direct_atomics(gpu, drrt_mod, kind):
    """The LDS gradient-window kernel (default) and the one-atomic-per-tap kernel
    (DRRT_FLAG_DIRECT_ATOMICS) add exactly the same terms: only the fp32 summation order differs.
    Incoherent (unsorted, six-view) waves exercise the global-atomic fallback of the window kernel."""
    R, span = 65, 1.0
    h = span / (R - 1); ds = h / 2
    rif = _t(_scene(kind, R), gpu)
    pos, vel = cases.cube_rays(4000, span, ds, seed=12, tilt=0.3)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    dx = torch.randn_like(xt); dv = torch.randn_like(vt)
    out = {}
    for sort in (True, False):
        for direct in (True, False):
            drrt_mod.options.sort_rays, drrt_mod.options.direct_atomics = sort, direct
            try:
                out[(sort, direct)] = T.backtrace(rif, rif.shape, xt, vt, dx, dv, h, ds).cpu().numpy()
                st = drrt_mod.read_stats()
            finally:
                drrt_mod.options.direct_atomics = False
            out[("steps", sort, direct)] = st["ray_steps"]
    base = out[(True, True)]
    for key in ((True, False), (False, True), (False, False)):
        assert cases.rel_l2(out[key], base) <= 2e-5, key
    assert len({out[("steps", s_, d_)] for s_ in (True, False) for d_ in (True, False)}) == 1


def test_linear_field_adjoint_within_1e4_of_fp64(gpu, oracle, drrt_mod):
    """north_star tolerance (adjoint within 1e-4 rel-L2) on a scene without cell-face
    discontinuities at moderate resolution: n = a + b.p is exact under trilinear interpolation."""
    R, span = 17, 1.0
    h = span / (R - 1); ds = h / 2
    g = np.linspace(0, span, R)
    Z, Y, X = np.meshgrid(g, g, g, indexing="ij")
    rif = (1.0 + 0.10 * X + 0.25 * Y - 0.05 * Z).astype(np.float32)
    pos, vel = cases.cube_rays(1000, span, ds, seed=8)
    drrt_mod.options.sort_rays = True
    T = drrt_mod.TracerC()
    xt, vt = T.trace(_t(rif, gpu), rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    rng = np.random.default_rng(9)
    dx = rng.normal(size=pos.shape).astype(np.float32); dv = rng.normal(size=pos.shape).astype(np.float32)
    gd = T.backtrace(_t(rif, gpu), rif.shape, xt, vt, _t(dx, gpu), _t(dv, gpu), h, ds).cpu().numpy()
    ref = oracle.backtrace(rif, rif.shape, xt.cpu().numpy(), vt.cpu().numpy(), dx, dv, h, ds, dtype=np.float64)
    assert cases.rel_l2(gd, ref["grad"]) <= 1e-4


def test_uniform_medium_kat(gpu, drrt_mod):
    """src/test.cpp:117-146 workload: rif=1, h=1, 33^3, rays on z=0 along +z, ds=0.5.
    Straight rays; exit at the first sample with z >= 32 => 64 steps, xt.z = 32 exactly."""
    R, h, ds = 33, 1.0, 0.5
    rif = np.ones((R, R, R), np.float32)
    g = np.linspace(0.5, 31.5, 16, dtype=np.float32)
    X, Y = np.meshgrid(g, g, indexing="ij")
    pos = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size, np.float32)], -1)
    vel = np.tile(np.array([[0, 0, 1]], np.float32), (len(pos), 1))
    for sort in (False, True):
        drrt_mod.options.sort_rays = sort
        xt, vt = drrt_mod.TracerC().trace(_t(rif, gpu), rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
        st = drrt_mod.read_stats()
        assert torch.equal(xt.cpu()[:, 2], torch.full((len(pos),), 32.0))
        assert torch.equal(xt.cpu()[:, :2], torch.from_numpy(pos[:, :2]))
        assert torch.equal(vt.cpu(), torch.from_numpy(vel))
        assert st["ray_steps"] == 64 * len(pos) and st["iters"] == 64 and st["n_failed"] == 0


def test_trace_plane_and_target(gpu, oracle, drrt_mod):
    R, span = 33, 1.0
    h = span / (R - 1); ds = h / 2
    rif = cases.luneburg(R)
    pos, vel = cases.plane_rays(5000, span, ds, seed=11)
    # sensor plane inside the volume at y = 0.7*span, normal +y
    po = np.tile(np.array([[0.5, 0.7, 0.5]], np.float32) * span, (len(pos), 1))
    pd = np.tile(np.array([[0, 1, 0]], np.float32), (len(pos), 1))
    drrt_mod.options.sort_rays = True
    T = drrt_mod.TracerC()
    with oracle.arith("factored"):
        ref = oracle.trace(rif, rif.shape, pos, vel, h, ds, dtype=np.float32, mode="plane", pln_o=po, pln_d=pd)
    for pair in (False, True):           # k_trace_flat<PAIR, 1>: plain grid and the pair copy of it
        drrt_mod.options.pair_grid = pair
        try:
            xt, vt, fm = T.trace_pln(_t(rif, gpu), rif.shape, _t(pos, gpu), _t(vel, gpu), _t(po, gpu), _t(pd, gpu), h, ds)
        finally:
            drrt_mod.options.pair_grid = "auto"
        assert np.array_equal(xt.cpu().numpy(), ref["xt"]) and np.array_equal(vt.cpu().numpy(), ref["vt"]), pair
        assert np.array_equal(fm.cpu().numpy().astype(bool), ref["failmask"]), pair
    # targets beyond the far face (closest approach in free flight AFTER escape: exercises the
    # global-loop-count coupling, src/tracer.cpp:225-227), inside the volume, and behind the source
    for tgt in ([0.5, 1.3, 0.5], [0.5, 0.6, 0.5], [-0.4, 0.2, 0.5]):
        tg = np.tile(np.array([tgt], np.float32) * span, (len(pos), 1))
        with oracle.arith("factored"):
            ref2 = oracle.trace_target(rif, rif.shape, pos, vel, tg, h, ds, dtype=np.float32)
        for pair in (False, True):       # k_target_a_flat<PAIR>
            drrt_mod.options.pair_grid = pair
            try:
                xt2, vt2, d2 = T.trace_target(_t(rif, gpu), rif.shape, _t(pos, gpu), _t(vel, gpu), _t(tg, gpu), h, ds)
            finally:
                drrt_mod.options.pair_grid = "auto"
            st = drrt_mod.read_stats()
            assert st["iters"] == ref2["iters"], pair
            assert np.array_equal(xt2.cpu().numpy(), ref2["xt"]) and np.array_equal(vt2.cpu().numpy(), ref2["vt"]), pair
            assert np.array_equal(d2.cpu().numpy(), ref2["dist2"]), pair
    ref64 = oracle.trace_target(rif, rif.shape, pos, vel, tg, h, ds, dtype=np.float64)
    assert np.mean(np.abs(d2.cpu().numpy() - ref64["dist2"]) <= 1e-5) >= 0.99


def test_sdf_variants(gpu, oracle, drrt_mod):
    R, span = 33, 1.0
    h = span / (R - 1); ds = h / 2
    rif = cases.luneburg(R)
    sdf = cases.sphere_sdf(R, span, 0.4)
    # rays start INSIDE the sdf<0 region (trace_sdf's `inside` is the sdf sign after step 1)
    rng = np.random.default_rng(2)
    d = rng.normal(size=(4000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = (0.5 * span + 0.2 * span * rng.uniform(0, 1, (4000, 1)) ** (1 / 3) * d).astype(np.float32)
    v = rng.normal(size=(4000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    vel = v.astype(np.float32)
    drrt_mod.options.sort_rays = True
    T = drrt_mod.TracerC()
    xt, vt = T.trace_sdf(_t(rif, gpu), _t(sdf, gpu), rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    with oracle.arith("factored"):
        ref = oracle.trace(rif, rif.shape, pos, vel, h, ds, dtype=np.float32, mode="sdf", sdf=sdf)
    assert np.array_equal(xt.cpu().numpy(), ref["xt"]) and np.array_equal(vt.cpu().numpy(), ref["vt"])
    dx = rng.normal(size=pos.shape).astype(np.float32); dv = rng.normal(size=pos.shape).astype(np.float32)
    g = T.backtrace_sdf(_t(rif, gpu), _t(sdf, gpu), rif.shape, xt, vt, _t(dx, gpu), _t(dv, gpu), h, ds)
    st = drrt_mod.read_stats()
    with oracle.arith("factored"):
        refb = oracle.backtrace(rif, rif.shape, ref["xt"], ref["vt"], dx, dv, h, ds, dtype=np.float32, sdf=sdf)
    assert st["ray_steps"] == refb["steps_total"]
    assert cases.rel_l2(g.cpu().numpy(), refb["grad"]) <= 2e-5
    for window in ("box", "ring"):                           # both window kernels have the sdf end condition (MODE 1)
        with drrt_mod.using(adjoint_window=window):
            gw = T.backtrace_sdf(_t(rif, gpu), _t(sdf, gpu), rif.shape, xt, vt, _t(dx, gpu), _t(dv, gpu), h, ds)
            assert drrt_mod.read_stats()["ray_steps"] == refb["steps_total"], window
            assert cases.rel_l2(gw.cpu().numpy(), refb["grad"]) <= 2e-5, window


def test_cable_variants(gpu, oracle, drrt_mod):
    rres, radius, length = 65, 1.0, 8.0
    ds = radius / rres / 2                                  # core/fiber_opt.py:156
    r = np.linspace(0, 1, rres)
    prof = np.sqrt(2.0 - r ** 2).astype(np.float32)         # Luneburg-like GRIN profile
    rng = np.random.default_rng(4)
    n = 3000
    ang = rng.uniform(0, 2 * np.pi, n); rad = 0.8 * radius * np.sqrt(rng.uniform(0, 1, n))
    pos = np.stack([radius + rad * np.cos(ang), np.full(n, 0.37 * ds), radius + rad * np.sin(ang)], -1).astype(np.float32)
    vel = rng.normal(0, 0.05, (n, 3)); vel[:, 1] = 1.0
    vel = (vel / np.linalg.norm(vel, axis=1, keepdims=True)).astype(np.float32)
    pos[0] = [radius, 0.37 * ds, radius]; vel[0] = [0, 1, 0]               # on the axis: r < eps branch
    tg = np.stack([np.full(n, radius), np.full(n, 0.75 * length), np.full(n, radius)], -1).astype(np.float32)
    T = drrt_mod.TracerC()
    xt, vt, d2 = T.trace_cable(_t(prof, gpu), radius, length, _t(pos, gpu), _t(vel, gpu), _t(tg, gpu), ds)
    st = drrt_mod.read_stats()
    with oracle.arith("factored"):
        ref = oracle.trace_cable(prof, radius, length, pos, vel, tg, ds, dtype=np.float32)
    assert np.array_equal(xt.cpu().numpy(), ref["xt"]) and np.array_equal(vt.cpu().numpy(), ref["vt"])
    assert np.array_equal(d2.cpu().numpy(), ref["dist2"]) and st["ray_steps"] == ref["steps_total"]
    dx = rng.normal(size=pos.shape).astype(np.float32); dv = rng.normal(size=pos.shape).astype(np.float32)
    g = T.backtrace_cable(_t(prof, gpu), radius, length, xt, vt, _t(dx, gpu), _t(dv, gpu), ds)
    with oracle.arith("factored"):
        refb = oracle.backtrace_cable(prof, radius, length, ref["xt"], ref["vt"], dx, dv, ds, dtype=np.float32)
    assert cases.rel_l2(g.cpu().numpy(), refb["grad"]) <= 2e-5
    # fp32-vs-fp64 spread of the ORACLE ITSELF on this case: 2.2e-2 (literal) / 3.2e-2 (factored) -- the
    # +-(grad.rhat)/h splat into a 65-bin profile is ill-conditioned in fp32; the fp32 bar is the line above
    ref64 = oracle.backtrace_cable(prof, radius, length, ref["xt"], ref["vt"], dx, dv, ds, dtype=np.float64)
    assert cases.rel_l2(g.cpu().numpy(), ref64["grad"]) <= 1e-1


def test_autograd_function_contract(gpu, oracle, drrt_mod):
    """core/tracer.py:294-335 contract: apply(rif,x,v,h,ds) -> (xt,vt); backward gives drif only;
    h, ds arrive as numpy float64 scalars (core/luneburg_opt.py:87, :48-49)."""
    from adjointnonlinearraytracing_amd import tracer
    R, span = 17, 20.0
    h = span / np.maximum(R - 1, 1); ds = h / 2
    rif_np = cases.luneburg(R, span=1.0)
    pos, vel = cases.plane_rays(2000, span, ds, seed=3)
    rif = _t(rif_np, gpu).requires_grad_(True)
    x = _t(pos, gpu).requires_grad_(True)
    xt, vt = tracer.BackTracerC.apply(rif, x, _t(vel, gpu), h, ds)
    assert xt.shape == (2000, 3) and vt.shape == (2000, 3) and xt.device == rif.device
    loss = ((xt - 0.5 * span) ** 2).sum() / 2000 / span + vt.sum() * 0.01
    loss.backward()
    assert rif.grad is not None and rif.grad.shape == rif.shape
    assert x.grad is None
    gx = (2 * (xt.detach() - 0.5 * span) / 2000 / span).cpu().numpy()
    gv = np.full_like(gx, 0.01)
    with oracle.arith("factored"):
        ref = oracle.backtrace(rif_np, rif_np.shape, xt.detach().cpu().numpy(), vt.detach().cpu().numpy(),
                               gx, gv, h, ds, dtype=np.float32)
    assert cases.rel_l2(rif.grad.cpu().numpy().ravel(), ref["grad"]) <= 2e-5


def test_errors_and_edge_cases(gpu, drrt_mod):
    T = drrt_mod.TracerC()
    rif = torch.ones(8, 8, 8, device=gpu)
    x = torch.rand(10, 3, device=gpu); v = torch.rand(10, 3, device=gpu)
    with pytest.raises(RuntimeError, match="Resolution doesn't match data"):     # src/volume.cpp:37
        T.trace(rif, (8, 8, 9), x, v, 1.0, 0.5)
    with pytest.raises(RuntimeError, match="invalid resolution"):                # src/volume.cpp:124
        T.trace(torch.ones(8, 1, 1, device=gpu), (1, 1, 8), x, v, 1.0, 0.5)
    # a non-positive / non-finite step would make the reference's max_steps expression undefined: refused
    for bad in ((1.0, 0.0), (1.0, -0.5), (0.0, 0.5), (float("nan"), 0.5), (1.0, float("inf"))):
        with pytest.raises(RuntimeError, match="positive and finite"):
            T.trace(rif, rif.shape, x, v, *bad)
    # empty ray set
    e = torch.empty(0, 3, device=gpu)
    xt, vt = T.trace(rif, rif.shape, e, e, 1.0, 0.5)
    assert xt.shape == (0, 3)
    g = T.backtrace(rif, rif.shape, e, e, e, e, 1.0, 0.5)
    assert g.shape == (512,) and float(g.abs().sum()) == 0.0
    # ragged: n not a multiple of the block; rays that never enter keep xt=pos, vt=vel (Q6)
    pos = torch.tensor([[-1.0, 3.0, 3.0], [3.0, 3.0, 3.0], [20.0, 3.0, 3.0]], device=gpu)
    vel = torch.tensor([[-1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [1.0, 0.0, 0.0]], device=gpu)
    xt, vt = T.trace(rif, rif.shape, pos, vel, 1.0, 0.5)
    assert torch.equal(xt[0], pos[0]) and torch.equal(xt[2], pos[2])
    assert float(xt[1, 0]) == 7.0
    # non-contiguous / strided views are accepted (core/luneburg_opt.py:57 builds them with cat/split)
    big = torch.rand(40, 6, device=gpu) * 6
    xt2, _ = T.trace(rif, rif.shape, big[:, :3], big[:, 3:], 1.0, 0.5)
    xt3, _ = T.trace(rif, rif.shape, big[:, :3].contiguous(), big[:, 3:].contiguous(), 1.0, 0.5)
    assert torch.equal(xt2, xt3)
    # CPU tensors are refused by TracerC (no silent fallback) and served by TracerS via the GPU
    with pytest.raises(RuntimeError):
        T.trace(rif.cpu(), rif.shape, x.cpu(), v.cpu(), 1.0, 0.5)
    xs, vs = drrt_mod.TracerS().trace(rif.cpu(), rif.shape, pos.cpu(), vel.cpu(), 1.0, 0.5)
    assert xs.device.type == "cpu" and torch.equal(xs, xt.cpu())


def test_shard_sum_equals_single(gpu, drrt_mod):
    """the sharded adjoint is a pure sum -- k shards accumulated with
    DRRT_FLAG_NO_ZERO semantics must equal the single-shard grid up to fp32 atomic ordering."""
    R, span = 33, 1.0
    h = span / (R - 1); ds = h / 2
    rif = _t(cases.smooth_field(R, seed=9), gpu)
    pos, vel = cases.cube_rays(1000, span, ds, seed=21)
    T = drrt_mod.TracerC()
    xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    dx = torch.randn_like(xt); dv = torch.randn_like(vt)
    full = T.backtrace(rif, rif.shape, xt, vt, dx, dv, h, ds)
    parts = sum(T.backtrace(rif, rif.shape, xt[s], vt[s], dx[s], dv[s], h, ds)
                for s in (slice(0, 1500), slice(1500, 4200), slice(4200, None)))
    assert cases.rel_l2(parts.cpu().numpy(), full.cpu().numpy()) <= 1e-5


# ---------------------------------------------------------------------------------------------
# committed golden fixtures (tests/golden, generated from the reference's torch helpers)
# ---------------------------------------------------------------------------------------------
def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_golden_luneburg_cube(gpu, oracle, drrt_mod):
    """Rays from core/source.py rand_rays_cube+random_rotate_ic, (grad_x, grad_v) from the
    core/sensor.py plane loss: the HIP path reproduces the stored float64 exit rays / dL/dn within
    the fp32 bar and the factored fp32 oracle bit-exactly."""
    z = _golden("luneburg_cube.npz")
    rif, h, ds = z["rif"], float(z["h"]), float(z["ds"])
    span = h * (rif.shape[0] - 1)
    drrt_mod.options.sort_rays = True
    T = drrt_mod.TracerC()
    xt, vt = T.trace(_t(rif, gpu), rif.shape, _t(z["x"], gpu), _t(z["v"], gpu), h, ds)
    with oracle.arith("factored"):
        o32 = oracle.trace(rif, rif.shape, z["x"], z["v"], h, ds, dtype=np.float32)
    assert np.array_equal(xt.cpu().numpy(), o32["xt"]) and np.array_equal(vt.cpu().numpy(), o32["vt"])
    d = np.linalg.norm(xt.cpu().numpy() - z["xt"], axis=1)
    assert np.mean(d <= 2e-5 * span) >= 0.99
    gx, gv = z["grad_x"].astype(np.float32), z["grad_v"].astype(np.float32)
    for corrected, key in ((False, "drif"), (True, "drif_corrected")):
        drrt_mod.options.corrected_h = corrected
        try:
            g = T.backtrace(_t(rif, gpu), rif.shape, _t(z["xt"].astype(np.float32), gpu),
                            _t(z["vt"].astype(np.float32), gpu), _t(gx, gpu), _t(gv, gpu), h, ds)
        finally:
            drrt_mod.options.corrected_h = False
        assert cases.rel_l2(g.cpu().numpy(), z[key]) <= 2e-2


def test_golden_fuel_injection(gpu, oracle, drrt_mod):
    """Real data: data/fuel_injection_64.npy padded to 65^3 (core/fuel_injection_opt.py:40-43)."""
    z = _golden("fuel_injection.npz")
    vol, h, ds = z["vol"], float(z["h"]), float(z["ds"])
    drrt_mod.options.sort_rays = True
    xt, vt = drrt_mod.TracerC().trace(_t(vol, gpu), vol.shape, _t(z["x"], gpu), _t(z["v"], gpu), h, ds)
    with oracle.arith("factored"):
        o32 = oracle.trace(vol, vol.shape, z["x"], z["v"], h, ds, dtype=np.float32)
    assert np.array_equal(xt.cpu().numpy(), o32["xt"]) and np.array_equal(vt.cpu().numpy(), o32["vt"])
    bad, unexplained = cases.step_flip_report(xt.cpu().numpy(), vt.cpu().numpy(), z["xt"], z["vt"], ds, tol=2e-5)
    assert unexplained <= 0.002 and bad <= 0.05


def test_non_cubic_grid(gpu, oracle, drrt_mod):
    """res = (W,H,D) all different; flat index (z*H + y)*W + x as src/volume.cpp:134-141 writes it."""
    W, H, D = 20, 14, 9
    h, ds = 0.25, 0.125
    rng = np.random.default_rng(5)
    rif = (1.0 + 0.3 * rng.random(W * H * D)).astype(np.float32)
    ext = np.array([(W - 1) * h, (H - 1) * h, (D - 1) * h])
    pos = (rng.random((5000, 3)) * ext * 1.2 - 0.1 * ext).astype(np.float32)
    vel = rng.normal(size=(5000, 3)); vel = (vel / np.linalg.norm(vel, axis=1, keepdims=True)).astype(np.float32)
    res = (W, H, D)
    T = drrt_mod.TracerC()
    for sort in (True, False):
        drrt_mod.options.sort_rays = sort
        xt, vt = T.trace(_t(rif, gpu), res, _t(pos, gpu), _t(vel, gpu), h, ds)
        with oracle.arith("factored"):
            o = oracle.trace(rif, res, pos, vel, h, ds, dtype=np.float32)
        assert np.array_equal(xt.cpu().numpy(), o["xt"]) and np.array_equal(vt.cpu().numpy(), o["vt"])
        dx = rng.normal(size=pos.shape).astype(np.float32); dv = rng.normal(size=pos.shape).astype(np.float32)
        g = T.backtrace(_t(rif, gpu), res, xt, vt, _t(dx, gpu), _t(dv, gpu), h, ds)
        st = drrt_mod.read_stats()
        with oracle.arith("factored"):
            ob = oracle.backtrace(rif, res, o["xt"], o["vt"], dx, dv, h, ds, dtype=np.float32)
        assert st["ray_steps"] == ob["steps_total"] and cases.rel_l2(g.cpu().numpy(), ob["grad"]) <= 2e-5


def test_full_size_properties(gpu, drrt_mod):
    """BASELINE.json's full size (256^3 grid, 1M rays, ~512 steps) through size-independent properties:
    determinism of the forward march, every ray exits with the expected step count, linearity of the
    adjoint in (dx, dv), shard-sum == whole, window kernel == direct atomics, sum(dL/dn) == 0 in a
    uniform medium."""
    import bench
    R, n = 256, 1024 * 1024
    rif, pos, vel, h, ds = bench.make_workload(R, n, gpu, seed=0)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    xt, vt = T.trace(rif, rif.shape, pos, vel, h, ds)
    st = drrt_mod.read_stats()
    order = drrt_mod.keep_order(drrt_mod.last_order)          # held across the two marches below
    # straight corner rays take 510-511 steps, rays through the ball (|v| = n > 1) fewer
    assert st["n_failed"] == 0 and 470 * n <= st["ray_steps"] <= 515 * n and 505 <= st["iters"] < 2048
    xt2, vt2 = T.trace(rif, rif.shape, pos, vel, h, ds)
    assert torch.equal(xt, xt2) and torch.equal(vt, vt2)                     # deterministic, order-independent
    drrt_mod.options.sort_rays = False
    xt3, _ = T.trace(rif, rif.shape, pos, vel, h, ds)
    drrt_mod.options.sort_rays = True
    assert torch.equal(xt, xt3)
    assert float((xt[:, 1] >= 1.0 - 1e-6).float().mean()) > 0.99            # exits through the far face
    dx1, dv1 = torch.randn_like(xt), torch.randn_like(vt)
    dx2, dv2 = torch.randn_like(xt), torch.randn_like(vt)
    g1 = T.backtrace(rif, rif.shape, xt, vt, dx1, dv1, h, ds, order=order)
    g2 = T.backtrace(rif, rif.shape, xt, vt, dx2, dv2, h, ds, order=order)
    g12 = T.backtrace(rif, rif.shape, xt, vt, 2 * dx1 - 3 * dx2, 2 * dv1 - 3 * dv2, h, ds, order=order)
    lin = 2 * g1 - 3 * g2
    assert float((g12 - lin).norm() / lin.norm()) <= 2e-5                    # linear in the seed
    half = n // 2
    ga = T.backtrace(rif, rif.shape, xt[:half], vt[:half], dx1[:half], dv1[:half], h, ds)
    gb = T.backtrace(rif, rif.shape, xt[half:], vt[half:], dx1[half:], dv1[half:], h, ds)
    assert float((ga + gb - g1).norm() / g1.norm()) <= 2e-5                  # shard sum == whole (multi-GPU reduction)
    sub = slice(0, n, 16)
    drrt_mod.options.direct_atomics = True
    try:
        gd = T.backtrace(rif, rif.shape, xt[sub], vt[sub], dx1[sub], dv1[sub], h, ds)
    finally:
        drrt_mod.options.direct_atomics = False
    gw = T.backtrace(rif, rif.shape, xt[sub], vt[sub], dx1[sub], dv1[sub], h, ds)
    assert float((gw - gd).norm() / gd.norm()) <= 2e-5                       # LDS windows == direct atomics
    uni = torch.full_like(rif, 1.25)
    xu, vu = T.trace(uni, uni.shape, pos, vel, h, ds)
    gu = T.backtrace(uni, uni.shape, xu, vu, dx1, dv1, h, ds)
    assert abs(float(gu.double().sum())) <= 1e-4 * float(gu.double().abs().sum())


def test_fp16_ray_state_mode(gpu, drrt_mod):
    """BASELINE config 5: fp16 ray state + fp32 adjoint accumulate (no counterpart in the reference,
    which is fp32-only, include/types.h:36-46).  Half inputs are widened exactly, the march is the
    fp32 march, outputs are rounded once: trace_f16io == half(trace_f32(float(inputs))) bit for bit;
    the adjoint from half inputs equals the fp32 adjoint from the same (widened) inputs."""
    R, span = 65, 1.0
    h = span / (R - 1); ds = h / 2
    rif = _t(cases.luneburg(R), gpu)
    pos, vel = cases.cube_rays(3000, span, ds, seed=17, tilt=0.2)
    pos16, vel16 = _t(pos, gpu).half(), _t(vel, gpu).half()
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    xt16, vt16 = T.trace(rif, rif.shape, pos16, vel16, h, ds)
    st16 = drrt_mod.read_stats()
    assert xt16.dtype == torch.float16 and vt16.dtype == torch.float16
    xt32, vt32 = T.trace(rif, rif.shape, pos16.float(), vel16.float(), h, ds)
    st32 = drrt_mod.read_stats()
    assert torch.equal(xt16, xt32.half()) and torch.equal(vt16, vt32.half()) and st16 == st32
    dx16, dv16 = torch.randn_like(xt16), torch.randn_like(vt16)
    g16 = T.backtrace(rif, rif.shape, xt16, vt16, dx16, dv16, h, ds)
    assert g16.dtype == torch.float32
    g32 = T.backtrace(rif, rif.shape, xt16.float(), vt16.float(), dx16.float(), dv16.float(), h, ds)
    assert cases.rel_l2(g16.cpu().numpy(), g32.cpu().numpy()) <= 2e-5


def test_calls_follow_the_current_torch_stream(gpu, drrt_mod):
    """Every launch, memset and sort of a call goes to the caller's stream: results on a side stream (with the
    inputs produced on that stream just before) equal the default-stream results."""
    R, span, n = 33, 1.0, 5000
    h = span / (R - 1); ds = h / 2
    rif_np = cases.smooth_field(R, seed=8)
    pos, vel = cases.plane_rays(n, span, ds, seed=4)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    rif = _t(rif_np, gpu)
    ref = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    gref = T.backtrace(rif, rif.shape, ref[0], ref[1], torch.ones_like(ref[0]), torch.ones_like(ref[0]), h, ds)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(side):
        p2 = _t(pos, gpu) * 1.0                       # produced on the side stream
        v2 = _t(vel, gpu) * 1.0
        out = T.trace(rif, rif.shape, p2, v2, h, ds)
        g = T.backtrace(rif, rif.shape, out[0], out[1], torch.ones_like(out[0]), torch.ones_like(out[0]), h, ds)
    side.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    assert cases.rel_l2(g.cpu().numpy(), gref.cpu().numpy()) <= 2e-6


def test_order_hint_lifetime_and_range_check(gpu, drrt_mod):
    """ADVICE r1 (order hint): (i) a call that fails validation consumes the hint -- the next successful call with
    the same n sorts for itself and is correct; (ii) a hint with out-of-range entries cannot make the kernels
    touch memory outside the ray arrays: those slots stay unvisited, every valid slot gets its exact result;
    (iii) trace_target ignores the hint (its state buffer would overlay an order living in the workspace)."""
    import ctypes as C
    from adjointnonlinearraytracing_amd import _lib
    lib = _lib.load()
    R, span, n = 33, 1.0, 3000
    h = span / (R - 1); ds = h / 2
    rif = _t(cases.luneburg(R), gpu)
    pos, vel = cases.cube_rays(n // 6, span, ds, seed=5)
    n = len(pos)
    pos_t, vel_t = _t(pos, gpu), _t(vel, gpu)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    xt0, vt0 = T.trace(rif, rif.shape, pos_t, vel_t, h, ds)
    dx = torch.ones_like(xt0); dv = torch.full_like(xt0, 0.1)
    g0 = T.backtrace(rif, rif.shape, xt0, vt0, dx, dv, h, ds)
    # (i) arm a hint full of garbage, fail a call, then run the same-n call: must match the unhinted result
    garbage = torch.full((n,), 0x7FFFFFF0, dtype=torch.int32, device=gpu)
    lib.drrt_set_order_hint(C.c_void_p(garbage.data_ptr()), n)
    with pytest.raises(RuntimeError, match="Resolution doesn't match data"):
        T.trace(rif, (R, R, R + 1), pos_t, vel_t, h, ds)
    assert lib.drrt_order_hint_pending() == 0
    xt1, vt1 = T.trace(rif, rif.shape, pos_t, vel_t, h, ds)
    assert torch.equal(xt1, xt0) and torch.equal(vt1, vt0)
    # (ii) a permutation with some entries replaced by out-of-range indices
    perm = torch.randperm(n, device=gpu).to(torch.int32)
    bad = perm.clone()
    hole = torch.arange(0, n, 7, device=gpu)
    bad[hole] = n + 12345
    skipped = perm[hole].long()
    xt2 = torch.full_like(xt0, -7.0); vt2 = torch.full_like(vt0, -7.0)
    st = torch.zeros(3, dtype=torch.int64, device=gpu)
    ws = torch.empty(int(lib.drrt_workspace_bytes_grid(n, rif.numel(), 1)), dtype=torch.uint8, device=gpu)
    res = (C.c_int * 3)(R, R, R)
    p = lambda t: C.c_void_p(t.data_ptr())
    lib.drrt_set_order_hint(p(bad), n)
    _lib.check(lib.drrt_trace_f32(p(rif), rif.numel(), res, n, p(pos_t), p(vel_t), h, ds, p(xt2), p(vt2), p(st),
                                  p(ws), ws.numel(), 1, None))
    torch.cuda.synchronize()
    keep = torch.ones(n, dtype=torch.bool, device=gpu); keep[skipped] = False
    assert torch.equal(xt2[keep], xt0[keep]) and torch.equal(vt2[keep], vt0[keep])
    assert bool((xt2[~keep] == -7.0).all())
    g2 = torch.empty_like(g0)
    lib.drrt_set_order_hint(p(bad), n)
    _lib.check(lib.drrt_backtrace_f32(p(rif), rif.numel(), res, n, p(xt0), p(vt0), p(dx), p(dv), h, ds, p(g2), p(st),
                                      p(ws), ws.numel(), 1, None))
    dxk = dx.clone(); dvk = dv.clone()
    # reference: the same adjoint over the kept rays only (skipped rays moved out of the box so they contribute nothing)
    xt_far = xt0.clone(); xt_far[~keep] = -50.0
    vt_far = vt0.clone(); vt_far[~keep] = torch.tensor([1.0, 0.0, 0.0], device=gpu)
    g_ref = T.backtrace(rif, rif.shape, xt_far, vt_far, dxk, dvk, h, ds)
    assert cases.rel_l2(g2.cpu().numpy(), g_ref.cpu().numpy()) <= 2e-5
    # (iii) trace_target with a hint pointing INTO the workspace region its state buffer overwrites
    tg = _t(np.tile(np.array([[0.5, 1.3, 0.5]], np.float32), (n, 1)), gpu)
    a = T.trace_target(rif, rif.shape, pos_t, vel_t, tg, h, ds)
    order = drrt_mod.last_order
    assert order is not None
    cnt = C.c_size_t(0)
    inws = lib.drrt_last_order(C.byref(cnt))
    lib.drrt_set_order_hint(C.c_void_p(inws), n)
    b = T.trace_target(rif, rif.shape, pos_t, vel_t, tg, h, ds)
    assert all(torch.equal(u, w) for u, w in zip(a, b))


def test_last_order_is_a_view_and_a_stale_one_is_ignored(gpu, oracle, drrt_mod):
    """drrt.last_order is a view into the forward call's workspace (no device-to-device copy per call, round-3 review):
    handed straight to the paired adjoint it is used as is; once another march has rewritten that workspace region the
    view is recognised as stale and the adjoint sorts for itself -- same gradient, never a wrong visit order;
    keep_order() gives a private copy that stays valid."""
    R, span = 33, 1.0
    h = span / (R - 1); ds = h / 2
    rif_np = cases.smooth_field(R, seed=2)
    rif = _t(rif_np, gpu)
    pos, vel = cases.cube_rays(700, span, ds, seed=3, tilt=0.2)
    pos_o, vel_o = cases.cube_rays(700, span, ds, seed=4, tilt=0.2)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    view = drrt_mod.last_order
    ws = drrt_mod._workspaces[drrt_mod._wkey(rif.device)]
    assert view is not None and ws.data_ptr() <= view.data_ptr() < ws.data_ptr() + ws.numel()       # no copy was made
    assert drrt_mod._valid_order(view) is view and getattr(view, "drrt_steps", None) is not None
    kept = drrt_mod.keep_order(view)
    assert kept.data_ptr() != view.data_ptr() and torch.equal(kept, view) and torch.equal(kept.drrt_steps, view.drrt_steps)
    ones = torch.ones_like(xt)
    g_view = T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=view)
    assert drrt_mod._valid_order(view) is view                      # an adjoint that was handed the order does not rewrite it
    T.trace(rif, rif.shape, _t(pos_o, gpu), _t(vel_o, gpu), h, ds)  # another march: the region now holds ITS order
    assert drrt_mod._valid_order(view) is None and drrt_mod.keep_order(view) is None
    g_stale = T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=view)
    g_kept = T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=kept)
    with oracle.arith("factored"):
        ob = oracle.backtrace(rif_np, rif_np.shape, xt.cpu().numpy(), vt.cpu().numpy(), np.ones_like(pos), np.ones_like(pos),
                              h, ds, dtype=np.float32)
    for g in (g_view, g_stale, g_kept):
        assert cases.rel_l2(g.cpu().numpy(), ob["grad"]) <= 2e-5


def test_plane_second_pass_on_plain_grid_and_pair_copy(gpu, oracle, drrt_mod):
    """trace_pln must flag the rays that can record a LATER exit (start past the plane, head back through it) so
    that the second pass re-marches them: both gather forms of k_trace_flat<., 1> against the oracle on such rays."""
    R, span = 17, 1.0
    h = span / (R - 1); ds = h / 2
    rif = cases.smooth_field(R, seed=9)
    rng = np.random.default_rng(3)
    n = 600
    pos = (rng.random((n, 3)) * 0.9 + 0.05).astype(np.float32) * span
    pos[:, 1] = 0.8 + 0.15 * rng.random(n).astype(np.float32)          # start PAST the plane y = 0.6 ...
    vel = rng.normal(size=(n, 3)).astype(np.float32) * 0.3
    vel[:, 1] = -np.abs(vel[:, 1]) - 0.7                                # ... heading back through it
    vel /= np.linalg.norm(vel, axis=1, keepdims=True)
    po = np.tile(np.array([[0.5, 0.6, 0.5]], np.float32) * span, (n, 1))
    pd = np.tile(np.array([[0, 1, 0]], np.float32), (n, 1))
    with oracle.arith("factored"):
        ref = oracle.trace(rif, rif.shape, pos, vel, h, ds, dtype=np.float32, mode="plane", pln_o=po, pln_d=pd)
    T = drrt_mod.TracerC()
    for sort in (False, True):
        drrt_mod.options.sort_rays = sort
        for pair in (False, True):
            drrt_mod.options.pair_grid = pair
            try:
                xt, vt, fm = T.trace_pln(_t(rif, gpu), rif.shape, _t(pos, gpu), _t(vel, gpu), _t(po, gpu), _t(pd, gpu), h, ds)
            finally:
                drrt_mod.options.pair_grid = "auto"
            assert np.array_equal(xt.cpu().numpy(), ref["xt"]) and np.array_equal(vt.cpu().numpy(), ref["vt"]), (sort, pair)
            assert np.array_equal(fm.cpu().numpy().astype(bool), ref["failmask"]), (sort, pair)
    drrt_mod.options.sort_rays = True


def test_failed_ray_warning_is_asynchronous_but_not_lost(gpu, drrt_mod, capsys):
    """src/tracer.cpp:89-90 prints "failed to exit all rays" when a ray is still live after max_steps.  The mirror
    prints the same line without a host sync per call: it may appear late, never get lost."""
    T = drrt_mod.TracerC()
    rif = torch.ones(8, 8, 8, device=gpu)
    pos = torch.tensor([[3.0, 3.0, 3.0], [3.0, 3.0, 3.0]], device=gpu)
    vel = torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0]], device=gpu)      # ray 0 is at rest: it can never exit
    drrt_mod.options.check_failed = True
    try:
        T.trace(rif, rif.shape, pos, vel, 1.0, 0.5)
        drrt_mod.flush_warnings()
        assert capsys.readouterr().out.count("failed to exit all rays") == 1
        T.trace(rif, rif.shape, pos[1:], vel[1:], 1.0, 0.5)                 # every ray exits: silent
        drrt_mod.flush_warnings()
        assert "failed" not in capsys.readouterr().out
    finally:
        drrt_mod.options.check_failed = False


@pytest.mark.parametrize("kind,R,step_res", [("luneburg", 65, 2), ("smooth", 33, 0.7), ("smooth", 5, 1.3)])
def test_adjoint_kernel_variants_agree(gpu, oracle, drrt_mod, kind, R, step_res):
    """The window kernels of drrt_backtrace_f32 -- k_backtrace_flat (box window), k_backtrace_ring (ring window; with the
    forward's visit order, and with the step hint that starts its rays on the forward march's clock), the device-side
    choice between the two, both sort keys -- and the one-atomic-per-tap kernel (DRRT_FLAG_DIRECT_ATOMICS: no windows, no
    register accumulators) run the same per-ray arithmetic (adj_sample / adj_contrib = adj_step): equal step counts,
    gradients equal up to the fp32 summation order, and each within 2e-5 of the oracle.  Steps larger than a cell and a 5^3 grid exercise the multi-face jumps and the
    clamped boundary cells; unsorted rays exercise the global-atomic fallback."""
    import ctypes as C
    from adjointnonlinearraytracing_amd import _lib
    lib = _lib.load()
    span = 1.0
    h = span / (R - 1); ds = h / step_res
    rif_np = _scene(kind, R) if kind != "smooth" else cases.smooth_field(R, seed=3)
    pos, vel = cases.cube_rays(1500, span, ds, seed=2, tilt=0.3)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    rif = _t(rif_np, gpu)
    xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    n = xt.shape[0]
    rng = np.random.default_rng(5)
    dx = _t(rng.normal(size=(n, 3)).astype(np.float32), gpu); dv = _t(rng.normal(size=(n, 3)).astype(np.float32), gpu)
    with oracle.arith("factored"):
        ob = oracle.backtrace(rif_np, rif_np.shape, xt.cpu().numpy(), vt.cpu().numpy(), dx.cpu().numpy(), dv.cpu().numpy(),
                              h, ds, dtype=np.float32)
    res = (C.c_int * 3)(R, R, R)
    p = lambda t: C.c_void_p(t.data_ptr())
    rif_flat = rif.reshape(-1).contiguous()
    grads = {}
    # the forward march's visit order and per-ray iteration counts, as a binding would hand them to the adjoint
    fsteps = getattr(drrt_mod.last_order, "drrt_steps", None)
    assert drrt_mod.last_order is not None and fsteps is not None and fsteps.numel() == n
    variants = [("flat", _lib.FLAG_STATIC_WINDOW), ("auto", 0), ("ring", _lib.FLAG_RING_WINDOW),
                ("ring_sparse", _lib.FLAG_RING_WINDOW | _lib.FLAG_RING_SPARSE),
                ("ring_direct", _lib.FLAG_RING_WINDOW | _lib.FLAG_RING_SPARSE | _lib.FLAG_RING_DIRECT),
                ("ring_chord", _lib.FLAG_RING_WINDOW | _lib.FLAG_CHORD_KEY), ("flat_chord", _lib.FLAG_STATIC_WINDOW | _lib.FLAG_CHORD_KEY),
                ("direct", _lib.FLAG_DIRECT_ATOMICS)]
    for name, fl in variants:
        for sort in (1, 0, "hint", "hint+steps"):
            flags = fl | (0 if sort == 0 else 1)
            ws = torch.empty(int(lib.drrt_workspace_bytes_grid(n, rif_flat.numel(), flags)) + 1024, dtype=torch.uint8, device=gpu)
            g = torch.empty_like(rif_flat)
            st = torch.zeros(3, dtype=torch.int64, device=gpu)
            if isinstance(sort, str):
                if not name.startswith(("ring", "auto")):
                    continue
                lib.drrt_set_order_hint(p(drrt_mod.last_order), n)
                if sort == "hint+steps":
                    lib.drrt_set_step_hint(p(fsteps), n)
            _lib.check(lib.drrt_backtrace_f32(p(rif_flat), rif_flat.numel(), res, n, p(xt), p(vt), p(dx), p(dv), h, ds, p(g), p(st),
                                              p(ws), ws.numel(), flags, None))
            assert lib.drrt_order_hint_pending() == 0                       # both hints are consumed by the call
            torch.cuda.synchronize()
            assert int(st[0]) == ob["steps_total"], (name, sort)
            assert cases.rel_l2(g.cpu().numpy(), ob["grad"]) <= 2e-5, (name, sort)
            grads[(name, sort)] = g.cpu().numpy()
    base = grads[("flat", 1)]
    for k, g in grads.items():
        assert cases.rel_l2(g, base) <= 2e-5, k


@pytest.mark.parametrize("seed", [0, 1])
def test_ring_window_with_rays_that_run_out_of_steps_beside_delayed_lanes(gpu, oracle, drrt_mod, seed):
    """Round-3 advisor finding: in k_backtrace_ring a ray that uses up its max_steps iterations while other lanes of its
    wave are still marching (step hint: lanes start up to 96 iterations apart) hands its cell over on its own; the flag
    that guards the wave's cooperative flushes must be wave-uniform again before the service of that same iteration.
    Slow rays (they never leave the volume, so they DO run out of steps) mixed into the waves of ordinary rays, ds << h
    so that windows stay clean for long stretches, arbitrary per-ray delays, ring kernel forced: step totals equal the
    oracle's and the gradient is within the summation-order bound; equal to the box kernel's and the direct kernel's."""
    import ctypes as C
    from adjointnonlinearraytracing_amd import _lib
    lib = _lib.load()
    R, span = 33, 1.0
    h = span / (R - 1); ds = h / 6
    rif_np = cases.smooth_field(R, seed=11)
    rng = np.random.default_rng(40 + seed)
    n = 4096
    xt = (rng.uniform(0.15, 0.85, (n, 3)) * span).astype(np.float32)
    vt = rng.normal(size=(n, 3)).astype(np.float32)
    vt /= np.linalg.norm(vt, axis=1, keepdims=True)
    slow = rng.random(n) < 0.3
    vt[slow] *= 0.002                                  # max_steps * ds * |v| << span: these rays end by running out of steps
    dx = rng.normal(size=(n, 3)).astype(np.float32); dv = rng.normal(size=(n, 3)).astype(np.float32)
    with oracle.arith("factored"):
        ob = oracle.backtrace(rif_np, rif_np.shape, xt, vt, dx, dv, h, ds, dtype=np.float32)
    max_steps = int(np.float32(2.0) * np.float32(h) * np.float32(R) / np.float32(ds))
    assert ob["steps_total"] >= int(slow.sum()) * max_steps          # the slow rays really use all their iterations
    res = (C.c_int * 3)(R, R, R)
    p = lambda t: C.c_void_p(t.data_ptr())
    rif = _t(rif_np, gpu).reshape(-1).contiguous()
    xt_d, vt_d, dx_d, dv_d = (_t(a, gpu) for a in (xt, vt, dx, dv))
    order = torch.arange(n, dtype=torch.int32, device=gpu)           # caller order: slow and ordinary rays share waves
    fsteps = _t(rng.integers(0, 300, n).astype(np.int32), gpu)       # arbitrary "forward iteration counts" -> delays 0..96
    grads = {}
    for name, fl, hint in (("ring+steps", _lib.FLAG_RING_WINDOW, True), ("ring", _lib.FLAG_RING_WINDOW, False),
                           ("ring_sparse+steps", _lib.FLAG_RING_WINDOW | _lib.FLAG_RING_SPARSE, True),
                           ("ring_sparse", _lib.FLAG_RING_WINDOW | _lib.FLAG_RING_SPARSE, False),
                           ("ring_direct+steps", _lib.FLAG_RING_WINDOW | _lib.FLAG_RING_SPARSE | _lib.FLAG_RING_DIRECT, True),
                           ("ring_direct", _lib.FLAG_RING_WINDOW | _lib.FLAG_RING_SPARSE | _lib.FLAG_RING_DIRECT, False),
                           ("box", _lib.FLAG_STATIC_WINDOW, False), ("direct", _lib.FLAG_DIRECT_ATOMICS, False)):
        flags = fl | _lib.FLAG_SORT_RAYS
        ws = torch.empty(int(lib.drrt_workspace_bytes_grid(n, rif.numel(), flags)) + 1024, dtype=torch.uint8, device=gpu)
        g = torch.empty_like(rif)
        st = torch.zeros(3, dtype=torch.int64, device=gpu)
        lib.drrt_set_order_hint(p(order), n)
        if hint:
            lib.drrt_set_step_hint(p(fsteps), n)
        _lib.check(lib.drrt_backtrace_f32(p(rif), rif.numel(), res, n, p(xt_d), p(vt_d), p(dx_d), p(dv_d), h, ds, p(g), p(st),
                                          p(ws), ws.numel(), flags, None))
        torch.cuda.synchronize()
        assert int(st[0]) == ob["steps_total"], name
        grads[name] = g.cpu().numpy()
        assert cases.rel_l2(grads[name], ob["grad"]) <= 2e-5, name
    for name in ("ring+steps", "ring", "ring_sparse+steps", "ring_sparse", "ring_direct+steps", "ring_direct", "box"):
        assert cases.rel_l2(grads[name], grads["direct"]) <= 2e-5, name


@pytest.mark.parametrize("sort", [True, False])
def test_chunked_adjoint_equals_the_one_launch_adjoint(gpu, oracle, drrt_mod, sort):
    """drrt_backtrace_chunk_f32 / TracerC.backtrace_chunked (include/drrt_hip.h): the adjoint march in K depth chunks --
    state handed from launch to launch, every accumulator handed over and every window flushed at a chunk's end -- gives
    the gradient of the one-launch march (same per-ray contributions; src/tracer.cpp:384-440) for K = 1, 3, 4, 7, with
    equal step totals, and the progress block of every chunk is consistent with where the rays are."""
    R, span = 33, 1.0
    h = span / (R - 1); ds = h / 2
    rif_np = cases.luneburg(R)
    rif = _t(rif_np, gpu)
    pos, vel = cases.plane_rays(5000, span, ds, seed=7, axis=1, tilt=0.05)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = sort
    try:
        xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
        order = drrt_mod.keep_order(drrt_mod.last_order)
        assert (order is not None) == sort
        rng = np.random.default_rng(3)
        dx = _t(rng.normal(size=pos.shape).astype(np.float32), gpu); dv = _t(rng.normal(size=pos.shape).astype(np.float32), gpu)
        g_one = T.backtrace(rif, rif.shape, xt, vt, dx, dv, h, ds, order=order)
        st_one = drrt_mod.read_stats()
        with oracle.arith("factored"):
            ob = oracle.backtrace(rif_np, rif_np.shape, xt.cpu().numpy(), vt.cpu().numpy(), dx.cpu().numpy(), dv.cpu().numpy(),
                                  h, ds, dtype=np.float32)
        assert st_one["ray_steps"] == ob["steps_total"]
        for K in (1, 3, 4, 7):
            seen = []
            g = T.backtrace_chunked(rif, rif.shape, xt, vt, dx, dv, h, ds, order=order, chunks=K,
                                    on_chunk=lambda k, grad, prog: seen.append(drrt_mod.decode_chunk_progress(prog)))
            st = drrt_mod.read_stats()
            assert st["ray_steps"] == ob["steps_total"], K
            assert cases.rel_l2(g.cpu().numpy(), ob["grad"]) <= 2e-5, K
            assert cases.rel_l2(g.cpu().numpy(), g_one.cpu().numpy()) <= 2e-5, K
            assert len(seen) == K
            # the rays travel towards -y in the adjoint: the deepest still-marching position can only move down, and every
            # chunk's samples lie at or below the previous chunk's
            ymax = [p["pos_max"][1] for p in seen if p["active"]]
            assert all(b <= a + 1e-6 for a, b in zip(ymax, ymax[1:])), (K, ymax)
            smax = [p["sample_max"][1] for p in seen if p["sample_max"] is not None]
            assert all(b <= a + 1e-6 for a, b in zip(smax, smax[1:])), (K, smax)
            assert seen[0]["sample_max"][1] <= float(xt[:, 1].max()) + 1e-6
            assert seen[-1]["active"] == 0                   # every ray has left the volume well before max_steps
    finally:
        drrt_mod.options.sort_rays = True


def test_chunked_adjoint_refuses_inconsistent_calls(gpu, drrt_mod):
    """A resumed chunk of a sorted march without the visit order of its first chunk, a state buffer that is too small and
    the one-atomic-per-tap mode are refused with a message (never a silently different gradient)."""
    import ctypes as C
    from adjointnonlinearraytracing_amd import _lib
    lib = _lib.load()
    R, n = 9, 300
    h = 1.0 / (R - 1); ds = h / 2
    rif = _t(cases.smooth_field(R, seed=1), gpu).reshape(-1).contiguous()
    pos, vel = cases.plane_rays(n, 1.0, ds, seed=1, axis=1, tilt=0.0)
    xt, vt = _t(pos, gpu), _t(vel, gpu)
    n = xt.shape[0]
    ones = torch.ones_like(xt)
    res = (C.c_int * 3)(R, R, R)
    p = lambda t: C.c_void_p(t.data_ptr())
    ws = torch.empty(int(lib.drrt_workspace_bytes_grid(n, rif.numel(), 1)) + 1024, dtype=torch.uint8, device=gpu)
    state = torch.empty(int(lib.drrt_backtrace_chunk_state_bytes(n)), dtype=torch.uint8, device=gpu)
    g = torch.empty_like(rif)

    def call(flags, it_begin, state_bytes=None):
        return lib.drrt_backtrace_chunk_f32(p(rif), rif.numel(), res, n, p(xt), p(vt), p(ones), p(ones), h, ds, p(g), None,
                                            p(ws), ws.numel(), flags, None, p(state),
                                            state.numel() if state_bytes is None else state_bytes, it_begin, 4, None)
    assert call(1, 0) == 0
    assert call(1, 4) == _lib.ERR_ARG and b"visit order" in lib.drrt_last_error()
    lib.drrt_set_order_hint(lib.drrt_last_order(None), n)
    assert call(1, 4) == 0
    assert call(0, 0, state_bytes=16) == _lib.ERR_ARG and b"state buffer" in lib.drrt_last_error()
    assert call(_lib.FLAG_DIRECT_ATOMICS, 0) == _lib.ERR_ARG
    assert lib.drrt_backtrace_max_steps(res, h, ds) == int(np.float32(2.0) * np.float32(h) * np.float32(R) / np.float32(ds))
    torch.cuda.synchronize()


def test_bundle_classification_picks_the_adjoint_kernel(gpu, oracle, drrt_mod):
    """drrt_last_bundle_counters (include/drrt_hip.h; csrc/drrt_march.h: bundles_want_ring / bundles_long /
    bundles_want_sparse): the ring-window kernel when a fifth of the sampled bundles' start cells do not fit the box window
    OR when 7.5 % of them left the forward march 12 or more cells of travel apart (24 iterations at ds = h / 2; counter [6], from the step hint); then its
    sparse-only instantiation, unless the call pinned the general one (DRRT_FLAG_RING_GENERAL -> counter [5]).  A sparse
    six-view set goes to the ring kernel; forced either way
    (A-B flags) and chosen by the counters the gradient is the oracle's, and the Python mirror's reading of the counters
    (read_bundle_counters) states the library's rule."""
    R, span = 65, 1.0
    h = span / (R - 1); ds = h / 2
    rif_np = cases.luneburg(R)
    rif = _t(rif_np, gpu)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    picked = {}
    for name, (pos, vel) in (("sparse", cases.cube_rays(2500, span, ds, seed=5, tilt=0.4)),
                             ("dense", cases.plane_rays(60000, span, ds, seed=6, axis=1, tilt=0.0))):
        xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
        order = drrt_mod.keep_order(drrt_mod.last_order)
        assert getattr(order, "drrt_steps", None) is not None
        ones = torch.ones_like(xt)
        with oracle.arith("factored"):
            ob = oracle.backtrace(rif_np, rif_np.shape, xt.cpu().numpy(), vt.cpu().numpy(), np.ones_like(pos), np.ones_like(pos),
                                  h, ds, dtype=np.float32)
        for mode in ("auto", "ring", "ring_sparse", "ring_direct", "ring_general", "box"):
            with drrt_mod.using(adjoint_window=mode):
                g = T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=order)
                c = drrt_mod.read_bundle_counters()
            assert drrt_mod.read_stats()["ray_steps"] == ob["steps_total"], (name, mode)
            assert cases.rel_l2(g.cpu().numpy(), ob["grad"]) <= 2e-5, (name, mode)
            if mode == "auto":
                assert c is not None and c["bundles"] > 0
                long_ = c["bundles_long"] > 0 and c["bundles_long"] * 1000 >= c["bundles"] * c["long_threshold_permille"]
                nofit = c["bundles_not_fitting"] > 0 and c["bundles_not_fitting"] * 100 >= c["bundles"] * c["ring_threshold_pct"]
                few = c["lanes"] > 0 and c["start_pair_share"] * 100 < c["direct_threshold_pct"]
                assert c["kernel"] == (("ring_direct" if few else "ring_sparse") if (long_ or nofit) else "box"), (name, c)
                picked[name] = c["kernel"]
            elif mode == "ring_general":                          # classifies, but never the sparse-only instantiation
                assert c is not None and c["kernel"] in ("box", "ring"), (name, c)
            else:
                assert c is None, (name, mode, c)                 # a forced kernel does not classify
        # without the step hint the counter of long bundles stays empty: only the start cells decide
        bare = order.clone()
        g = T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=bare)
        c = drrt_mod.read_bundle_counters()
        assert c is not None and c["bundles_long"] == 0, c
        assert cases.rel_l2(g.cpu().numpy(), ob["grad"]) <= 2e-5
    # the criterion is a LENGTH (iterations * ds / h): the same rays marched at half the step read about the same share
    pos, vel = cases.cube_rays(2500, span, ds, seed=5, tilt=0.4)
    share = []
    for dsx in (ds, ds / 2):
        xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, dsx)
        order = drrt_mod.keep_order(drrt_mod.last_order)
        T.backtrace(rif, rif.shape, xt, vt, torch.ones_like(xt), torch.ones_like(xt), h, dsx, order=order)
        share.append(drrt_mod.read_bundle_counters()["long_bundle_share"])
    assert abs(share[0] - share[1]) <= 0.2, share
    # (the dense source at this size: whatever its counters say, asserted above; the metric's plane source at full size
    # stays with the box window: tests/test_baseline_configs.py, tests/test_bench_contract.py)
    assert picked["sparse"] in ("ring_sparse", "ring_direct"), picked


@pytest.mark.parametrize("case", ["wide_range", "growing", "zero_seeds", "one_nan"])
def test_fixed_point_window_of_the_sparse_ring_kernel(gpu, oracle, drrt_mod, case):
    """The sparse-only instantiation of k_backtrace_ring keeps its window in 32-bit fixed point with one exponent per wave
    (csrc/drrt_adjoint_ring.hip): scale chosen at the first contributing step, re-chosen when the window is empty and the
    magnitudes have drifted, hand-overs out of range sent to the grid, a complete flush before any slot could overflow.
    Against the oracle (and the general, fp64-window instantiation) with adjoint seeds that stress exactly that:
      wide_range  seeds spread over 9 decades inside every wave (the small ones lie below the wave's quantum: the GLOBAL
                  rel-L2 bound holds, and nothing is off by more than a quantum of the largest);
      growing     a strongly focusing medium and long marches: lambda / mu grow by orders of magnitude along a ray;
      zero_seeds  nothing to accumulate: the scale is never set, every hand-over is zero;
      one_nan     one ray with NaN seeds: its own contributions are NaN, every voxel it does not touch is as without it."""
    R, span = 65, 1.0
    h = span / (R - 1); ds = h / 2
    rng = np.random.default_rng(77)
    rif_np = cases.luneburg(R) if case != "growing" else (1.0 + 1.5 * np.exp(-8.0 * ((np.indices((R, R, R), dtype=np.float32) / (R - 1) - 0.5) ** 2).sum(0))).astype(np.float32)
    pos, vel = cases.cube_rays(1500, span, ds, seed=9, tilt=0.35)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    rif = _t(rif_np, gpu)
    xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    order = drrt_mod.keep_order(drrt_mod.last_order)
    n = xt.shape[0]
    dx = rng.normal(size=(n, 3)).astype(np.float32); dv = rng.normal(size=(n, 3)).astype(np.float32)
    if case == "wide_range":
        sc = (10.0 ** rng.uniform(-6, 3, size=(n, 1))).astype(np.float32)
        dx *= sc; dv *= sc
    if case == "zero_seeds":
        dx[:] = 0; dv[:] = 0
    if case == "one_nan":
        dx[n // 2] = np.nan
    xt_n, vt_n = xt.cpu().numpy(), vt.cpu().numpy()
    with oracle.arith("factored"):
        ob = oracle.backtrace(rif_np, rif_np.shape, xt_n, vt_n, dx, dv, h, ds, dtype=np.float32)
    grads = {}
    for mode in ("ring_sparse", "ring_direct", "ring"):
        with drrt_mod.using(adjoint_window=mode):
            grads[mode] = T.backtrace(rif, rif.shape, xt, vt, _t(dx, gpu), _t(dv, gpu), h, ds, order=order).cpu().numpy()
        assert drrt_mod.read_stats()["ray_steps"] == ob["steps_total"], mode
    ref = ob["grad"]
    if case == "zero_seeds":
        assert not grads["ring_sparse"].any() and not grads["ring_direct"].any() and not grads["ring"].any()
        return
    if case == "one_nan":
        ok = np.isfinite(ref)
        assert (~ok).any() and np.array_equal(np.isfinite(grads["ring_sparse"]), ok)
        assert np.array_equal(np.isfinite(grads["ring_direct"]), ok)
        with oracle.arith("factored"):            # the same march without that ray: the finite voxels must agree with it
            keep = np.arange(n) != n // 2
            ob2 = oracle.backtrace(rif_np, rif_np.shape, xt_n[keep], vt_n[keep], dx[keep], dv[keep], h, ds, dtype=np.float32)
        assert cases.rel_l2(grads["ring_sparse"][ok], ob2["grad"][ok]) <= 2e-5
        assert cases.rel_l2(grads["ring_direct"][ok], ob2["grad"][ok]) <= 2e-5
        return
    for mode in ("ring_sparse", "ring_direct", "ring"):
        assert cases.rel_l2(grads[mode], ref) <= 2e-5, (case, mode, cases.rel_l2(grads[mode], ref))
    for mode in ("ring_sparse", "ring_direct"):
        err = np.abs(grads[mode].astype(np.float64) - ref)
        assert err.max() <= 2e-4 * np.abs(ref).max(), (case, mode, err.max(), np.abs(ref).max())


def test_q16_ray_state_mode(gpu, drrt_mod):
    """16-bit ray state "q16" (include/drrt_hip.h): trace_q16io / backtrace_q16io widen exactly, march in fp32 and round
    once -- trace_q16io(enc(x), enc(v)) == enc(trace_f32(dec(enc(x)), dec(enc(v)))) bit for bit, and the adjoint from
    q16 exit rays + half seeds equals the fp32 adjoint fed the decoded / widened arrays (same arithmetic, fp32 sums)."""
    R, span, n = 65, 1.0, 30000
    h = span / (R - 1); ds = h / 2
    rif = _t(cases.luneburg(R), gpu)
    pos, vel = cases.cube_rays(n // 6, span, ds, seed=8, tilt=0.2)
    pos, vel = _t(pos, gpu), _t(vel, gpu)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    xq, vq = drrt_mod.encode_rays16(rif.shape, h, pos, vel)
    x0, v0 = drrt_mod.decode_rays16(rif.shape, h, xq, vq)
    # codes round-trip exactly; decode(encode(.)) is within half a code of the input; saturation at the range ends
    xq2, vq2 = drrt_mod.encode_rays16(rif.shape, h, x0, v0)
    assert torch.equal(xq2, xq) and torch.equal(vq2, vq)
    far = torch.tensor([[-50.0, 0.5, 50.0]], device=gpu); big = torch.tensor([[3.0, -3.0, float("nan")]], device=gpu)
    fq, bq = drrt_mod.encode_rays16(rif.shape, h, far, big)
    assert fq.view(torch.uint16).tolist()[0][0] == 0 and fq.view(torch.uint16).tolist()[0][2] == 65535
    assert bq.tolist()[0] == [32767, -32768, -32768]
    xt_q, vt_q = T.trace(rif, rif.shape, xq, vq, h, ds)
    st_q = drrt_mod.read_stats()
    order = drrt_mod.keep_order(drrt_mod.last_order)
    xt_f, vt_f = T.trace(rif, rif.shape, x0, v0, h, ds)
    assert drrt_mod.read_stats() == st_q
    ex, ev = drrt_mod.encode_rays16(rif.shape, h, xt_f, vt_f)
    assert torch.equal(xt_q, ex) and torch.equal(vt_q, ev)
    dx = torch.randn(xt_q.shape, device=gpu).half(); dv = torch.randn(xt_q.shape, device=gpu).half()
    g_q = T.backtrace(rif, rif.shape, xt_q, vt_q, dx, dv, h, ds, order=order)
    xd, vd = drrt_mod.decode_rays16(rif.shape, h, xt_q, vt_q)
    g_f = T.backtrace(rif, rif.shape, xd, vd, dx.float(), dv.float(), h, ds, order=order)
    assert cases.rel_l2(g_q.cpu().numpy(), g_f.cpu().numpy()) <= 2e-5
    with pytest.raises(RuntimeError, match="float16 seeds"):
        T.backtrace(rif, rif.shape, xt_q, vt_q, dx.float(), dv.float(), h, ds)


def test_fitted_windows_for_oblique_views(gpu):
    """A plane view oblique to the grid with few rays per voxel column: the 64-ray bundles start the adjoint staggered on
    an oblique exit face and do not fit the default 9^3 gradient window.  The call must pick the kernel with run-time
    window dimensions (debug counter [3]), a dense axis-aligned view must not, and both must return the gradient of
    the one-atomic-per-tap kernel and of the compile-time-window kernel (fp32 summation order only)."""
    import ctypes as C
    from adjointnonlinearraytracing_amd import _lib, source
    lib = _lib.load()
    R = 96
    span = 1.0; h = span / (R - 1); ds = h / 2
    g = torch.linspace(0.0, 1.0, R, device=gpu)
    z, y, x = torch.meshgrid(g, g, g, indexing="ij")
    rif = (1.0 + 0.05 * torch.exp(-((x - 0.45) ** 2 + (y - 0.55) ** 2 + (z - 0.5) ** 2) / 0.03)).contiguous()
    nvox = rif.numel(); res = (C.c_int * 3)(R, R, R)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    fitted_waves = {}
    for ang, pix in ((0.0, 2 * R), (40.0, R)):                       # 16 / 4 rays per voxel column
        xs, vs, _ = source.plane_source3_rand(torch.tensor(ang), (pix, pix), 4, span, sensor_dist=0.2 * span, device=gpu)
        xs, vs = xs.contiguous(), vs.contiguous()
        nr = xs.shape[0]
        flags = _lib.FLAG_SORT_RAYS
        ws = torch.empty(int(lib.drrt_workspace_bytes_grid(nr, nvox, flags)), dtype=torch.uint8, device=gpu)
        xt, vt = torch.empty_like(xs), torch.empty_like(vs)
        st = torch.zeros(3, dtype=torch.int64, device=gpu)
        _lib.check(lib.drrt_trace_f32(p(rif), nvox, res, nr, p(xs), p(vs), h, ds, p(xt), p(vt), p(st), p(ws), ws.numel(),
                                      flags, stream))
        cnt = C.c_size_t(0)
        src = lib.drrt_last_order(C.byref(cnt))
        assert cnt.value == nr
        order = torch.empty(nr, dtype=torch.int32, device=gpu)
        torch.cuda.synchronize()
        wsi = ws.view(torch.int32)
        off = (src - ws.data_ptr()) // 4
        order.copy_(wsi[off:off + nr])
        gen = torch.Generator().manual_seed(3)
        dx = torch.randn(nr, 3, generator=gen).to(gpu); dv = torch.randn(nr, 3, generator=gen).to(gpu)
        grads = {}
        for name, fl in (("auto", flags | _lib.FLAG_DEBUG_COUNTERS), ("static", flags | _lib.FLAG_STATIC_WINDOW),
                         ("direct", flags | _lib.FLAG_DIRECT_ATOMICS)):
            grad = torch.empty(nvox, dtype=torch.float32, device=gpu)
            lib.drrt_set_order_hint(p(order), nr)
            _lib.check(lib.drrt_backtrace_f32(p(rif), nvox, res, nr, p(xt), p(vt), p(dx), p(dv), h, ds, p(grad), p(st),
                                              p(ws), ws.numel(), fl, stream))
            torch.cuda.synchronize()
            grads[name] = grad.cpu().numpy().astype(np.float64)
            if name == "auto":
                o = (ws.numel() - 512) & ~7
                fitted_waves[ang] = int(ws[o:o + 512].view(torch.int64)[3])
        assert cases.rel_l2(grads["auto"], grads["direct"]) <= 2e-6, ang
        assert cases.rel_l2(grads["static"], grads["direct"]) <= 2e-6, ang
    assert fitted_waves[0.0] == 0, fitted_waves                       # dense axis-aligned bundles fit the default window
    assert fitted_waves[40.0] > 0, fitted_waves                       # the oblique view runs with fitted windows


@pytest.mark.parametrize("ds_div", [2.0, 0.9])
def test_fitted_windows_sweep(gpu, ds_div):
    """Views at several angles about both axes, dense and sparse, half-cell and multi-cell steps, on a 48^3 grid: whatever
    window variant the call picks, the gradient equals the one-atomic-per-tap kernel's (a wider sweep -- 126 cases on
    48^3 / 96^3 / 130^3 grids, 122 of them with fitted windows -- was run once during development: worst 2.2e-7)."""
    import ctypes as C
    from adjointnonlinearraytracing_amd import _lib, source
    lib = _lib.load()
    R = 48
    span = 1.0; h = span / (R - 1); ds = h / ds_div
    g = torch.linspace(0.0, 1.0, R, device=gpu)
    z, y, x = torch.meshgrid(g, g, g, indexing="ij")
    rif = (1.0 + 0.08 * torch.exp(-((x - 0.45) ** 2 + (y - 0.55) ** 2 + (z - 0.5) ** 2) / 0.03)).contiguous()
    nvox = rif.numel(); res = (C.c_int * 3)(R, R, R)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    fitted = 0
    for k, (ang, pix) in enumerate(((17.0, 24), (17.0, 96), (61.0, 48), (123.0, 24), (123.0, 48), (88.0, 96))):
        xs, vs, _ = source.plane_source3_rand(torch.tensor(ang), (pix, pix), 2, span, xaxis=bool(k % 2),
                                              sensor_dist=0.2 * span, device=gpu)
        xs, vs = xs.contiguous(), vs.contiguous()
        nr = xs.shape[0]
        flags = _lib.FLAG_SORT_RAYS
        ws = torch.empty(int(lib.drrt_workspace_bytes_grid(nr, nvox, flags)), dtype=torch.uint8, device=gpu)
        xt, vt = torch.empty_like(xs), torch.empty_like(vs)
        st = torch.zeros(3, dtype=torch.int64, device=gpu)
        _lib.check(lib.drrt_trace_f32(p(rif), nvox, res, nr, p(xs), p(vs), h, ds, p(xt), p(vt), p(st), p(ws), ws.numel(),
                                      flags, stream))
        cnt = C.c_size_t(0)
        src = lib.drrt_last_order(C.byref(cnt))
        torch.cuda.synchronize()
        off = (src - ws.data_ptr()) // 4
        order = ws.view(torch.int32)[off:off + nr].clone()
        gen = torch.Generator().manual_seed(int(ang) + pix)
        dx = torch.randn(nr, 3, generator=gen).to(gpu); dv = torch.randn(nr, 3, generator=gen).to(gpu)
        grads = {}
        for name, fl in (("auto", flags | _lib.FLAG_DEBUG_COUNTERS), ("direct", flags | _lib.FLAG_DIRECT_ATOMICS)):
            grad = torch.empty(nvox, dtype=torch.float32, device=gpu)
            lib.drrt_set_order_hint(p(order), nr)
            _lib.check(lib.drrt_backtrace_f32(p(rif), nvox, res, nr, p(xt), p(vt), p(dx), p(dv), h, ds, p(grad), p(st),
                                              p(ws), ws.numel(), fl, stream))
            torch.cuda.synchronize()
            grads[name] = grad.cpu().numpy().astype(np.float64)
            if name == "auto":
                o = (ws.numel() - 512) & ~7
                fitted += int(ws[o:o + 512].view(torch.int64)[3]) > 0
        assert cases.rel_l2(grads["auto"], grads["direct"]) <= 2e-6, (ang, pix)
    assert fitted > 0


def test_call_sequence_is_graph_capturable(gpu, drrt_mod):
    """trace + paired backtrace issue only stream-ordered work (kernels, memsets) after their first call, so the
    sequence can be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed: same exit rays, same gradient.
    (Measured: no faster than eager -- 0.24 ms either way at 33^3 / 16k rays -- the short kernels, not the launches,
    are the time there.)"""
    drrt_mod.options.check_failed = False
    T = drrt_mod.TracerC()
    R = 33; h = 1.0 / (R - 1); ds = h / 2; n = 16384
    g = torch.linspace(0, 1, R, device=gpu)
    z, y, x = torch.meshgrid(g, g, g, indexing="ij")
    rif = (1.0 + 0.1 * torch.exp(-((x - .5) ** 2 + (y - .5) ** 2 + (z - .5) ** 2) / .05)).contiguous()
    gen = torch.Generator().manual_seed(0)
    pos = torch.rand(n, 3, generator=gen).to(gpu); pos[:, 1] = 0
    vel = torch.zeros(n, 3, device=gpu); vel[:, 1] = 1
    dx = torch.ones(n, 3, device=gpu); dv = torch.ones(n, 3, device=gpu)

    def step():
        xt, vt = T.trace(rif, (R, R, R), pos, vel, h, ds)
        return xt, T.backtrace(rif, (R, R, R), xt, vt, dx, dv, h, ds, order=drrt_mod.last_order)

    try:
        xt0, g0 = step()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            xt1, g1 = step()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(xt1, xt0)
        assert cases.rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) <= 2e-6
    finally:
        drrt_mod.options.check_failed = True


@pytest.mark.parametrize("blocks,extra", [(1, 0), (7, 13), (8, 0), (9, 1), (127, 255), (128, 0), (129, 77), (263, 5)])
def test_xcd_block_order_visits_every_ray_once(gpu, drrt_mod, blocks, extra):
    """xcd_block (csrc/drrt_kernels.hip) hands the launch's blocks the visit order XCD by XCD -- runs of 16 per XCD in the
    forward march and in the ring-window adjoint.  A remap that is not a bijection for some block count would leave rays
    unmarched or march them twice: block counts around the multiples of 8 and of 128 (the remap's group size), with a
    ragged last block.  Forward: bit-identical to DRRT_FLAG_DISPATCH_IN_ORDER and to the unsorted
    call; adjoint (ring kernel forced, with both hints): same step total and the same gradient up to summation order."""
    import ctypes as C
    from adjointnonlinearraytracing_amd import _lib
    lib = _lib.load()
    R = 17; span = 1.0; h = span / (R - 1); ds = h / 2
    rif_np = cases.luneburg(R)
    n_want = (blocks - 1) * 256 + (extra if extra else 256)
    pos, vel = cases.cube_rays(n_want // 6 + 1, span, ds, seed=blocks, tilt=0.2)
    pos, vel = pos[:n_want], vel[:n_want]
    n = pos.shape[0]
    assert n == n_want and (n + 255) // 256 == blocks
    rif = _t(rif_np, gpu).reshape(-1).contiguous()
    res = (C.c_int * 3)(R, R, R)
    p = lambda t: C.c_void_p(t.data_ptr())
    xs, vs = _t(pos, gpu), _t(vel, gpu)
    outs = {}
    for name, flags in (("unsorted", 0), ("in_order", 1 | _lib.FLAG_DISPATCH_IN_ORDER), ("xcd", 1)):
        ws = torch.empty(int(lib.drrt_workspace_bytes_grid(n, rif.numel(), flags)) + 1024, dtype=torch.uint8, device=gpu)
        xt = torch.full((n, 3), float("nan"), device=gpu); vt = torch.full((n, 3), float("nan"), device=gpu)
        st = torch.zeros(3, dtype=torch.int64, device=gpu)
        _lib.check(lib.drrt_trace_f32(p(rif), rif.numel(), res, n, p(xs), p(vs), h, ds, p(xt), p(vt), p(st), p(ws), ws.numel(), flags, None))
        torch.cuda.synchronize()
        outs[name] = (xt.cpu().numpy(), vt.cpu().numpy(), int(st[0]))
        if name == "xcd":
            cnt = C.c_size_t(0)
            order = torch.empty(n, dtype=torch.int32, device=gpu)
            ptr = lib.drrt_last_order(C.byref(cnt))
            assert ptr and cnt.value == n
            off = ptr - ws.data_ptr()
            order.copy_(ws[off:off + 4 * n].view(torch.int32))
            ptr_s = lib.drrt_last_steps(C.byref(cnt))
            assert ptr_s and cnt.value == n
            off = ptr_s - ws.data_ptr()
            fsteps = ws[off:off + 4 * n].view(torch.int32).clone()
    for name in ("in_order", "xcd"):
        assert not np.isnan(outs[name][0]).any()
        assert np.array_equal(outs[name][0], outs["unsorted"][0]) and np.array_equal(outs[name][1], outs["unsorted"][1]), name
        assert outs[name][2] == outs["unsorted"][2], name
    xt, vt = _t(outs["xcd"][0], gpu), _t(outs["xcd"][1], gpu)
    dx, dv = torch.ones_like(xt), torch.ones_like(vt)
    grads = {}
    for name, flags in (("in_order", 1 | _lib.FLAG_RING_WINDOW | _lib.FLAG_DISPATCH_IN_ORDER), ("xcd", 1 | _lib.FLAG_RING_WINDOW)):
        ws = torch.empty(int(lib.drrt_workspace_bytes_grid(n, rif.numel(), flags)) + 1024, dtype=torch.uint8, device=gpu)
        g = torch.empty_like(rif)
        st = torch.zeros(3, dtype=torch.int64, device=gpu)
        lib.drrt_set_order_hint(p(order), n)
        lib.drrt_set_step_hint(p(fsteps), n)
        _lib.check(lib.drrt_backtrace_f32(p(rif), rif.numel(), res, n, p(xt), p(vt), p(dx), p(dv), h, ds, p(g), p(st), p(ws), ws.numel(), flags, None))
        torch.cuda.synchronize()
        grads[name] = (g.cpu().numpy(), int(st[0]))
    assert grads["xcd"][1] == grads["in_order"][1]
    assert cases.rel_l2(grads["xcd"][0], grads["in_order"][0]) <= 2e-5


def test_bundle_classification_counters(gpu, drrt_mod):
    """drrt_last_bundle_counters (include/drrt_hip.h): the adjoint's device-side choice between the box-window and the
    ring-window kernel is readable after the call.  Rays of an axis-aligned plane view on a 65^3 grid form compact
    bundles (box kernel); forcing a kernel skips the classification (no counters); the gradient does not depend on it."""
    R = 65; span = 1.0; h = span / (R - 1); ds = h / 2
    rif = _t(cases.luneburg(R), gpu)
    pos, vel = cases.plane_rays(40000, span, ds, seed=4, axis=1, tilt=0.0)
    T = drrt_mod.TracerC()
    drrt_mod.options.sort_rays = True
    xt, vt = T.trace(rif, rif.shape, _t(pos, gpu), _t(vel, gpu), h, ds)
    order = drrt_mod.last_order
    n = xt.shape[0]
    ones = torch.ones_like(xt)
    g_auto = T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=order)
    c = drrt_mod.read_bundle_counters()
    assert c is not None
    blocks = (n + 255) // 256
    assert 0 < c["bundles"] <= 4 * ((blocks + 15) // 16)
    assert 0 < c["lanes"] <= 64 * c["bundles"] and c["lanes_outside"] <= c["lanes"] and c["bundles_not_fitting"] <= c["bundles"]
    assert c["kernel"] == "box" and c["not_fitting_share"] < 0.2
    with drrt_mod.using(adjoint_window="ring"):
        g_ring = T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=order)
        assert drrt_mod.read_bundle_counters() is None
    assert cases.rel_l2(g_ring.cpu().numpy(), g_auto.cpu().numpy()) <= 2e-5
