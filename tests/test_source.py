"""Ray generation: core/source.py plane-source generators.

CPU tier: the numpy restatement (oracle/source_ref.py) against tests/golden/source_rays.npz, which was
produced by RUNNING the reference's rand_rays_cube / rand_rays_in_sphere / random_rotate_ic on
recorded torch.rand draws.  GPU tier: drrt_gen_plane_rays_f32 (through the python mirror
adjointnonlinearraytracing_amd.source) against the same fixture and against the restatement on larger
seeded inputs.

Tolerance: the ray COUNT per view and the ray order are exact; coordinates agree to 4 fp32 ulps of the
scene width (the two 3x3 products run through the host BLAS in the reference, whose accumulation order
is not specified; everything else is the same fp32 expression order).
"""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PIX, SPP = (12, 10), 2


def ulps(width):
    return 4 * np.finfo(np.float32).eps * width


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "source_rays.npz"))


@pytest.fixture(scope="module")
def SR():
    from oracle import source_ref
    return source_ref


def _close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    assert np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)), initial=0.0) <= tol


# ------------------------------------------------------------------------------------ CPU tier
def test_restatement_matches_reference_cube(gold, SR):
    (x, v, pl), nr = SR.views(gold["cube_u"], SR.cube_mats(), PIX, SPP, 20.0, circle=True, sensor_dist=0.0)
    assert nr == gold["cube_nrays"].tolist()
    _close(x, gold["cube_x"], ulps(20.0)); _close(v, gold["cube_v"], 0.0); _close(pl, gold["cube_planes"], ulps(20.0))
    xr, vr, plr = SR.rotate_ic(x, v, pl, 20.0, gold["cube_M"])
    _close(xr, gold["cube_xr"], ulps(20.0)); _close(vr, gold["cube_vr"], ulps(1.0))
    _close(plr, gold["cube_planes_r"], ulps(20.0))


def test_restatement_matches_reference_sphere_and_independent(gold, SR):
    (x, v, pl), nr = SR.views(gold["sph_u"], SR.sphere_mats(5, 300), PIX, SPP, 20.0, circle=False, sensor_dist=0.7)
    assert nr == gold["sph_nrays"].tolist() == [240] * 5
    _close(x, gold["sph_x"], ulps(20.0)); _close(v, gold["sph_v"], 0.0); _close(pl, gold["sph_planes"], ulps(20.0))
    (x, v, pl), nr = SR.views(gold["ind_u"], SR.sphere_mats(3, 360, xaxis=True), PIX, SPP, 0.3, circle=True,
                              sensor_dist=1.0, independent=True)
    assert nr == gold["ind_nrays"].tolist()
    _close(x, gold["ind_x"], ulps(0.3)); _close(v, gold["ind_v"], 0.0); _close(pl, gold["ind_planes"], ulps(1.3))


def test_restatement_matches_reference_point_sources(gold, SR):
    (x, v, pl), nr = SR.views(gold["pt_u"], SR.sphere_mats(4, 360), PIX, SPP, 20.0, circle=False, sensor_dist=0.5,
                              kind="point")
    assert nr == gold["pt_nrays"].tolist() == [240] * 4
    _close(x, gold["pt_x"], ulps(20.0)); _close(v, gold["pt_v"], ulps(1.0)); _close(pl, gold["pt_planes"], ulps(20.0))
    (x, v, pl), nr = SR.views(gold["ptc_u"], SR.sphere_mats(3, 270, xaxis=True), PIX, SPP, 3.0, circle=True,
                              sensor_dist=0.0, kind="point")
    assert nr == gold["ptc_nrays"].tolist()
    _close(x, gold["ptc_x"], ulps(3.0)); _close(v, gold["ptc_v"], ulps(1.0)); _close(pl, gold["ptc_planes"], ulps(3.0))


# ------------------------------------------------------------------------------------ GPU tier
@pytest.fixture(scope="module")
def S():
    from adjointnonlinearraytracing_amd import source
    return source


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.gpu
def test_hip_cube_matches_reference(gold, S):
    import torch
    (x, v, pl), nr = S.rand_rays_cube(PIX, SPP, 20.0, circle=True, offset=torch.from_numpy(gold["cube_u"]))
    assert nr == gold["cube_nrays"].tolist()
    _close(_np(x), gold["cube_x"], ulps(20.0)); _close(_np(v), gold["cube_v"], 0.0)
    _close(_np(pl), gold["cube_planes"], ulps(20.0))
    # fused random_rotate_ic == the reference's two-stage result; and == the stand-alone torch mirror
    (xr, vr, plr), nr2 = S.rand_rays_cube(PIX, SPP, 20.0, circle=True, offset=torch.from_numpy(gold["cube_u"]),
                                          rotmat=torch.from_numpy(gold["cube_M"]), span=20.0)
    assert nr2 == nr
    _close(_np(xr), gold["cube_xr"], ulps(20.0)); _close(_np(vr), gold["cube_vr"], ulps(1.0))
    _close(_np(plr), gold["cube_planes_r"], ulps(20.0))
    x2, v2, pl2 = S.random_rotate_ic(x, v, pl, 20.0, rotmat=torch.from_numpy(gold["cube_M"]))
    _close(_np(x2), gold["cube_xr"], ulps(20.0)); _close(_np(pl2), gold["cube_planes_r"], ulps(20.0))


@pytest.mark.gpu
def test_hip_sphere_and_independent_match_reference(gold, S):
    import torch
    (x, v, pl), nr = S.rand_rays_in_sphere(5, PIX, SPP, 20.0, angle_span=300, sensor_dist=0.7,
                                           offset=torch.from_numpy(gold["sph_u"]))
    assert nr == [240] * 5
    _close(_np(x), gold["sph_x"], ulps(20.0)); _close(_np(v), gold["sph_v"], 0.0)
    _close(_np(pl), gold["sph_planes"], ulps(20.0))
    (x, v, pl), nr = S.rand_rays_in_sphere(3, PIX, SPP, 0.3, circle=True, xaxis=True, sensor_dist=1.0, indep=True,
                                           offset=torch.from_numpy(gold["ind_u"]))
    assert nr == gold["ind_nrays"].tolist()
    _close(_np(x), gold["ind_x"], ulps(0.3)); _close(_np(pl), gold["ind_planes"], ulps(1.3))
    # single view entry point
    x1, v1, pl1 = S.plane_source3_rand(torch.tensor(0.0), PIX, SPP, 20.0, sensor_dist=0.7,
                                       offset=torch.from_numpy(gold["sph_u"][0]))
    _close(_np(x1), gold["sph_x"][:240], ulps(20.0))


@pytest.mark.gpu
def test_hip_point_sources_match_reference(gold, S, SR):
    import torch
    (x, v, pl), nr = S.rand_ptrays_in_sphere(4, PIX, SPP, 20.0, sensor_dist=0.5, offset=torch.from_numpy(gold["pt_u"]))
    assert nr == [240] * 4
    _close(_np(x), gold["pt_x"], ulps(20.0)); _close(_np(v), gold["pt_v"], ulps(1.0))
    _close(_np(pl), gold["pt_planes"], ulps(20.0))
    (x, v, pl), nr = S.rand_ptrays_in_sphere(3, PIX, SPP, 3.0, angle_span=270, circle=True, xaxis=True,
                                             sensor_dist=0.0, offset=torch.from_numpy(gold["ptc_u"]))
    assert nr == gold["ptc_nrays"].tolist()
    _close(_np(x), gold["ptc_x"], ulps(3.0)); _close(_np(v), gold["ptc_v"], ulps(1.0))
    _close(_np(pl), gold["ptc_planes"], ulps(3.0))
    x1, v1, pl1 = S.point_source3_rand(torch.tensor(0.0), PIX, SPP, 20.0, sensor_dist=0.5,
                                       offset=torch.from_numpy(gold["pt_u"][0]))
    _close(_np(v1), gold["pt_v"][:240], ulps(1.0))
    # at size, against the restatement (bit-exact: same fp32 operation order), with a fused random_rotate_ic
    rng = np.random.default_rng(8)
    u = rng.random((5, 6, 200, 173), dtype=np.float32)
    (x, v, pl), nr = S.rand_ptrays_in_sphere(5, (200, 173), 3, 7.0, angle_span=300, circle=True, sensor_dist=0.3,
                                             offset=torch.from_numpy(u))
    (xo, vo, plo), nro = SR.views(u, SR.sphere_mats(5, 300), (200, 173), 3, 7.0, circle=True, sensor_dist=0.3,
                                  kind="point")
    assert nr == nro
    assert np.array_equal(_np(x), xo) and np.array_equal(_np(v), vo) and np.array_equal(_np(pl), plo)


@pytest.mark.gpu
@pytest.mark.parametrize("pix,spp,circle", [((512, 512), 1, True), ((257, 131), 3, True), ((64, 64), 4, False),
                                            ((1, 1), 1, True), ((33, 1), 1, False)])
def test_hip_matches_restatement_at_size(S, SR, pix, spp, circle):
    """Compaction across many blocks / ragged tails / several views: bit-exact against the restatement
    (same fp32 operation order), which the fixture above pins to the reference."""
    import torch
    rng = np.random.default_rng(5)
    u = rng.random((6, 2 * spp, *pix), dtype=np.float32)
    (x, v, pl), nr = S.rand_rays_cube(pix, spp, 2.0, circle=circle, offset=torch.from_numpy(u))
    (xo, vo, plo), nro = SR.views(u, SR.cube_mats(), pix, spp, 2.0, circle=circle, sensor_dist=0.0)
    assert nr == nro
    assert np.array_equal(_np(x), xo) and np.array_equal(_np(v), vo) and np.array_equal(_np(pl), plo)


@pytest.mark.gpu
def test_hip_generated_rays_feed_the_march(S):
    """Device-drawn jitter (no offset given): every ray starts inside the cube's bounding sphere, points along
    its view direction and marches through the tracer."""
    import torch
    from adjointnonlinearraytracing_amd import drrt
    torch.manual_seed(0)
    span, R = 20.0, 33
    (x, v, pl), nr = S.rand_rays_cube((64, 64), 1, span, circle=True, rotmat=S.random_rotmat(), span=span)
    assert sum(nr) == x.shape[0] and x.is_cuda
    assert torch.allclose(v.norm(dim=1), torch.ones_like(v[:, 0]), atol=1e-5)
    assert ((x - span / 2).norm(dim=1) <= span / 2 * 1.4143).all()
    rif = torch.ones(R, R, R, device="cuda")
    h = span / (R - 1)
    xt, vt = drrt.TracerC().trace(rif.flatten(), rif.shape, x, v, h, h / 2)
    assert torch.isfinite(xt).all() and torch.allclose(vt, v, atol=1e-6)


@pytest.mark.gpu
def test_hip_source_argument_errors(S):
    import torch
    with pytest.raises(RuntimeError):
        S.rand_rays_cube((8, 8), 1, -1.0)
    with pytest.raises(RuntimeError):
        S.plane_source3_rand(0.0, (8, 8), 1, 1.0, device="cpu")


# ---------------------------------------------------------------------------------------- cone source
@pytest.fixture(scope="module")
def cone_gold():
    return np.load(os.path.join(G, "cone_rays.npz"))


def test_restatement_matches_reference_cone_source(cone_gold, SR):
    """core/source.py:186-203 cone_source3_rand + hatbox_sample (:531-545) RUN AS IS (fixture) vs the numpy restatement:
    single view (angle 35 deg, cone 100 deg) and rand_rays_cube(src_type='cone', cone_ang=60)."""
    g = cone_gold
    pix, spp, width = tuple(int(p) for p in g["pix"]), int(g["spp"]), float(g["width"])
    x, v, pl = SR.cone_view(g["single_u"], SR.view_matrix(np.float32(35.0), False), pix, spp, width, sensor_dist=0.7, cone_angle=100.0)
    _close(x, g["single_x"], ulps(width)); _close(v, g["single_v"], ulps(1.0)); _close(pl, g["single_planes"], ulps(width))
    # all directions lie inside the cone about R e_y
    axis = g["single_planes"][0, 1]
    assert np.all(g["single_v"] @ axis >= np.cos(np.deg2rad(50.0)) - 1e-6)
    (x, v, pl), nr = SR.views(g["cube_u"], SR.cube_mats(), pix, spp, width, sensor_dist=0.0, kind="cone", cone_angle=60)
    assert nr == g["cube_nrays"].tolist() == [pix[0] * pix[1] * spp] * 6
    _close(x, g["cube_x"], ulps(width)); _close(v, g["cube_v"], ulps(1.0)); _close(pl, g["cube_planes"], ulps(width))


@pytest.mark.gpu
def test_hip_cone_source_matches_reference_run(gpu, cone_gold, S):
    import torch
    g = cone_gold
    pix, spp, width = tuple(int(p) for p in g["pix"]), int(g["spp"]), float(g["width"])
    x, v, pl = S.cone_source3_rand(torch.tensor(35.0), pix, spp, width, sensor_dist=0.7, cone_angle=100.0,
                                   offset=torch.from_numpy(g["single_u"]), device=gpu)
    _close(_np(x), g["single_x"], ulps(width)); _close(_np(v), g["single_v"], 4 * ulps(1.0)); _close(_np(pl), g["single_planes"], ulps(width))
    (x, v, pl), nr = S.rand_rays_cube(pix, spp, width, src_type='cone', cone_ang=60, offset=torch.from_numpy(g["cube_u"]), device=gpu)
    assert nr == g["cube_nrays"].tolist()
    _close(_np(x), g["cube_x"], ulps(width)); _close(_np(v), g["cube_v"], 4 * ulps(1.0)); _close(_np(pl), g["cube_planes"], ulps(width))
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "point_rays.npz"))
    (x, v, pl), nr = S.rand_rays_cube((4, 3), 4, 1.0, src_type='point', cone_ang=50, device=gpu)   # torch ops on the device
    assert nr == z["cube_nrays"].tolist() and x.is_cuda
    _close(_np(x), z["cube_x"], 4 * ulps(1.0)); _close(_np(v), z["cube_v"], 8 * ulps(1.0)); _close(_np(pl), z["cube_planes"], 4 * ulps(1.0))


@pytest.mark.gpu
def test_hip_cone_source_full_size_statistics(gpu, S, SR):
    """Fibre-experiment size (core/fiber_opt.py:131): 1M rays in one call, bit-compared with the restatement on the
    same draws (cosf / sinf may differ from numpy's by an ulp: 4-ulp tolerance), unit directions inside the cone."""
    import torch
    pix, spp, width = (512, 512), 4, 2.0
    n = pix[0] * pix[1] * spp
    u = torch.rand(2, n, generator=torch.Generator().manual_seed(3))
    x, v, pl = S.cone_source3_rand(torch.tensor(0.0), pix, spp, width, sensor_dist=1.0, cone_angle=40.0, offset=u, device=gpu)
    assert x.shape == (n, 3) and v.shape == (n, 3) and pl.shape == (n, 3, 3)
    vn = _np(v)
    assert np.abs(np.linalg.norm(vn, axis=1) - 1.0).max() < 1e-6
    assert vn[:, 1].min() >= np.cos(np.deg2rad(20.0)) - 1e-6
    xr, vr, plr = SR.cone_view(u.numpy(), SR.view_matrix(np.float32(0.0), False), pix, spp, width, sensor_dist=1.0, cone_angle=40.0)
    _close(vn, vr, 4 * ulps(1.0)); _close(_np(x), xr, ulps(width))


# ---- area sources (core/source.py:107-183, :368-385) and sum_norm: the reference's torch expressions on any device ---
GA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "area_rays.npz")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_area_sources_match_reference_run(tag):
    """Fixture made by RUNNING area_source3_rand_bias / area_source3_cone with the host generator's draws recorded."""
    from adjointnonlinearraytracing_amd import source
    z = np.load(GA)
    ang, p0, p1, spp, width, sd, circle, xaxis = z[f"bias_{tag}_args"].tolist()
    kw = dict(circle=bool(circle), xaxis=bool(xaxis), sensor_dist=sd, device="cpu")
    pix = (int(p0), int(p1))
    (x, v, pl), xt, tpv = source.area_source3_rand_bias(torch.tensor(ang), pix, int(spp), width,
                                                       offset=torch.from_numpy(z[f"bias_{tag}_uoff"]),
                                                       tosense=torch.from_numpy(z[f"bias_{tag}_uts"]), **kw)
    for got, key in ((x, "x"), (v, "v"), (pl, "planes"), (xt, "xt"), (tpv, "tpv")):
        want = z[f"bias_{tag}_{key}"]
        assert got.shape == want.shape and np.abs(got.numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), key
    (x, v, pl), tpv = source.area_source3_cone(torch.tensor(ang), pix, int(spp), width, cone_angle=70.0,
                                               offset=torch.from_numpy(z[f"cone_{tag}_uoff"]),
                                               hatbox=torch.from_numpy(z[f"cone_{tag}_uhat"]), **kw)
    for got, key in ((x, "x"), (v, "v"), (pl, "planes"), (tpv, "tpv")):
        want = z[f"cone_{tag}_{key}"]
        assert got.shape == want.shape, key       # (case b: the disc mask of the reference rejects every sample of the
        if want.size:                             #  near-face source -- an empty ray set, reproduced as such)
            assert np.abs(got.numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), key


def test_area_view_collections_and_sum_norm():
    from adjointnonlinearraytracing_amd import source
    (x, v, pl), targets, dists, nrays = source.rand_area_in_sphere(3, (5, 4), 2, 1.0, angle_span=180, sensor_dist=0.3,
                                                                   device="cpu")
    assert nrays == [40, 40, 40] and x.shape == (120, 3) and pl.shape == (120, 3, 3) and targets.shape == (120, 3)
    assert torch.allclose(v.norm(dim=1), torch.ones(120), atol=1e-5) and dists.shape == (120,)
    (x, v, pl), dists, nrays = source.rand_cone_in_sphere(2, (4, 4), 3, 1.0, sensor_dist=0.0, cone_angle=60.0, device="cpu")
    assert nrays == [48, 48] and torch.allclose(v.norm(dim=1), torch.ones(96), atol=1e-5)
    im = torch.rand(6, 6) + 0.1
    out, sc = source.sum_norm(im, scale=True)
    assert abs(float(out.mean()) - 1.0) < 1e-6 and torch.allclose(out, source.sum_norm(im)) and float(sc) > 0


GP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "point_rays.npz")


def test_point_source_matches_reference_run():
    """Fixture made by RUNNING point_source3 / rand_rays_cube(src_type='point') (deterministic): same torch ops on the
    CPU, so the match is exact up to the last place of the rotation matmul."""
    from adjointnonlinearraytracing_amd import source
    z = np.load(GP)
    for tag in ("a", "b"):
        ang, p0, p1, spp, width, cone, xaxis, sd = z[f"{tag}_args"].tolist()
        x, v, pl = source.point_source3(torch.tensor(ang), (int(p0), int(p1)), int(spp), width, cone_angle=cone,
                                        xaxis=bool(xaxis), sensor_dist=sd)
        for got, key in ((x, "x"), (v, "v"), (pl, "planes")):
            want = z[f"{tag}_{key}"]
            assert got.shape == want.shape and np.abs(got.numpy() - want).max() <= 1e-6, key
    (x, v, pl), nrays = source.rand_rays_cube((4, 3), 4, 1.0, src_type='point', cone_ang=50, device="cpu")
    assert nrays == z["cube_nrays"].tolist()
    for got, key in ((x, "x"), (v, "v"), (pl, "planes")):
        assert np.abs(got.numpy() - z[f"cube_{key}"]).max() <= 1e-6, key
