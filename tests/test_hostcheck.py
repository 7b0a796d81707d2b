"""CPU tier: the product's own per-ray code (csrc/drrt_device.h: the __host__ __device__ step
functions and whole-ray drivers the HIP kernels call), compiled for the host by tests/hostcheck,
must agree BIT FOR BIT with the oracle's `factored` float32 arithmetic on trajectories, exit steps
and closest-approach records; adjoint grids agree up to summation order.  This also validates the
per-ray termination / two-phase trace_target restructuring against the reference's global-loop
semantics, which the oracle keeps."""
import numpy as np
import pytest

import cases
import hostcheck_lib as H
H_ = H


@pytest.fixture(scope="module")
def scene():
    R, span = 33, 1.0
    h = span / (R - 1); ds = h / 2
    pos, vel = cases.cube_rays(400, span, ds, seed=1, tilt=0.15)
    rng = np.random.default_rng(3)
    return dict(R=R, span=span, h=h, ds=ds, pos=pos, vel=vel,
                dx=rng.normal(size=pos.shape).astype(np.float32), dv=rng.normal(size=pos.shape).astype(np.float32))


@pytest.mark.parametrize("kind", ["luneburg", "smooth", "uniform"])
def test_trace_and_backtrace_bitwise(oracle, scene, kind):
    R, h, ds, pos, vel = scene["R"], scene["h"], scene["ds"], scene["pos"], scene["vel"]
    rif = {"luneburg": cases.luneburg(R), "smooth": cases.smooth_field(R, 3),
           "uniform": np.full((R, R, R), 1.2, np.float32)}[kind]
    with oracle.arith("factored"):
        o = oracle.trace(rif, rif.shape, pos, vel, h, ds, dtype=np.float32)
    k = H.trace(rif, rif.shape, pos, vel, h, ds)
    assert np.array_equal(o["xt"], k["xt"]) and np.array_equal(o["vt"], k["vt"])
    assert np.array_equal(o["steps"], k["steps"]) and o["n_failed"] == k["n_failed"]
    for corr in (False, True):
        with oracle.arith("factored"):
            ob = oracle.backtrace(rif, rif.shape, k["xt"], k["vt"], scene["dx"], scene["dv"], h, ds,
                                  dtype=np.float32, corrected_h=corr)
        kb = H.backtrace(rif, rif.shape, k["xt"], k["vt"], scene["dx"], scene["dv"], h, ds, corrected_h=corr)
        assert ob["steps_total"] == kb["steps_total"]
        assert cases.rel_l2(kb["grad"], ob["grad"]) < 2e-6          # same terms, different summation order


def test_plane_target_sdf_bitwise(oracle, scene):
    R, span, h, ds, pos, vel = scene["R"], scene["span"], scene["h"], scene["ds"], scene["pos"], scene["vel"]
    rif = cases.luneburg(R)
    po = np.tile(np.array([[0.5, 0.7, 0.5]], np.float32), (len(pos), 1))
    pd = np.tile(np.array([[0, 1, 0]], np.float32), (len(pos), 1))
    with oracle.arith("factored"):
        op = oracle.trace(rif, rif.shape, pos, vel, h, ds, dtype=np.float32, mode="plane", pln_o=po, pln_d=pd)
    kp = H.trace(rif, rif.shape, pos, vel, h, ds, mode="plane", pln_o=po, pln_d=pd)
    assert np.array_equal(op["xt"], kp["xt"]) and np.array_equal(op["vt"], kp["vt"])
    assert np.array_equal(op["failmask"], kp["failmask"])
    # target beyond the far face AND a target inside the volume: both sides of the global-loop coupling
    for tgt in ([0.5, 1.3, 0.5], [0.5, 0.6, 0.5], [-0.4, 0.2, 0.5]):
        tg = np.tile(np.array([tgt], np.float32), (len(pos), 1))
        with oracle.arith("factored"):
            ot = oracle.trace_target(rif, rif.shape, pos, vel, tg, h, ds, dtype=np.float32)
        kt = H.trace_target(rif, rif.shape, pos, vel, tg, h, ds)
        assert ot["iters"] == kt["iters"]
        assert np.array_equal(ot["xt"], kt["xt"]) and np.array_equal(ot["vt"], kt["vt"])
        assert np.array_equal(ot["dist2"], kt["dist2"])
    sdf = cases.sphere_sdf(R, span, 0.4)
    rng = np.random.default_rng(2)
    d = rng.normal(size=(len(pos), 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    ins = (0.5 * span + 0.2 * span * rng.uniform(0, 1, (len(pos), 1)) ** (1 / 3) * d).astype(np.float32)
    with oracle.arith("factored"):
        os_ = oracle.trace(rif, rif.shape, ins, vel, h, ds, dtype=np.float32, mode="sdf", sdf=sdf)
    ks = H.trace(rif, rif.shape, ins, vel, h, ds, mode="sdf", sdf=sdf)
    assert np.array_equal(os_["xt"], ks["xt"]) and np.array_equal(os_["vt"], ks["vt"])
    assert os_["n_failed"] == ks["n_failed"]
    with oracle.arith("factored"):
        ob = oracle.backtrace(rif, rif.shape, ks["xt"], ks["vt"], scene["dx"], scene["dv"], h, ds,
                              dtype=np.float32, sdf=sdf)
    kb = H.backtrace(rif, rif.shape, ks["xt"], ks["vt"], scene["dx"], scene["dv"], h, ds, sdf=sdf)
    assert ob["steps_total"] == kb["steps_total"] and cases.rel_l2(kb["grad"], ob["grad"]) < 2e-6


def test_cable_bitwise(oracle):
    rres, radius, length = 65, 1.0, 6.0
    ds = radius / rres / 2
    prof = np.sqrt(2.0 - np.linspace(0, 1, rres) ** 2).astype(np.float32)
    rng = np.random.default_rng(4)
    n = 500
    ang = rng.uniform(0, 2 * np.pi, n); rad = 0.8 * radius * np.sqrt(rng.uniform(0, 1, n))
    pos = np.stack([radius + rad * np.cos(ang), np.full(n, 0.37 * ds), radius + rad * np.sin(ang)], -1).astype(np.float32)
    pos[0] = [radius, 0.37 * ds, radius]                                   # on the axis: r < eps branch
    vel = rng.normal(0, 0.05, (n, 3)); vel[:, 1] = 1.0
    vel = (vel / np.linalg.norm(vel, axis=1, keepdims=True)).astype(np.float32)
    vel[0] = [0, 1, 0]
    tg = np.stack([np.full(n, radius), np.full(n, 0.75 * length), np.full(n, radius)], -1).astype(np.float32)
    with oracle.arith("factored"):
        o = oracle.trace_cable(prof, radius, length, pos, vel, tg, ds, dtype=np.float32)
    k = H.trace_cable(prof, radius, length, pos, vel, tg, ds)
    assert np.array_equal(o["xt"], k["xt"]) and np.array_equal(o["vt"], k["vt"]) and np.array_equal(o["dist2"], k["dist2"])
    assert o["steps_total"] == k["steps_total"]
    dx = rng.normal(size=pos.shape).astype(np.float32); dv = rng.normal(size=pos.shape).astype(np.float32)
    with oracle.arith("factored"):
        ob = oracle.backtrace_cable(prof, radius, length, k["xt"], k["vt"], dx, dv, ds, dtype=np.float32)
    kb = H.backtrace_cable(prof, radius, length, k["xt"], k["vt"], dx, dv, ds)
    assert ob["steps_total"] == kb["steps_total"]
    assert np.array_equal(ob["grad"], kb["grad"])       # same ray-major summation order: bitwise


def test_non_cubic_grid_bitwise(oracle):
    """res = (W,H,D) with W != H != D: flat index (z*H + y)*W + x with (W,H,D) = res exactly as
    src/volume.cpp:110-112,134-141 writes it (the reference's scripts only use cubes, Q2)."""
    W, H, D = 12, 9, 7
    h, ds = 0.25, 0.125
    rng = np.random.default_rng(5)
    rif = (1.0 + 0.3 * rng.random(W * H * D)).astype(np.float32)
    ext = np.array([(W - 1) * h, (H - 1) * h, (D - 1) * h])
    pos = (rng.random((600, 3)) * ext * 1.2 - 0.1 * ext).astype(np.float32)
    vel = rng.normal(size=(600, 3)); vel = (vel / np.linalg.norm(vel, axis=1, keepdims=True)).astype(np.float32)
    res = (W, H, D)
    with oracle.arith("factored"):
        o = oracle.trace(rif, res, pos, vel, h, ds, dtype=np.float32)
    k = H_.trace(rif, res, pos, vel, h, ds)
    assert np.array_equal(o["xt"], k["xt"]) and np.array_equal(o["vt"], k["vt"]) and np.array_equal(o["steps"], k["steps"])
    dx = rng.normal(size=pos.shape).astype(np.float32); dv = rng.normal(size=pos.shape).astype(np.float32)
    with oracle.arith("factored"):
        ob = oracle.backtrace(rif, res, k["xt"], k["vt"], dx, dv, h, ds, dtype=np.float32)
    kb = H_.backtrace(rif, res, k["xt"], k["vt"], dx, dv, h, ds)
    assert ob["steps_total"] == kb["steps_total"] and cases.rel_l2(kb["grad"], ob["grad"]) < 2e-6
    lit = oracle.trace(rif, res, pos, vel, h, ds, dtype=np.float64)
    assert np.mean(np.linalg.norm(k["xt"] - lit["xt"], axis=1) <= 2e-5) >= 0.98


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_hostcheck(oracle, seed):
    """The per-ray code on the nasty seeded configurations of cases.fuzz_config (non-cubic grids down to 2 voxels
    per axis, steps up to 1.7 cells, rays outside / exactly on faces, zero velocities): bit-exact forward for all
    variants, adjoint equal up to summation order."""
    c = cases.fuzz_config(seed)
    res, h, ds = c["res"], c["h"], c["ds"]
    with oracle.arith("factored"):
        o = oracle.trace(c["rif"], res, c["pos"], c["vel"], h, ds, dtype=np.float32)
        k = H.trace(c["rif"], res, c["pos"], c["vel"], h, ds)
        assert np.array_equal(o["xt"], k["xt"]) and np.array_equal(o["vt"], k["vt"]) and np.array_equal(o["steps"], k["steps"])
        ob = oracle.backtrace(c["rif"], res, o["xt"], o["vt"], c["dx"], c["dv"], h, ds, dtype=np.float32)
        kb = H.backtrace(c["rif"], res, k["xt"], k["vt"], c["dx"], c["dv"], h, ds)
        assert ob["steps_total"] == kb["steps_total"]
        assert cases.rel_l2(kb["grad"], ob["grad"]) < 2e-5 or np.abs(ob["grad"]).max() < 1e-20
        op = oracle.trace(c["rif"], res, c["pos"], c["vel"], h, ds, dtype=np.float32, mode="plane", pln_o=c["po"], pln_d=c["pd"])
        kp = H.trace(c["rif"], res, c["pos"], c["vel"], h, ds, mode="plane", pln_o=c["po"], pln_d=c["pd"])
        assert np.array_equal(op["xt"], kp["xt"]) and np.array_equal(op["failmask"], kp["failmask"])
        ot = oracle.trace_target(c["rif"], res, c["pos"], c["vel"], c["tg"], h, ds, dtype=np.float32)
        kt = H.trace_target(c["rif"], res, c["pos"], c["vel"], c["tg"], h, ds)
        assert ot["iters"] == kt["iters"] and np.array_equal(ot["xt"], kt["xt"]) and np.array_equal(ot["dist2"], kt["dist2"])
        os_ = oracle.trace(c["rif"], res, c["pos"], c["vel"], h, ds, dtype=np.float32, mode="sdf", sdf=c["sdf"])
        ks = H.trace(c["rif"], res, c["pos"], c["vel"], h, ds, mode="sdf", sdf=c["sdf"])
        assert np.array_equal(os_["xt"], ks["xt"]) and np.array_equal(os_["vt"], ks["vt"])
        obs = oracle.backtrace(c["rif"], res, os_["xt"], os_["vt"], c["dx"], c["dv"], h, ds, dtype=np.float32, sdf=c["sdf"])
        kbs = H.backtrace(c["rif"], res, ks["xt"], ks["vt"], c["dx"], c["dv"], h, ds, sdf=c["sdf"])
        assert cases.rel_l2(kbs["grad"], obs["grad"]) < 2e-5 or np.abs(obs["grad"]).max() < 1e-20
        # the adjoint started from ARBITRARY rays (not exit states of a forward march)
        obr = oracle.backtrace(c["rif"], res, c["pos"], c["vel"], c["dx"], c["dv"], h, ds, dtype=np.float32)
        kbr = H.backtrace(c["rif"], res, c["pos"], c["vel"], c["dx"], c["dv"], h, ds)
        assert obr["steps_total"] == kbr["steps_total"]
        assert cases.grads_agree(kbr["grad"], obr["grad"])


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_cable_hostcheck(oracle, seed):
    c = cases.fuzz_cable_config(seed)
    a = (c["prof"], c["radius"], c["length"])
    with oracle.arith("factored"):
        o = oracle.trace_cable(*a, c["pos"], c["vel"], c["tg"], c["ds"], dtype=np.float32)
        k = H.trace_cable(*a, c["pos"], c["vel"], c["tg"], c["ds"])
        assert np.array_equal(o["xt"], k["xt"]) and np.array_equal(o["vt"], k["vt"]) and np.array_equal(o["dist2"], k["dist2"])
        assert o["steps_total"] == k["steps_total"]
        for xt, vt in ((k["xt"], k["vt"]), (c["pos"], c["vel"])):
            ob = oracle.backtrace_cable(*a, xt, vt, c["dx"], c["dv"], c["ds"], dtype=np.float32)
            kb = H.backtrace_cable(*a, xt, vt, c["dx"], c["dv"], c["ds"])
            assert ob["steps_total"] == kb["steps_total"]
            assert cases.grads_agree(kb["grad"], ob["grad"])


def test_sanitized_nonfinite_inputs(tmp_path):
    """ASan + UBSan over the per-ray code with NaN / Inf / huge values planted in rays, seeds and grid: no
    out-of-bounds tap, no integer overflow, every march terminates (tests/hostcheck/sanitize_run.py)."""
    import glob
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    rt = glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no clang sanitizer runtime in this image")
    so = str(tmp_path / "libhostcheck_asan.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-mfma", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", so,
                    os.path.join(here, "hostcheck", "hostcheck.hip")], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, os.path.join(here, "hostcheck", "sanitize_run.py"), so], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "finished without reports" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
