"""examples/pybind_drrt: the reference-style pybind11 module over the C ABI must build,
expose the reference's `TracerC` method set (src/drrt.cpp:47-58) and -- on the GPU -- return exactly what the
shipped ctypes mirror returns."""
import importlib.util
import os
import sys

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METHODS = ["trace", "trace_pln", "trace_sdf", "trace_target", "trace_cable", "backtrace", "backtrace_sdf",
           "backtrace_cable"]


@pytest.fixture(scope="module")
def native():
    import torch  # noqa: F401  (its HIP runtime first, as for the ctypes loader)
    spec = importlib.util.spec_from_file_location("pybind_build", os.path.join(ROOT, "examples", "pybind_drrt", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    so = b.build()
    spec = importlib.util.spec_from_file_location("drrt_native", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pybind_module_builds_and_mirrors_the_reference_class(native):
    assert native.version().startswith("drrt_hip")
    t = native.TracerC()
    for m in METHODS:
        assert callable(getattr(t, m)), m


@pytest.mark.gpu
def test_pybind_module_matches_the_ctypes_mirror(gpu, native):
    import torch
    from adjointnonlinearraytracing_amd import drrt
    drrt.options.check_failed = False
    drrt.options.sort_rays = True
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    c = cases.fuzz_config(4)
    res, h, ds = list(c["res"]), c["h"], c["ds"]
    R, S = f(c["rif"]).reshape(-1), f(c["sdf"]).reshape(-1)
    P, V, DX, DV = f(c["pos"]), f(c["vel"]), f(c["dx"]), f(c["dv"])
    A, B = native.TracerC(), drrt.TracerC()
    xa, va = A.trace(R, res, P, V, h, ds); xb, vb = B.trace(R, res, P, V, h, ds)
    assert torch.equal(xa, xb) and torch.equal(va, vb)
    pa = A.trace_pln(R, res, P, V, f(c["po"]), f(c["pd"]), h, ds); pb = B.trace_pln(R, res, P, V, f(c["po"]), f(c["pd"]), h, ds)
    assert torch.equal(pa[0], pb[0]) and torch.equal(pa[1], pb[1]) and torch.equal(pa[2], pb[2].to(torch.bool))
    sa = A.trace_sdf(R, S, res, P, V, h, ds); sb = B.trace_sdf(R, S, res, P, V, h, ds)
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1])
    ta = A.trace_target(R, res, P, V, f(c["tg"]), h, ds); tb = B.trace_target(R, res, P, V, f(c["tg"]), h, ds)
    assert all(torch.equal(a, b) for a, b in zip(ta, tb))
    ga = A.backtrace(R, res, xa, va, DX, DV, h, ds); gb = B.backtrace(R, res, xb, vb, DX, DV, h, ds)
    assert cases.grads_agree(ga.cpu().numpy(), gb.cpu().numpy(), tol=2e-6)
    gsa = A.backtrace_sdf(R, S, res, sa[0], sa[1], DX, DV, h, ds); gsb = B.backtrace_sdf(R, S, res, sb[0], sb[1], DX, DV, h, ds)
    assert cases.grads_agree(gsa.cpu().numpy(), gsb.cpu().numpy(), tol=2e-6)
    k = cases.fuzz_cable_config(2)
    prof = f(k["prof"])
    ca = A.trace_cable(prof, k["radius"], k["length"], f(k["pos"]), f(k["vel"]), f(k["tg"]), k["ds"])
    cb = B.trace_cable(prof, k["radius"], k["length"], f(k["pos"]), f(k["vel"]), f(k["tg"]), k["ds"])
    assert all(torch.equal(a, b) for a, b in zip(ca, cb))
    gca = A.backtrace_cable(prof, k["radius"], k["length"], ca[0], ca[1], f(k["dx"]), f(k["dv"]), k["ds"])
    gcb = B.backtrace_cable(prof, k["radius"], k["length"], cb[0], cb[1], f(k["dx"]), f(k["dv"]), k["ds"])
    assert cases.grads_agree(gca.cpu().numpy(), gcb.cpu().numpy(), tol=1e-4)
    with pytest.raises(RuntimeError, match="Resolution doesn't match data"):
        A.trace(R, [res[0], res[1], res[2] + 1], P, V, h, ds)
