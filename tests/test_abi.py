"""CPU tier: the C-ABI library loads without a GPU and exports every symbol include/drrt_hip.h
declares; the Python mirrors expose the reference's names and call shapes.  No compute calls."""
import inspect
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "drrt_hip.h")


@pytest.fixture(scope="module")
def lib():
    from adjointnonlinearraytracing_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "adjointnonlinearraytracing_amd", "csrc")], check=True)
    return _lib


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"DRRT_API\s+[\w\s\*]+?\b(drrt_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 14 and "drrt_trace_f32" in syms and "drrt_backtrace_cable_f32" in syms
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (drrt_\w+)", out))
    assert set(syms) <= exported, f"missing: {set(syms) - exported}"
    assert set(lib.SIGNATURES) == set(syms), "python binding table out of sync with the header"
    handle = lib.load()
    assert handle.drrt_version().decode().startswith("drrt_hip") and "gfx950" in handle.drrt_version().decode()
    assert handle.drrt_last_error().decode() == ""


def test_header_cites_reference_interfaces():
    text = open(HEADER).read()
    for cite in ("src/tracer.cpp:35-100", "src/tracer.cpp:384-440", "src/drrt.cpp:47-58", "include/tracer.h:15-89",
                 "src/tracer.cpp:511-567", "src/tracer.cpp:174-242"):
        assert cite in text


def test_gfx950_code_object_is_embedded(lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", lib.LIB_PATH],
                         capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_python_mirrors_have_reference_call_shapes(lib):
    from adjointnonlinearraytracing_amd import drrt, tracer
    # src/drrt.cpp:47-58 (TracerC) and :38-45 (TracerS: only these four)
    for m in ("test", "testscale", "trace", "trace_pln", "trace_sdf", "trace_target", "trace_cable",
              "backtrace", "backtrace_sdf", "backtrace_cable"):
        assert callable(getattr(drrt.TracerC, m))
    for m in ("trace", "trace_sdf", "trace_target", "backtrace"):
        assert callable(getattr(drrt.TracerS, m))
    assert not hasattr(drrt.TracerS, "trace_cable") and not hasattr(drrt.TracerS, "backtrace_sdf")
    def params(f):
        return list(inspect.signature(f).parameters)[1:]
    # include/tracer.h:15-89 argument order
    assert params(drrt.TracerC.trace) == ["rif", "res", "pos", "vel", "h", "ds"]
    assert params(drrt.TracerC.trace_pln) == ["rif", "res", "pos", "vel", "pln_o", "pln_d", "h", "ds"]
    assert params(drrt.TracerC.trace_target) == ["rif", "res", "pos", "vel", "target", "h", "ds"]
    assert params(drrt.TracerC.trace_sdf) == ["rif", "sdf", "res", "pos", "vel", "h", "ds"]
    assert params(drrt.TracerC.trace_cable) == ["rif", "radius", "length", "pos", "vel", "target", "ds"]
    # (+ one optional keyword the reference does not have: the paired forward call's visit order)
    assert params(drrt.TracerC.backtrace) == ["rif", "res", "xt", "vt", "dx", "dv", "h", "ds", "order"]
    assert params(drrt.TracerC.backtrace_sdf) == ["rif", "sdf", "res", "xt", "vt", "dx", "dv", "h", "ds", "order"]
    assert params(drrt.TracerC.backtrace_cable) == ["rif", "radius", "length", "xt", "vt", "dx", "dv", "ds"]
    # core/tracer.py:294-526 forward signatures
    assert params(tracer.BackTracerC.forward) == ["rif", "x", "v", "h", "ds"]
    assert params(tracer.BackPlaneTracerC.forward) == ["rif", "x", "v", "sp", "sn", "h", "ds"]
    assert params(tracer.BackTargetTracerC.forward) == ["rif", "x", "v", "sp", "h", "ds"]
    assert params(tracer.BackSDFTracerC.forward) == ["rif", "sdf", "x", "v", "h", "ds"]
    assert params(tracer.BackCableTracerC.forward) == ["rif", "radius", "length", "x", "v", "sp", "ds"]
    with pytest.raises(NotImplementedError):
        drrt.TracerD()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import or load it."""
    pkg = os.path.join(ROOT, "adjointnonlinearraytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "libdrrt_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_tracerc_refuses_cpu_tensors_without_gpu(lib):
    import torch
    from adjointnonlinearraytracing_amd import drrt
    with pytest.raises(RuntimeError, match="cuda"):
        drrt.TracerC().trace(torch.ones(4, 4, 4), (4, 4, 4), torch.zeros(2, 3), torch.zeros(2, 3), 1.0, 0.5)


def test_order_hint_never_outlives_one_call(lib):
    """a march call that fails validation must still consume the visit-order hint, so that a stale
    device pointer cannot be picked up by a later call with the same ray count.  Validation failures happen
    before any HIP call, so this runs without a GPU (null data pointers are never dereferenced)."""
    import ctypes as C
    h = lib.load()
    res_bad = (C.c_int * 3)(4, 4, 5)            # 4*4*5 != nvox = 64 -> "Resolution doesn't match data"
    res_ok = (C.c_int * 3)(4, 4, 4)
    fake = C.c_void_p(0xDEAD0000)
    one = C.c_void_p(0x1000)                    # non-null stand-in for rif (make_vol only checks for null)
    n = 128

    def armed():
        h.drrt_set_order_hint(fake, n)
        assert h.drrt_order_hint_pending() == n

    calls = [
        lambda: h.drrt_trace_f32(one, 64, res_bad, n, None, None, 1.0, 0.5, None, None, None, None, 0, 0, None),
        lambda: h.drrt_trace_f16io(one, 64, res_bad, n, None, None, 1.0, 0.5, None, None, None, None, 0, 0, None),
        lambda: h.drrt_trace_pln_f32(one, 64, res_bad, n, None, None, None, None, 1.0, 0.5, None, None, None,
                                     None, None, 0, 0, None),
        lambda: h.drrt_trace_sdf_f32(one, one, 64, res_bad, n, None, None, 1.0, 0.5, None, None, None, None, 0, 0, None),
        lambda: h.drrt_trace_target_f32(one, 64, res_bad, n, None, None, None, 1.0, 0.5, None, None, None, None,
                                        None, 0, 0, None),
        lambda: h.drrt_backtrace_f32(one, 64, res_bad, n, None, None, None, None, 1.0, 0.5, None, None, None, 0, 0, None),
        lambda: h.drrt_backtrace_f16io(one, 64, res_bad, n, None, None, None, None, 1.0, 0.5, None, None, None, 0, 0, None),
        lambda: h.drrt_backtrace_sdf_f32(one, one, 64, res_bad, n, None, None, None, None, 1.0, 0.5, None, None,
                                         None, 0, 0, None),
        # valid grid, invalid step: fails in check_steps, after make_vol
        lambda: h.drrt_trace_f32(one, 64, res_ok, n, None, None, 1.0, -0.5, None, None, None, None, 0, 0, None),
        # cable calls: rres < 2
        lambda: h.drrt_trace_cable_f32(one, 1, 1.0, 1.0, n, None, None, None, 0.1, None, None, None, None, None, 0, 0, None),
        lambda: h.drrt_backtrace_cable_f32(one, 1, 1.0, 1.0, n, None, None, None, None, 0.1, one, None, None, 0, 0, None),
    ]
    for call in calls:
        armed()
        assert call() < 0 and h.drrt_last_error()
        assert h.drrt_order_hint_pending() == 0
    h.drrt_set_order_hint(fake, n)
    h.drrt_set_order_hint(None, n)              # a null order disarms whatever n says
    assert h.drrt_order_hint_pending() == 0


def test_q16_params_host_side(lib):
    """drrt_q16_params is host-only: q_min = -E/16, q_step = 1.125 E / 65535 with E the largest box extent, 2^-14."""
    import ctypes as C
    h = lib.load()
    out = (C.c_float * 3)()
    assert h.drrt_q16_params((C.c_int * 3)(256, 256, 256), C.c_float(1.0 / 255), out) == 0
    E = 255 * float(C.c_float(1.0 / 255).value)
    assert abs(out[0] + E / 16) < 1e-6 and abs(out[1] - 1.125 * E / 65535) < 1e-9 and out[2] == 2.0 ** -14
    assert h.drrt_q16_params((C.c_int * 3)(4, 9, 5), C.c_float(0.5), out) == 0 and abs(out[0] + 4.0 / 16) < 1e-7
    assert h.drrt_q16_params((C.c_int * 3)(4, 0, 5), C.c_float(0.5), out) < 0


def test_options_context_manager_is_per_thread():
    """drrt.using(...) overrides the options of the calling thread only, nests, and restores; the module-level
    `options` stays the process default."""
    import threading
    from adjointnonlinearraytracing_amd import _lib, drrt
    base = drrt._flags(adjoint=True)
    seen = {}

    def other():
        seen["other"] = drrt._flags(adjoint=True)

    with drrt.using(sort_rays=False, corrected_h=True, adjoint_window="ring"):
        inner = drrt._flags(adjoint=True)
        t = threading.Thread(target=other); t.start(); t.join()
        with drrt.using(adjoint_window="box"):
            nested = drrt._flags(adjoint=True)
        assert drrt._flags(adjoint=True) == inner
    assert drrt._flags(adjoint=True) == base == seen["other"]
    assert inner & _lib.FLAG_RING_WINDOW and inner & _lib.FLAG_CORRECTED_H and not inner & _lib.FLAG_SORT_RAYS
    assert nested & _lib.FLAG_STATIC_WINDOW and not nested & _lib.FLAG_RING_WINDOW and nested & _lib.FLAG_CORRECTED_H
