"""ctypes front-end of the CPU parity oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module (as the checker / reported baseline); the product package
``adjointnonlinearraytracing_amd`` never does.

Functions mirror the reference's ``Tracer`` methods (``/root/reference/include/tracer.h:15-89``)
on numpy arrays; ``dtype`` float32 or float64 selects the instantiation.
Parity status: "parity unpinned" against the reference's native enoki build (see
``drrt_oracle.c``); pinned pieces are listed in ``tests/golden/README.md``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdrrt_oracle.so")
_lib: Optional[C.CDLL] = None


def build(force: bool = False) -> str:
    """Compile the oracle (gcc via oracle/Makefile).  Returns the .so path."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("drrt_oracle.c", "drrt_oracle_impl.h", "Makefile")):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


class arith:
    """Context manager selecting the oracle's arithmetic mode: 'literal' (the reference's
    expression order; default) or 'factored' (the explicit IEEE sequence of the HIP kernels)."""

    def __init__(self, mode: str):
        if mode not in ("literal", "factored"):
            raise ValueError(mode)
        self.mode = 1 if mode == "factored" else 0

    def __enter__(self):
        self.prev = lib().oracle_get_arith()
        lib().oracle_set_arith(self.mode)
        return self

    def __exit__(self, *exc):
        lib().oracle_set_arith(self.prev)
        return False


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _real(dtype):
    return C.c_float if np.dtype(dtype) == np.float32 else C.c_double


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype, shape_last=None) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a, dtype=dtype))
    if shape_last is not None and (a.ndim != 2 or a.shape[1] != shape_last):
        raise ValueError(f"expected (N,{shape_last}) array, got {a.shape}")
    return a


def _res(res):
    r = np.asarray(list(res), dtype=np.int32)
    if r.shape != (3,):
        raise ValueError("res must be a 3-sequence")
    return r


def _check(rc: int):
    if rc == -1:
        raise RuntimeError("Resolution doesn't match data")   # src/volume.cpp:37
    if rc == -2:
        raise RuntimeError("volume: invalid resolution!")     # src/volume.cpp:124
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}")


# ----------------------------------------------------------------------------- forward
def trace(rif, res, pos, vel, h, ds, dtype=np.float32, mode="trace", sdf=None,
          pln_o=None, pln_d=None):
    """Tracer::trace / trace_plane / trace_sdf (src/tracer.cpp:35-100,102-172,244-310).

    Returns dict(xt, vt, steps (per ray), n_failed, iters[, failmask])."""
    s, R = _sfx(dtype), _real(dtype)
    rif = _c(np.asarray(rif).reshape(-1), dtype)
    res = _res(res)
    pos, vel = _c(pos, dtype, 3), _c(vel, dtype, 3)
    n = pos.shape[0]
    xt, vt = np.empty_like(pos), np.empty_like(vel)
    steps = np.zeros(n, dtype=np.int32)
    nf, iters = C.c_longlong(0), C.c_int(0)
    L = lib()
    out = {}
    if mode == "trace":
        rc = getattr(L, f"oracle_trace_{s}")(
            _p(rif), _p(res), C.c_longlong(rif.size), C.c_size_t(n), _p(pos), _p(vel),
            R(h), R(ds), _p(xt), _p(vt), _p(steps), C.byref(nf), C.byref(iters))
    elif mode == "plane":
        po, pd = _c(pln_o, dtype, 3), _c(pln_d, dtype, 3)
        fm = np.zeros(n, dtype=np.uint8)
        rc = getattr(L, f"oracle_trace_pln_{s}")(
            _p(rif), _p(res), C.c_longlong(rif.size), C.c_size_t(n), _p(pos), _p(vel),
            _p(po), _p(pd), R(h), R(ds), _p(xt), _p(vt), _p(fm), _p(steps),
            C.byref(nf), C.byref(iters))
        out["failmask"] = fm.astype(bool)
    elif mode == "sdf":
        sdf = _c(np.asarray(sdf).reshape(-1), dtype)
        if sdf.size != rif.size:
            raise RuntimeError("Resolution doesn't match data")
        rc = getattr(L, f"oracle_trace_sdf_{s}")(
            _p(rif), _p(sdf), _p(res), C.c_longlong(rif.size), C.c_size_t(n), _p(pos), _p(vel),
            R(h), R(ds), _p(xt), _p(vt), _p(steps), C.byref(nf), C.byref(iters))
    else:
        raise ValueError(mode)
    _check(rc)
    out.update(xt=xt, vt=vt, steps=steps, n_failed=nf.value, iters=iters.value)
    return out


def trace_target(rif, res, pos, vel, target, h, ds, dtype=np.float32):
    """Tracer::trace_target (src/tracer.cpp:174-242)."""
    s, R = _sfx(dtype), _real(dtype)
    rif = _c(np.asarray(rif).reshape(-1), dtype)
    res = _res(res)
    pos, vel, target = _c(pos, dtype, 3), _c(vel, dtype, 3), _c(target, dtype, 3)
    n = pos.shape[0]
    xt, vt = np.empty_like(pos), np.empty_like(vel)
    d2 = np.empty(n, dtype=dtype)
    nf, iters = C.c_longlong(0), C.c_int(0)
    rc = getattr(lib(), f"oracle_trace_target_{s}")(
        _p(rif), _p(res), C.c_longlong(rif.size), C.c_size_t(n), _p(pos), _p(vel), _p(target),
        R(h), R(ds), _p(xt), _p(vt), _p(d2), C.byref(nf), C.byref(iters))
    _check(rc)
    return dict(xt=xt, vt=vt, dist2=d2, n_failed=nf.value, iters=iters.value)


def trace_cable(rif, radius, length, pos, vel, target, ds, dtype=np.float32):
    """Tracer::trace_cable (src/tracer.cpp:312-382)."""
    s, R = _sfx(dtype), _real(dtype)
    rif = _c(np.asarray(rif).reshape(-1), dtype)
    pos, vel, target = _c(pos, dtype, 3), _c(vel, dtype, 3), _c(target, dtype, 3)
    n = pos.shape[0]
    xt, vt = np.empty_like(pos), np.empty_like(vel)
    d2 = np.empty(n, dtype=dtype)
    nf, st = C.c_longlong(0), C.c_longlong(0)
    rc = getattr(lib(), f"oracle_trace_cable_{s}")(
        _p(rif), C.c_size_t(rif.size), R(radius), R(length), C.c_size_t(n), _p(pos), _p(vel),
        _p(target), R(ds), _p(xt), _p(vt), _p(d2), C.byref(nf), C.byref(st))
    _check(rc)
    return dict(xt=xt, vt=vt, dist2=d2, n_failed=nf.value, steps_total=st.value)


# ----------------------------------------------------------------------------- adjoint
def backtrace(rif, res, xt, vt, dx, dv, h, ds, dtype=np.float32, sdf=None,
              corrected_h: bool = False):
    """Tracer::backtrace / backtrace_sdf (src/tracer.cpp:384-440,443-509).

    ``corrected_h=False`` reproduces the reference as written (gradient splat without
    1/h); True divides the gradient-splat term by h (exact discrete adjoint).
    Returns dict(grad (flat), steps_total)."""
    s, R = _sfx(dtype), _real(dtype)
    rif = _c(np.asarray(rif).reshape(-1), dtype)
    res = _res(res)
    xt, vt, dx, dv = (_c(a, dtype, 3) for a in (xt, vt, dx, dv))
    n = xt.shape[0]
    grad = np.zeros(rif.size, dtype=dtype)
    st = C.c_longlong(0)
    gs = (1.0 / float(np.dtype(dtype).type(h))) if corrected_h else 1.0
    if sdf is None:
        rc = getattr(lib(), f"oracle_backtrace_{s}")(
            _p(rif), _p(res), C.c_longlong(rif.size), C.c_size_t(n), _p(xt), _p(vt), _p(dx),
            _p(dv), R(h), R(ds), R(gs), _p(grad), C.byref(st))
    else:
        sdf = _c(np.asarray(sdf).reshape(-1), dtype)
        rc = getattr(lib(), f"oracle_backtrace_sdf_{s}")(
            _p(rif), _p(sdf), _p(res), C.c_longlong(rif.size), C.c_size_t(n), _p(xt), _p(vt),
            _p(dx), _p(dv), R(h), R(ds), R(gs), _p(grad), C.byref(st))
    _check(rc)
    return dict(grad=grad, steps_total=st.value)


class trajectory_signatures:
    """Context manager: while active, ``backtrace`` records for every ray a signature of the integer cells it
    contributes in and its number of contributing steps (``.sig`` uint64[n], ``.steps`` int32[n]).  Rays whose
    signatures agree between two arithmetics (fp32 factored vs fp64 literal) are free of cell-face tie events."""

    def __init__(self, n: int):
        self.sig = np.zeros(n, dtype=np.uint64)
        self.steps = np.zeros(n, dtype=np.int32)

    def __enter__(self):
        lib().oracle_set_trajectory_sink(_p(self.sig), _p(self.steps))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_trajectory_sink(None, None)
        return False


def backtrace_cable(rif, radius, length, xt, vt, dx, dv, ds, dtype=np.float32):
    """Tracer::backtrace_cable (src/tracer.cpp:511-567)."""
    s, R = _sfx(dtype), _real(dtype)
    rif = _c(np.asarray(rif).reshape(-1), dtype)
    xt, vt, dx, dv = (_c(a, dtype, 3) for a in (xt, vt, dx, dv))
    n = xt.shape[0]
    grad = np.zeros(rif.size, dtype=dtype)
    st = C.c_longlong(0)
    rc = getattr(lib(), f"oracle_backtrace_cable_{s}")(
        _p(rif), C.c_size_t(rif.size), R(radius), R(length), C.c_size_t(n), _p(xt), _p(vt),
        _p(dx), _p(dv), R(ds), _p(grad), C.byref(st))
    _check(rc)
    return dict(grad=grad, steps_total=st.value)


def bench_allcores(rif, res, pos, vel, h, ds, nthreads=0, want_rays=False):
    """Timing harness: fp32 trace + backtrace(dx=dv=1) over contiguous ray chunks on `nthreads` OpenMP
    threads (0 = all), private gradient grids summed at the end.  Runs in the arithmetic mode that is
    selected (``arith``).  Returns dict(t_fwd, t_adj, fwd_steps, adj_steps, threads, grad) and, with
    ``want_rays``, the exit rays ``xt``, ``vt`` and the per-ray forward step counts ``steps`` -- what
    bench.py's ``parity_check`` compares the GPU's results of the SAME rays with."""
    rif = _c(np.asarray(rif).reshape(-1), np.float32)
    pos, vel = _c(pos, np.float32, 3), _c(vel, np.float32, 3)
    grad = np.zeros(rif.size, np.float32)
    n = len(pos)
    xt = np.empty((n, 3), np.float32) if want_rays else None
    vt = np.empty((n, 3), np.float32) if want_rays else None
    steps = np.empty(n, np.int32) if want_rays else None
    tf, ta, fs, th, ast = C.c_double(0), C.c_double(0), C.c_longlong(0), C.c_int(0), C.c_longlong(0)
    rc = lib().oracle_bench_allcores_f32(_p(rif), _p(_res(res)), C.c_longlong(rif.size), C.c_size_t(n), _p(pos),
                                         _p(vel), C.c_float(h), C.c_float(ds), C.c_int(nthreads), _p(grad),
                                         C.byref(tf), C.byref(ta), C.byref(fs), C.byref(th),
                                         _p(xt), _p(vt), _p(steps), C.byref(ast))
    _check(rc)
    out = dict(t_fwd=tf.value, t_adj=ta.value, fwd_steps=fs.value, adj_steps=ast.value, threads=th.value, grad=grad)
    if want_rays:
        out.update(xt=xt, vt=vt, steps=steps)
    return out


# ----------------------------------------------------------------------------- samplers
def eval_grad(data, res, h, pts, mask=None, dtype=np.float32):
    """volume::eval_grad (src/volume.cpp:101-181) at points (N,3) -> (n (N,), grad (N,3))."""
    s, R = _sfx(dtype), _real(dtype)
    data = _c(np.asarray(data).reshape(-1), dtype)
    pts = _c(pts, dtype, 3)
    n = pts.shape[0]
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
    on, og = np.empty(n, dtype=dtype), np.empty((n, 3), dtype=dtype)
    getattr(lib(), f"oracle_eval_grad_{s}")(_p(data), _p(_res(res)), R(h), C.c_size_t(n),
                                            _p(pts), _p(m), _p(on), _p(og))
    return on, og


def eval_hess(data, res, h, pts, mask=None, dtype=np.float32):
    """volume::eval_hess (src/volume.cpp:40-99) -> (N,3) = (dxdy, dxdz, dydz)."""
    s, R = _sfx(dtype), _real(dtype)
    data = _c(np.asarray(data).reshape(-1), dtype)
    pts = _c(pts, dtype, 3)
    n = pts.shape[0]
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
    oh = np.empty((n, 3), dtype=dtype)
    getattr(lib(), f"oracle_eval_hess_{s}")(_p(data), _p(_res(res)), R(h), C.c_size_t(n),
                                            _p(pts), _p(m), _p(oh))
    return oh


def splat(nvox_or_data, res, h, pts, val, grad, mask=None, dtype=np.float32,
          grad_scale=1.0):
    """volume::splat (src/volume.cpp:182-244) into a zero grid (or a copy of the given one)."""
    s, R = _sfx(dtype), _real(dtype)
    if np.isscalar(nvox_or_data):
        data = np.zeros(int(nvox_or_data), dtype=dtype)
    else:
        data = _c(np.asarray(nvox_or_data).reshape(-1), dtype).copy()
    pts, grad = _c(pts, dtype, 3), _c(grad, dtype, 3)
    val = _c(val, dtype)
    n = pts.shape[0]
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
    getattr(lib(), f"oracle_splat_{s}")(_p(data), _p(_res(res)), R(h), C.c_size_t(n), _p(pts),
                                        _p(val), _p(grad), _p(m), R(grad_scale))
    return data


def cyl_eval_grad(data, radius, pts, dtype=np.float32):
    """cylinder_volume::eval_grad (src/cylinder_volume.cpp:26-59)."""
    s, R = _sfx(dtype), _real(dtype)
    data = _c(np.asarray(data).reshape(-1), dtype)
    pts = _c(pts, dtype, 3)
    n = pts.shape[0]
    on, og = np.empty(n, dtype=dtype), np.empty((n, 3), dtype=dtype)
    getattr(lib(), f"oracle_cyl_eval_grad_{s}")(_p(data), C.c_size_t(data.size), R(radius),
                                                C.c_size_t(n), _p(pts), _p(on), _p(og))
    return on, og


def cyl_eval_hess(data, radius, pts, dtype=np.float32):
    """cylinder_volume::eval_hess (src/cylinder_volume.cpp:61-111) -> (N,4) H00,H02,H20,H22."""
    s, R = _sfx(dtype), _real(dtype)
    data = _c(np.asarray(data).reshape(-1), dtype)
    pts = _c(pts, dtype, 3)
    n = pts.shape[0]
    oh = np.empty((n, 4), dtype=dtype)
    getattr(lib(), f"oracle_cyl_eval_hess_{s}")(_p(data), C.c_size_t(data.size), R(radius),
                                                C.c_size_t(n), _p(pts), _p(oh))
    return oh
