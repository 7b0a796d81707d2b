"""Loader for tests/hostcheck/hostcheck.hip (TEST INFRASTRUCTURE ONLY): the product's own
__host__ __device__ per-ray code compiled for the host, so CPU-only tests can compare it with the
oracle's `factored` arithmetic bit for bit.  Never imported by the package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "hostcheck", "hostcheck.hip")
_HDR = os.path.join(_HERE, "..", "adjointnonlinearraytracing_amd", "csrc", "drrt_device.h")
_SO = os.path.join(_HERE, "hostcheck", "_build", "libhostcheck.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        if (not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(_SRC), os.path.getmtime(_HDR))):
            subprocess.run(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC",
                            "-ffp-contract=off", "-mfma", "-shared", "-fvisibility=hidden", "-o", _SO, _SRC],
                           check=True, capture_output=True)
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f(a, cols=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a


def _res(res):
    return np.asarray(list(res), dtype=np.int32)


def trace(rif, res, pos, vel, h, ds, mode="trace", sdf=None, pln_o=None, pln_d=None):
    m = {"trace": 0, "plane": 1, "sdf": 2}[mode]
    rif, pos, vel = _f(rif).reshape(-1), _f(pos), _f(vel)
    sdf = None if sdf is None else _f(sdf).reshape(-1)
    po = None if pln_o is None else _f(pln_o); pd = None if pln_d is None else _f(pln_d)
    n = len(pos)
    xt, vt = np.empty_like(pos), np.empty_like(vel)
    fm = np.zeros(n, np.uint8); steps = np.zeros(n, np.int32); nf = C.c_longlong(0)
    lib().hostcheck_trace(m, _p(rif), _p(sdf), _p(_res(res)), C.c_size_t(n), _p(pos), _p(vel), _p(po), _p(pd),
                          C.c_float(h), C.c_float(ds), _p(xt), _p(vt), _p(fm), _p(steps), C.byref(nf))
    return dict(xt=xt, vt=vt, failmask=fm.astype(bool), steps=steps, n_failed=nf.value)


def trace_target(rif, res, pos, vel, target, h, ds):
    rif, pos, vel, target = _f(rif).reshape(-1), _f(pos), _f(vel), _f(target)
    n = len(pos)
    xt, vt, d2 = np.empty_like(pos), np.empty_like(vel), np.empty(n, np.float32)
    it = C.c_int(0)
    lib().hostcheck_trace_target(_p(rif), _p(_res(res)), C.c_size_t(n), _p(pos), _p(vel), _p(target),
                                 C.c_float(h), C.c_float(ds), _p(xt), _p(vt), _p(d2), C.byref(it))
    return dict(xt=xt, vt=vt, dist2=d2, iters=it.value)


def backtrace(rif, res, xt, vt, dx, dv, h, ds, sdf=None, corrected_h=False):
    rif = _f(rif).reshape(-1)
    sdf_ = None if sdf is None else _f(sdf).reshape(-1)
    xt, vt, dx, dv = _f(xt), _f(vt), _f(dx), _f(dv)
    grad = np.zeros(rif.size, np.float32); st = C.c_longlong(0)
    gs = float(np.float32(1.0) / np.float32(h)) if corrected_h else 1.0
    lib().hostcheck_backtrace(0 if sdf is None else 1, _p(rif), _p(sdf_), _p(_res(res)), C.c_size_t(len(xt)),
                              _p(xt), _p(vt), _p(dx), _p(dv), C.c_float(h), C.c_float(ds), C.c_float(gs),
                              _p(grad), C.byref(st))
    return dict(grad=grad, steps_total=st.value)


def trace_cable(rif, radius, length, pos, vel, target, ds):
    rif, pos, vel, target = _f(rif).reshape(-1), _f(pos), _f(vel), _f(target)
    n = len(pos)
    xt, vt, d2 = np.empty_like(pos), np.empty_like(vel), np.empty(n, np.float32)
    st = C.c_longlong(0)
    lib().hostcheck_trace_cable(_p(rif), C.c_int(rif.size), C.c_float(radius), C.c_float(length), C.c_size_t(n),
                                _p(pos), _p(vel), _p(target), C.c_float(ds), _p(xt), _p(vt), _p(d2), C.byref(st))
    return dict(xt=xt, vt=vt, dist2=d2, steps_total=st.value)


def backtrace_cable(rif, radius, length, xt, vt, dx, dv, ds):
    rif = _f(rif).reshape(-1)
    xt, vt, dx, dv = _f(xt), _f(vt), _f(dx), _f(dv)
    grad = np.zeros(rif.size, np.float32); st = C.c_longlong(0)
    lib().hostcheck_backtrace_cable(_p(rif), C.c_int(rif.size), C.c_float(radius), C.c_float(length),
                                    C.c_size_t(len(xt)), _p(xt), _p(vt), _p(dx), _p(dv), C.c_float(ds),
                                    _p(grad), C.byref(st))
    return dict(grad=grad, steps_total=st.value)
