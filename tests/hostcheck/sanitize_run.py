"""Runs the host build of the per-ray code (csrc/drrt_device.h via tests/hostcheck) under AddressSanitizer +
UndefinedBehaviorSanitizer on the fuzz configurations with NaN / Inf / huge / denormal values planted in the rays,
the seeds and the grid.  Started by tests/test_hostcheck.py::test_sanitized_nonfinite_inputs with the sanitizer
runtime preloaded; argv[1] = the instrumented library.  (GPU sanitizers are not available on the pool.)"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np          # noqa: E402
import hostcheck_lib as H   # noqa: E402
H._lib = C.CDLL(sys.argv[1])
import cases                # noqa: E402
bad = [np.nan, np.inf, -np.inf, 3e38, -3e38, 1e30, -1e30, 1e-40, 0.0]
rng = np.random.default_rng(0)
for seed in range(6):
    c = cases.fuzz_config(seed)
    n = len(c["pos"])
    pos, vel = c["pos"].copy(), c["vel"].copy()
    dx, dv = c["dx"].copy(), c["dv"].copy()
    for arr in (pos, vel, dx, dv):
        idx = rng.integers(0, n, 60); comp = rng.integers(0, 3, 60)
        arr[idx, comp] = rng.choice(bad, 60).astype(np.float32)
    res, h, ds = c["res"], c["h"], c["ds"]
    k = H.trace(c["rif"], res, pos, vel, h, ds)
    H.trace(c["rif"], res, pos, vel, h, ds, mode="plane", pln_o=c["po"], pln_d=c["pd"])
    H.trace(c["rif"], res, pos, vel, h, ds, mode="sdf", sdf=c["sdf"])
    H.trace_target(c["rif"], res, pos, vel, c["tg"], h, ds)
    H.backtrace(c["rif"], res, pos, vel, dx, dv, h, ds)
    H.backtrace(c["rif"], res, k["xt"], k["vt"], dx, dv, h, ds, sdf=c["sdf"])
    rifbad = c["rif"].copy(); rifbad.reshape(-1)[rng.integers(0, rifbad.size, 5)] = np.nan
    H.trace(rifbad, res, pos, vel, h, ds); H.backtrace(rifbad, res, pos, vel, dx, dv, h, ds)
for seed in range(6):
    c = cases.fuzz_cable_config(seed)
    n = len(c["pos"])
    pos, vel, dx, dv = c["pos"].copy(), c["vel"].copy(), c["dx"].copy(), c["dv"].copy()
    for arr in (pos, vel, dx, dv):
        idx = rng.integers(0, n, 40); comp = rng.integers(0, 3, 40)
        arr[idx, comp] = rng.choice(bad, 40).astype(np.float32)
    k = H.trace_cable(c["prof"], c["radius"], c["length"], pos, vel, c["tg"], c["ds"])
    H.backtrace_cable(c["prof"], c["radius"], c["length"], pos, vel, dx, dv, c["ds"])
    H.backtrace_cable(c["prof"], c["radius"], c["length"], k["xt"], k["vt"], dx, dv, c["ds"])
print("sanitizer run finished without reports")
