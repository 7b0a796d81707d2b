#!/usr/bin/env python3
"""Diagnostic (GPU box): where does a wave of k_backtrace_ring spend its TIME?  Needs a library variant built with
-DDRRT_RING_STAMPS (tools/build_variant.sh stamps "-DDRRT_RING_STAMPS"); runs the six-rotated-views workload of bench.py
once through the C ABI and prints the share of wave-cycles per region of an iteration (s_memtime brackets; the stamps
themselves cost ~10 % -- read the shares, not the absolute time).
usage: DRRT_HIP_LIB=.../_variants/stamps.so python tools/ring_stamps.py [workload]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
import bench                                                    # noqa: E402
from adjointnonlinearraytracing_amd import _lib, drrt           # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cube6_rotated"
dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.drrt_debug_ring_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
R, n = 256, 1 << 20
h = 1.0 / (R - 1); ds = h / 2
rif = bench.make_grid_tomo(R, dev) if workload == "tomo_weak" else bench.make_grid(R, dev)
if workload == "metric":
    pos, vel = (t.to(dev) for t in bench.make_rays(n, 0))
else:
    pos, vel, _ = bench.make_rays_cube6(n, 0, dev)
T = drrt.TracerC()
if workload == "metric":
    drrt.options.adjoint_window = "ring"
fe = getattr(lib, "drrt_debug_ring_events", None)
if fe is not None:
    fe.restype, fe.argtypes = C.c_int, [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 8)()
ev = (C.c_ulonglong * 8)()
for rep in range(3):
    xt, vt = T.trace(rif.reshape(-1), (R, R, R), pos, vel, h, ds)
    order = drrt.last_order
    ones = torch.ones_like(xt)
    torch.cuda.synchronize()
    assert fn(None, 1) == 0
    if fe is not None:
        fe(None, 1)
    g = T.backtrace(rif.reshape(-1), (R, R, R), xt, vt, ones, ones, h, ds, order=order)
    torch.cuda.synchronize()
    assert fn(out, 0) == 0
    if fe is not None:
        fe(ev, 0)
v = [int(x) for x in out]
names = ["top + step hint", "window service", "sample + weights (waits for taps)", "step + locate + gather issue + lambda/mu",
         "leave: hand-over + new slot", "epilogue"]
tot = v[0]
print(json.dumps({"workload": workload, "waves": v[7], "wave_cycles_total": tot, "cycles_per_wave": tot / max(v[7], 1),
                  "share": {nm: round(v[1 + k] / tot, 4) for k, nm in enumerate(names)},
                  "adjoint_kernel": drrt.read_bundle_counters(),
                  "fixed_point_events": dict(zip(["forced_flush_budget", "forced_flush_asked", "looked_and_kept", "rescales",
                                                  "lane_handovers_to_grid_by_guard", "big_class_lane_emits"], [int(x) for x in ev][:6]))}))
