#!/usr/bin/env python3
"""Why is an oblique view through the Luneburg ball 1.8x the aligned one in the box-window adjoint?  Ablations of the
box kernel (development instantiation: 7 = nothing ablated, 1 = no hand-over at all, 2 = no global atomics, 3 = no LDS
adds, 5 = never flush before the end) on one 1M-ray plane view at 0 and 45 degrees."""
import sys, json, torch
sys.path.insert(0, ".")
from adjointnonlinearraytracing_amd import drrt, source
import bench
dev = torch.device("cuda:0")
drrt.options.check_failed = False
R = 256; span = 1.0; h = span / (R - 1); ds = h / 2
rif = bench.make_grid(R, dev)
T = drrt.TracerC()
def timeit(f, k=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for ang in (0.0, 45.0):
    xs, vs, _ = source.plane_source3_rand(torch.tensor(ang), (512, 512), 4, span, sensor_dist=0.2 * span, device=dev)
    xt, vt = T.trace(rif, rif.shape, xs, vs, h, ds)
    fwd = timeit(lambda: T.trace(rif, rif.shape, xs, vs, h, ds))
    order = drrt.last_order
    ones = torch.ones_like(xt)
    out = {"view_deg": ang, "fwd_ms": round(fwd, 3), "fwd_ray_steps": drrt.read_stats()["ray_steps"]}
    with drrt.using(adjoint_window="box"):
        for x in (0, 7, 1, 2, 3, 5):
            drrt._EXPERIMENT = x
            out[f"x{x}"] = round(timeit(lambda: T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=order)), 3)
        drrt._EXPERIMENT = 0
    out["adj_ray_steps"] = drrt.read_stats()["ray_steps"]
    print(json.dumps(out), flush=True)
