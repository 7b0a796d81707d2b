#!/usr/bin/env python3
"""The multi-view lines of tools/probe_views.py only (A/B of library variants)."""
import sys, json, torch
sys.path.insert(0, ".")
from adjointnonlinearraytracing_amd import drrt, source
dev = torch.device("cuda:0")
drrt.options.check_failed = False
R = 256; span = 1.0; h = span / (R - 1); ds = h / 2
g = torch.linspace(0.0, 1.0, R, device=dev)
z, y, x = torch.meshgrid(g, g, g, indexing="ij")
n = (1.0 + 0.05 * torch.exp(-((x - 0.45) ** 2 + (y - 0.55) ** 2 + (z - 0.5) ** 2) / 0.03)).contiguous()
del x, y, z
T = drrt.TracerC()
def timeit(f, k=8):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
def probe(name, xs, vs):
    res = (R, R, R)
    xt, vt = T.trace(n, res, xs, vs, h, ds)
    order = drrt.last_order
    dx, dv = torch.ones_like(xt), torch.ones_like(vt)
    tf = timeit(lambda: T.trace(n, res, xs, vs, h, ds))
    ta = timeit(lambda: T.backtrace(n, res, xt, vt, dx, dv, h, ds, order=order))
    print(json.dumps({"case": name, "fwd": round(tf, 3), "adj": round(ta, 3)}), flush=True)
(xs, vs, planes), rpv = source.rand_rays_in_sphere(4, (256, 256), 4, span, angle_span=180, circle=False, xaxis=False, sensor_dist=0.2 * span, device=dev)
probe("4 views", xs, vs)
xs1, vs1, _ = source.plane_source3_rand(torch.tensor(45.0), (512, 512), 4, span, sensor_dist=0.2 * span, device=dev)
probe("45 deg 1M", xs1, vs1)
