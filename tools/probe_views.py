#!/usr/bin/env python3
"""Adjoint / forward kernel time vs ray-set geometry (development probe)."""
import sys, json
import torch
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

def timeit(f, k=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k

def probe(name, xs, vs, seed="ones"):
    res = (R, R, R)
    xt, vt = T.trace(n, res, xs, vs, h, ds)
    st = drrt.read_stats(); order = drrt.last_order
    if seed == "ones":
        dx, dv = torch.ones_like(xt), torch.ones_like(vt)
    else:
        dx, dv = torch.randn_like(xt), torch.randn_like(vt)
    tf = timeit(lambda: T.trace(n, res, xs, vs, h, ds))
    ta = timeit(lambda: T.backtrace(n, res, xt, vt, dx, dv, h, ds, order=order))
    st2 = drrt.read_stats()
    print(json.dumps({"case": name, "rays": int(xs.shape[0]), "fwd_ray_steps": st["ray_steps"], "adj_ray_steps": st2["ray_steps"],
                      "trace_call_ms": round(tf, 3), "backtrace_call_ms": round(ta, 3),
                      "adj_ns_per_ray_step": round(ta * 1e6 / max(st2["ray_steps"], 1), 4)}))

for ang in (0.0, 45.0, 90.0):
    xs, vs, planes = source.plane_source3_rand(torch.tensor(ang), (512, 512), 4, span, sensor_dist=0.2 * span, device=dev)
    probe(f"plane view {ang:.0f} deg, 512^2 x 4", xs, vs)
(xs, vs, planes), rpv = source.rand_rays_in_sphere(4, (256, 256), 4, span, angle_span=180, circle=False, xaxis=False,
                                                   sensor_dist=0.2 * span, device=dev)
probe("4 views 0/45/90/135, 256^2 x 4 each", xs, vs)
probe("4 views, random seeds", xs, vs, seed="randn")
tot = 0.0
for k, ang in enumerate((0.0, 45.0, 90.0, 135.0)):
    xs1, vs1, _ = source.plane_source3_rand(torch.tensor(ang), (256, 256), 4, span, sensor_dist=0.2 * span, device=dev)
    probe(f"single view {ang:.0f} deg, 256^2 x 4", xs1, vs1)
drrt.options.sort_rays = False
probe("4 views, NO sort", xs, vs)
drrt.options.sort_rays = True
drrt.options.pair_grid = False
probe("4 views, no pair copy", xs, vs)
drrt.options.pair_grid = "auto"
with drrt.using(adjoint_window="box"):
    probe("4 views, box-window adjoint kernel forced", xs, vs)
with drrt.using(adjoint_window="ring"):
    probe("4 views, ring-window adjoint kernel forced", xs, vs)
with drrt.using(chord_key=True):
    probe("4 views, rounds-1/2 chord sort key", xs, vs)
