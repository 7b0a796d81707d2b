#!/usr/bin/env python3
"""Calibration of the adjoint's kernel choice: for a spread of ray sets the classification counters
(drrt.read_bundle_counters) next to the time of the box-window and of the ring-window kernel forced."""
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
def probe(name, xs, vs, own_sort=False, grid=None):
    n = rif if grid is None else grid
    xt, vt = T.trace(n, n.shape, xs, vs, h, ds)
    order = None if own_sort else drrt.last_order
    ones = torch.ones_like(xt)
    out = {"case": name, "rays": int(xt.shape[0])}
    for w in ("auto", "box", "ring"):
        with drrt.using(adjoint_window=w):
            out[w + "_ms"] = round(timeit(lambda: T.backtrace(n, n.shape, xt, vt, ones, ones, h, ds, order=order)), 3)
            if w == "auto":
                out["counters"] = drrt.read_bundle_counters()
    print(json.dumps(out), flush=True)
N = 1 << 20
pos, vel = (t.to(dev) for t in bench.make_rays(N, 0))
probe("metric", pos, vel)
probe("metric, adjoint sorts its exit rays itself", pos, vel, own_sort=True)
ps, vs_ = (t.to(dev) for t in bench.make_rays_shifted(N, 0))
probe("plane_shifted", ps, vs_)
x6, v6, _ = bench.make_rays_cube6(N, 0, dev)
probe("cube6_rotated", x6, v6)
probe("cube6_rotated, own sort", x6, v6, own_sort=True)
g = torch.linspace(0.0, 1.0, R, device=dev)
z, y, x = torch.meshgrid(g, g, g, indexing="ij")
weak = (1.0 + 0.05 * torch.exp(-((x - 0.45) ** 2 + (y - 0.55) ** 2 + (z - 0.5) ** 2) / 0.03)).contiguous()
del x, y, z
(xs, vs, planes), rpv = source.rand_rays_in_sphere(4, (256, 256), 4, span, angle_span=180, circle=False, xaxis=False, sensor_dist=0.2 * span, device=dev)
probe("4 views (weak lens)", xs, vs, grid=weak)
for ang in (0.0, 20.0, 45.0):
    xs1, vs1, _ = source.plane_source3_rand(torch.tensor(ang), (512, 512), 4, span, sensor_dist=0.2 * span, device=dev)
    probe(f"plane view {ang:.0f} deg 1M (weak lens)", xs1, vs1, grid=weak)
    probe(f"plane view {ang:.0f} deg 1M (Luneburg)", xs1, vs1)
# exit rays on a sphere inside the lens (the sdf case of tools/run_configs.py), plain backtrace
p2 = pos.clone(); p2[:, 1] = 0.5
keep = ((p2 - 0.5).norm(dim=1) < 0.4)
p2, v2 = p2[keep].contiguous(), vel[keep].contiguous()
sdf = (torch.sqrt(((torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1) - 0.5) ** 2).sum(-1)) - 0.45).contiguous()
xt, vt = T.trace_sdf(rif, sdf, rif.shape, p2, v2, h, ds)
ones = torch.ones_like(xt)
out = {"case": "rays ending on a sphere (trace_sdf exits), backtrace_sdf sorts them itself", "rays": int(xt.shape[0])}
for w in ("auto", "box", "ring"):
    with drrt.using(adjoint_window=w):
        out[w + "_ms"] = round(timeit(lambda: T.backtrace_sdf(rif, sdf, rif.shape, xt, vt, ones, ones, h, ds)), 3)
        if w == "auto":
            out["counters"] = drrt.read_bundle_counters()
print(json.dumps(out), flush=True)
