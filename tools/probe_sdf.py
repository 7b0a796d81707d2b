#!/usr/bin/env python3
"""backtrace_sdf at the size of tools/run_configs.py's sdf case: which adjoint kernel runs, and how fast (development probe)."""
import sys, json, torch
sys.path.insert(0, ".")
from adjointnonlinearraytracing_amd import drrt
import bench
dev = torch.device("cuda:0")
drrt.options.check_failed = False
R = 256; span = 1.0; h = span / (R - 1); ds = h / 2
rif = bench.make_grid(R, dev)
pos, vel = (t.to(dev) for t in bench.make_rays(1 << 20, 0))
g = torch.linspace(0, span, R, device=dev)
Z, Y, X = torch.meshgrid(g, g, g, indexing="ij")
sdf = (torch.sqrt((X - .5) ** 2 + (Y - .5) ** 2 + (Z - .5) ** 2) - 0.45).contiguous()
p2 = pos.clone(); p2[:, 1] = 0.5
keep = ((p2 - 0.5).norm(dim=1) < 0.4)
p2, v2 = p2[keep].contiguous(), vel[keep].contiguous()
T = drrt.TracerC()
def timeit(f, k=6):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
xt, vt = T.trace_sdf(rif, sdf, rif.shape, p2, v2, h, ds)
order = drrt.last_order
ones = torch.ones_like(xt)
for name, kw in (("auto", {}), ("box", {"adjoint_window": "box"}), ("ring", {"adjoint_window": "ring"})):
    with drrt.using(**kw):
        t0 = timeit(lambda: T.backtrace_sdf(rif, sdf, rif.shape, xt, vt, ones, ones, h, ds))
        st = drrt.read_stats()["ray_steps"]
        t1 = timeit(lambda: T.backtrace_sdf(rif, sdf, rif.shape, xt, vt, ones, ones, h, ds, order=order)) if order is not None else float("nan")
    print(json.dumps({"window": name, "rays": int(xt.shape[0]), "adj_ray_steps": st, "own_sort_ms": round(t0, 3), "forward_order_ms": round(t1, 3)}), flush=True)
for name, kw in (("auto", {}), ("box", {"adjoint_window": "box"}), ("ring", {"adjoint_window": "ring"})):
    with drrt.using(**kw):
        t0 = timeit(lambda: T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds))
        st = drrt.read_stats()["ray_steps"]
    print(json.dumps({"plain backtrace of the same exit rays, window": name, "adj_ray_steps": st, "own_sort_ms": round(t0, 3)}), flush=True)
