#!/usr/bin/env python3
"""Which adjoint kernel for which obliqueness?  One plane view (512^2 x 4 spp) through the Luneburg ball and through the weak
medium at a sweep of angles about z, plus the metric's own source and its shifted copy: adjoint time with the box window, the
general ring instantiation and the sparse-only one (fixed-point window), each forced.  Development probe for the kernel choice.
usage: [DRRT_HIP_LIB=...] python tools/probe_angle_sweep.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
import bench                                                    # noqa: E402
from adjointnonlinearraytracing_amd import drrt, source         # noqa: E402

dev = torch.device("cuda:0")
drrt.options.check_failed = False
R = 256; span = 1.0; h = span / (R - 1); ds = h / 2
ball = bench.make_grid(R, dev)
weak = bench.make_grid_tomo(R, dev)
T = drrt.TracerC()


def timeit(f, k=4):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k


def probe(name, n, xs, vs):
    res = (R, R, R)
    xt, vt = T.trace(n, res, xs, vs, h, ds)
    order = drrt.keep_order(drrt.last_order)
    ones = torch.ones_like(xt)
    out = {"case": name}
    v = vt.abs()
    out["minor_over_major"] = round(float((v.sort(dim=1).values[:, 1] / v.max(dim=1).values).mean()), 3)
    for mode in ("auto", "box", "ring_general", "ring_sparse", "ring_direct"):
        with drrt.using(adjoint_window=mode):
            ta = timeit(lambda: T.backtrace(n, res, xt, vt, ones, ones, h, ds, order=order))
            c = drrt.read_bundle_counters()
        out[mode] = round(ta, 3)
        if mode == "auto" and c is not None:
            out["auto_kernel"] = c["kernel"]
            out["not_fitting"], out["long_share"] = round(c["not_fitting_share"], 3), round(c["long_bundle_share"], 3)
    ns = drrt.read_stats()["ray_steps"]
    out["best"] = min(("box", "ring_general", "ring_sparse", "ring_direct"), key=lambda m: out[m])
    out["ray_steps"] = ns
    print(json.dumps(out), flush=True)


for gname, grid in (("ball", ball), ("weak", weak)):
    for name, gen in (("metric source", bench.make_rays), ("metric source shifted", bench.make_rays_shifted)):
        x, v = gen(1 << 20, 0)
        probe(f"{name}, {gname}", grid, x.to(dev), v.to(dev))
    for ang in (0.0, 1.0, 2.0, 5.0, 10.0, 20.0, 30.0, 45.0):
        x1, v1, _ = source.plane_source3_rand(torch.tensor(ang), (512, 512), 4, span, sensor_dist=0.2 * span, device=dev)
        probe(f"one view {ang:g} deg, 512^2 x 4 spp, {gname}", grid, x1, v1)
