#!/usr/bin/env python3
"""Adjoint kernel time on the ray sets that the bundle classification sends to the ring-window kernel (development probe for
its per-wave dense / sparse rule and window parameters): bench.py's six rotated views through the Luneburg ball and through
the weak medium, four tomography views (256^2 x 4 spp) through a Gaussian blob and through the ball, one 45-degree view.
usage: [DRRT_HIP_LIB=.../_variants/X.so] python tools/probe_ring_sets.py"""
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
g = torch.linspace(0.0, 1.0, R, device=dev)
z, y, x = torch.meshgrid(g, g, g, indexing="ij")
blob = (1.0 + 0.05 * torch.exp(-((x - 0.45) ** 2 + (y - 0.55) ** 2 + (z - 0.5) ** 2) / 0.03)).contiguous()
del x, y, z
T = drrt.TracerC()


def timeit(f, k=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k


def probe(name, n, xs, vs, force=None):
    res = (R, R, R)
    xt, vt = T.trace(n, res, xs, vs, h, ds)
    order = drrt.keep_order(drrt.last_order)
    ones = torch.ones_like(xt)
    out = {"case": name, "rays": int(xs.shape[0])}
    # bundle statistics at the adjoint's start (exit cells, visit order): lanes per distinct cell, per distinct (x, y) column,
    # share of bundles whose lanes all move along one dominant axis
    with torch.no_grad():
        m = (order.numel() // 64) * 64
        o = order[:m].long()
        c = torch.floor(xt[o] / h).long().clamp(0, R - 2)
        cid = (c[:, 2] * R + c[:, 1]) * R + c[:, 0]
        srt = cid.view(-1, 64).sort(dim=1).values
        distinct = 1 + (srt[:, 1:] != srt[:, :-1]).sum(dim=1)
        out["lanes_per_cell"] = round(float((64.0 / distinct.float()).mean()), 2)
        v = vt[o].abs()
        out["mean_minor_over_major"] = round(float((v.sort(dim=1).values[:, 1] / v.max(dim=1).values).mean()), 2)
        out["mean_least_over_major"] = round(float((v.min(dim=1).values / v.max(dim=1).values).mean()), 2)
    for mode in ([force] if force else ["auto", "ring_general", "ring_sparse", "ring_direct", "box"]):
        with drrt.using(adjoint_window=mode):
            ta = timeit(lambda: T.backtrace(n, res, xt, vt, ones, ones, h, ds, order=order))
            c = drrt.read_bundle_counters()
        out[mode + "_ms"] = round(ta, 3)
        if c is not None:
            out[mode + "_kernel"] = c["kernel"]
            out["not_fitting_share"], out["long_share"] = round(c["not_fitting_share"], 3), round(c["long_bundle_share"], 3)
    out["adj_ray_steps"] = drrt.read_stats()["ray_steps"]
    print(json.dumps(out), flush=True)


x6, v6, _ = bench.make_rays_cube6(1 << 20, 0, dev)
probe("six rotated views, Luneburg ball", ball, x6, v6)
probe("six rotated views, weak medium", weak, x6, v6)
(xs, vs, planes), rpv = source.rand_rays_in_sphere(4, (256, 256), 4, span, angle_span=180, circle=False, xaxis=False,
                                                   sensor_dist=0.2 * span, device=dev)
probe("4 views 0/45/90/135, 256^2 x 4 spp, Gaussian blob", blob, xs, vs)
probe("4 views 0/45/90/135, 256^2 x 4 spp, Luneburg ball", ball, xs, vs)
x1, v1, _ = source.plane_source3_rand(torch.tensor(45.0), (512, 512), 4, span, sensor_dist=0.2 * span, device=dev)
probe("one view 45 deg, 512^2 x 4 spp, Luneburg ball", ball, x1, v1)
x2, v2, _ = source.plane_source3_rand(torch.tensor(45.0), (1024, 1024), 1, span, sensor_dist=0.2 * span, device=dev)
probe("one view 45 deg, 1024^2 x 1 spp, Luneburg ball", ball, x2, v2)
x3, v3, _ = source.plane_source3_rand(torch.tensor(20.0), (724, 724), 2, span, sensor_dist=0.2 * span, device=dev)
probe("one view 20 deg, 724^2 x 2 spp, Luneburg ball", ball, x3, v3)
