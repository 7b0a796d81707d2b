#!/usr/bin/env python3
"""Per-bundle statistics at the adjoint's start for the angle sweep of probe_angle_sweep.py: what separates the sets the
sparse-only ring instantiation wins from those the box window wins?  (64 consecutive rays of the visit order = one bundle.)
usage: python tools/probe_bundle_stats.py"""
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


def q(t, p):
    return round(float(torch.quantile(t.float(), p)), 3)


def probe(name, n, xs, vs):
    res = (R, R, R)
    xt, vt = T.trace(n, res, xs, vs, h, ds)
    order = drrt.last_order
    steps = order.drrt_steps.long()
    m = (order.numel() // 64) * 64
    o = order[:m].long()
    x = (xt[o] / h).view(-1, 64, 3)
    v = vt[o].view(-1, 64, 3)
    k = steps[o].view(-1, 64)
    out = {"case": name}
    ext = (x.max(dim=1).values.floor() - x.min(dim=1).values.floor() + 2)          # slots per axis
    em = ext.max(dim=1).values
    out["box_extent_max_axis_mean_p90"] = [round(float(em.mean()), 2), q(em, 0.9)]
    out["mean_capped_12_16_24_32_64"] = [round(float(em.clamp(max=c).mean()), 2) for c in (12, 16, 24, 32, 64)]
    out["share_gt_7_10_12_16_32"] = [round(float((em > c).float().mean()), 3) for c in (7, 10, 12, 16, 32)]
    samp = em.view(-1)[torch.arange(0, em.numel(), 16, device=em.device)]                # (roughly what the classification samples)
    out["sampled_mean_capped_16_32"] = [round(float(samp.clamp(max=c).mean()), 2) for c in (16, 32)]
    ks = (k.max(dim=1).values - k.min(dim=1).values)
    out["steps_spread_mean_p50_p90"] = [round(float(ks.float().mean()), 1), q(ks, 0.5), q(ks, 0.9)]
    out["steps_spread_capped96_mean"] = round(float(ks.clamp(max=96).float().mean()), 2)
    out["steps_spread_share_ge_12_16_18_20_24_32"] = [round(float((ks >= c).float().mean()), 3) for c in (12, 16, 18, 20, 24, 32)]
    vm = v.mean(dim=1, keepdim=True)
    dv = ((v - vm).norm(dim=2) / vm.norm(dim=2).clamp_min(1e-9)).max(dim=1).values
    out["dir_spread_mean_p50_p90"] = [round(float(dv.mean()), 4), q(dv, 0.5), q(dv, 0.9)]
    # where the lanes stand when all have been put back on the forward clock (step hint): x + (kmax - k) * ds * v
    back = (k.max(dim=1, keepdim=True).values - k).clamp(max=96).unsqueeze(-1).float()
    xh = x + back * (ds / h) * v
    exth = (xh.max(dim=1).values.floor() - xh.min(dim=1).values.floor() + 2)
    out["hinted_extent_max_axis_mean_p90"] = [round(float(exth.max(dim=1).values.mean()), 2), q(exth.max(dim=1).values, 0.9)]
    print(json.dumps(out), flush=True)


blob = None
for gname, grid in (("ball", ball), ("weak", weak)):
    for nm, gen in (("metric source", bench.make_rays), ("metric source shifted", bench.make_rays_shifted)):
        x, v = gen(1 << 20, 0)
        probe(f"{nm}, {gname}", grid, x.to(dev), v.to(dev))
    for ang in (0.0, 1.0, 2.0, 5.0, 10.0, 20.0, 30.0, 45.0):
        x1, v1, _ = source.plane_source3_rand(torch.tensor(ang), (512, 512), 4, span, sensor_dist=0.2 * span, device=dev)
        probe(f"one view {ang:g} deg, {gname}", grid, x1, v1)
x6, v6, _ = bench.make_rays_cube6(1 << 20, 0, dev)
probe("six rotated views, ball", ball, x6, v6)
probe("six rotated views, weak", weak, x6, v6)
g = torch.linspace(0.0, 1.0, R, device=dev)
z, y, x = torch.meshgrid(g, g, g, indexing="ij")
blob = (1.0 + 0.05 * torch.exp(-((x - 0.45) ** 2 + (y - 0.55) ** 2 + (z - 0.5) ** 2) / 0.03)).contiguous()
(xs, vs, planes), rpv = source.rand_rays_in_sphere(4, (256, 256), 4, span, angle_span=180, circle=False, xaxis=False,
                                                   sensor_dist=0.2 * span, device=dev)
probe("4 views, blob", blob, xs, vs)
probe("4 views, ball", ball, xs, vs)
