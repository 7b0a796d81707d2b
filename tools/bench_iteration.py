#!/usr/bin/env python3
"""Times ONE WHOLE optimisation iteration of the reference's image experiment (core/image_opt.py:88-119 +
core/optimizer.py:56-69) at the metric's size through this package's drop-in API, phase by phase:

    ray generation (source.rand_rays_in_sphere, HIP)  ->  tracer.BackTracerC forward (sort, pair copy, march)
    ->  sensor.trace_rays_to_plane + sensor.generate_sensor per view (HIP splat)  ->  MSE loss
    ->  backward (sensor backward, adjoint march)  ->  optimizer.MaskedAdam (mask + Adam + clamp, HIP)

usage: bench_iteration.py [--grid 256] [--views 4] [--nbins 256] [--spp 4] [--iters 10]      (views*nbins^2*spp rays)
Prints one JSON object (ms per phase, averaged; the phases are separated by device syncs, so their sum is a little
above the unsynchronised iteration time that is also reported)."""
import argparse
import json
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from adjointnonlinearraytracing_amd import drrt, optimizer, sensor, source, tracer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--nbins", type=int, default=256)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    drrt.options.check_failed = False
    R, span = a.grid, 1.0
    h = span / (R - 1)
    ds = h / 2
    g = torch.linspace(0.0, 1.0, R, device=dev)
    z, y, x = torch.meshgrid(g, g, g, indexing="ij")
    n = (1.0 + 0.05 * torch.exp(-((x - 0.45) ** 2 + (y - 0.55) ** 2 + (z - 0.5) ** 2) / 0.03)).contiguous().requires_grad_(True)
    del x, y, z
    n0 = n.detach().clone()               # the field is put back after every step: this times an iteration on a smooth
                                          # medium, not Adam's first sign-like steps against an arbitrary target (which
                                          # turn the volume into noise and the ray bundles incoherent)
    opt = optimizer.MaskedAdam([n], lr=1e-3)
    target = [torch.ones(a.nbins, a.nbins, device=dev) for _ in range(a.views)]
    ph = {k: 0.0 for k in ("ray_generation", "forward_march", "sensor_and_loss", "backward", "adam_mask_clamp")}

    def sync():
        torch.cuda.synchronize(dev)
        return time.perf_counter()

    def iteration(timed):
        t0 = sync() if timed else 0.0
        rays, rpv = source.rand_rays_in_sphere(a.views, (a.nbins, a.nbins), a.spp, span, angle_span=180, circle=False,
                                               xaxis=False, sensor_dist=0.2 * span, device=dev)
        xs, vs, planes = rays
        t1 = sync() if timed else 0.0
        opt.zero_grad()
        xt, vt = tracer.BackTracerC.apply(n, xs, vs, h, ds)
        t2 = sync() if timed else 0.0
        sp, sn = planes[:, 0, :], planes[:, 1, :]
        xp, vp = sensor.trace_rays_to_plane((xt, vt), (sp, sn))
        loss, off = 0.0, 0
        for k, cnt in enumerate(rpv):
            pl = planes[off]
            img = sensor.generate_sensor((xp[off:off + cnt], vp[off:off + cnt]), 1.0, (pl[None, 0], pl[None, 1]), a.nbins,
                                         span, pl[None, 2])
            loss = loss + F.mse_loss(img * (img.numel() / img.sum()), target[k])
            off += cnt
        t3 = sync() if timed else 0.0
        loss.backward()
        t4 = sync() if timed else 0.0
        opt.step()
        t5 = sync() if timed else 0.0
        with torch.no_grad():
            n.copy_(n0)
        if timed:
            for key, d in zip(ph, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                ph[key] += d * 1e3
        return xs.shape[0]

    for _ in range(3):
        nrays = iteration(False)
    for _ in range(a.iters):
        iteration(True)
    t0 = sync()
    for _ in range(a.iters):
        iteration(False)
    whole = (sync() - t0) / a.iters * 1e3
    st = drrt.read_stats()
    print(json.dumps({"grid": R, "rays_per_iteration": int(nrays), "views": a.views, "sensor": a.nbins,
                      "phase_ms": {k: v / a.iters for k, v in ph.items()}, "iteration_ms_unsynchronised": whole,
                      "last_call_ray_steps": st["ray_steps"]}))


if __name__ == "__main__":
    main()
