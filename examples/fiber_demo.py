#!/usr/bin/env python3
"""The reference's fibre experiment (core/fiber_opt.py:100-280) on this package's mirrors: optimise the RADIAL index profile
of a cylindrical fibre so that light entering through a cone (or as a plane wave) refocuses on the axis point it left
from, one "hop" further down, and again a hop later -- `tracer.BackCableTracerC` (HIP cable march + adjoint),
a three-line radial lookup for the boundary index (the reference: `cable.Cable.GetLinear`, core/fiber_opt.py:158-160), `source.cone_source3_rand` / `plane_source3_rand`, Adam with the
experiment's own midpoint up-sampling between levels.

    python examples/fiber_demo.py [--res 5 9 17] [--iters 30] [--nbins 32] [--src cone|planar]
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.optim as optim

from adjointnonlinearraytracing_amd import drrt, source, tracer


def radial_index(profile: torch.Tensor, radius: float, pts: torch.Tensor) -> torch.Tensor:
    """n at the points' distance from the fibre axis (x = z = radius): the profile's samples lie h = radius / (len - 1)
    apart, linear in between, constant beyond the last one.  Differentiable w.r.t. the profile."""
    r = torch.hypot(pts[:, 0] - radius, pts[:, 2] - radius) * ((profile.shape[0] - 1) / radius)
    lo = r.floor().long().clamp(0, profile.shape[0] - 1)
    hi = (lo + 1).clamp(max=profile.shape[0] - 1)
    return torch.lerp(profile[lo], profile[hi], (r - lo).clamp(0, 1))


def upres_scene(n: torch.Tensor) -> torch.Tensor:
    """core/fiber_opt.py:60-68: insert the midpoints (k samples -> 2k - 1)."""
    nn = torch.zeros((n.shape[0] - 1) * 2 + 1, device=n.device, dtype=n.dtype)
    nn[::2] = n
    nn[1::2] = (n[1:] + n[:-1]) / 2
    return nn.requires_grad_(True)


def run(res_list=(5, 9, 17), iters=30, nbins=32, spp=1, src_type="cone", cable_length=5.0, cable_radius=1.0,
        camera_span=0.1, cone_ang=60.0, sensor_dist=1.57, hop_dist=3.14, hop_weight=0.1, lr=0.01, plane_eps=0.001,
        seed=0, verbose=True):
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    drrt.options.check_failed = False

    def gen_start_rays():                                                 # :124-130
        if src_type == "planar":
            return source.plane_source3_rand(torch.tensor([0.0]), (nbins, nbins), spp, cable_radius * 2, circle=True,
                                             sensor_dist=sensor_dist - cable_radius * 2, device=dev)
        return source.cone_source3_rand(torch.tensor(0.0), (nbins, nbins), spp, cable_radius * 2,
                                        sensor_dist=sensor_dist, cone_angle=cone_ang, device=dev)

    def trace(nt, rays, target):                                          # :152-163
        x, v = rays
        sds = cable_radius / nt.shape[0] / 2
        v = v / radial_index(nt, cable_radius, x)[:, None]
        return tracer.BackCableTracerC.apply(nt, cable_radius, cable_length, x, v, target, sds)

    n = torch.ones(res_list[0], device=dev).requires_grad_(True)
    opto = optim.Adam([n], lr=lr)
    hist = []
    for level in range(len(res_list)):
        for _ in range(iters * (level + 1)):                              # :176
            opto.zero_grad()
            x, v, planes = gen_start_rays()
            nrays = x.shape[0]
            sp, sn = planes[:, 0, :], planes[:, 1, :]
            total = 0.0
            for target, weight in ((sp, 1.0), (sp + hop_dist * sn, hop_weight)):     # :196-216
                xm, vm, dist2 = trace(n, (x, v), target)
                keep = dist2 > plane_eps ** 2
                loss = weight * torch.sum((xm[keep] - target[keep]) ** 2 / nrays / cable_radius) / camera_span
                loss.backward()
                total += float(loss.detach())
            with torch.no_grad():
                n.grad[-1] = 0                                            # :240 the cladding sample stays fixed
            opto.step()
            hist.append(total)
            if verbose and len(hist) % 10 == 1:
                print(f"level {n.shape[0]:3d}  iter {len(hist):4d}  loss {total:.5f}")
        if level < len(res_list) - 1:                                     # :253-256
            n = upres_scene(n.detach())
            opto = optim.Adam([n], lr=(0.5 ** level) * lr)
    drrt.options.check_failed = True
    return n.detach(), hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, nargs="+", default=[5, 9, 17])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--nbins", type=int, default=32)
    ap.add_argument("--src", default="cone")
    a = ap.parse_args()
    n, hist = run(tuple(a.res), a.iters, a.nbins, src_type=a.src)
    print(f"loss {hist[0]:.5f} -> {hist[-1]:.5f};  profile (axis -> cladding): "
          + " ".join(f"{float(t):.3f}" for t in n[:: max(1, n.shape[0] // 8)]))
