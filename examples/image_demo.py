#!/usr/bin/env python3
"""The reference's image experiment (core/image_opt.py:23-150) written against this package's mirrors, end to end on the
device: design a volume whose near-field images (`sensor.generate_sensor`) and far-field images
(`sensor.generate_inf_sensor`) under area illumination (`source.rand_area_in_sphere`) match targets rendered from a
hidden volume -- `optimizer.multires_opt` around `tracer.BackTracerC`, `sensor.trace_rays_to_plane`, `source.sum_norm`,
optionally with the SDF-texture term `sensor.get_sdf_vals_near` (the `sdf_loss` mode of :107-110).

    python examples/image_demo.py [--res 17 33] [--iters 30] [--views 2] [--nbins 32]
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.nn.functional as F

from adjointnonlinearraytracing_amd import drrt, optimizer, sensor, source, tracer


def hidden_field(res: int, device) -> torch.Tensor:
    g = torch.linspace(0.0, 1.0, res, device=device)
    z, y, x = torch.meshgrid(g, g, g, indexing="ij")
    return (1.0 + 0.04 * torch.exp(-((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) / 0.03)).contiguous()


def run(res_list=(17, 33), views=2, iters=30, nbins=32, spp=16, span=1.0, lr=2e-3, far_span=60.0, defl_weight=0.1,
        sdf_weight=0.0, seed=0, verbose=True):
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    drrt.options.corrected_h = True            # exact discrete adjoint at h << 1 (see examples/tomography_demo.py)
    ds = span / (res_list[-1] - 1) / 2
    sensor_dist = 0.2 * span

    def gen_start_rays(samples):                                           # image_opt.py:43-53, src_type 'area'
        iv, _, tpv, rpv = source.rand_area_in_sphere(views, (nbins, nbins), samples, span, angle_span=180, circle=False,
                                                     xaxis=False, sensor_dist=sensor_dist, device=dev)
        return iv, rpv, tpv

    def sensor_list(planes, rpv):                                          # :57-66
        sp, sn, st, off = [], [], [], 0
        for cnt in rpv:
            sp.append(planes[None, off, 0, :]); sn.append(planes[None, off, 1, :]); st.append(planes[None, off, 2, :])
            off += cnt
        return sp, sn, st

    def render(n):                                                         # :88-117
        (x, v, planes), rpv, tpv = gen_start_rays(spp)
        sp_l, sn_l, st_l = sensor_list(planes, rpv)
        h = span / np.maximum(n.shape[0] - 1, 1)
        xm, vm = tracer.BackTracerC.apply(n, x, v, h, ds)
        xmp, vmp = sensor.trace_rays_to_plane((xm, vm), (planes[:, 0, :], planes[:, 1, :]))
        xs, vs, ds_ = xmp.split(rpv), vmp.split(rpv), (1 / (tpv ** 2)).split(rpv)
        near = [source.sum_norm(sensor.generate_sensor((a, b), d, (p, q), nbins, span, t))
                for a, b, p, q, t, d in zip(xs, vs, sp_l, sn_l, st_l, ds_)]
        far = [source.sum_norm(sensor.generate_inf_sensor((a, b), 1, (p, q), nbins, far_span, t))
               for a, b, p, q, t in zip(xs, vs, sp_l, sn_l, st_l)]
        return near, far, (xs, vs, sp_l, sn_l, st_l)

    truth = hidden_field(res_list[-1], dev)
    with torch.no_grad():                                                  # targets: average a few jittered renderings
        acc_n, acc_f = None, None
        for _ in range(4):
            near, far, _ = render(truth)
            acc_n = near if acc_n is None else [a + b for a, b in zip(acc_n, near)]
            acc_f = far if acc_f is None else [a + b for a, b in zip(acc_f, far)]
        disp_ims, defl_ims = [a / 4 for a in acc_n], [a / 4 for a in acc_f]
        # a smooth "distance to the bright region" texture per view for the optional SDF term (:107-110)
        sdf_tex = [F.avg_pool2d((im.max() - im)[None, None], 5, 1, 2)[0, 0].contiguous() for im in disp_ims]

    def loss_function(n):                                                  # :84-125
        near, far, (xs, vs, sp_l, sn_l, st_l) = render(n)
        loss = sum(F.mse_loss(a, b) for a, b in zip(near, disp_ims)) / len(disp_ims)
        loss = loss + defl_weight * sum(F.mse_loss(a, b) for a, b in zip(far, defl_ims))
        if sdf_weight:
            vals = [sensor.get_sdf_vals_near((a, b), tex, (p, q), span, t)
                    for a, b, tex, p, q, t in zip(xs, vs, sdf_tex, sp_l, sn_l, st_l)]
            loss = loss + sdf_weight * sum((s ** 2).sum() / s.numel() for s in vals)
        return loss

    err = []

    def log(it, n):                                                        # rms distance to the hidden volume (noise-free)
        up = n if n.shape == truth.shape else F.interpolate(n.detach()[None, None], size=truth.shape, mode="trilinear",
                                                            align_corners=True)[0, 0]
        err.append(float(((up.detach() - truth) ** 2).mean().sqrt()))
        if verbose and it % 10 == 0:
            print(f"iter {it:4d}  {n.shape[0]:3d}^3  rms(n - truth) {err[-1]:.5f}")

    eta = torch.ones((res_list[0],) * 3, device=dev)
    try:
        with tempfile.TemporaryDirectory() as tmp:
            n, hist = optimizer.multires_opt(loss_function, eta, iters, list(res_list), log_func=log, lr=lr,
                                             statename=os.path.join(tmp, "state.pt"))
    finally:
        drrt.options.corrected_h = False
    return n.detach(), truth, hist, err


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, nargs="+", default=[17, 33])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--nbins", type=int, default=32)
    ap.add_argument("--sdf-weight", type=float, default=0.0)
    a = ap.parse_args()
    n, truth, hist, err = run(tuple(a.res), a.views, a.iters, a.nbins, sdf_weight=a.sdf_weight)
    print(f"loss {hist[0]:.5f} -> {hist[-1]:.5f};  rms(n - truth) {err[0]:.5f} -> {err[-1]:.5f};  "
          f"n range [{float(n.min()):.4f}, {float(n.max()):.4f}]")
