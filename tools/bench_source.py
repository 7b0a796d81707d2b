#!/usr/bin/env python3
"""Times the device ray generation (csrc/drrt_source.hip) on the metric workload's source: six 512x512
views (rand_rays_cube, disc mask, fused random_rotate_ic) -- 1.57M candidates, ~1.24M rays.  Prints one
JSON line with the average call time (HIP events, excluding the count read-back) and the HBM write rate
(60 B per kept ray out + 8 B per candidate in)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adjointnonlinearraytracing_amd import source as S   # noqa: E402


def main():
    pix, spp, span = (512, 512), 1, 20.0
    M = S.random_rotmat()
    u = torch.rand(6, 2 * spp, *pix, device="cuda")
    for _ in range(3):
        (x, v, pl), nr = S.rand_rays_cube(pix, spp, span, circle=True, offset=u, rotmat=M, span=span)
    torch.cuda.synchronize()
    reps = 20
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        (x, v, pl), nr = S.rand_rays_cube(pix, spp, span, circle=True, offset=u, rotmat=M, span=span)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    n, cand = int(x.shape[0]), 6 * spp * pix[0] * pix[1]
    byts = 60 * n + 8 * cand
    print(json.dumps({"workload": "rand_rays_cube 6x512x512 circle + rotate_ic", "rays": n, "candidates": cand,
                      "ms_per_call_incl_host": ms, "GBps_algorithmic": byts / ms / 1e6}))


if __name__ == "__main__":
    main()
