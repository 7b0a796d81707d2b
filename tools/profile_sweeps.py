#!/usr/bin/env python3
"""Counterpart of the reference's profiling executable (src/test.cpp:117-318, target `run`): wall
time and device memory of trace + backtrace ("BA" columns; the enoki-autodiff "AD" columns have no
counterpart here) as a function of 1/ds (`profile_stepsize`, :148-206: 33^3 grid, 512^2 rays, h = 1)
and of the grid resolution (`profile_resolution`, :241-318: R in {3..257}, 256^2 rays, ds = 0.5).
Workload = `compare_back` (:117-146): rif = 1, rays on the z = 0 face along +z, dx = dv = 1.

    python tools/profile_sweeps.py > profiles/r1_sweeps.txt        (needs a GPU)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from adjointnonlinearraytracing_amd import drrt


def compare_back(nres, nrays, ds, dev, reps=3):
    h = 1.0
    rif = torch.ones(nres, nres, nres, device=dev)
    g = torch.linspace(0.0, float(nres), nrays, device=dev)               # :127-129 (as written: up to nres)
    X, Y = torch.meshgrid(g, g, indexing="ij")
    pos = torch.stack([X.flatten(), Y.flatten(), torch.zeros(nrays * nrays, device=dev)], -1)
    vel = torch.zeros_like(pos)
    vel[:, 2] = 1.0
    ones = torch.ones_like(pos)
    T = drrt.TracerC()
    best = 1e9
    steps = 0
    for _ in range(reps):
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        xt, vt = T.trace(rif, rif.shape, pos, vel, h, ds)
        fs = drrt.last_stats
        order = drrt.last_order
        g_ = T.backtrace(rif, rif.shape, xt, vt, ones, ones, h, ds, order=order)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        steps = int(fs[0].item())
    return best, torch.cuda.max_memory_allocated() / 2 ** 20, steps


def main():
    dev = torch.device("cuda:0")
    drrt.options.check_failed = False
    compare_back(9, 64, 0.5, dev)                                         # warm-up (library load, allocator)
    print("# profile_stepsize (src/test.cpp:148-206): 33^3, 512^2 rays, h=1")
    print("1/ds  back_time_s  back_mem_MiB  fwd_ray_steps  ray_steps_per_s")
    for ds in (0.3, 0.33, 0.37, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1, 1.2):
        t, m, s = compare_back(33, 512, ds, dev)
        print(f"{1 / ds:.3f}  {t:.6f}  {m:.1f}  {s}  {s / t:.3e}")
    print("# profile_resolution (src/test.cpp:241-318): 256^2 rays, ds=0.5, h=1")
    print("nres  back_time_s  back_mem_MiB  fwd_ray_steps  ray_steps_per_s")
    for nres in (3, 5, 9, 17, 33, 65, 129, 257):
        t, m, s = compare_back(nres, 256, 0.5, dev)
        print(f"{nres}  {t:.6f}  {m:.1f}  {s}  {s / t:.3e}")


if __name__ == "__main__":
    main()
