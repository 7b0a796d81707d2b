"""Profiling-harness parity: tools/profile_sweeps.py re-creates the reference's `compare_back`
workload (src/test.cpp:117-146: rif = 1, h = 1, rays on the z = 0 face along +z, dx = dv = 1) and its two sweeps.
This 5-second smoke test guards the tool against API drift and checks the workload's closed form."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
def test_compare_back_smoke(gpu, oracle):
    import profile_sweeps
    from adjointnonlinearraytracing_amd import drrt
    drrt.options.check_failed = False
    nres, nrays, ds = 9, 64, 0.5
    t, mem, steps = profile_sweeps.compare_back(nres, nrays, ds, gpu, reps=1)
    assert t > 0 and mem > 0
    # the same workload through the CPU oracle: equal forward step totals (uniform medium: straight rays, every ray --
    # also those that start on or beyond the far x / y faces, linspace runs to nres as written in src/test.cpp:127-129 --
    # marches until z >= (nres-1)*h, i.e. ceil((nres-1)/ds) = 16 steps)
    import numpy as np
    g = np.linspace(0.0, float(nres), nrays, dtype=np.float32)
    X, Y = np.meshgrid(g, g, indexing="ij")
    pos = np.stack([X.ravel(), Y.ravel(), np.zeros(nrays * nrays, np.float32)], -1)
    vel = np.zeros_like(pos); vel[:, 2] = 1.0
    o = oracle.trace(np.ones((nres, nres, nres), np.float32), (nres, nres, nres), pos, vel, 1.0, ds, dtype=np.float32)
    assert steps == int(o["steps"].sum()) == nrays * nrays * 16
