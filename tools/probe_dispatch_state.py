#!/usr/bin/env python3
"""Does what ran BEFORE the adjoint change its speed at low occupancy?  (NOTES.md, round 4: the adjoint of a 1/8 shard of the
metric runs 8 % slower when a differently configured sort ran 0.5 ms earlier -- same kernel binary, same inputs.)  Shard 3 of
8 of the metric's rays; between the forward march and the adjoint a dummy elementwise kernel over `m` floats is launched
(m / 256-thread blocks of various counts: the hardware's workgroup dispatcher keeps its round-robin position across kernels);
adjoint time by HIP events around the call.
usage: python tools/probe_dispatch_state.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
import bench                                                    # noqa: E402
from adjointnonlinearraytracing_amd import drrt                 # noqa: E402

dev = torch.device("cuda:0")
drrt.options.check_failed = False
drrt.options.sort_rays = True
R = 256; h = 1.0 / (R - 1); ds = h / 2
rif = bench.make_grid(R, dev)
x, v = bench.make_rays(1 << 20, 0)
n = (1 << 20) // 8
x = x[3 * n:4 * n].to(dev).contiguous(); v = v[3 * n:4 * n].to(dev).contiguous()
T = drrt.TracerC()
ones = torch.ones_like(x)
scratch = torch.zeros(1 << 22, device=dev)


def run(m, reps=6):
    ts = []
    for _ in range(reps):
        xt, vt = T.trace(rif, (R, R, R), x, v, h, ds)
        order = drrt.last_order
        if m > 0:
            scratch[:m].add_(1.0)                               # the dummy kernel
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        T.backtrace(rif, (R, R, R), xt, vt, ones, ones, h, ds, order=order)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return round(ts[len(ts) // 2], 3), round(ts[0], 3)


out = {}
for m in (0, 1, 256, 512, 768, 1024, 1280, 1536, 1792, 2048, 4096, 65536, 1 << 20, 1 << 22):
    out[m] = run(m)
    print(m, out[m], flush=True)
print(json.dumps({"kernel": drrt.read_bundle_counters()["kernel"], "median_min_ms_by_dummy_elements": out}))
