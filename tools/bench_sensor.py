#!/usr/bin/env python3
"""Times the fused sensor splat (csrc/drrt_sensor.hip) for 1M rays onto a 512^2 image in its two regimes:
spread-out rays (a few rays per pixel) and Luneburg-focused rays (most rays on a handful of pixels), forward and
backward.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from adjointnonlinearraytracing_amd import drrt, sensor        # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    dev = torch.device("cuda:0")
    drrt.options.check_failed = False
    n, res, span = 1 << 20, 512, 1.0
    torch.manual_seed(0)
    out = {}
    # spread-out
    x = torch.rand(n, 3, device=dev) * 0.8 + 0.1
    x[:, 1] = span
    v = torch.randn(n, 3, device=dev) * 0.1
    v[:, 1] = 1.0
    p = torch.tensor([[0.5, 1.1, 0.5]], device=dev)
    nn = torch.tensor([[0.0, 1.0, 0.0]], device=dev)
    tt = torch.tensor([[0.0, 0.0, 1.0]], device=dev)
    gI = torch.randn(res, res, device=dev)
    for tag, (xs, vs, pp) in {"spread": (x, v, p)}.items():
        out[tag + "_fwd_ms"] = timed(lambda: sensor.generate_sensor((xs, vs), 1.0, (pp, nn), res, span, tt))
    # focused: exit rays of the benchmark's Luneburg march, sensor on the far face
    rif, pos, vel, h, ds = bench.make_workload(256, 1024 * 1024, dev, seed=0)
    xt, vt = drrt.TracerC().trace(rif, (256, 256, 256), pos, vel, h, ds)
    pf = torch.tensor([[0.5, 1.0, 0.5]], device=dev)
    out["focused_fwd_ms"] = timed(lambda: sensor.generate_sensor((xt, vt), 1.0, (pf, nn), res, span, tt))
    for tag, (xs, vs, pp) in {"spread": (x, v, p), "focused": (xt, vt, pf)}.items():
        xg, vg = xs.clone().requires_grad_(True), vs.clone().requires_grad_(True)

        def both():
            xg.grad = None; vg.grad = None
            (sensor.generate_sensor((xg, vg), 1.0, (pp, nn), res, span, tt) * gI).sum().backward()
        out[tag + "_fwd_bwd_ms"] = timed(both)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
