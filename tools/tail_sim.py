#!/usr/bin/env python3
"""How much of the adjoint kernel's time is dispatch order?  The forward march leaves each ray's iteration count
(`last_order.drrt_steps`), so the length of every adjoint wave is known before the adjoint is launched.  This probe list-
schedules the blocks (the 4-wave blocks the adjoint kernels had when this was measured; duration = its longest wave) on the chip's block slots in
dispatch order and longest-first, for the bench's ray sets.  -> stdout (JSON lines)"""
import sys, json, heapq
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from adjointnonlinearraytracing_amd import drrt

dev = torch.device("cuda:0")
R = 256; h = 1.0 / (R - 1); ds = h / 2
n = bench.make_grid(R, dev)
T = drrt.TracerC()

def makespan(dur, slots):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for d in dur:
        t = heapq.heappop(free) + d
        end = max(end, t)
        heapq.heappush(free, t)
    return end

def probe(name, xs, vs, waves_per_simd):
    T.trace(n, (R, R, R), xs, vs, h, ds)
    order = drrt.last_order
    steps = order.drrt_steps.cpu().numpy().astype(np.int64)      # indexed by ray
    o = order.cpu().numpy().astype(np.int64)
    k = steps[o]                                                  # in visit order: lane t of the launch
    pad = (-len(k)) % 256
    k = np.concatenate([k, np.zeros(pad, np.int64)]).reshape(-1, 4, 64)
    wave = k.max(axis=2)                                          # a wave runs until its longest lane is done
    blk = wave.max(axis=1)
    slots = 256 * waves_per_simd                                  # blocks resident on the chip (4 waves each, 4 SIMDs per CU)
    tot = float(blk.sum())
    out = {"case": name, "blocks": int(len(blk)), "block_slots": slots,
           "wave_len_min_mean_max": [int(wave.min()), round(float(wave.mean()), 1), int(wave.max())],
           "lane_idle_in_wave": round(1.0 - k.sum() / (wave.sum() * 64.0), 4),
           "wave_idle_in_block": round(1.0 - wave.sum() / (blk.sum() * 4.0), 4),
           "ideal": round(tot / slots, 1), "in_order": round(makespan(blk, slots), 1),
           "longest_first": round(makespan(np.sort(blk)[::-1], slots), 1)}
    out["in_order_over_ideal"] = round(out["in_order"] / out["ideal"], 3)
    out["lpt_over_ideal"] = round(out["longest_first"] / out["ideal"], 3)
    print(json.dumps(out), flush=True)

N = 1 << 20
pos, vel = bench.make_rays(N, 0)
probe("metric", pos.to(dev), vel.to(dev), 5)
x, v, _ = bench.make_rays_cube6(N, 0, dev)
probe("cube6_rotated", x, v, 4)
