#!/usr/bin/env python3
"""Times the tail of one multires_opt iteration (core/optimizer.py:61-66: boundary mask, Adam step, clamp) at R^3:
the reference's torch statements vs the fused HIP pass (optimizer.MaskedAdam).  usage: bench_optim.py [R] [iters]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from adjointnonlinearraytracing_amd import optimizer  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
n0 = 1.0 + 0.3 * torch.rand(R, R, R, device=dev)
grad = torch.randn(R, R, R, device=dev)


def run(fused):
    n = n0.clone().requires_grad_(True)
    o = optimizer.MaskedAdam([n], lr=1e-3) if fused else torch.optim.Adam([n], lr=1e-3)
    mask = torch.ones_like(n, dtype=torch.bool)
    mask[1:-1, 1:-1, 1:-1] = 0
    def it():
        n.grad = grad.clone()
        with torch.no_grad():
            if not fused:
                n.grad[mask] = 0
        o.step()
        with torch.no_grad():
            if not fused:
                n.clamp_(min=1)
    for _ in range(3):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        it()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


tc = None
n = n0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    g = grad.clone()
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / K * 1e3
print(json.dumps({"grid": R, "iters": K, "grad_clone_ms": tc, "torch_mask_adam_clamp_ms": run(False) - tc,
                  "fused_masked_adam_ms": run(True) - tc}))
