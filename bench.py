#!/usr/bin/env python3
"""Headline benchmark: ray-steps/sec (fwd + adjoint) on the synthetic 256^3 / 1M-ray / 512-step
workload of BASELINE.json.

One "step" = one forward march (drrt_trace_f32) + one adjoint march (drrt_backtrace_f32) over
one batch of rays, inputs resident in HBM, called through the C ABI (ctypes) on torch's current
stream.  With N > 1 ranks (torchrun, one process per GPU, backend nccl = RCCL) every rank marches
its own batch of `--rays` rays (weak scaling) against a replicated grid and the per-rank dL/dn
grids are summed by ONE all-reduce per step -- the path's only exchange.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      -- the dominant kernel (adjoint march): algorithmic bytes / launch duration,
                   measured live with HIP events recorded around the kernel inside the library
  roofline_fwd  -- same for the forward march kernel (the 40 %-of-HBM target of north_star)
  cpu_baseline  -- the CPU oracle (plain-C port of the reference, 1 thread) timed on a bounded
                   sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.29 TB/s measured copy
B_FWD = 32.0                   # algorithmic bytes per forward ray-step: 8 fp32 taps
B_ADJ = 64.0                   # adjoint: 8 taps read + 8 fp32 atomic adds


def make_workload(R: int, n_rays: int, device, seed: int):
    """Luneburg ball on an R^3 grid + jittered plane source on the y=0 face, v=(0,1,0)
    (plane_source3_rand-equivalent, /root/reference/core/source.py:54-69: pixel (i,j) -> position
    (x,0,z) with j (z) the fast index).  Rays are NOT pre-sorted."""
    span = 1.0
    h = span / (R - 1)
    ds = h / 2                                  # step_res = 2 (core/luneburg_opt.py:38,48-49)
    g = torch.linspace(0.0, span, R, device=device)
    Z, Y, X = torch.meshgrid(g, g, g, indexing="ij")
    r = torch.sqrt((X - span / 2) ** 2 + (Y - span / 2) ** 2 + (Z - span / 2) ** 2) / (span / 2)
    rif = torch.sqrt(2.0 - torch.clamp(r, max=1.0) ** 2).to(torch.float32).contiguous()
    del X, Y, Z, r
    side = int(round(n_rays ** 0.5))
    assert side * side == n_rays, "--rays must be a perfect square (plane source pixels)"
    gen = torch.Generator(device="cpu").manual_seed(seed)
    off = torch.rand(2, side, side, generator=gen)
    i = torch.arange(side, dtype=torch.float32)[:, None].expand(side, side)
    j = torch.arange(side, dtype=torch.float32)[None, :].expand(side, side)
    x = (i + off[0]) / side * span
    z = (j + off[1]) / side * span
    pos = torch.stack([x.flatten(), torch.zeros(n_rays), z.flatten()], dim=-1).clamp_(0.0, span * (1 - 1e-6))
    pos[:, 1] = 0.0
    vel = torch.zeros(n_rays, 3)
    vel[:, 1] = 1.0
    return rif, pos.to(device), vel.to(device), h, ds


def measured_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel, from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc.json, produced by tools/profile_bench.sh + tools/condense_profile.py on the SAME
    command: separate --pmc passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when no summary matches."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=os.path.getmtime)
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if k.startswith(kernel_prefix) and "hbm_traffic_bytes_per_launch" in v:
                return {"bytes": v["hbm_traffic_bytes_per_launch"], "source": os.path.basename(f)}
    return None


def cpu_baseline(R, h, ds, rif_np, pos_np, vel_np, target_seconds=15.0):
    """Time the CPU oracle (kind 'port': plain-C restatement of the reference, single thread --
    the reference's CPU path is single-threaded) on a ray sub-sample."""
    from oracle import oracle as O
    O.build()
    n_probe = 512
    sel = np.linspace(0, len(pos_np) - 1, n_probe).astype(np.int64)
    t0 = time.perf_counter()
    o = O.trace(rif_np, (R, R, R), pos_np[sel], vel_np[sel], h, ds, dtype=np.float32)
    b = O.backtrace(rif_np, (R, R, R), o["xt"], o["vt"], np.ones_like(o["xt"]), np.ones_like(o["xt"]), h, ds)
    t_probe = time.perf_counter() - t0
    n = int(min(len(pos_np), max(n_probe, n_probe * target_seconds / max(t_probe, 1e-3))))
    sel = np.linspace(0, len(pos_np) - 1, n).astype(np.int64)
    t0 = time.perf_counter()
    o = O.trace(rif_np, (R, R, R), pos_np[sel], vel_np[sel], h, ds, dtype=np.float32)
    t1 = time.perf_counter()
    b = O.backtrace(rif_np, (R, R, R), o["xt"], o["vt"], np.ones_like(o["xt"]), np.ones_like(o["xt"]), h, ds)
    t2 = time.perf_counter()
    steps = int(o["steps"].sum())
    # all-cores variant of the same port (OpenMP over contiguous ray chunks, private gradient grids):
    # what a maintainer would get from the host without a GPU; reported beside the 1-thread figure.
    threads = max(1, min(len(os.sched_getaffinity(0)), 32))
    n_all = int(min(len(pos_np), n * max(1, threads // 2)))
    sel_all = np.linspace(0, len(pos_np) - 1, n_all).astype(np.int64)
    a = O.bench_allcores(rif_np, (R, R, R), pos_np[sel_all], vel_np[sel_all], h, ds, threads)
    allcores = {"value": a["fwd_steps"] / (a["t_fwd"] + a["t_adj"]), "unit": "ray-steps/s", "cores": a["threads"],
                "sample": f"{n_all} rays, fwd {a['t_fwd']:.2f}s + adjoint {a['t_adj']:.2f}s (incl. summing "
                          f"{a['threads']} private grids)"}
    return {
        "allcores": allcores,
        "value": steps / (t2 - t0), "unit": "ray-steps/s", "cores": 1, "kind": "port",
        "sample": f"{n} of the workload's rays (evenly strided), {R}^3 grid, fwd {t1 - t0:.2f}s + adjoint {t2 - t1:.2f}s, "
                  f"{steps} fwd ray-steps",
        "fwd_ray_steps_per_s": steps / (t1 - t0), "adj_ray_steps_per_s": b["steps_total"] / (t2 - t1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--rays", type=int, default=1024 * 1024, help="rays per GPU (perfect square)")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--direct-atomics", action="store_true")
    ap.add_argument("--lds-bricks", action="store_true", help="forward: opt-in LDS-staged grid bricks")
    ap.add_argument("--no-order-reuse", action="store_true",
                    help="adjoint computes its own visit order instead of reusing the forward's")
    ap.add_argument("--quad", action="store_true", help="opt-in: build / use the 16-byte quad copy of the grid "
                                                         "(DRRT_FLAG_QUAD_GRID; forward builds it, adjoint reuses it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-counters", action="store_true", help="print LDS-window counters (stderr)")
    ap.add_argument("--experiment", type=int, default=0, help="development ablation id (0 = product)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only "
                                                        "to rehearse the multi-rank flow on a 1-GPU box)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL / gloo print banners to the process's stdout from
    # native code (e.g. "RCCL version : ..."), so file descriptor 1 is pointed at stderr until the final
    # print and restored just for it.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but only {ndev} GPUs visible")
    local = local % ndev                         # gloo rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun: also with a single rank (rehearses RCCL)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    from adjointnonlinearraytracing_amd import _lib
    lib = _lib.load()            # loud failure if the HIP library is missing

    R = args.grid
    rif, pos, vel, h, ds = make_workload(R, args.rays, dev, seed=rank)
    n = pos.shape[0]
    nvox = rif.numel()
    res = (C.c_int * 3)(R, R, R)
    flags = 0 if args.no_sort else _lib.FLAG_SORT_RAYS
    if args.quad and not args.lds_bricks:
        flags |= _lib.FLAG_QUAD_GRID                   # forward builds the quad copy, the paired adjoint reuses it
    fflags = flags | (_lib.FLAG_LDS_BRICKS if args.lds_bricks else 0)
    aflags = flags | (_lib.FLAG_DIRECT_ATOMICS if args.direct_atomics else 0)
    if flags & _lib.FLAG_QUAD_GRID:
        aflags |= _lib.FLAG_QUAD_REUSE
    aflags |= (_lib.FLAG_DEBUG_COUNTERS if args.debug_counters else 0) | ((args.experiment & 0xff) << 8)
    ws = torch.empty(int(lib.drrt_workspace_bytes_grid(n, nvox, flags)) + 1024, dtype=torch.uint8, device=dev)
    xt, vt = torch.empty_like(pos), torch.empty_like(vel)
    dx, dv = torch.ones_like(pos), torch.ones_like(vel)          # adjoint seed dx=dv=1 (src/test.cpp:142-144)
    grad = torch.empty(nvox, dtype=torch.float32, device=dev)
    st_f = torch.zeros(3, dtype=torch.int64, device=dev)
    st_a = torch.zeros(3, dtype=torch.int64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())

    def step():
        _lib.check(lib.drrt_trace_f32(p(rif), nvox, res, n, p(pos), p(vel), h, ds, p(xt), p(vt),
                                      p(st_f), p(ws), ws.numel(), fflags, stream))
        if (flags & _lib.FLAG_SORT_RAYS) and not args.no_order_reuse:        # adjoint visits rays in the forward's bundle order
            lib.drrt_set_order_hint(lib.drrt_last_order(None), n)
        _lib.check(lib.drrt_backtrace_f32(p(rif), nvox, res, n, p(xt), p(vt), p(dx), p(dv), h, ds, p(grad),
                                          p(st_a), p(ws), ws.numel(), aflags, stream))
        if use_dist:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    _lib.check(lib.drrt_profile_begin(8 * args.steps + 8))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_collect()
    lib.drrt_profile_end()

    if args.debug_counters:
        off = (ws.numel() - 512) & ~7
        dbg = ws[off:off + 512].view(torch.int64).cpu().tolist()
        print(f"[debug] window flushes {dbg[0]}, ray-steps via LDS window {dbg[1]}, via global fallback {dbg[2]}",
              file=sys.stderr)
    fwd_steps = int(st_f[0].item()); adj_steps = int(st_a[0].item())
    n_failed = int(st_f[1].item())
    t = torch.tensor([elapsed, float(fwd_steps), float(adj_steps)], dtype=torch.float64, device=dev)
    if use_dist:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0]); fwd_total = float(tsum[1]); adj_total = float(tsum[2])
    else:
        fwd_total, adj_total = float(fwd_steps), float(adj_steps)

    if rank == 0:
        def avg(name):
            v = [ms for k, ms in prof if k == name]
            return (sum(v) / len(v)) if v else float("nan")
        ms_fwd, ms_adj, ms_sort, ms_zero = avg("trace"), avg("backtrace"), avg("sort"), avg("zero")
        ms_quad = avg("quad")
        ach_adj = adj_steps * B_ADJ / (ms_adj * 1e-3) / 1e9
        ach_fwd = fwd_steps * B_FWD / (ms_fwd * 1e-3) / 1e9
        default_cfg = (R == 256 and n == 1024 * 1024 and not args.no_sort and not args.direct_atomics
                       and not args.experiment and not args.quad and not args.lds_bricks)
        tr_adj = measured_traffic("drrt::k_backtrace_win") if default_cfg else None
        tr_fwd = measured_traffic("drrt::k_trace") if default_cfg else None
        out = {
            "metric": "ray-steps/sec (fwd+adjoint), 256^3 RIF grid, 1M rays x 512 steps",
            "value": fwd_total * args.steps / elapsed,
            "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Luneburg ball {R}^3 fp32 grid (replicated), {n} rays per GPU from a seeded jittered "
                                   f"plane source on the y=0 face, ds=h/2, fwd trace + adjoint backtrace (dx=dv=1)"
                                   + (", one RCCL all-reduce(sum) of the grid per step" if world > 1 else ""),
                       "grid": R, "rays_per_gpu": n, "fwd_ray_steps_per_gpu": fwd_steps,
                       "adj_ray_steps_per_gpu": adj_steps, "n_failed": n_failed,
                       "sort_rays": not args.no_sort, "quad_grid": bool(flags & _lib.FLAG_QUAD_GRID), "parallelism": f"ray-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": "adjoint march (k_backtrace)", "achieved": ach_adj,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_adj / HBM_PEAK_GBS, "traffic": tr_adj and tr_adj["bytes"],
                         "traffic_source": tr_adj and tr_adj["source"],
                         "algorithmic_bytes_per_ray_step": B_ADJ, "ray_steps_per_launch": adj_steps,
                         "avg_kernel_ms": ms_adj},
            "roofline_fwd": {"bound": "hbm", "kernel": "forward march (k_trace)", "achieved": ach_fwd,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_fwd / HBM_PEAK_GBS, "traffic": tr_fwd and tr_fwd["bytes"],
                             "traffic_source": tr_fwd and tr_fwd["source"],
                             "algorithmic_bytes_per_ray_step": B_FWD, "ray_steps_per_launch": fwd_steps,
                             "avg_kernel_ms": ms_fwd},
            "phase_ms": {"sort_avg": ms_sort, "zero_grid": ms_zero, "quad_copy": None if ms_quad != ms_quad else ms_quad,
                         "trace": ms_fwd, "backtrace": ms_adj},
            "fwd_only_ray_steps_per_s_per_gpu": fwd_steps / (ms_fwd * 1e-3),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(R, h, ds, rif.cpu().numpy(), pos.cpu().numpy(), vel.cpu().numpy(),
                                               target_seconds=args.cpu_seconds)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
