#!/usr/bin/env python3
"""Headline benchmark: ray-steps/sec (fwd + adjoint) on the synthetic 256^3 / 1M-ray / 512-step
workload of BASELINE.json.

One "step" = one forward march (drrt_trace_f32) + one adjoint march (drrt_backtrace_f32) over
one batch of rays, inputs resident in HBM, called through the C ABI (ctypes) on torch's current
stream.

Multi-GPU (one process per GPU, backend nccl = RCCL): the refractive-index grid is replicated, rays
are sharded, and the per-rank dL/dn grids are summed by ONE all-reduce per step -- the path's only
exchange.
  * `python bench.py --gpus N` with no launcher starts the N ranks ITSELF (child processes, before
    anything touches the GPU) and relays rank 0's JSON line; it exits non-zero when fewer than N
    GPUs are visible (nccl) or when any rank fails -- it never falls back to fewer ranks;
  * under `torch.distributed.run` (RANK / WORLD_SIZE in the environment) it runs as one rank and
    refuses a WORLD_SIZE that differs from --gpus.
  * `--scaling strong` (default): the metric's ONE 1M-ray set is split into contiguous shards
    (dist.shard_bounds); value = global forward ray-steps / max-over-ranks time incl. the
    all-reduce.  `--scaling weak`: every rank marches its own `--rays` rays.  `--scaling both`
    (default for N > 1): value/ms_per_step are the strong figures, `weak_scaling` carries the weak.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      -- the dominant kernel (adjoint march): SURVEY 8.6's algorithmic bytes / launch
                   duration (HIP events recorded around the kernel inside the library) against the
                   HBM peak, PLUS the bound that physically applies (`physical_bound`,
                   `physical_frac`), derived from the committed PMC summary named in `pmc_source`
  roofline_fwd  -- same for the forward march kernel (the 40 %-of-HBM target of north_star)
  cpu_baseline  -- the CPU oracle (plain-C port of the reference, 1 thread) timed on a bounded
                   sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.29 TB/s measured copy
B_FWD = 32.0                   # algorithmic bytes per forward ray-step: 8 fp32 taps
B_ADJ = 64.0                   # adjoint: 8 taps read + 8 fp32 atomic adds
# Committed rocprofv3 PMC summary (tools/profile_bench.sh + tools/condense_profile.py on THIS command) that
# `roofline.traffic` and the physical-bound figures are read from.  It is NOT measured in the run: the
# bench line says so in `traffic_source` / `pmc_source`.
PMC_PROFILE = os.path.join("profiles", "r4_pmc.json")
PMC_FALLBACK = os.path.join("profiles", "r3_pmc.json")
N_SIMD = 256 * 4               # MI355X: 256 CUs x 4 SIMDs
CLK_HZ = 2.4e9                 # nominal shader clock: only used when the PMC summary has no measured effective clock
VALU_CYCLES = 4.0              # a wave64 VALU instruction occupies its SIMD for 4 cycles (SQ_ACTIVE_INST_VALU counts
                               # exactly one quad-cycle per instruction on these kernels; tools/valu_bench.hip: 3.9)
TA_CYCLES_PER_GATHER = 30.0    # a divergent 64-lane gather instruction occupies its CU's texture addresser for ~30-37 cycles,
                               # with ONE active lane as with 64 (tools/chain_bench.hip: 15 ns per instruction per CU at 8
                               # waves per SIMD); the low end is used -- an estimate, the instruction COUNT is from the PMC
N_CU = 256


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--rays", type=int, default=1024 * 1024,
                    help="rays of the GLOBAL set (strong scaling) / per GPU (weak scaling); a perfect square")
    ap.add_argument("--scaling", choices=("strong", "weak", "both"), default=None,
                    help="default: strong for --gpus 1, both for --gpus > 1")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--direct-atomics", action="store_true")
    ap.add_argument("--no-step-hint", action="store_true", help="adjoint rays start at their own exit sample (A-B)")
    ap.add_argument("--no-order-reuse", action="store_true",
                    help="adjoint computes its own visit order instead of reusing the forward's")
    ap.add_argument("--pair", choices=["auto", "on", "off"], default="auto",
                    help="pair copy of the grid (DRRT_FLAG_PAIR_GRID: two 16-byte gathers per cell; the forward builds "
                         "it -- inside the timed step -- and the adjoint reuses it).  auto = the rule of "
                         "adjointnonlinearraytracing_amd.drrt (enough rays to fill the GPU, enough ray-steps per voxel)")
    ap.add_argument("--quad", action="store_true", help="same as --pair on (round-1 name)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-counters", action="store_true", help="print LDS-window counters (stderr)")
    ap.add_argument("--experiment", type=int, default=0, help="development ablation id (0 = product)")
    ap.add_argument("--fwd-flags", type=lambda v: int(v, 0), default=0, help="extra DRRT_FLAG_* bits for the forward call")
    ap.add_argument("--adj-flags", type=lambda v: int(v, 0), default=0, help="extra DRRT_FLAG_* bits for the adjoint call")
    ap.add_argument("--shard-of", type=int, default=0, metavar="G",
                    help="single process: march only shard --shard-index of G of the strong-scaling ray set (what ONE rank of "
                         "a G-GPU run does, without the all-reduce) -- per-shard timings for the scaling projection in "
                         "DESIGN.md (tools/run_shards.sh times every index and takes the max)")
    ap.add_argument("--shard-index", type=int, default=0, metavar="K", help="which shard of --shard-of (default 0)")
    ap.add_argument("--adjoint-chunks", type=int, default=0, metavar="K",
                    help="run the adjoint as K depth chunks (drrt_backtrace_chunk_f32: K launches of max_steps / K iterations, ray "
                         "state carried through HBM, windows flushed at chunk ends) -- what the slab-wise all-reduce of "
                         "dist.py needs; the line then carries `chunking` (the share of the grid that is final after each chunk)")
    ap.add_argument("--overlap-reduce", action="store_true",
                    help="N > 1 with --adjoint-chunks K: reduce the planes of dL/dn that are final after each chunk on a side "
                         "stream while the next chunk marches (dist.SlabReducer) instead of one whole-grid all-reduce at the end; "
                         "phase_ms.allreduce_exposed is then what is left after the last chunk")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the `variants` leg (the reference's six rotated views and the shifted plane source)")
    ap.add_argument("--variant-steps", type=int, default=5)
    ap.add_argument("--variants", default="cube6_rotated,plane_shifted,tomo_weak", help="comma list of the variants to run")
    ap.add_argument("--workload", choices=("metric", "cube6_rotated", "plane_shifted", "tomo_weak"), default="metric",
                    help="ray set (and medium) of the MAIN run (default: the metric's plane source through the Luneburg ball).  "
                         "The others are the `variants` run as the main workload -- for profiling them on their own and for "
                         "their shard timings / multi-GPU runs (multi-view sets are sharded view by view, dist.shard_views)")
    ap.add_argument("--source-axis", choices=("x", "y", "z"), default="y",
                    help="A-B: the metric's plane source on the x=0 / z=0 face instead of y=0 (rays along that axis; the ball is "
                         "symmetric, so the work is the same and only the memory order of the cells along the rays differs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only "
                                                        "to rehearse the multi-rank flow on a 1-GPU box)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.shard_of and not (0 <= args.shard_index < args.shard_of):
        ap.error("--shard-index must lie in [0, --shard-of)")
    if args.scaling is None:
        args.scaling = "strong" if args.gpus == 1 else "both"
    return args


# ------------------------------------------------------------------------------------------------
# self-launch: N child ranks, one GPU each, started BEFORE this process touches the GPU
# ------------------------------------------------------------------------------------------------
def launch_ranks(args) -> int:
    import torch          # device_count() does not initialise the GPU on this image
    ndev = torch.cuda.device_count()
    if ndev == 0:
        print("bench.py: no GPU visible (there is no CPU path)", file=sys.stderr)
        return 2
    if args.backend == "nccl" and ndev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible; refusing to run fewer ranks "
              f"(use --backend gloo to rehearse the multi-rank flow on shared GPUs)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):      # one rank failed: do not leave the others waiting
                break
            time.sleep(0.1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    out0 = procs[0].stdout.read()      # one JSON line: far below the pipe buffer, safe to read after exit
    lines = [l for l in out0.decode(errors="replace").splitlines() if l.startswith("{")]
    if rc != 0 or not lines:
        print(f"bench.py: a rank failed (exit codes {[p.returncode for p in procs]})", file=sys.stderr)
        return rc or 1
    d = json.loads(lines[-1])
    if d.get("n_gpus") != args.gpus:
        print(f"bench.py: ranks reported n_gpus={d.get('n_gpus')} != --gpus {args.gpus}", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


# ------------------------------------------------------------------------------------------------
def make_grid(R: int, device):
    import torch
    span = 1.0
    g = torch.linspace(0.0, span, R, device=device)
    Z, Y, X = torch.meshgrid(g, g, g, indexing="ij")
    r = torch.sqrt((X - span / 2) ** 2 + (Y - span / 2) ** 2 + (Z - span / 2) ** 2) / (span / 2)
    return torch.sqrt(2.0 - torch.clamp(r, max=1.0) ** 2).to(torch.float32).contiguous()


def make_grid_tomo(R: int, device, seed: int = 0):
    """SURVEY 8.6's second medium, the weak-deflection regime of the tomography experiment (value range of
    data/fuel_injection_64.npy, core/fuel_injection_opt.py:41-43): n = 1 + 3e-4 U, U = torch.rand (seed 0) low-pass
    filtered (three passes of a 9^3 box filter = a smooth kernel ~16 voxels wide) and rescaled to [0, 1]."""
    import torch
    gen = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand(R, R, R, generator=gen).to(device)[None, None]
    for _ in range(3):
        u = torch.nn.functional.avg_pool3d(torch.nn.functional.pad(u, (4,) * 6, mode="replicate"), 9, stride=1)
    u = u[0, 0]
    u = (u - u.min()) / (u.max() - u.min())
    return (1.0 + 3e-4 * u).to(torch.float32).contiguous()


def truncated_views(rays_per_view, n):
    """Per-view ray counts of the first n rays of a multi-view set (the last views lose what the truncation removes)."""
    out, left = [], n
    for c in rays_per_view:
        k = min(int(c), left)
        out.append(k)
        left -= k
    return out


def make_rays(n_rays: int, seed: int):
    """Jittered plane source on the y=0 face, v=(0,1,0) (plane_source3_rand-equivalent,
    /root/reference/core/source.py:54-69: pixel (i,j) -> position (x,0,z) with j (z) the fast index).
    Host tensors; rays are NOT pre-sorted."""
    import torch
    span = 1.0
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
    return pos, vel


def make_rays_shifted(n_rays: int, seed: int):
    """The headline source moved by a third of a pixel in x and z: the same rays, no longer aligned with the
    voxel columns (1024 pixels over 255 cells: the headline's 8 x 8-pixel sort tiles sit on 2 x 2 cells)."""
    pos, vel = make_rays(n_rays, seed)
    side = int(round(n_rays ** 0.5))
    pos[:, 0] += 1.0 / side / 3.0
    pos[:, 2] += 1.0 / side / 3.0
    pos.clamp_(0.0, 1.0 - 1e-6)
    pos[:, 1] = 0.0
    return pos, vel


def make_rays_cube6(n_rays: int, seed: int, device):
    """What the reference's Luneburg experiment feeds the path every iteration (core/luneburg_opt.py:53-57,94):
    `source.rand_rays_cube((P, P), spp, span, circle=True)` -- six plane views, four about z and two about x
    (core/source.py:398-412) -- turned by ONE random rotation (`random_rotate_ic`, core/source.py:555-563;
    scipy Rotation.random with random_state = seed), generated by the package's device generators with the
    rotation fused.  P is chosen so that the disc-masked views hold a little more than n_rays rays; the first
    n_rays are kept (the last view loses < 1 %).  Rays start on planes tangent to the inscribed sphere, i.e.
    partly outside the box, exactly as in the reference."""
    import numpy as np
    import torch
    from scipy.spatial.transform import Rotation
    from adjointnonlinearraytracing_amd import source
    rot = torch.from_numpy(Rotation.random(random_state=seed).as_matrix())
    P = int(np.ceil(np.sqrt(n_rays * 1.004 / (6.0 * np.pi / 4.0))))
    gen = torch.Generator(device="cpu").manual_seed(seed)
    while True:
        off = torch.rand(6, 2, P, P, generator=gen)
        (x, v, planes), nrays = source.rand_rays_cube((P, P), 1, 1.0, circle=True, offset=off, device=device,
                                                      rotmat=rot, span=1.0)
        if x.shape[0] >= n_rays:
            break
        P += 2
    return x[:n_rays].contiguous(), v[:n_rays].contiguous(), {"pixels_per_view": P, "rays_per_view": nrays,
                                                              "rotation_seed": seed}


def make_workload(R: int, n_rays: int, device, seed: int):
    """Luneburg ball on an R^3 grid + the plane source above (kept for tools/ and tests)."""
    span = 1.0
    h = span / (R - 1)
    ds = h / 2                                  # step_res = 2 (core/luneburg_opt.py:38,48-49)
    pos, vel = make_rays(n_rays, seed)
    return make_grid(R, device), pos.to(device), vel.to(device), h, ds


def load_pmc():
    """The committed PMC summary (see PMC_PROFILE).  -> (dict, basename) or (None, None)."""
    for rel in (PMC_PROFILE, PMC_FALLBACK):
        f = os.path.join(ROOT, rel)
        if os.path.exists(f):
            try:
                return json.load(open(f)), rel
            except Exception:
                continue
    return None, None


def pmc_kernel(pmc, *prefixes):
    """The kernel of the PMC summary whose name starts with one of `prefixes` (in order of preference)."""
    if not pmc:
        return None
    for prefix in prefixes:
        hits = [v for k, v in pmc.items() if k.startswith(prefix) and isinstance(v, dict)]
        if hits:      # several instantiations may be launched (one returns at once): the one that did the work
            return max(hits, key=lambda v: v.get("SQ_INSTS_VALU", 0.0))
    return None


def physical_bound(pk, ms):
    """What physically limits a march kernel, from the PMC summary `pk` of the same command and the live kernel time `ms`.
    The clock is the MEASURED effective clock of the profiled dispatch (GRBM_GUI_ACTIVE / 8 / duration, tools/
    condense_profile.py) when the summary has it; the nominal 2.4 GHz otherwise, and the line says which.
      valu_issue_frac   VALU instructions x 4 cycles / (SIMDs x kernel cycles): the share of the kernel the vector pipes need
                        if nothing else stood in the way (an upper bound on how much faster FEWER instructions could make it)
      valu_busy_frac    the same from SQ_ACTIVE_INST_VALU (quad-cycles the pipes were actually busy)
      wait_frac         SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of resident wave-cycles spent waiting (dependent chain:
                        gather and LDS round trips, branches)
      ta_gather_frac_est  gather INSTRUCTIONS x ~30 cycles per CU (the texture addresser's cost does not depend on how
                        many lanes are active)
    DESIGN.md 5: the adjoint responds to added VALU work at about a quarter of its issue time, i.e. it is bound by the
    wave's dependent chain per step WITH the vector pipes ~80 % busy -- hence the name of the bound."""
    out = {}
    if ms != ms or ms <= 0 or not pk:
        return out
    clk_ghz = pk.get("effective_clock_ghz")
    out["clock_ghz"] = clk_ghz if clk_ghz else CLK_HZ / 1e9
    out["clock_source"] = ("measured: GRBM_GUI_ACTIVE / 8 / dispatch duration of the profiled run (committed PMC summary)"
                           if clk_ghz else "ASSUMED nominal clock (the PMC summary carries no GRBM_GUI_ACTIVE pass)")
    cyc = ms * 1e-3 * out["clock_ghz"] * 1e9
    valu = ta = None
    if "SQ_INSTS_VALU" in pk:
        valu = pk["SQ_INSTS_VALU"] * VALU_CYCLES / (N_SIMD * cyc)
        out["valu_issue_frac"] = valu
        out["valu_insts_per_launch_pmc"] = pk["SQ_INSTS_VALU"]
    if "SQ_ACTIVE_INST_VALU" in pk:
        out["valu_busy_frac"] = pk["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * cyc)
    if "wait_frac" in pk:
        out["wait_frac"] = pk["wait_frac"]
    elif "SQ_WAIT_INST_ANY" in pk and pk.get("SQ_WAVE_CYCLES"):
        out["wait_frac"] = pk["SQ_WAIT_INST_ANY"] / pk["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VMEM_RD" in pk:
        ta = pk["SQ_INSTS_VMEM_RD"] * TA_CYCLES_PER_GATHER / (N_CU * cyc)
        out["ta_gather_frac_est"] = ta
        out["gather_insts_per_launch_pmc"] = pk["SQ_INSTS_VMEM_RD"]
    if valu is None and ta is None:
        return out
    if ta is None or (valu is not None and valu >= ta):
        out["physical_bound"], out["physical_frac"] = "dependent_chain+valu_issue", valu
    else:
        out["physical_bound"], out["physical_frac"] = "texture_addresser_gather_instructions", ta
    out["physical_note"] = (f"taps are served by L1/L2/Infinity Cache (the 64 MiB grid is cache-resident), so HBM is not the "
                            f"physical limiter; VALU = {VALU_CYCLES:.0f} cycles per wave64 instruction per SIMD "
                            f"(tools/valu_bench.hip: 3.9-4.8), a gather instruction = {TA_CYCLES_PER_GATHER:.0f} cycles of its "
                            f"CU's texture addresser (tools/chain_bench.hip: 30-37); instruction counts and clock from the "
                            f"committed PMC summary; physical_frac = the VALU issue share, the rest of the kernel time is the "
                            f"wave's dependent chain (wait_frac) that the resident waves do not hide")
    return out


def cpu_baseline(R, h, ds, rif_np, pos_np, vel_np, target_seconds=15.0, gpu=None):
    """Time the CPU oracle (kind 'port': plain-C restatement of the reference, single thread --
    the reference's CPU path is single-threaded) on a ray sub-sample.

    gpu = (callable rays -> dict(xt, vt, grad, fwd_steps, adj_steps)) runs the benchmark's OWN kernel configuration on
    a given ray sub-set; when given, the all-cores run of the oracle (factored arithmetic = the op sequence of the
    kernels) doubles as the checker: -> (cpu_baseline, parity_check)."""
    import numpy as np
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
    if n_all * 10 >= len(pos_np) * 9:
        n_all = len(pos_np)                     # close to the whole workload: take all of it (parity at full size)
    sel_all = np.arange(len(pos_np)) if n_all == len(pos_np) else np.linspace(0, len(pos_np) - 1, n_all).astype(np.int64)
    with O.arith("factored"):
        a = O.bench_allcores(rif_np, (R, R, R), pos_np[sel_all], vel_np[sel_all], h, ds, threads, want_rays=gpu is not None)
    allcores = {"value": a["fwd_steps"] / (a["t_fwd"] + a["t_adj"]), "unit": "ray-steps/s", "cores": a["threads"],
                "sample": f"{n_all} rays, fwd {a['t_fwd']:.2f}s + adjoint {a['t_adj']:.2f}s (incl. summing "
                          f"{a['threads']} private grids), factored fp32 arithmetic"}
    base = {
        "allcores": allcores,
        "value": steps / (t2 - t0), "unit": "ray-steps/s", "cores": 1, "kind": "port",
        "sample": f"{n} of the workload's rays (evenly strided), {R}^3 grid, fwd {t1 - t0:.2f}s + adjoint {t2 - t1:.2f}s, "
                  f"{steps} fwd ray-steps",
        "fwd_ray_steps_per_s": steps / (t1 - t0), "adj_ray_steps_per_s": b["steps_total"] / (t2 - t1),
    }
    if gpu is None:
        return base, None
    g = gpu(sel_all)
    gn = float(np.linalg.norm(a["grad"].astype(np.float64)))
    parity = {
        "checker": "oracle/ (plain-C restatement of src/tracer.cpp:35-100,384-440), factored fp32 arithmetic, all host "
                   "cores over contiguous ray chunks, private grids summed",
        "rays": int(n_all), "full_workload": bool(n_all == len(pos_np)),
        "kernel_flags": g["flags_note"],
        "xt_bit_exact": bool(np.array_equal(g["xt"], a["xt"])), "vt_bit_exact": bool(np.array_equal(g["vt"], a["vt"])),
        "rays_differing": int(np.count_nonzero(np.any(g["xt"] != a["xt"], axis=1) | np.any(g["vt"] != a["vt"], axis=1))),
        "fwd_ray_steps_gpu": int(g["fwd_steps"]), "fwd_ray_steps_oracle": int(a["fwd_steps"]),
        "adj_ray_steps_gpu": int(g["adj_steps"]), "adj_ray_steps_oracle": int(a["adj_steps"]),
        "rel_l2": float(np.linalg.norm(g["grad"].astype(np.float64) - a["grad"].astype(np.float64)) / max(gn, 1e-300)),
        "rel_l2_bound": 2e-5,
    }
    parity["ok"] = bool(parity["xt_bit_exact"] and parity["vt_bit_exact"]
                        and parity["fwd_ray_steps_gpu"] == parity["fwd_ray_steps_oracle"]
                        and parity["adj_ray_steps_gpu"] == parity["adj_ray_steps_oracle"]
                        and parity["rel_l2"] <= parity["rel_l2_bound"])
    return base, parity


def run_rank(args) -> int:
    import torch
    import torch.distributed as dist

    # The contract is ONE JSON line on stdout.  RCCL / gloo print banners to the process's stdout from
    # native code (e.g. "RCCL version : ..."), so file descriptor 1 is pointed at stderr until the final
    # print and restored just for it.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to benchmark a different rank count",
              file=sys.stderr)
        return 2
    ndev = torch.cuda.device_count()
    if ndev == 0:
        print("bench.py needs a GPU (there is no CPU path)", file=sys.stderr)
        return 2
    if args.backend == "nccl" and world > ndev:
        print(f"bench.py: {world} ranks but only {ndev} GPUs visible", file=sys.stderr)
        return 2
    local = local % ndev                         # gloo rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ          # under a launcher: also with a single rank (rehearses RCCL)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            return 2

    from adjointnonlinearraytracing_amd import _lib
    from adjointnonlinearraytracing_amd import dist as drrt_dist
    lib = _lib.load()            # loud failure if the HIP library is missing

    R = args.grid
    span = 1.0
    h = span / (R - 1)
    ds = h / 2                                   # step_res = 2 (core/luneburg_opt.py:38,48-49)
    rif = make_grid(R, dev)
    nvox = rif.numel()
    res = (C.c_int * 3)(R, R, R)
    flags = 0 if args.no_sort else _lib.FLAG_SORT_RAYS
    if args.quad:
        args.pair = "on"
    fflags0 = flags | args.fwd_flags
    aflags0 = flags | (_lib.FLAG_DIRECT_ATOMICS if args.direct_atomics else 0) | args.adj_flags
    aflags0 |= (_lib.FLAG_DEBUG_COUNTERS if args.debug_counters else 0) | ((args.experiment & 0xff) << 8)

    def pair_flags(n):
        """(forward flags, adjoint flags, pair copy in use) for a rank marching n rays: the forward builds the pair copy,
        the paired adjoint reuses it; `auto` applies the product's own rule (drrt._march_workspace)."""
        from adjointnonlinearraytracing_amd import drrt as _drrt
        use = args.pair == "on" or (args.pair == "auto" and n >= _drrt._PAIR_AUTO_MIN_RAYS
                                    and n * R * (h / ds) >= 8.0 * nvox)
        if not use or n == 0:
            return fflags0, aflags0, False
        return fflags0 | _lib.FLAG_PAIR_GRID, aflags0 | _lib.FLAG_PAIR_GRID | _lib.FLAG_PAIR_REUSE, True
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    grad = torch.empty(nvox, dtype=torch.float32, device=dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ring_pct = int(lib.drrt_ring_threshold_pct())

    def bench_rays(pos, vel, steps, warmup, force_flags=None, keep=False, rif=rif, robust=False):
        """Time `steps` fwd + adjoint passes over the rays (pos, vel) resident on the device; -> this rank's measurements.
        force_flags = (fflags, aflags, pair) overrides the pair-copy rule (parity runs on a sub-sample keep the flags)."""
        n = pos.shape[0]
        fflags, aflags, pair = force_flags if force_flags is not None else pair_flags(n)
        ws = torch.empty(int(lib.drrt_workspace_bytes_grid(n, nvox, fflags | aflags)) + 1024, dtype=torch.uint8, device=dev)
        xt, vt = torch.empty_like(pos), torch.empty_like(vel)
        dx, dv = torch.ones_like(pos), torch.ones_like(vel)          # adjoint seed dx=dv=1 (src/test.cpp:142-144)
        st_f = torch.zeros(3, dtype=torch.int64, device=dev)
        st_a = torch.zeros(3, dtype=torch.int64, device=dev)
        K = max(0, args.adjoint_chunks)
        if K > 1:
            total_it = int(lib.drrt_backtrace_max_steps(res, h, ds))
            cb = [total_it * k // K for k in range(K + 1)]
            cstate = torch.empty(int(lib.drrt_backtrace_chunk_state_bytes(n)), dtype=torch.uint8, device=dev)
            cprog = torch.zeros(K, 20, dtype=torch.int32, device=dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        overlap_info = {}

        def step(k=None):
            _lib.check(lib.drrt_trace_f32(p(rif), nvox, res, n, p(pos), p(vel), h, ds, p(xt), p(vt),
                                          p(st_f), p(ws), ws.numel(), fflags, stream))
            if (flags & _lib.FLAG_SORT_RAYS) and not args.no_order_reuse:    # adjoint visits rays in the forward's bundle order
                lib.drrt_set_order_hint(lib.drrt_last_order(None), n)
                if not args.no_step_hint:                                    # ... on the forward march's clock
                    lib.drrt_set_step_hint(lib.drrt_last_steps(None), n)
            overlap = K > 1 and use_dist and args.overlap_reduce
            if K > 1:                                                        # the adjoint in K depth chunks
                cev = []
                for c in range(K):
                    if c > 0 and (flags & _lib.FLAG_SORT_RAYS):              # every chunk visits the rays in the same order
                        lib.drrt_set_order_hint(lib.drrt_last_order(None), n)
                    _lib.check(lib.drrt_backtrace_chunk_f32(p(rif), nvox, res, n, p(xt), p(vt), p(dx), p(dv), h, ds, p(grad),
                                                            p(st_a), p(ws), ws.numel(), aflags, stream, p(cstate),
                                                            cstate.numel(), cb[c], cb[c + 1] - cb[c], p(cprog[c])))
                    if overlap:
                        e_ = torch.cuda.Event(); e_.record(); cev.append(e_)
                if overlap:                                                  # all chunks are queued: follow them as they finish
                    from adjointnonlinearraytracing_amd import drrt as _drrt
                    red = drrt_dist.SlabReducer(grad, (R, R, R), h)
                    for c in range(K):
                        cev[c].synchronize()
                        red.after_chunk(_drrt.decode_chunk_progress(cprog[c]), cev[c])
                    early = int(sum(int(b_.numel()) for _, b_, _ in red.parts))   # voxels whose reduce is already under way
                    if k is not None:
                        ev[k][0].record()                                    # the last chunk is done: from here on the reduce is exposed
                    red.finish(cev[-1])
                    if k is not None:
                        ev[k][1].record()
                        overlap_info.update(share_of_grid_reduced_under_the_march=early / float(nvox),
                                            violated=bool(red.violated), axis_and_direction=red.choice)
                    return
            else:
                _lib.check(lib.drrt_backtrace_f32(p(rif), nvox, res, n, p(xt), p(vt), p(dx), p(dv), h, ds, p(grad),
                                                  p(st_a), p(ws), ws.numel(), aflags, stream))
            if use_dist:
                if k is not None:
                    ev[k][0].record()
                dist.all_reduce(grad, op=dist.ReduceOp.SUM)
                if k is not None:
                    ev[k][1].record()

        for _ in range(warmup):
            step()
        barrier()
        ms_ar_plain = None
        if K > 1 and use_dist and args.overlap_reduce:
            # what ONE whole-grid all-reduce costs in this run (the figure the slab-wise reduce is compared with): timed once,
            # outside the timed steps, on a scratch copy of the grid
            scratch = grad.clone()
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier(); e0_.record(); dist.all_reduce(scratch, op=dist.ReduceOp.SUM); e1_.record(); barrier()
            ms_ar_plain = e0_.elapsed_time(e1_)
            del scratch
        _lib.check(lib.drrt_profile_begin(8 * steps + 8))
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        barrier()
        elapsed = time.perf_counter() - t0
        prof = _lib.profile_collect()
        lib.drrt_profile_end()
        ms_ar = (sum(a.elapsed_time(b) for a, b in ev) / len(ev)) if use_dist else 0.0

        dbg = None
        if args.debug_counters:
            off = (ws.numel() - 512) & ~7
            dbg = ws[off:off + 512].view(torch.int64).cpu().tolist()
            print(f"[debug] window flushes {dbg[0]}, ray-steps via LDS window {dbg[1]}, via global fallback {dbg[2]}, "
                  f"fitted waves {dbg[3]}; leaves: one face -> window {dbg[4]} (lanes adding after pre-reduction {dbg[5]}), "
                  f"one face -> global {dbg[6]}, all eight {dbg[7]}; wave-steps {dbg[8]}, with >= 2 axes leaving {dbg[9]}; "
                  f"ring kernel: [0..2] = full flushes / slides / fits; no window for the box {dbg[10]}, services {dbg[11]}, "
                  f"lanes left outside {dbg[12]}, fitted volume sum {dbg[13]}, all eight -> global {dbg[14]}, "
                  f"ray-steps in clamped cells {dbg[15]}; sample of boxes without a window (step, ex, ey, ez, lanes, block): "
                  + str([(dbg[17 + 4 * k], dbg[18 + 4 * k] & 0xffff, (dbg[18 + 4 * k] >> 16) & 0xffff, dbg[18 + 4 * k] >> 32,
                          dbg[19 + 4 * k], dbg[20 + 4 * k]) for k in range(11)]), file=sys.stderr)
        fwd_steps = int(st_f[0].item()); adj_steps = int(st_a[0].item())
        n_failed = int(st_f[1].item())
        # which adjoint kernel the device-side bundle classification of the LAST step chose (drrt_last_bundle_counters)
        choice = None
        cptr = lib.drrt_last_bundle_counters()
        if cptr:
            off = int(cptr) - ws.data_ptr()
            if 0 <= off and off + 32 <= ws.numel():
                c = ws[off:off + 32].view(torch.int32).cpu().tolist()
                ext_min = int(lib.drrt_ring_long_threshold_permille())
                long_ = bool(c[6] and c[6] * 1000 >= c[1] * ext_min)
                ring = bool(c[0] and c[0] * 100 >= c[1] * ring_pct) or long_
                sparse = ring and c[5] == 0
                direct_pct = int(lib.drrt_ring_direct_threshold_pct())
                direct = sparse and c[3] != 0 and c[4] * 100 < c[3] * direct_pct
                choice = {"kernel": ("ring_direct" if direct else "ring_sparse" if sparse else "ring") if ring else "box",
                          "start_pair_share": round(c[4] / c[3], 4) if c[3] else None, "direct_max_pair_pct": direct_pct,
                          "bundles_not_fitting": c[0], "bundles_sampled": c[1], "ring_threshold_pct": ring_pct,
                          "lanes_far_from_bundle": c[2], "lanes_sampled": c[3],
                          "long_bundle_share": round(c[6] / c[1], 4) if c[1] else None, "ring_min_long_permille": ext_min}
        t = torch.tensor([elapsed, float(fwd_steps), float(adj_steps)], dtype=torch.float64, device=dev)
        if use_dist:
            tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            elapsed = float(tmax[0]); fwd_total = float(tsum[1]); adj_total = float(tsum[2])
        else:
            fwd_total, adj_total = float(fwd_steps), float(adj_steps)

        def avg(name):
            v = [ms for k, ms in prof if k == name]
            if name == "backtrace" and K > 1:            # K launches per step: their sum
                return (sum(v) / len(v) * K) if v else float("nan")
            if robust and v:                             # `variants`: the median (twice in round 4 one step's sort of the first
                v = sorted(v)                            # variant showed 70 ms between its events -- a host-side stall while the
                return v[len(v) // 2]                    # 22 sort kernels were being enqueued; the headline keeps the mean)
            return (sum(v) / len(v)) if v else float("nan")
        chunking = None
        if K > 1:
            # Which part of the grid is final after each chunk?  The adjoint moves a ray by -ds * v per iteration: along the
            # source axis every plane beyond the deepest still-marching ray (+ the upper tap) is final while no ray heads back.
            from adjointnonlinearraytracing_amd import drrt as _drrt
            ax = {"x": 0, "y": 1, "z": 2}[args.source_axis]
            fin = []
            for c in range(K):
                pr = _drrt.decode_chunk_progress(cprog[c])
                if pr["active"] == 0:
                    fin.append(1.0)
                elif pr["vel_min"][ax] > 0.0:
                    top = int(pr["pos_max"][ax] / h) + 2                   # first final plane
                    fin.append(max(0.0, min(1.0, (R - top) / R)))
                else:
                    fin.append(0.0)
            chunking = {"chunks": K, "iteration_bounds": cb, "grid_final_after_chunk": fin,
                        "note": "share of the grid planes along the source axis that no still-marching ray can reach after "
                                "chunk k (drrt_backtrace_chunk_f32 progress block); the slab-wise all-reduce of dist.py may "
                                "start on that share while chunk k + 1 marches"}
        out = dict(n=n, steps=steps, elapsed=elapsed, fwd_total=fwd_total, adj_total=adj_total, fwd_steps=fwd_steps,
                   adj_steps=adj_steps, n_failed=n_failed, ms_fwd=avg("trace"), ms_adj=avg("backtrace"),
                   ms_sort=avg("sort"), ms_zero=avg("zero"), ms_quad=avg("quad"), ms_allreduce=ms_ar, pair=pair,
                   pos=pos, vel=vel, flags=(fflags, aflags, pair), dbg=dbg, adjoint_kernel=choice, chunking=chunking,
                   overlap=(overlap_info if (K > 1 and use_dist and args.overlap_reduce) else None), ms_allreduce_plain=ms_ar_plain)
        if keep:                                   # results of the LAST step (the adjoint's grid holds this rank's gradient
            out.update(xt=xt, vt=vt,               # only when there is no all-reduce: parity_check runs at world == 1)
                       grad=grad.clone())
        return out

    def direct_atomics_grad(m, rif=rif):
        """dL/dn of the rays of measurement `m` (its exit rays `xt`, `vt`, dx = dv = 1) by the one-atomic-per-tap kernel:
        no windows, no register accumulators, no visit order -- the in-library cross-check of the windowed adjoint."""
        n = m["pos"].shape[0]
        g = torch.empty(nvox, dtype=torch.float32, device=dev)
        ones = torch.ones_like(m["xt"])
        _lib.check(lib.drrt_backtrace_f32(p(rif), nvox, res, n, p(m["xt"]), p(m["vt"]), p(ones), p(ones), h, ds, p(g),
                                          None, None, 0, _lib.FLAG_DIRECT_ATOMICS, stream))
        torch.cuda.synchronize(dev)
        return g

    def rel_l2(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))

    grids = {}

    def workload_set(name, n_rays, seed):
        """-> (grid, pos, vel, per-view ray counts or None, description) of a named workload, rays resident on the device."""
        if name == "metric":
            gpos, gvel = make_rays(n_rays, seed=seed)
            if args.source_axis != "y":                            # A-B: the same source on another face (the ball is symmetric)
                perm = {"x": [1, 0, 2], "z": [0, 2, 1]}[args.source_axis]
                gpos, gvel = gpos[:, perm].contiguous(), gvel[:, perm].contiguous()
            return rif, gpos.to(dev), gvel.to(dev), None, {}
        if name == "plane_shifted":
            gpos, gvel = make_rays_shifted(n_rays, seed=seed)
            return rif, gpos.to(dev), gvel.to(dev), None, {"rays": "the headline's plane source moved by 1/3 pixel in x and z"}
        gpos, gvel, info = make_rays_cube6(n_rays, seed, dev)
        views = truncated_views(info["rays_per_view"], n_rays)
        meta = {"rays": "source.rand_rays_cube((P, P), 1, span, circle=True) + random_rotate_ic "
                        "(core/source.py:398-412,555-563; core/luneburg_opt.py:53-57), device generators, first "
                        f"{n_rays} rays", **info}
        if name == "tomo_weak":
            if "tomo" not in grids:
                grids["tomo"] = make_grid_tomo(R, dev)
            meta["medium"] = ("n = 1 + 3e-4 U, U = low-pass filtered torch.rand (seed 0) rescaled to [0, 1]: the weak-deflection "
                              "regime of core/fuel_injection_opt.py:41-43 (SURVEY 8.6, second medium)")
            return grids["tomo"], gpos, gvel, views, meta
        return rif, gpos, gvel, views, meta

    def shard_of_set(pos, vel, views, k, G):
        """Shard k of G: one contiguous piece of a single-view set, a strip of EVERY view of a multi-view set."""
        if G <= 1:
            return pos, vel
        if views is None:
            lo, hi = drrt_dist.shard_bounds(pos.shape[0], k, G)
            return pos[lo:hi].contiguous(), vel[lo:hi].contiguous()
        spans = drrt_dist.shard_views(views, k, G)
        return (torch.cat([pos[lo:hi] for lo, hi in spans]).contiguous(),
                torch.cat([vel[lo:hi] for lo, hi in spans]).contiguous())

    def run_mode(mode):
        """-> dict of this rank's measurements for one scaling mode."""
        if mode == "strong":
            g, gpos, gvel, views, _ = workload_set(args.workload, args.rays, 0)     # ONE global ray set, same on every rank
            if args.shard_of > 1 and world == 1:
                pos, vel = shard_of_set(gpos, gvel, views, args.shard_index, args.shard_of)
            else:
                pos, vel = shard_of_set(gpos, gvel, views, rank, world)
        else:
            g, pos, vel, views, _ = workload_set(args.workload, args.rays, rank)
        out = bench_rays(pos, vel, args.steps, args.warmup,
                         keep=(world == 1 and mode == "strong" and args.workload == "metric"), rif=g)
        out["mode"] = mode
        out["multi_view"] = views is not None
        return out

    def run_variant(name):
        """The other ray distributions / media of SURVEY 8.6 on the same grid size (1 GPU; after the headline run, not part
        of `value`)."""
        g, pos, vel, views, meta = workload_set(name, args.rays, 0)
        v = bench_rays(pos, vel, args.variant_steps, 3, keep=True, rif=g, robust=True)
        g_ref = direct_atomics_grad(v, rif=g)
        fl = v["flags"]
        res_ = {"n_rays": v["n"], "steps": v["steps"], "ms_per_step": v["elapsed"] / v["steps"] * 1e3,
                "trace": v["ms_fwd"], "backtrace": v["ms_adj"], "sort": v["ms_sort"],
                "fwd_ray_steps": v["fwd_steps"], "adj_ray_steps": v["adj_steps"], "n_failed": v["n_failed"],
                "value": v["fwd_steps"] * v["steps"] / v["elapsed"],
                "fwd_ns_per_ray_step": v["ms_fwd"] * 1e6 / max(v["fwd_steps"], 1),
                "adj_ns_per_ray_step": v["ms_adj"] * 1e6 / max(v["adj_steps"], 1),
                "pair_grid": bool(fl[2]), "adjoint_kernel": v["adjoint_kernel"],
                "grad_rel_l2_vs_direct_atomics": rel_l2(v["grad"], g_ref)}
        res_.update(meta)
        return res_

    rc = 0
    modes = ["strong", "weak"] if args.scaling == "both" else [args.scaling]
    results = {m: run_mode(m) for m in modes}
    main_mode = modes[0]
    m = results[main_mode]

    if rank == 0:
        n, fwd_steps, adj_steps = m["n"], m["fwd_steps"], m["adj_steps"]
        ms_fwd, ms_adj = m["ms_fwd"], m["ms_adj"]
        ach_adj = adj_steps * B_ADJ / (ms_adj * 1e-3) / 1e9
        ach_fwd = fwd_steps * B_FWD / (ms_fwd * 1e-3) / 1e9
        default_cfg = (R == 256 and n == 1024 * 1024 and world == 1 and not args.no_sort and not args.direct_atomics
                       and not args.experiment and args.pair == "auto" and not args.fwd_flags and not args.shard_of
                       and not args.adjoint_chunks
                       and not args.adj_flags and args.workload == "metric" and args.source_axis == "y")
        pmc, pmc_src = load_pmc() if default_cfg else (None, None)
        pk_adj = pmc_kernel(pmc, "drrt::k_backtrace_flat", "drrt::k_backtrace_win")
        pk_fwd = pmc_kernel(pmc, "drrt::k_trace_flat", "drrt::k_trace<0>")
        tr_adj = pk_adj and pk_adj.get("hbm_traffic_bytes_per_launch")
        tr_fwd = pk_fwd and pk_fwd.get("hbm_traffic_bytes_per_launch")
        lib_version = lib.drrt_version().decode()
        pmc_lib = ((pmc or {}).get("_meta") or {}).get("lib_version")
        pmc_stale = bool(pmc_src) and pmc_lib != lib_version
        if pmc_stale:
            print(f"bench.py: WARNING: {pmc_src} was recorded with library '{pmc_lib}', this run uses '{lib_version}': the "
                  f"PMC-derived fields (traffic, physical_bound) describe OTHER kernels; re-run tools/profile_bench.sh",
                  file=sys.stderr)
        src_note = (f"{pmc_src} (recorded with library '{pmc_lib}'{'; STALE: this run uses ' + repr(lib_version) if pmc_stale else ''}): "
                    f"rocprofv3 --pmc passes of this command, committed; read from that file, NOT "
                    f"measured in this run") if pmc_src else None
        shard = (f"the metric's single set of {args.rays} rays split into {world} contiguous shards "
                 f"({n} on rank 0)") if main_mode == "strong" else f"{n} rays per GPU (own seed per rank)"
        if args.shard_of > 1 and world == 1:
            shard += f"; THIS RUN: shard {args.shard_index} of {args.shard_of} only"
        if args.workload != "metric":
            shard = (f"NOT the metric's source: the `{args.workload}` workload of `variants` ({n} rays on rank 0"
                     + (", a strip of every view per rank" if m.get("multi_view") else "") + ")"
                     + (f"; THIS RUN: shard {args.shard_index} of {args.shard_of} only" if args.shard_of > 1 and world == 1 else ""))
        roof = {"bound": "hbm", "kernel": "adjoint march (k_backtrace_flat)", "achieved": ach_adj,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_adj / HBM_PEAK_GBS, "traffic": tr_adj,
                "traffic_source": src_note, "pmc_source": pmc_src, "pmc_lib_version": pmc_lib, "pmc_stale": pmc_stale,
                "hbm_measured_gbps": (tr_adj / (ms_adj * 1e-3) / 1e9) if tr_adj else None,
                "algorithmic_bytes_per_ray_step": B_ADJ, "ray_steps_per_launch": adj_steps, "avg_kernel_ms": ms_adj}
        roof.update(physical_bound(pk_adj, ms_adj))
        roof_f = {"bound": "hbm", "kernel": "forward march (k_trace_flat)", "achieved": ach_fwd,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_fwd / HBM_PEAK_GBS, "traffic": tr_fwd,
                  "traffic_source": src_note, "pmc_source": pmc_src,
                  "hbm_measured_gbps": (tr_fwd / (ms_fwd * 1e-3) / 1e9) if tr_fwd else None,
                  "algorithmic_bytes_per_ray_step": B_FWD, "ray_steps_per_launch": fwd_steps, "avg_kernel_ms": ms_fwd,
                  "note": "frac can exceed 1: SURVEY 8.6 counts every tap as an HBM read, but the taps are cache-served"}
        roof_f.update(physical_bound(pk_fwd, ms_fwd))
        out = {
            "metric": "ray-steps/sec (fwd+adjoint), 256^3 RIF grid, 1M rays x 512 steps",
            "value": m["fwd_total"] * args.steps / m["elapsed"],
            "unit": "ray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": main_mode, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "lib_version": lib_version,
            "config": {"workload": f"Luneburg ball {R}^3 fp32 grid (replicated), {shard} from a seeded jittered "
                                   f"plane source on the y=0 face, ds=h/2, fwd trace + adjoint backtrace (dx=dv=1)"
                                   + (", one all-reduce(sum) of the dL/dn grid per step "
                                      f"(backend {args.backend})" if use_dist else ""),
                       "grid": R, "global_rays": int(args.rays if main_mode == "strong" else args.rays * world),
                       "rays_rank0": n, "fwd_ray_steps_rank0": fwd_steps, "adj_ray_steps_rank0": adj_steps,
                       "fwd_ray_steps_global": m["fwd_total"], "n_failed": m["n_failed"],
                       "sort_rays": not args.no_sort, "pair_grid": bool(m["pair"]),
                       "parallelism": f"ray-shard x{world}", "backend": args.backend if use_dist else None,
                       "shard_of": args.shard_of or None, "shard_index": args.shard_index if args.shard_of else None,
                       "adjoint_kernel": m["adjoint_kernel"], "adjoint_chunks": args.adjoint_chunks or None},
            "roofline": roof,
            "roofline_fwd": roof_f,
            "phase_ms": {"sort_avg": m["ms_sort"], "zero_grid": m["ms_zero"],
                         "pair_copy": None if m["ms_quad"] != m["ms_quad"] else m["ms_quad"],
                         "trace": ms_fwd, "backtrace": ms_adj,
                         "allreduce": (m["ms_allreduce_plain"] if m.get("ms_allreduce_plain") is not None else m["ms_allreduce"]) if use_dist else None,
                         # one whole-grid reduce after the adjoint: all of it is exposed.  With --adjoint-chunks K
                         # --overlap-reduce (plane-source sets) the planes that are final after each chunk are reduced on a
                         # side stream under the next chunk; `allreduce_exposed` is then what remains after the last chunk and
                         # `allreduce` one whole-grid reduce timed once in the same run
                         "allreduce_exposed": m["ms_allreduce"] if use_dist else None,
                         "allreduce_overlapped": bool(m.get("overlap") is not None)},
            "fwd_only_ray_steps_per_s_per_gpu": fwd_steps / (ms_fwd * 1e-3),
            # the whole step's algorithmic bytes (32 B per forward + 64 B per adjoint ray-step) per second over
            # the HBM peak: above 1 means SURVEY's byte model is exhausted as a yardstick (the taps are cache-served) --
            # the bound that physically applies is in roofline.physical_bound / valu_issue_frac / wait_frac
            "whole_step_algorithmic_over_peak": (fwd_steps * B_FWD + adj_steps * B_ADJ) / (m["elapsed"] / args.steps)
                                                / 1e9 / HBM_PEAK_GBS,
        }
        if m.get("chunking"):
            out["chunking"] = m["chunking"]
        if m.get("overlap"):
            out["overlap"] = m["overlap"]
        if "weak" in results and main_mode != "weak":
            w = results["weak"]
            out["weak_scaling"] = {"value": w["fwd_total"] * args.steps / w["elapsed"], "unit": "ray-steps/s",
                                   "ms_per_step": w["elapsed"] / args.steps * 1e3, "rays_per_gpu": w["n"],
                                   "phase_ms": {"sort_avg": w["ms_sort"], "trace": w["ms_fwd"], "backtrace": w["ms_adj"],
                                                "allreduce": w["ms_allreduce"] if use_dist else None}}
        if world == 1 and not args.no_variants and not args.shard_of and args.workload == "metric":
            base_ns = ms_adj * 1e6 / max(adj_steps, 1)
            out["variants"] = {"headline_adj_ns_per_ray_step": base_ns,
                               "headline_fwd_ns_per_ray_step": ms_fwd * 1e6 / max(fwd_steps, 1)}
            for name in [v for v in args.variants.split(",") if v]:
                v = run_variant(name)
                v["adj_ns_ratio_to_headline"] = v["adj_ns_per_ray_step"] / base_ns
                out["variants"][name] = v
        if world == 1 and not args.no_cpu_baseline and args.workload == "metric":
            def gpu_subset(sel):
                """The benchmark's own kernel configuration (same flags as the timed run) on the rays `sel`."""
                import numpy as np
                fl = m["flags"]
                note = (f"fwd 0x{fl[0]:x} adj 0x{fl[1]:x} (sort + order hand-over, pair copy "
                        f"{'built by the forward, reused by the adjoint' if fl[2] else 'off'})")
                if len(sel) == m["n"]:
                    r = m                                      # the timed run's last step IS the result
                else:
                    idx = torch.from_numpy(np.asarray(sel)).to(dev)
                    r = bench_rays(m["pos"][idx].contiguous(), m["vel"][idx].contiguous(), 1, 0, force_flags=fl, keep=True)
                return {"xt": r["xt"].cpu().numpy(), "vt": r["vt"].cpu().numpy(), "grad": r["grad"].cpu().numpy(),
                        "fwd_steps": r["fwd_steps"], "adj_steps": r["adj_steps"], "flags_note": note}
            base, parity = cpu_baseline(R, h, ds, rif.cpu().numpy(), m["pos"].cpu().numpy(), m["vel"].cpu().numpy(),
                                        target_seconds=args.cpu_seconds, gpu=gpu_subset if "xt" in m else None)
            out["cpu_baseline"] = base
            if parity is not None:
                out["parity_check"] = parity
                if not parity["ok"]:
                    # a number whose kernels disagree with the oracle is not a result: no `value`, non-zero exit
                    print("bench.py: PARITY CHECK FAILED: " + json.dumps(parity), file=sys.stderr)
                    out = {"metric": out["metric"], "value": None, "unit": out["unit"], "n_gpus": world,
                           "error": "parity_check failed: the timed kernels disagree with the oracle; no throughput is reported",
                           "lib_version": lib_version, "parity_check": parity}
                    rc = 3
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
