"""CPU tier: the ray-sharded multi-process path (adjointnonlinearraytracing_amd/dist.py) with
world_size 2 over gloo.  The per-rank march is injected (the CPU oracle stands in for the HIP
kernels, which need a GPU); what is under test is the sharding, the hand-over of the forward's visit
order to the adjoint, and the single all-reduce: every rank must end up with the gradient of the GLOBAL
ray set."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _oracle_trace(rif_flat, shape, x, v, h, ds):
    """Stand-in for the HIP forward: also returns a visit order (here: a fixed permutation of the shard), as
    dist._hip_trace does, so that the order hand-over to the adjoint is under test."""
    from oracle import oracle as O
    o = O.trace(rif_flat.numpy(), tuple(shape), x.numpy(), v.numpy(), h, ds, dtype=np.float32)
    order = torch.arange(x.shape[0] - 1, -1, -1, dtype=torch.int32)
    return torch.from_numpy(o["xt"]), torch.from_numpy(o["vt"]), order


_seen_orders = []


def _oracle_backtrace(rif_flat, shape, xt, vt, gx, gv, h, ds, order=None):
    """Stand-in for the HIP adjoint: visits the rays in `order` (the sum must not depend on it) and records that
    it received the forward's order."""
    from oracle import oracle as O
    assert order is not None and order.dtype == torch.int32 and order.numel() == xt.shape[0], \
        "ShardedBackTracerC dropped the forward's visit order"
    _seen_orders.append(order)
    idx = order.long().numpy()
    b = O.backtrace(rif_flat.numpy(), tuple(shape), xt.numpy()[idx], vt.numpy()[idx], gx.numpy()[idx], gv.numpy()[idx],
                    h, ds, dtype=np.float32)
    return torch.from_numpy(b["grad"])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import cases
    from adjointnonlinearraytracing_amd import dist as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    R, span = 17, 1.0
    h = span / (R - 1); ds = h / 2
    rif = torch.from_numpy(cases.smooth_field(R, seed=5)).requires_grad_(True)
    pos, vel = cases.cube_rays(101, span, ds, seed=2)          # 606 rays: not divisible by 4
    pos, vel = torch.from_numpy(pos), torch.from_numpy(vel)
    x, v = D.shard_rays(rank, world, pos, vel)
    xt, vt = D.ShardedBackTracerC.apply(rif, x, v, h, ds, None, _oracle_trace, _oracle_backtrace)
    loss = (xt ** 2).sum() + vt.sum()                          # ray-separable loss (core/luneburg_opt.py:102)
    loss.backward()
    assert len(_seen_orders) == 1 and _seen_orders[0].numel() == x.shape[0]     # the adjoint got this shard's order
    tot = loss.detach().clone(); dist.all_reduce(tot)
    np.save(os.path.join(out_dir, f"grad_{rank}.npy"), rif.grad.numpy())
    np.save(os.path.join(out_dir, f"loss_{rank}.npy"), tot.numpy())
    dist.destroy_process_group()


def test_shard_bounds_cover_all_rays():
    from adjointnonlinearraytracing_amd.dist import shard_bounds
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_world2_gloo_allreduce_gives_global_gradient(tmp_path, oracle):
    import cases
    world, port = 2, 29611 + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "grad_0.npy"), np.load(tmp_path / "grad_1.npy")
    assert np.array_equal(g0, g1)                             # replicated optimisers stay in lock-step
    # single-process reference over the whole ray set
    R, span = 17, 1.0
    h = span / (R - 1); ds = h / 2
    rif = cases.smooth_field(R, seed=5)
    pos, vel = cases.cube_rays(101, span, ds, seed=2)
    o = oracle.trace(rif, rif.shape, pos, vel, h, ds, dtype=np.float32)
    b = oracle.backtrace(rif, rif.shape, o["xt"], o["vt"], 2 * o["xt"], np.ones_like(o["vt"]), h, ds, dtype=np.float32)
    assert cases.rel_l2(g0.ravel(), b["grad"]) < 1e-5        # shard sum == global, up to fp32 reassociation
    l0 = float(np.load(tmp_path / "loss_0.npy"))
    assert abs(l0 - float((o["xt"].astype(np.float64) ** 2).sum() + o["vt"].astype(np.float64).sum())) < 1e-2
