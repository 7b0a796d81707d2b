"""CPU tier: the ray-sharded multi-process path (adjointnonlinearraytracing_amd/dist.py) with
world_size 2 over gloo.  In the worker processes the two module functions of dist.py that call the HIP march are
replaced by the CPU oracle (the kernels need a GPU); what is under test is the sharding, the hand-over of the forward's visit
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
    D._hip_trace, D._hip_backtrace = _oracle_trace, _oracle_backtrace      # this worker process only: the CPU stand-ins
    xt, vt = D.ShardedBackTracerC.apply(rif, x, v, h, ds, None)
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


# ---- image losses on ray shards (core/image_opt.py:99-112) --------------------------------------------------
class _CpuSensor(torch.autograd.Function):
    """Test stand-in for sensor.generate_sensor (HIP): oracle/sensor_ref.py's numpy splat + analytic backward."""

    @staticmethod
    def forward(ctx, x, v, p, n, res, span):
        from oracle import sensor_ref as SR
        ctx.args = (x.detach().numpy(), v.detach().numpy(), p.numpy(), n.numpy(), res, span)
        return torch.from_numpy(SR.generate_sensor(ctx.args[0], ctx.args[1], 1.0, ctx.args[2], ctx.args[3], res, span))

    @staticmethod
    def backward(ctx, g):
        from oracle import sensor_ref as SR
        x, v, p, n, res, span = ctx.args
        gx, gv = SR.generate_sensor_backward(x, v, 1.0, p, n, res, span, g.numpy())
        return torch.from_numpy(gx), torch.from_numpy(gv), None, None, None, None


_IMG = dict(R=17, span=1.0, res=12, views=2, per_view=150)


def _image_problem():
    """Two plane views (+y and +x), a fixed target image per view."""
    import cases
    R, span = _IMG["R"], _IMG["span"]
    h = span / (R - 1); ds = h / 2
    rif = cases.smooth_field(R, seed=7, amp=0.05)
    xs, vs, planes = [], [], []
    for k, axis in enumerate((1, 0)):
        p, v = cases.plane_rays(_IMG["per_view"], span, ds, seed=11 + k, axis=axis, tilt=0.02, lo=0.25, hi=0.75)
        xs.append(p); vs.append(v)
        nrm = np.zeros(3); nrm[axis] = 1.0
        planes.append((np.full(3, span / 2) + nrm * span * 0.6, nrm))
    rng = np.random.default_rng(3)
    targets = [rng.uniform(0.5, 1.5, (_IMG["res"], _IMG["res"])) for _ in planes]
    return rif, np.concatenate(xs).astype(np.float32), np.concatenate(vs).astype(np.float32), planes, targets, h, ds


def _image_loss(xt, vt, counts, planes, targets, reduce_fn):
    """core/image_opt.py:99-112 in small: per view, splat the view's rays into its sensor image, sum over ranks
    (`reduce_fn`), normalise to unit mean (source.sum_norm), MSE against the measurement; mean over views."""
    loss = 0.0
    for xv, vv, (p, n), tgt in zip(xt.split(counts), vt.split(counts), planes, targets):
        img = _CpuSensor.apply(xv.double(), vv.double(), torch.from_numpy(p), torch.from_numpy(n), _IMG["res"], _IMG["span"])
        img = reduce_fn(img)
        img = img * (img.numel() / img.sum())
        loss = loss + torch.mean((img - torch.from_numpy(tgt)) ** 2)
    return loss / len(planes)


def _image_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from adjointnonlinearraytracing_amd import dist as D
    D.init_from_env(backend="gloo")
    rif_np, pos, vel, planes, targets, h, ds = _image_problem()
    rif = torch.from_numpy(rif_np).requires_grad_(True)
    pos, vel = torch.from_numpy(pos), torch.from_numpy(vel)
    n, views = pos.shape[0], _IMG["views"]
    x, v = D.shard_rays(rank, world, pos, vel, views=views)               # a strip of EVERY view on every rank
    counts = D.local_views(n, views, rank, world)
    assert sum(counts) == x.shape[0] and min(counts) > 0
    D._hip_trace, D._hip_backtrace = _oracle_trace, _oracle_backtrace      # this worker process only: the CPU stand-ins
    xt, vt = D.ShardedBackTracerC.apply(rif, x, v, h, ds, None)
    loss = _image_loss(xt, vt, counts, planes, targets, D.allreduce_image)
    loss.backward()
    np.save(os.path.join(out_dir, f"igrad_{rank}.npy"), rif.grad.numpy())
    np.save(os.path.join(out_dir, f"iloss_{rank}.npy"), np.asarray(float(loss)))
    np.save(os.path.join(out_dir, f"icounts_{rank}.npy"), np.asarray(counts))
    dist.destroy_process_group()


def test_shard_rays_interleaves_views():
    from adjointnonlinearraytracing_amd import dist as D
    x = torch.arange(26)
    per = [10, 9, 7]
    got = [D.shard_rays(r, 3, x, views=per)[0].tolist() for r in range(3)]
    assert sorted(sum(got, [])) == list(range(26))                         # a partition of the set
    bounds = np.cumsum([0] + per)
    for r in range(3):
        lv = D.local_views(26, per, r, 3)
        assert all(c > 0 for c in lv) and sum(lv) == len(got[r])          # every rank holds rays of every view
        k = 0
        for vi, c in enumerate(lv):                                       # ... in view order, contiguous strips
            assert all(bounds[vi] <= e < bounds[vi + 1] for e in got[r][k:k + c]); k += c
    assert D.shard_rays(1, 2, torch.arange(12), views=3)[0].tolist() == [2, 3, 6, 7, 10, 11]
    with pytest.raises(ValueError):
        D.shard_rays(0, 2, torch.arange(10), views=3)


@pytest.mark.timeout(300)
def test_world2_image_loss_on_ray_shards(tmp_path, oracle):
    """An image-MSE loss through per-view sensor images on ray shards: with dist.allreduce_image (forward all-reduce,
    backward identity) and dist.ShardedBackTracerC every rank gets the loss and the dL/dn of the single-process run."""
    world, port = 2, 29811 + (os.getpid() % 150)
    mp.spawn(_image_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "igrad_0.npy"), np.load(tmp_path / "igrad_1.npy")
    assert np.array_equal(g0, g1)
    assert float(np.load(tmp_path / "iloss_0.npy")) == pytest.approx(float(np.load(tmp_path / "iloss_1.npy")), rel=1e-12)
    # single process, all rays
    rif_np, pos, vel, planes, targets, h, ds = _image_problem()
    rif = torch.from_numpy(rif_np).requires_grad_(True)

    class _Single(torch.autograd.Function):
        @staticmethod
        def forward(ctx, r, x, v):
            o = _oracle_trace(r.detach().flatten(), r.shape, x, v, h, ds)
            ctx.save_for_backward(r, o[0], o[1]); ctx.order = o[2]
            return o[0], o[1]

        @staticmethod
        def backward(ctx, gx, gv):
            r, xt, vt = ctx.saved_tensors
            return _oracle_backtrace(r.detach().flatten(), r.shape, xt, vt, gx, gv, h, ds, order=ctx.order).reshape(r.shape), None, None

    xt, vt = _Single.apply(rif, torch.from_numpy(pos), torch.from_numpy(vel))
    counts = [_IMG["per_view"]] * _IMG["views"]
    loss = _image_loss(xt, vt, counts, planes, targets, lambda im: im)
    loss.backward()
    import cases
    assert float(np.load(tmp_path / "iloss_0.npy")) == pytest.approx(float(loss), rel=1e-9)
    assert cases.rel_l2(g0, rif.grad.numpy()) < 1e-5
    assert float(np.abs(g0).sum()) > 0


# ---- slab-wise all-reduce under a depth-chunked adjoint (dist.SlabReducer) ----------------------------------------------
def _prog(active, pos_lo, pos_hi, vel_lo, vel_hi, s_lo, s_hi):
    return dict(active=active, pos_min=pos_lo, pos_max=pos_hi, vel_min=vel_lo, vel_max=vel_hi, sample_min=s_lo, sample_max=s_hi)


def _slab_scenarios(rank, shape, h):
    """Per scenario: the progress blocks this rank reports after each of 4 chunks.  The rays march towards -y (adjoint of a
    +y plane source), rank 1 a little behind rank 0; shape = (D, H, W)."""
    D, H, W = shape
    top = (H - 1) * h
    lag = 0.6 * h * rank
    def down(k, turned=False, sample_hi=None):                  # after chunk k: the deepest marching ray stands at y_k
        y_hi = top * (1.0 - 0.25 * (k + 1)) + lag
        y_lo = max(0.0, y_hi - 1.5 * h)
        prev_hi = top * (1.0 - 0.25 * k) + lag if k else top
        vy = (-0.2, 1.0) if turned else (0.7, 1.0)
        return _prog(50 if k < 3 else 0, [0.1, y_lo, 0.1], [0.9 * (W - 1) * h, y_hi, 0.9 * (D - 1) * h],
                     [-.1, vy[0], -.1], [.1, vy[1], .1], [0.1, y_hi, 0.1],
                     [0.9 * (W - 1) * h, prev_hi if sample_hi is None else sample_hi, 0.9 * (D - 1) * h])
    def up_z(k):                                                # rays marching towards +z (v_z < 0)
        zt = (D - 1) * h
        z_lo = zt * 0.25 * (k + 1) - lag
        prev = zt * 0.25 * k - lag if k else 0.0
        return _prog(40 if k < 3 else 0, [0.1, 0.1, max(0.0, z_lo)], [0.5, 0.5, min(zt, z_lo + 1.2 * h)], [-.1, -.1, -1.0], [.1, .1, -0.6],
                     [0.1, 0.1, max(0.0, prev)], [0.5, 0.5, min(zt, z_lo + 1.2 * h)])
    return {
        "down_y": [down(k) for k in range(4)],
        "up_z": [up_z(k) for k in range(4)],
        # a ray of rank 1 turns around in chunk 2: nothing more is handed in early, the rest goes at the end
        "turned": [down(0), down(1), down(2, turned=(rank == 1)), down(3)],
        # rank 0's third chunk contributes ABOVE planes that were handed in after chunk 1: the slab results must be dropped
        "violated": [down(0), down(1), down(2, sample_hi=(top if rank == 0 else None)), down(3)],
        # no axis on which all rays agree (a multi-view set): the plain whole-grid reduce
        "no_axis": [_prog(30, [0, 0, 0], [.5, .5, .5], [-1, -1, -1], [1, 1, 1], [0, 0, 0], [.5, .5, .5])] * 4,
    }


def _slab_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from adjointnonlinearraytracing_amd import dist as D
    D.init_from_env(backend="gloo")
    shape, h = (7, 13, 9), 0.1                                   # (D, H, W): non-cubic on purpose
    report = {}
    for name, seq in _slab_scenarios(rank, shape, h).items():
        g = torch.from_numpy(np.random.default_rng(100 + rank).normal(size=shape).astype(np.float32)).reshape(-1).clone()
        want = g.clone(); dist.all_reduce(want)                 # the whole-grid reduce: the reference result
        red = D.SlabReducer(g, shape, h)
        for pr in seq:
            red.after_chunk(pr)
        early = sum(int(np.prod(b.shape)) for _, b, _ in red.parts)          # voxels handed in BEFORE the end
        out = red.finish()
        assert torch.equal(out, want), name                      # slab-wise == whole-grid, bit for bit (same two summands)
        report[name] = dict(early=early, violated=red.violated, stopped=red.stopped, choice=red.choice)
    np.save(os.path.join(out_dir, f"slab_{rank}.npy"), np.asarray([repr(report)]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_slabwise_allreduce_equals_whole_grid_allreduce(tmp_path):
    """dist.SlabReducer over gloo, world 2: whatever the ranks report -- planes final early, a ray that turns around, a
    chunk that contributes into planes already handed in, no common axis -- every rank ends with exactly the whole-grid
    all-reduce; and in the regular case most of the grid is reduced BEFORE the last chunk has finished."""
    world, port = 2, 29911 + (os.getpid() % 60)
    mp.spawn(_slab_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rep = [eval(str(np.load(tmp_path / f"slab_{r}.npy")[0])) for r in range(world)]
    assert rep[0] == rep[1]                                      # the ranks agreed on every decision
    r = rep[0]
    nvox = 7 * 13 * 9
    assert r["down_y"]["choice"] == (1, True) and r["down_y"]["early"] >= 0.6 * nvox and not r["down_y"]["violated"]
    assert r["up_z"]["choice"] == (2, False) and r["up_z"]["early"] >= 0.5 * nvox
    assert r["turned"]["stopped"] and 0 < r["turned"]["early"] < r["down_y"]["early"]
    assert r["violated"]["violated"]
    assert r["no_axis"]["choice"] is None and r["no_axis"]["early"] == 0


def _chunked_standin(rif_flat, shape, xt, vt, gx, gv, h, ds, order, chunks, on_chunk):
    """CPU stand-in for TracerC.backtrace_chunked: the oracle's adjoint in one go, then `chunks` progress reports of rays
    marching down the y axis (the grid is complete from the start, so any slab partition must reproduce it)."""
    g = _oracle_backtrace(rif_flat, shape, xt, vt, gx, gv, h, ds, order=order)
    H = shape[1]
    for k in range(chunks):
        y_hi = (H - 1) * h * (1.0 - (k + 1) / chunks)
        on_chunk(k, g, _prog(10 if k < chunks - 1 else 0, [0, max(0.0, y_hi - h), 0], [1, y_hi, 1], [-.1, .8, -.1], [.1, 1, .1],
                             [0, y_hi, 0], [1, (H - 1) * h * (1.0 - k / chunks), 1]))
    return g


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import cases
    from adjointnonlinearraytracing_amd import dist as D
    D.init_from_env(backend="gloo")
    R, span = 17, 1.0
    h = span / (R - 1); ds = h / 2
    pos, vel = cases.plane_rays(400, span, ds, seed=2, axis=1, tilt=0.05)
    pos, vel = torch.from_numpy(pos), torch.from_numpy(vel)
    x, v = D.shard_rays(rank, world, pos, vel)
    D._hip_trace, D._hip_backtrace = _oracle_trace, _oracle_backtrace
    D._hip_backtrace_chunked, D._decode_progress = _chunked_standin, (lambda p: p)
    grads = {}
    for chunks in (0, 4):
        rif = torch.from_numpy(cases.smooth_field(R, seed=5)).requires_grad_(True)
        xt, vt = D.ShardedBackTracerC.apply(rif, x, v, h, ds, None, chunks)
        ((xt ** 2).sum() + vt.sum()).backward()
        grads[chunks] = rif.grad.numpy().copy()
    assert np.array_equal(grads[0], grads[4])
    np.save(os.path.join(out_dir, f"ov_{rank}.npy"), grads[4])
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_sharded_tracer_with_overlapped_reduce(tmp_path, oracle):
    """dist.ShardedBackTracerC(..., overlap_chunks=4) == the plain sharded tracer (one whole-grid all-reduce) on every rank."""
    world, port = 2, 29711 + (os.getpid() % 90)
    mp.spawn(_overlap_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert np.array_equal(np.load(tmp_path / "ov_0.npy"), np.load(tmp_path / "ov_1.npy"))
