"""`drrt` module mirror: ``TracerC`` / ``TracerS`` objects with the reference's method names and
argument order (``/root/reference/src/drrt.cpp:21-59``, ``include/tracer.h:15-89``), operating on
torch tensors instead of enoki arrays and backed by the hand-written HIP kernels in
``csrc/`` through the C ABI (``include/drrt_hip.h``).

* ``TracerC`` -- tensors on the ``cuda`` (ROCm) device; asynchronous on torch's current stream.
* ``TracerS`` -- the reference's CPU class (``Tracer<false,false>``, binds trace / trace_sdf /
  trace_target / backtrace only, ``src/drrt.cpp:38-45``).  Here it accepts CPU tensors, stages
  them to the GPU, runs the SAME HIP kernels and copies the results back: there is deliberately
  no CPU compute path in this package.
* ``TracerD`` (enoki autodiff, ``src/drrt.cpp:28-36``) is out of scope; constructing it raises.

All methods ``detach()`` / ``contiguous()`` / cast to fp32 exactly where the reference narrows
(``core/tracer.py:299-301``, ``include/tracer.h:20-21``); ``res`` may be any 3-sequence
(``torch.Size`` is what the scripts pass, ``core/tracer.py:298,307``).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import dataclasses
import threading
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib


@dataclass
class Options:
    sort_rays: bool = True        # locality-sort rays by entry voxel (DRRT_FLAG_SORT_RAYS)
    corrected_h: bool = False     # adjoint: divide gradient splat by h; default = as written
    check_failed: bool = True     # print "failed to exit all rays" like src/tracer.cpp:89-90 (asynchronously: no host
                                  # sync per call; the message may appear one call late -- see flush_warnings())
    direct_atomics: bool = False  # adjoint: one global atomic per tap (debug / A-B)
    adjoint_window: str = "auto"  # adjoint: "auto" = the bundles of the call are classified on the device and the box-window kernel
                                  # (k_backtrace_flat) or the ring-window kernel (k_backtrace_ring; its general or its
                                  # sparse-only instantiations) runs; "box" / "ring" / "ring_sparse" / "ring_direct" force one, "ring_general"
                                  # keeps the choice but never takes the sparse-only instantiation (A-B)
    chord_key: bool = False       # locality sort with the rounds-1/2 key (DRRT_FLAG_CHORD_KEY, A-B)
    pair_grid: object = "auto"    # the "pair copy" of the grid in the workspace (DRRT_FLAG_PAIR_GRID; 8 bytes per voxel, two
                                  # 16-byte gathers per cell instead of four 8-byte ones).  True, False, or "auto" = a
                                  # forward march builds it when the call keeps the whole GPU busy and does enough
                                  # ray-steps per voxel to pay for the copy, and the adjoint paired with that forward
                                  # reuses it.  Bit-identical; 256^3 / 1M rays: forward 1.31 -> 1.06 ms (+0.05 ms for the
                                  # copy), adjoint -2 %

    @property
    def quad_grid(self):          # round-1 name
        return self.pair_grid

    @quad_grid.setter
    def quad_grid(self, v):
        self.pair_grid = v


options = Options()          # the process-wide defaults

# Per-thread overrides: `with drrt.using(sort_rays=False): ...` changes the options of the calls made by THIS thread inside
# the block and nothing else (two threads with different settings do not race on the module global; the visit-order hint
# of the C ABI is per thread in the same way).  Calls outside any block read the module-level `options`.
_tls = threading.local()


def _opt() -> Options:
    return getattr(_tls, "options", None) or options


@contextlib.contextmanager
def using(**overrides):
    """Context manager: the tracer calls of the current thread run with `overrides` applied on top of the options in
    effect (``with drrt.using(corrected_h=True, pair_grid=False): ...``); restored on exit, nestable."""
    prev = getattr(_tls, "options", None)
    _tls.options = dataclasses.replace(prev or options, **overrides)
    try:
        yield _tls.options
    finally:
        _tls.options = prev

# last call's statistics (ray_steps, n_failed, iters) as a device tensor of 3 int64 words
last_stats: Optional[torch.Tensor] = None

# Visit order (int32 ray indices) used by the last SORTED forward call: a VIEW into that call's workspace, valid until
# the next call on the same (device, stream) that sorts rays or stores state there -- any trace*, or a backtrace* that is
# not given an order.  A stale view is recognised (generation stamp `drrt_gen`) and ignored by the calls it is handed
# to, which then sort for themselves: slower, never wrong.  Holders that outlive the next call -- tracer.Back*TracerC's
# ctx, dist.ShardedBackTracerC -- take a private copy with keep_order().  Handed to the paired backtrace
# (drrt_set_order_hint) the adjoint visits rays in the forward's bundle order (include/drrt_hip.h, "visit order hand-over").
last_order: Optional[torch.Tensor] = None
_order_gen: Dict[tuple, int] = {}            # per workspace key: how often its order / state region has been rewritten

# one scratch buffer per (device, stream): calls queued on different streams must not share scratch
_workspaces: Dict[tuple, torch.Tensor] = {}


def _wkey(device: torch.device) -> tuple:
    return (device, torch.cuda.current_stream(device).cuda_stream)


def _flags(adjoint: bool = False) -> int:
    f = 0
    if _opt().sort_rays:
        f |= _lib.FLAG_SORT_RAYS
    if adjoint and _opt().corrected_h:
        f |= _lib.FLAG_CORRECTED_H
    if adjoint and _opt().direct_atomics:
        f |= _lib.FLAG_DIRECT_ATOMICS
    if adjoint and _opt().adjoint_window == "box":
        f |= _lib.FLAG_STATIC_WINDOW
    if adjoint and _opt().adjoint_window == "ring":
        f |= _lib.FLAG_RING_WINDOW
    if adjoint and _opt().adjoint_window == "ring_sparse":       # A-B: the ring kernel's sparse-only instantiation, forced
        f |= _lib.FLAG_RING_WINDOW | _lib.FLAG_RING_SPARSE
    if adjoint and _opt().adjoint_window == "ring_direct":       # A-B: ... its direct sparse-only instantiation, forced
        f |= _lib.FLAG_RING_WINDOW | _lib.FLAG_RING_SPARSE | _lib.FLAG_RING_DIRECT
    if adjoint and _opt().adjoint_window == "ring_general":      # A-B: device-side choice between box and the GENERAL ring kernel
        f |= _lib.FLAG_RING_GENERAL
    if _opt().chord_key:
        f |= _lib.FLAG_CHORD_KEY
    if adjoint and _EXPERIMENT:
        f |= (_EXPERIMENT & 0xFF) << 8          # development ablations of the adjoint kernel (include/drrt_hip.h)
    return f


_EXPERIMENT = 0


def _workspace(n: int, flags: int, device: torch.device, nvox: int = 0) -> torch.Tensor:
    need = int(_lib.load().drrt_workspace_bytes_grid(n, nvox, flags))
    key = _wkey(device)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
        _quad_tokens.pop(key, None)
    if not (flags & _lib.FLAG_PAIR_GRID):
        _quad_tokens.pop(key, None)             # this call may overwrite the region a pair copy lived in
    return ws


# What the pair copy in a device's workspace was built from: (rif tensor, key).  Holding the tensor keeps its
# storage alive, so equal (data_ptr, version counter) means "same contents"; n and the sort bit fix where in
# the workspace the copy lives.
_quad_tokens: Dict[tuple, tuple] = {}


def _march_workspace(rif_: torch.Tensor, res, n: int, h: float, ds: float, flags: int, device: torch.device,
                     paired: bool = False, adjoint: bool = False):
    """Workspace + final flags of a grid march call: decides on DRRT_FLAG_PAIR_GRID (options.pair_grid).
    Forward calls always rebuild the copy.  An adjoint the caller explicitly pairs with its forward
    (`paired`: it passed the forward's visit order) adds DRRT_FLAG_PAIR_REUSE when the workspace still holds the
    copy built from this very tensor (same storage, same version counter, same layout) and no other call has
    used the workspace since.  With "auto" an adjoint never builds the copy itself (it gains ~2 %, less than the
    copy costs): it uses it only when it can reuse the forward's."""
    q = _opt().pair_grid
    auto = q == "auto"
    if auto:
        # the copy moves 12 B per voxel; the gathers it halves only bound the march when the GPU is full of waves
        ok = float(ds) > 0.0 and float(h) > 0.0               # invalid steps are the library's to report
        q = ok and n >= _PAIR_AUTO_MIN_RAYS and \
            n * max(int(r) for r in res) * (float(h) / float(ds)) >= 8.0 * rif_.numel()
    if not q or n == 0:
        return flags, _workspace(n, flags, device)
    pflags = flags | _lib.FLAG_PAIR_GRID
    key = (rif_.data_ptr(), rif_._version, rif_.numel(), n, flags & _lib.FLAG_SORT_RAYS)
    if auto and adjoint:
        tok = _quad_tokens.get(_wkey(device))
        need = int(_lib.load().drrt_workspace_bytes_grid(n, rif_.numel(), pflags))
        ws = _workspaces.get(_wkey(device))
        if not (paired and tok is not None and tok[1] == key and ws is not None and ws.numel() >= need):
            return flags, _workspace(n, flags, device)
    ws = _workspace(n, pflags, device, rif_.numel())         # may reallocate -> drops the token
    tok = _quad_tokens.get(_wkey(device))
    if paired and tok is not None and tok[1] == key:
        pflags |= _lib.FLAG_PAIR_REUSE
    _quad_tokens[_wkey(device)] = (rif_, key)
    return pflags, ws


_PAIR_AUTO_MIN_RAYS = 384 * 1024     # about 6 resident waves per SIMD on 256 CUs


def _dev(t: torch.Tensor) -> torch.device:
    if not t.is_cuda:
        raise RuntimeError("TracerC expects tensors on the cuda (ROCm) device; use TracerS for host tensors")
    return t.device


def _f32(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _is_half(*ts: torch.Tensor) -> bool:
    """fp16 ray-state mode (BASELINE config 5): every ray tensor of the call is float16."""
    return all(t.dtype == torch.float16 for t in ts)


def _is_q16(*ts: torch.Tensor) -> bool:
    """16-bit ray state "q16" (include/drrt_hip.h): every position / direction tensor of the call is int16 codes."""
    return all(t.dtype == torch.int16 for t in ts)


def encode_rays16(res: Sequence[int], h: float, pos: Optional[torch.Tensor] = None, vel: Optional[torch.Tensor] = None):
    """fp32 (n,3) positions / directions -> q16 codes (int16 tensors; position codes are unsigned 16-bit values stored
    in int16 storage).  Rounded on the device exactly as the kernels round their outputs (drrt_q16_encode)."""
    ref = pos if pos is not None else vel
    dev = _dev(ref)
    with torch.cuda.device(dev):
        p_ = None if pos is None else _rays(pos, dev)
        v_ = None if vel is None else _rays(vel, dev)
        n = (p_ if p_ is not None else v_).shape[0]
        pq = None if p_ is None else torch.empty(n, 3, dtype=torch.int16, device=dev)
        vq = None if v_ is None else torch.empty(n, 3, dtype=torch.int16, device=dev)
        _lib.check(_lib.load().drrt_q16_encode(_res3(res), float(h), n, _p(p_), _p(v_), _p(pq), _p(vq), _stream(dev)))
    return tuple(t for t in (pq, vq) if t is not None) if (pos is not None and vel is not None) else (pq if pq is not None else vq)


def decode_rays16(res: Sequence[int], h: float, pos_q: Optional[torch.Tensor] = None, vel_q: Optional[torch.Tensor] = None):
    """q16 codes -> fp32 (exact widening, drrt_q16_decode)."""
    ref = pos_q if pos_q is not None else vel_q
    dev = _dev(ref)
    with torch.cuda.device(dev):
        n = ref.shape[0]
        pq = None if pos_q is None else pos_q.detach().contiguous()
        vq = None if vel_q is None else vel_q.detach().contiguous()
        p_ = None if pq is None else torch.empty(n, 3, dtype=torch.float32, device=dev)
        v_ = None if vq is None else torch.empty(n, 3, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().drrt_q16_decode(_res3(res), float(h), n, _p(pq), _p(vq), _p(p_), _p(v_), _stream(dev)))
    return tuple(t for t in (p_, v_) if t is not None) if (pos_q is not None and vel_q is not None) else (p_ if p_ is not None else v_)


def _rays(t: torch.Tensor, device: torch.device, n: Optional[int] = None, half: bool = False, q16: bool = False) -> torch.Tensor:
    if q16:
        t = t.detach().to(device=device).contiguous()
    else:
        t = t.detach().to(device=device, dtype=torch.float16).contiguous() if half else _f32(t, device)
    if t.dim() != 2 or t.shape[1] != 3 or (n is not None and t.shape[0] != n):
        raise RuntimeError(f"expected a ({'N' if n is None else n},3) ray tensor, got {tuple(t.shape)}")
    return t


def _res3(res: Sequence[int]):
    r = [int(v) for v in res]
    if len(r) != 3:
        raise RuntimeError("res must have 3 entries")
    return (C.c_int * 3)(*r)


def _stream(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _new_stats(device: torch.device) -> torch.Tensor:
    global last_stats
    last_stats = torch.empty(3, dtype=torch.int64, device=device)
    return last_stats


def read_stats(stats: Optional[torch.Tensor] = None) -> Dict[str, int]:
    """Synchronising read of a stats block -> dict(ray_steps, n_failed, iters)."""
    s = (last_stats if stats is None else stats).cpu()
    return dict(ray_steps=int(s[0]), n_failed=int(s[1]), iters=int(s[2]) & 0xFFFFFFFF)


def _bump_order_gen(device: torch.device) -> None:
    """The call about to be made rewrites the order / state region of this (device, stream)'s workspace."""
    k = _wkey(device)
    _order_gen[k] = _order_gen.get(k, 0) + 1


def _capture_order(n: int, device: torch.device) -> None:
    """Hand out the permutation (and the per-ray iteration counts) the library just left in the workspace: views, no
    copies -- see `last_order`."""
    global last_order
    last_order = None
    if not _opt().sort_rays or n < 2:
        return
    cnt = C.c_size_t(0)
    ptr = _lib.load().drrt_last_order(C.byref(cnt))
    if not ptr or cnt.value != n:
        return
    key = _wkey(device)
    ws = _workspaces[key]
    off = int(ptr) - ws.data_ptr()
    if 0 <= off and off + 4 * n <= ws.numel():
        last_order = ws[off:off + 4 * n].view(torch.int32)
        last_order.drrt_gen = (key, _order_gen.get(key, 0))
        # the forward march's per-ray iteration counts ride along ON the order tensor (attribute `drrt_steps`), so every
        # holder of the order hands both to the paired adjoint: its rays then start on the forward march's clock (step
        # hint, include/drrt_hip.h)
        ptr_s = _lib.load().drrt_last_steps(C.byref(cnt))
        off_s = int(ptr_s) - ws.data_ptr() if ptr_s else -1
        if ptr_s and cnt.value == n and 0 <= off_s and off_s + 4 * n <= ws.numel():
            last_order.drrt_steps = ws[off_s:off_s + 4 * n].view(torch.int32)


def keep_order(order: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """A private copy of a visit order (with its iteration counts) that stays valid whatever is called next; None for a
    stale or missing order.  For holders that keep the order across other tracer calls (autograd ctx)."""
    order = _valid_order(order)
    if order is None or getattr(order, "drrt_gen", None) is None:
        return order
    kept = order.clone()
    steps = getattr(order, "drrt_steps", None)
    if steps is not None:
        kept.drrt_steps = steps.clone()
    return kept


def _valid_order(order: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """`order` unless it is a workspace view whose region has been rewritten since it was handed out."""
    if order is None:
        return None
    gen = getattr(order, "drrt_gen", None)
    if gen is not None and _order_gen.get(gen[0], 0) != gen[1]:
        return None
    return order


last_bundle_counters: Optional[torch.Tensor] = None


def _capture_counters(ws: torch.Tensor) -> None:
    """The bundle classification of the adjoint call just made (drrt_last_bundle_counters, include/drrt_hip.h): four int32
    copied out of its workspace (device-to-device, async) -> `last_bundle_counters`, or None when the call did not classify."""
    global last_bundle_counters
    last_bundle_counters = None
    ptr = _lib.load().drrt_last_bundle_counters()
    if not ptr:
        return
    off = int(ptr) - ws.data_ptr()
    if 0 <= off and off + 32 <= ws.numel():
        last_bundle_counters = ws[off:off + 32].view(torch.int32).clone()


def read_bundle_counters() -> Optional[Dict[str, int]]:
    """Synchronising read of `last_bundle_counters` -> which adjoint kernel the last backtrace* call chose, and why.
    The rule is the library's own (`drrt_ring_threshold_pct()`, `drrt_ring_long_threshold_permille()`: its compile-time
    thresholds, so variant builds report what they ran): the ring-window kernel when a fifth of the bundles' START cells do
    not fit the box window or when 7.5 % of the bundles left the forward march 12 or more cells of travel apart (24 iterations at ds = h / 2); its sparse-only
    instantiation unless the call pinned the general one (counter [5])."""
    if last_bundle_counters is None:
        return None
    c = [int(v) for v in last_bundle_counters.cpu()]
    share = c[0] / c[1] if c[1] else 0.0
    pct = int(_lib.load().drrt_ring_threshold_pct())
    ext = int(_lib.load().drrt_ring_long_threshold_permille())
    long_ = bool(c[6] and c[6] * 1000 >= c[1] * ext)
    ring = bool(c[0] and c[0] * 100 >= c[1] * pct) or long_
    sparse = ring and c[5] == 0
    dpct = int(_lib.load().drrt_ring_direct_threshold_pct())
    direct = sparse and c[3] != 0 and c[4] * 100 < c[3] * dpct
    return dict(bundles_not_fitting=c[0], bundles=c[1], lanes_outside=c[2], lanes=c[3], not_fitting_share=share,
                start_pair_share=(c[4] / c[3] if c[3] else 0.0),
                bundles_long=c[6], long_bundle_share=(c[6] / c[1] if c[1] else 0.0), long_threshold_permille=ext,
                ring_threshold_pct=pct,
                direct_threshold_pct=dpct,
                kernel=("ring_direct" if direct else "ring_sparse" if sparse else "ring") if ring else "box")


def decode_chunk_progress(progress: torch.Tensor) -> Dict[str, object]:
    """Synchronising read of a chunk's progress block (drrt_backtrace_chunk_f32) -> dict(active, pos_min, pos_max, vel_min,
    vel_max, sample_min, sample_max): the bounding boxes of where the still-marching rays stand and head, and of the
    samples the chunk contributed at (lists of 3 floats; None when there is no such ray / sample)."""
    k = progress.cpu().to(torch.int64)
    n_active = int(k[12]) & 0xFFFFFFFF
    bits = torch.where(k >= 0, k, k ^ 0x7FFFFFFF).to(torch.int32)
    f = bits.view(torch.float32).tolist()
    out = dict(active=n_active, pos_min=None, pos_max=None, vel_min=None, vel_max=None, sample_min=None, sample_max=None)
    if n_active:
        out.update(pos_min=f[0:3], pos_max=f[3:6], vel_min=f[6:9], vel_max=f[9:12])
    if int(k[13]) != 0x7FFFFFFF:
        out.update(sample_min=f[13:16], sample_max=f[16:19])
    return out


def _hint(order: Optional[torch.Tensor], n: int) -> bool:
    """Arm the library's order (and step) hint for the next march call; -> whether an order was handed over."""
    if order is not None and order.numel() == n and order.dtype == torch.int32 and order.is_cuda:
        _lib.load().drrt_set_order_hint(C.c_void_p(order.data_ptr()), n)
        steps = getattr(order, "drrt_steps", None)
        if steps is not None and steps.numel() == n and steps.dtype == torch.int32 and steps.device == order.device:
            _lib.load().drrt_set_step_hint(C.c_void_p(steps.data_ptr()), n)
        return True
    return False


def _clear_hint() -> None:
    """The library consumes a hint at the entry of the next march call; this covers the paths on which that call
    is never reached (an exception while marshalling arguments)."""
    _lib.load().drrt_set_order_hint(None, 0)
    _lib.load().drrt_set_step_hint(None, 0)


# "failed to exit all rays" (src/tracer.cpp:90) without a host sync per call: the stats block is copied to pinned
# host memory asynchronously behind the kernels, and looked at when the copy has landed -- at the next tracer call,
# at flush_warnings(), or at interpreter exit.  The message can therefore appear one call late; it is never lost.
_pending_warn: list = []      # (event, pinned host tensor)
_pinned_pool: list = []


def _capturing() -> bool:
    """True while the current stream is being captured into a HIP graph: event queries, pinned allocations and host
    copies are not capturable, so the failed-ray bookkeeping stands still during a capture."""
    try:
        return torch.cuda.is_current_stream_capturing()
    except Exception:
        return False


def _drain_warnings(block: bool = False) -> None:
    if _capturing():
        return
    while _pending_warn and (block or _pending_warn[0][0].query()):
        ev, host = _pending_warn.pop(0)
        if block:
            ev.synchronize()
        if int(host[1]) > 0:
            print("failed to exit all rays")            # src/tracer.cpp:90
        _pinned_pool.append(host)


def flush_warnings() -> None:
    """Wait for the outstanding marches and print any pending "failed to exit all rays" message."""
    _drain_warnings(block=True)


def _at_exit() -> None:
    try:
        if _pending_warn:
            _drain_warnings(block=True)
    except Exception:              # the HIP runtime may already be gone at interpreter teardown
        pass


import atexit as _atexit   # noqa: E402
_atexit.register(_at_exit)


def _warn_failed(stats: torch.Tensor) -> None:
    _drain_warnings()
    if not _opt().check_failed or _capturing():      # a captured march reports through its stats block only
        return
    host = _pinned_pool.pop() if _pinned_pool else torch.empty(3, dtype=torch.int64, pin_memory=True)
    host.copy_(stats, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(stats.device))
    _pending_warn.append((ev, host))
    if len(_pending_warn) > 64:                          # bounded backlog
        _drain_warnings(block=True)


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class TracerC:
    """GPU tracer without autodiff -- mirror of ``drrt.TracerC`` (``src/drrt.cpp:47-58``)."""

    # ---- forward ------------------------------------------------------------------------
    def trace(self, rif, res, pos, vel, h, ds) -> Tuple[torch.Tensor, torch.Tensor]:
        """Tracer::trace, src/tracer.cpp:35-100.  float16 pos AND vel select the IEEE-half ray-state variant
        (drrt_trace_f16io: half in, fp32 march, half out); int16 pos AND vel (codes from ``encode_rays16``) select the
        16-bit ray state "q16" (drrt_trace_q16io), which keeps sub-voxel positions -- see include/drrt_hip.h."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            half, q16 = _is_half(pos, vel), _is_q16(pos, vel)
            qpos = (not q16) and pos.dtype == torch.int16          # q16 positions with fp32 directions
            rif_, pos_ = _f32(rif, dev).reshape(-1), _rays(pos, dev, half=half, q16=q16 or qpos)
            n = pos_.shape[0]
            vel_ = _rays(vel, dev, n, half=half, q16=q16)
            xt, vt = torch.empty_like(pos_), torch.empty_like(vel_)
            fl = _flags() | (_lib.FLAG_Q16_POS_ONLY if qpos else 0)
            q16 = q16 or qpos
            (fl, ws), st = _march_workspace(rif_, res, n, h, ds, fl, dev), _new_stats(dev)
            _bump_order_gen(dev)
            fn = _lib.load().drrt_trace_q16io if q16 else (_lib.load().drrt_trace_f16io if half else _lib.load().drrt_trace_f32)
            _lib.check(fn(
                _p(rif_), rif_.numel(), _res3(res), n, _p(pos_), _p(vel_), float(h), float(ds),
                _p(xt), _p(vt), _p(st), _p(ws), ws.numel(), fl, _stream(dev)))
            _capture_order(n, dev)
            _warn_failed(st)
        return xt, vt

    def trace_pln(self, rif, res, pos, vel, pln_o, pln_d, h, ds):
        """Tracer::trace_plane, src/tracer.cpp:102-172 -> (xt, vt, failmask uint8)."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            rif_, pos_ = _f32(rif, dev).reshape(-1), _rays(pos, dev)
            n = pos_.shape[0]
            vel_, po, pd = _rays(vel, dev, n), _rays(pln_o, dev, n), _rays(pln_d, dev, n)
            xt, vt = torch.empty_like(pos_), torch.empty_like(vel_)
            fm = torch.empty(n, dtype=torch.uint8, device=dev)
            fl = _flags()
            (fl, ws), st = _march_workspace(rif_, res, n, h, ds, fl, dev), _new_stats(dev)
            _bump_order_gen(dev)
            _lib.check(_lib.load().drrt_trace_pln_f32(
                _p(rif_), rif_.numel(), _res3(res), n, _p(pos_), _p(vel_), _p(po), _p(pd),
                float(h), float(ds), _p(xt), _p(vt), _p(fm), _p(st), _p(ws), ws.numel(), fl,
                _stream(dev)))
            _capture_order(n, dev)
            _warn_failed(st)
        return xt, vt, fm

    def trace_target(self, rif, res, pos, vel, target, h, ds):
        """Tracer::trace_target, src/tracer.cpp:174-242 -> (xt, vt, dist2)."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            rif_, pos_ = _f32(rif, dev).reshape(-1), _rays(pos, dev)
            n = pos_.shape[0]
            vel_, tg = _rays(vel, dev, n), _rays(target, dev, n)
            xt, vt = torch.empty_like(pos_), torch.empty_like(vel_)
            d2 = torch.empty(n, dtype=torch.float32, device=dev)
            fl = _flags()
            (fl, ws), st = _march_workspace(rif_, res, n, h, ds, fl, dev), _new_stats(dev)
            _bump_order_gen(dev)
            _lib.check(_lib.load().drrt_trace_target_f32(
                _p(rif_), rif_.numel(), _res3(res), n, _p(pos_), _p(vel_), _p(tg),
                float(h), float(ds), _p(xt), _p(vt), _p(d2), _p(st), _p(ws), ws.numel(), fl,
                _stream(dev)))
            _capture_order(n, dev)
            _warn_failed(st)
        return xt, vt, d2

    def trace_sdf(self, rif, sdf, res, pos, vel, h, ds):
        """Tracer::trace_sdf, src/tracer.cpp:244-310."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            rif_, sdf_ = _f32(rif, dev).reshape(-1), _f32(sdf, dev).reshape(-1)
            if sdf_.numel() != rif_.numel():
                raise RuntimeError("Resolution doesn't match data")      # src/volume.cpp:37
            pos_ = _rays(pos, dev)
            n = pos_.shape[0]
            vel_ = _rays(vel, dev, n)
            xt, vt = torch.empty_like(pos_), torch.empty_like(vel_)
            fl = _flags()
            (fl, ws), st = _march_workspace(rif_, res, n, h, ds, fl, dev), _new_stats(dev)
            _bump_order_gen(dev)
            _lib.check(_lib.load().drrt_trace_sdf_f32(
                _p(rif_), _p(sdf_), rif_.numel(), _res3(res), n, _p(pos_), _p(vel_),
                float(h), float(ds), _p(xt), _p(vt), _p(st), _p(ws), ws.numel(), fl, _stream(dev)))
            _capture_order(n, dev)
        return xt, vt

    def trace_cable(self, rif, radius, length, pos, vel, target, ds):
        """Tracer::trace_cable, src/tracer.cpp:312-382 -> (xt, vt, dist2)."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            rif_, pos_ = _f32(rif, dev).reshape(-1), _rays(pos, dev)
            n = pos_.shape[0]
            vel_, tg = _rays(vel, dev, n), _rays(target, dev, n)
            xt, vt = torch.empty_like(pos_), torch.empty_like(vel_)
            d2 = torch.empty(n, dtype=torch.float32, device=dev)
            ws, st = _workspace(n, 0, dev), _new_stats(dev)
            _lib.check(_lib.load().drrt_trace_cable_f32(
                _p(rif_), rif_.numel(), float(radius), float(length), n, _p(pos_), _p(vel_), _p(tg),
                float(ds), _p(xt), _p(vt), _p(d2), _p(st), _p(ws), ws.numel(), 0, _stream(dev)))
            _warn_failed(st)
        return xt, vt, d2

    # ---- adjoint ------------------------------------------------------------------------
    def backtrace(self, rif, res, xt, vt, dx, dv, h, ds, order: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Tracer::backtrace, src/tracer.cpp:384-440 -> flat dL/dn (fp32[nvox]).
        `order` (optional, not in the reference): visit order of the paired forward call.
        float16 xt, vt, dx, dv select the fp16 ray-state variant (fp32 recurrences and accumulation)."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            order = _valid_order(order)
            half = _is_half(xt, vt, dx, dv)
            q16 = _is_q16(xt, vt)                      # q16 exit rays + IEEE-half seeds (drrt_backtrace_q16io)
            qpos = (not q16) and xt.dtype == torch.int16           # q16 positions; directions and seeds fp32
            if q16 and not _is_half(dx, dv):
                raise RuntimeError("q16 exit rays (int16) go with float16 seeds dx, dv")
            rif_, xt_ = _f32(rif, dev).reshape(-1), _rays(xt, dev, half=half, q16=q16 or qpos)
            n = xt_.shape[0]
            vt_ = _rays(vt, dev, n, half=half, q16=q16)
            dx_, dv_ = _rays(dx, dev, n, half=half or q16), _rays(dv, dev, n, half=half or q16)
            grad = torch.empty_like(rif_)
            fl = _flags(adjoint=True) | (_lib.FLAG_Q16_POS_ONLY if qpos else 0)
            q16 = q16 or qpos
            (fl, ws), st = _march_workspace(rif_, res, n, h, ds, fl, dev, paired=order is not None, adjoint=True), _new_stats(dev)
            fn = _lib.load().drrt_backtrace_q16io if q16 else (_lib.load().drrt_backtrace_f16io if half else _lib.load().drrt_backtrace_f32)
            try:
                if not _hint(order, n):
                    _bump_order_gen(dev)               # the adjoint sorts for itself: it rewrites the order region
                _lib.check(fn(
                    _p(rif_), rif_.numel(), _res3(res), n, _p(xt_), _p(vt_), _p(dx_), _p(dv_),
                    float(h), float(ds), _p(grad), _p(st), _p(ws), ws.numel(), fl, _stream(dev)))
                _capture_counters(ws)
            finally:
                _clear_hint()
        return grad

    def backtrace_chunked(self, rif, res, xt, vt, dx, dv, h, ds, order: Optional[torch.Tensor] = None, chunks: int = 4,
                          on_chunk=None) -> torch.Tensor:
        """Tracer::backtrace in `chunks` depth chunks (drrt_backtrace_chunk_f32, include/drrt_hip.h): the same gradient as
        ``backtrace`` (same per-ray contributions, another summation order), computed by `chunks` launches of
        max_steps / chunks iterations each.  After every chunk ``on_chunk(k, grad, progress)`` is called with the running
        gradient (flat fp32[nvox], still being accumulated into by the later chunks) and ``progress`` = a device tensor of
        20 int32 (decode with ``decode_chunk_progress``): the bounding box of the positions and velocities of the rays
        that are still marching -- what lies behind all of them, with none heading back, is final -- and the box of the
        samples this chunk contributed at.  Not in the reference;
        used by ``dist`` to reduce final slabs of dL/dn across ranks while the next chunk marches.  fp32 rays only."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            order = _valid_order(order)
            rif_, xt_ = _f32(rif, dev).reshape(-1), _rays(xt, dev)
            n = xt_.shape[0]
            vt_, dx_, dv_ = _rays(vt, dev, n), _rays(dx, dev, n), _rays(dv, dev, n)
            grad = torch.empty_like(rif_)
            fl = _flags(adjoint=True)
            (fl, ws), st = _march_workspace(rif_, res, n, h, ds, fl, dev, paired=order is not None, adjoint=True), _new_stats(dev)
            lib = _lib.load()
            total = int(lib.drrt_backtrace_max_steps(_res3(res), float(h), float(ds)))
            if total < 0:
                raise RuntimeError("h and ds must be positive and finite")
            chunks = max(1, min(int(chunks), max(total, 1)))
            state = torch.empty(int(lib.drrt_backtrace_chunk_state_bytes(n)), dtype=torch.uint8, device=dev)
            bounds = [total * k // chunks for k in range(chunks + 1)]
            try:
                for k in range(chunks):
                    progress = torch.empty(20, dtype=torch.int32, device=dev)
                    if k == 0:
                        if not _hint(order, n):
                            _bump_order_gen(dev)           # the first chunk sorts for itself
                    else:                                  # later chunks: the order the first chunk used
                        if order is not None:
                            _hint(order, n)
                        elif fl & _lib.FLAG_SORT_RAYS:
                            lib.drrt_set_order_hint(lib.drrt_last_order(None), n)
                    _lib.check(lib.drrt_backtrace_chunk_f32(
                        _p(rif_), rif_.numel(), _res3(res), n, _p(xt_), _p(vt_), _p(dx_), _p(dv_), float(h), float(ds),
                        _p(grad), _p(st), _p(ws), ws.numel(), fl, _stream(dev), _p(state), state.numel(),
                        bounds[k], bounds[k + 1] - bounds[k], _p(progress)))
                    if on_chunk is not None:
                        on_chunk(k, grad, progress)
            finally:
                _clear_hint()
        return grad

    def backtrace_sdf(self, rif, sdf, res, xt, vt, dx, dv, h, ds, order: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Tracer::backtrace_sdf, src/tracer.cpp:443-509."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            order = _valid_order(order)
            rif_, sdf_ = _f32(rif, dev).reshape(-1), _f32(sdf, dev).reshape(-1)
            if sdf_.numel() != rif_.numel():
                raise RuntimeError("Resolution doesn't match data")
            xt_ = _rays(xt, dev)
            n = xt_.shape[0]
            vt_, dx_, dv_ = _rays(vt, dev, n), _rays(dx, dev, n), _rays(dv, dev, n)
            grad = torch.empty_like(rif_)
            fl = _flags(adjoint=True)
            (fl, ws), st = _march_workspace(rif_, res, n, h, ds, fl, dev, paired=order is not None, adjoint=True), _new_stats(dev)
            try:
                if not _hint(order, n):
                    _bump_order_gen(dev)
                _lib.check(_lib.load().drrt_backtrace_sdf_f32(
                    _p(rif_), _p(sdf_), rif_.numel(), _res3(res), n, _p(xt_), _p(vt_), _p(dx_), _p(dv_),
                    float(h), float(ds), _p(grad), _p(st), _p(ws), ws.numel(), fl, _stream(dev)))
                _capture_counters(ws)
            finally:
                _clear_hint()
        return grad

    def backtrace_cable(self, rif, radius, length, xt, vt, dx, dv, ds) -> torch.Tensor:
        """Tracer::backtrace_cable, src/tracer.cpp:511-567 -> dL/d(profile) fp32[rres]."""
        dev = _dev(rif)
        with torch.cuda.device(dev):
            rif_, xt_ = _f32(rif, dev).reshape(-1), _rays(xt, dev)
            n = xt_.shape[0]
            vt_, dx_, dv_ = _rays(vt, dev, n), _rays(dx, dev, n), _rays(dv, dev, n)
            grad = torch.empty_like(rif_)
            ws, st = _workspace(n, 0, dev), _new_stats(dev)
            _lib.check(_lib.load().drrt_backtrace_cable_f32(
                _p(rif_), rif_.numel(), float(radius), float(length), n, _p(xt_), _p(vt_), _p(dx_),
                _p(dv_), float(ds), _p(grad), _p(st), _p(ws), ws.numel(), 0, _stream(dev)))
        return grad

    # ---- print-only smoke methods of the reference (src/tracer.cpp:16-33) ------------------
    def test(self) -> torch.Tensor:
        """Tracer::tester: returns a zero 3-vector."""
        return torch.zeros(1, 3)

    def testscale(self, p) -> None:
        """Tracer::test_in: prints 1.1+0.5 and its floor2int (print-only in the reference)."""
        one = torch.full((1, 3), 1.1) + 0.5
        print(one)
        print(torch.floor(one).to(torch.int32))


class TracerS:
    """Host-tensor tracer -- mirror of ``drrt.TracerS`` (``src/drrt.cpp:38-45``): binds only
    ``trace``, ``trace_sdf``, ``trace_target``, ``backtrace``.  Inputs may live on
    the CPU; compute happens on ``cuda:0`` through ``TracerC`` and results return to the
    inputs' device.  No CPU compute path exists in this package."""

    def __init__(self, device: str = "cuda:0"):
        self._dev = torch.device(device)
        self._c = TracerC()

    def _up(self, *ts):
        return [t.to(self._dev) if isinstance(t, torch.Tensor) else t for t in ts]

    def trace(self, rif, res, pos, vel, h, ds):
        out_dev = pos.device
        r, p, v = self._up(rif, pos, vel)
        return tuple(t.to(out_dev) for t in self._c.trace(r, res, p, v, h, ds))

    def trace_sdf(self, rif, sdf, res, pos, vel, h, ds):
        out_dev = pos.device
        r, s, p, v = self._up(rif, sdf, pos, vel)
        return tuple(t.to(out_dev) for t in self._c.trace_sdf(r, s, res, p, v, h, ds))

    def trace_target(self, rif, res, pos, vel, target, h, ds):
        out_dev = pos.device
        r, p, v, tg = self._up(rif, pos, vel, target)
        return tuple(t.to(out_dev) for t in self._c.trace_target(r, res, p, v, tg, h, ds))

    def backtrace(self, rif, res, xt, vt, dx, dv, h, ds):
        out_dev = rif.device
        r, a, b, c, d = self._up(rif, xt, vt, dx, dv)
        return self._c.backtrace(r, res, a, b, c, d, h, ds).to(out_dev)

    test = TracerC.test
    testscale = TracerC.testscale


class TracerD:
    """enoki-autodiff tracer of the reference (``src/drrt.cpp:28-36``): not provided -- the
    hand-written adjoint (``TracerC.backtrace*``) is the supported gradient path."""

    def __init__(self, *a, **k):
        raise NotImplementedError(
            "drrt.TracerD (enoki autodiff) is out of scope; use TracerC + tracer.BackTracerC")
