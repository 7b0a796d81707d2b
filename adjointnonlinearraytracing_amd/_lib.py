"""ctypes binding of ``libdrrt_hip.so`` -- the C ABI declared in ``include/drrt_hip.h``.

There is NO fallback: if the shared library is missing or a call fails, a ``RuntimeError`` is
raised (the reference turns ``std::runtime_error`` into Python ``RuntimeError`` the same way,
pybind default translator, ``/root/reference/src/volume.cpp:28,37,115,124``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRRT_HIP_LIB: developer override (A-B builds of the same sources, e.g. other compiler flags); default = the in-tree build
LIB_PATH = os.environ.get("DRRT_HIP_LIB") or os.path.join(_HERE, "libdrrt_hip.so")

# flags (include/drrt_hip.h)
FLAG_NONE = 0
FLAG_SORT_RAYS = 1
FLAG_CORRECTED_H = 2
FLAG_NO_ZERO = 4
FLAG_DIRECT_ATOMICS = 8
FLAG_DEBUG_COUNTERS = 16
FLAG_PAIR_GRID = 64
FLAG_PAIR_REUSE = 128
FLAG_QUAD_GRID, FLAG_QUAD_REUSE = FLAG_PAIR_GRID, FLAG_PAIR_REUSE      # round-1 names
FLAG_Q16_POS_ONLY = 0x200000
FLAG_STATIC_WINDOW = 0x400000
FLAG_CHORD_KEY = 0x800000
FLAG_RING_WINDOW = 0x1000000
FLAG_DISPATCH_IN_ORDER = 0x2000000
FLAG_RING_SPARSE, FLAG_RING_GENERAL, FLAG_RING_DIRECT = 0x4000000, 0x8000000, 0x10000000
ADAM_MASK_BOUNDARY, ADAM_CLAMP_MIN = 1, 2

ERR_RES_MISMATCH, ERR_BAD_RES, ERR_ARG, ERR_HIP = -1, -2, -3, -4


class Stats(C.Structure):
    """Mirror of ``drrt_stats`` (device-resident; copy back to read)."""
    _fields_ = [("ray_steps", C.c_ulonglong), ("n_failed", C.c_ulonglong),
                ("iters", C.c_uint), ("reserved", C.c_uint)]


STATS_BYTES = C.sizeof(Stats)

_vp, _f, _sz, _u, _ll, _i = C.c_void_p, C.c_float, C.c_size_t, C.c_uint, C.c_longlong, C.c_int
_d = C.c_double
_tail = [_vp, _vp, _sz, _u, _vp]          # stats, workspace, workspace_bytes, flags, stream

SIGNATURES = {
    # name: (restype, argtypes)   -- must stay in sync with include/drrt_hip.h
    "drrt_workspace_bytes": (_sz, [_sz, _u]),
    "drrt_workspace_bytes_grid": (_sz, [_sz, _ll, _u]),
    "drrt_last_error": (C.c_char_p, []),
    "drrt_version": (C.c_char_p, []),
    "drrt_trace_f32": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _f, _f, _vp, _vp] + _tail),
    "drrt_trace_f16io": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _f, _f, _vp, _vp] + _tail),
    "drrt_backtrace_f16io": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _vp, _vp, _f, _f, _vp] + _tail),
    "drrt_trace_q16io": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _f, _f, _vp, _vp] + _tail),
    "drrt_backtrace_q16io": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _vp, _vp, _f, _f, _vp] + _tail),
    "drrt_q16_params": (_i, [_vp, _f, _vp]),
    "drrt_q16_encode": (_i, [_vp, _f, _sz, _vp, _vp, _vp, _vp, _vp]),
    "drrt_q16_decode": (_i, [_vp, _f, _sz, _vp, _vp, _vp, _vp, _vp]),
    "drrt_trace_pln_f32": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp] + _tail),
    "drrt_trace_target_f32": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp] + _tail),
    "drrt_trace_sdf_f32": (_i, [_vp, _vp, _ll, _vp, _sz, _vp, _vp, _f, _f, _vp, _vp] + _tail),
    "drrt_trace_cable_f32": (_i, [_vp, _sz, _f, _f, _sz, _vp, _vp, _vp, _f, _vp, _vp, _vp] + _tail),
    "drrt_backtrace_f32": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _vp, _vp, _f, _f, _vp] + _tail),
    "drrt_backtrace_chunk_state_bytes": (_sz, [_sz]),
    "drrt_backtrace_max_steps": (_i, [_vp, _f, _f]),
    "drrt_backtrace_chunk_f32": (_i, [_vp, _ll, _vp, _sz, _vp, _vp, _vp, _vp, _f, _f, _vp] + _tail + [_vp, _sz, _i, _i, _vp]),
    "drrt_backtrace_sdf_f32": (_i, [_vp, _vp, _ll, _vp, _sz, _vp, _vp, _vp, _vp, _f, _f, _vp] + _tail),
    "drrt_backtrace_cable_f32": (_i, [_vp, _sz, _f, _f, _sz, _vp, _vp, _vp, _vp, _f, _vp] + _tail),
    "drrt_sensor_splat_f32": (_i, [_sz, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _f, _vp, _u, _vp]),
    "drrt_sensor_splat_bwd_f32": (_i, [_sz, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "drrt_sensor_far_splat_f32": (_i, [_sz, _vp, _vp, _f, _vp, _vp, _i, _f, _vp, _u, _vp]),
    "drrt_sensor_far_splat_bwd_f32": (_i, [_sz, _vp, _vp, _f, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "drrt_sensor_tex_get_f32": (_i, [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp]),
    "drrt_sensor_tex_get_bwd_f32": (_i, [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "drrt_sensor_splat_dframe_f32": (_i, [_sz, _vp, _vp, _vp, _f, _vp, _i, _f, _vp, _u, _vp]),
    "drrt_sensor_splat_dframe_bwd_f32": (_i, [_sz, _vp, _vp, _vp, _f, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "drrt_sensor_far_splat_dframe_f32": (_i, [_sz, _vp, _vp, _f, _vp, _i, _f, _vp, _u, _vp]),
    "drrt_sensor_far_splat_dframe_bwd_f32": (_i, [_sz, _vp, _vp, _f, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "drrt_sensor_tex_get_dframe_f32": (_i, [_sz, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp]),
    "drrt_sensor_tex_get_dframe_bwd_f32": (_i, [_sz, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "drrt_rays_to_plane_f32": (_i, [_sz, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "drrt_rays_to_plane_bwd_f32": (_i, [_sz, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "drrt_upres_volume_f32": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "drrt_adam_step_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _d, _d, _d, _d, _d, _u, _vp]),
    "drrt_gen_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "drrt_gen_rays_f32": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _d, _d, _i, _i, _vp, _d, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "drrt_gen_cone_rays_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _d, _d, _vp, _d, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "drrt_last_order": (_vp, [_vp]),
    "drrt_set_order_hint": (None, [_vp, _sz]),
    "drrt_order_hint_pending": (_sz, []),
    "drrt_last_steps": (_vp, [_vp]),
    "drrt_set_step_hint": (None, [_vp, _sz]),
    "drrt_last_bundle_counters": (_vp, []),
    "drrt_ring_threshold_pct": (_i, []),
    "drrt_ring_long_threshold_permille": (_i, []),
    "drrt_ring_direct_threshold_pct": (_i, []),
    "drrt_profile_begin": (_i, [_i]),
    "drrt_profile_collect": (_i, [_vp, _vp, _i]),
    "drrt_profile_end": (None, []),
}

PROF_NAMES = {1: "trace", 2: "backtrace", 3: "sort", 4: "zero", 5: "quad"}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libdrrt_hip.so (after torch, so that both share ONE HIP runtime in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libdrrt_hip.so not found at {LIB_PATH}: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C adjointnonlinearraytracing_amd/csrc`.  There is no CPU fallback.")
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first; ours binds to the same SONAME)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch: loud
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def profile_collect(max_out: int = 4096):
    """-> list of (kernel name, ms) in launch order since the last collect (synchronises)."""
    ids = (C.c_int * max_out)()
    ms = (C.c_float * max_out)()
    n = load().drrt_profile_collect(ids, ms, max_out)
    return [(PROF_NAMES.get(ids[i], str(ids[i])), float(ms[i])) for i in range(n)]


def last_error() -> str:
    return load().drrt_last_error().decode()


def check(rc: int) -> None:
    """Negative status -> RuntimeError carrying the library's message (reference: pybind maps
    std::runtime_error to RuntimeError)."""
    if rc != 0:
        raise RuntimeError(last_error() or f"drrt_hip error {rc}")
