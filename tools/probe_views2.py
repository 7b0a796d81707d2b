#!/usr/bin/env python3
"""Window statistics of the adjoint for one plane view at a given angle (development probe; C ABI + debug counters)."""
import ctypes as C, json, sys
import torch
sys.path.insert(0, ".")
from adjointnonlinearraytracing_amd import _lib, source

lib = _lib.load()
dev = torch.device("cuda:0")
R = 256; span = 1.0; h = span / (R - 1); ds = h / 2
n = torch.ones(R, R, R, device=dev)
nvox = n.numel(); res = (C.c_int * 3)(R, R, R)
p = lambda t: C.c_void_p(t.data_ptr())
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for ang, pix, spp in ((0.0, 256, 4), (45.0, 256, 4), (45.0, 512, 4), (30.0, 256, 4), (45.0, 256, 16)):
    xs, vs, _ = source.plane_source3_rand(torch.tensor(ang), (pix, pix), spp, span, sensor_dist=0.2 * span, device=dev)
    xs = xs.contiguous(); vs = vs.contiguous(); nr = xs.shape[0]
    flags = _lib.FLAG_SORT_RAYS
    ws = torch.empty(int(lib.drrt_workspace_bytes_grid(nr, nvox, flags)) + 1024, dtype=torch.uint8, device=dev)
    xt, vt = torch.empty_like(xs), torch.empty_like(vs)
    st = torch.zeros(3, dtype=torch.int64, device=dev)
    _lib.check(lib.drrt_trace_f32(p(n), nvox, res, nr, p(xs), p(vs), h, ds, p(xt), p(vt), p(st), p(ws), ws.numel(), flags, stream))
    order = torch.empty(nr, dtype=torch.int32, device=dev)
    cnt = C.c_size_t(0)
    src = lib.drrt_last_order(C.byref(cnt))
    C.memmove  # noqa
    torch.cuda.synchronize()
    import ctypes
    # copy the order out of the workspace (device to device)
    o = (ctypes.c_char * 1)  # noqa
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(C.c_void_p(order.data_ptr()), C.c_void_p(src), C.c_size_t(nr * 4), 3)
    grad = torch.empty(nvox, dtype=torch.float32, device=dev)
    dx, dv = torch.ones_like(xt), torch.ones_like(vt)
    out = {"angle": ang, "pixels": pix, "spp": spp, "rays": nr}
    for name, fl in (("flat", flags | _lib.FLAG_DEBUG_COUNTERS), ("legacy", flags | _lib.FLAG_DEBUG_COUNTERS | _lib.FLAG_LEGACY_ADJOINT)):
        lib.drrt_set_order_hint(C.c_void_p(order.data_ptr()), nr)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.drrt_backtrace_f32(p(n), nvox, res, nr, p(xt), p(vt), p(dx), p(dv), h, ds, p(grad), p(st), p(ws), ws.numel(), fl, stream))
        b.record(); torch.cuda.synchronize()
        off = (ws.numel() - 512) & ~7
        dbg = ws[off:off + 512].view(torch.int64).cpu().tolist()
        out[name] = {"ms": round(a.elapsed_time(b), 3), "flushes": dbg[0], "lds_steps": dbg[1], "global_steps": dbg[2],
                     "ray_steps": int(st[0].item())}
    print(json.dumps(out))
