// drrt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the eikonal ray march and its adjoint,
// plus the C ABI declared in include/drrt_hip.h.
//
// Reference semantics: /root/reference/src/tracer.cpp (march loops :35-382, adjoint loops
// :384-567), src/volume.cpp, src/cylinder_volume.cpp.  Quirk numbers (Q1..Q16) refer to
// SURVEY.md section 8.1.
//
// Execution model (MI355X-first, not the reference's array-at-a-time enoki JIT):
//   * one ray per lane, wave64; the WHOLE march of a ray runs in registers inside one kernel
//     (the reference launches one fused kernel + one reduction + one host sync PER STEP and
//     streams x, v, xt, vt and masks through DRAM every step);
//   * per-ray termination: a ray stops as soon as it is flagged escaped.  This is exact for
//     trace / trace_plane / trace_sdf / backtrace*: an escaped ray flies straight outside the
//     convex box, can never produce another `cross`, and the adjoint masks every contribution
//     with `active` (proof sketch in DESIGN.md).  trace_target is the one variant whose result
//     depends on the GLOBAL loop count (its closest-approach update is not gated by `escaped`,
//     src/tracer.cpp:225-227), so it runs as two kernels around a device-side max reduction;
//   * rays are visited through an optional permutation (locality sort by entry voxel,
//     drrt_sort.hip) so that the 64 lanes of a wave touch a handful of 128-B lines per tap.
//
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/drrt_hip.h"
#include "drrt_device.h"

namespace drrt {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// stats: block-level reduction, then 3 atomics per block
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_stats(drrt_stats* stats, unsigned steps, unsigned failed) {
  if (!stats) return;
  __shared__ unsigned s_sum[kBlock / kWave], s_max[kBlock / kWave], s_fail[kBlock / kWave];
  unsigned ws = wave_sum_u32(steps), wm = wave_max_u32(steps), wf = wave_sum_u32(failed);
  int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) { s_sum[wid] = ws; s_max[wid] = wm; s_fail[wid] = wf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum = 0, fail = 0; unsigned mx = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) { sum += s_sum[w]; fail += s_fail[w]; mx = max(mx, s_max[w]); }
    if (sum)  atomicAdd(&stats->ray_steps, sum);
    if (fail) atomicAdd(&stats->n_failed, fail);
    if (mx)   atomicMax(&stats->iters, mx);
  }
}

struct Ray3 { float x, y, z; };
__device__ __forceinline__ Ray3 ld3(const float* p, size_t i) { return Ray3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(float* p, size_t i, float a, float b, float c) {
  p[3 * i] = a; p[3 * i + 1] = b; p[3 * i + 2] = c;
}

// ---------------------------------------------------------------------------------------------
// forward march: trace (MODE 0), trace_plane (MODE 1), trace_sdf (MODE 2)
// ---------------------------------------------------------------------------------------------
struct TraceArgs {
  Vol vol;
  const float* sdf;            // MODE 2
  const float* pos; const float* vel;
  const float* pln_o; const float* pln_d;   // MODE 1
  float* xt; float* vt; uint8_t* failmask;
  const uint32_t* perm;        // nullable: visit order
  drrt_stats* stats;
  size_t n;
  float ds;
  int max_steps;
};

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_trace(TraceArgs a) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned steps = 0, failed = 0;
  if (t < a.n) {
    const size_t i = a.perm ? (size_t)a.perm[t] : t;
    const Vol& V = a.vol;
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i);
    float x = p.x, y = p.y, z = p.z, vx = u.x, vy = u.y, vz = u.z;
    float xtx = x, xty = y, xtz = z, vtx = vx, vty = vy, vtz = vz;          // :56-57
    float pox = 0, poy = 0, poz = 0, pdx = 0, pdy = 0, pdz = 0;
    if (MODE == 1) {
      Ray3 o = ld3(a.pln_o, i), d = ld3(a.pln_d, i);
      pox = o.x; poy = o.y; poz = o.z; pdx = d.x; pdy = d.y; pdz = d.z;
    }
    bool inside = inbounds(V, x, y, z);                                    // :61
    bool esc = false;                                                       // :62
    bool act = true;
    if (MODE == 2) {                                                        // :276-277
      Cell c = locate(V, x, y, z);
      Sample s = interp<false>(fetch(a.sdf, c), c.wx, c.wy, c.wz);
      act = s.n < 0.f;
    }
    const float ds = a.ds, inv_h = V.inv_h;
    for (int it = 0; it < a.max_steps; ++it) {
      float n = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
      if (inside) {                                                         // masked gather (Q4)
        Cell c = locate(V, x, y, z);
        Sample s = interp<false>(fetch(V.data, c), c.wx, c.wy, c.wz);
        n = s.n; gx = s.gx * inv_h; gy = s.gy * inv_h; gz = s.gz * inv_h;
      }
      const float dsn = ds * n;
      vx = fmaf(dsn, gx, vx); vy = fmaf(dsn, gy, vy); vz = fmaf(dsn, gz, vz);     // :70
      x = fmaf(ds, vx, x); y = fmaf(ds, vy, y); z = fmaf(ds, vz, z);               // :71
      bool cur_inside;
      if (MODE == 2) {                                                      // :287-288
        float d = 0.f;
        if (inside) {
          Cell c = locate(V, x, y, z);
          d = interp<false>(fetch(a.sdf, c), c.wx, c.wy, c.wz).n;
        }
        cur_inside = d < 0.f;
      } else {
        cur_inside = inbounds(V, x, y, z);                                  // :73
        if (MODE == 1) {                                                    // :144-145
          float dot = (x - pox) * pdx + (y - poy) * pdy + (z - poz) * pdz;
          cur_inside = cur_inside & !(dot > 0.f);
        }
      }
      const bool cross = inside & !cur_inside;                              // :74
      esc = esc | cross | escaped(V, x, y, z, vx, vy, vz);                  // :75-76
      if (cross) { xtx = x; xty = y; xtz = z; vtx = vx; vty = vy; vtz = vz; }  // :79-80
      ++steps;
      if (esc) break;                                                       // per-ray form of :82
      inside = cur_inside;                                                  // :86
    }
    act = act & !esc;                                                       // :77
    if (MODE != 2 && !esc) { xtx = x; xty = y; xtz = z; }                   // :95 (vt stays, Q6)
    failed = act ? 1u : 0u;
    st3(a.xt, i, xtx, xty, xtz);
    st3(a.vt, i, vtx, vty, vtz);
    if (MODE == 1) a.failmask[i] = esc ? 0 : 1;                             // :171
  }
  block_stats(a.stats, steps, failed);
}

// ---------------------------------------------------------------------------------------------
// trace_target (src/tracer.cpp:174-242): phase A marches until escaped, phase B continues the
// (now straight) flight up to the global iteration count, tracking the closest approach.
// ---------------------------------------------------------------------------------------------
struct TargetArgs {
  Vol vol;
  const float* pos; const float* vel; const float* target;
  float* xt; float* vt; float* dist2;
  float* state;                // workspace: n * 7 floats (x,v,steps) -- SoA
  const uint32_t* perm;
  drrt_stats* stats;
  size_t n;
  float ds;
  int max_steps;
};

__global__ void __launch_bounds__(kBlock) k_target_a(TargetArgs a) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned steps = 0, failed = 0;
  if (t < a.n) {
    const size_t i = a.perm ? (size_t)a.perm[t] : t;
    const Vol& V = a.vol;
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i), tg = ld3(a.target, i);
    float x = p.x, y = p.y, z = p.z, vx = u.x, vy = u.y, vz = u.z;
    float xtx = x, xty = y, xtz = z, vtx = vx, vty = vy, vtz = vz;
    float ex = x - tg.x, ey = y - tg.y, ez = z - tg.z;
    float best = ex * ex + ey * ey + ez * ez;                               // :200
    bool inside = inbounds(V, x, y, z);
    bool esc = false;
    const float ds = a.ds, inv_h = V.inv_h;
    for (int it = 0; it < a.max_steps; ++it) {
      float n = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
      if (inside) {
        Cell c = locate(V, x, y, z);
        Sample s = interp<false>(fetch(V.data, c), c.wx, c.wy, c.wz);
        n = s.n; gx = s.gx * inv_h; gy = s.gy * inv_h; gz = s.gz * inv_h;
      }
      const float dsn = ds * n;
      vx = fmaf(dsn, gx, vx); vy = fmaf(dsn, gy, vy); vz = fmaf(dsn, gz, vz);
      x = fmaf(ds, vx, x); y = fmaf(ds, vy, y); z = fmaf(ds, vz, z);
      ex = x - tg.x; ey = y - tg.y; ez = z - tg.z;
      float cur = ex * ex + ey * ey + ez * ez;                              // :216
      bool cur_inside = inbounds(V, x, y, z);
      bool cross = inside & !cur_inside;
      esc = esc | cross | escaped(V, x, y, z, vx, vy, vz);
      if (cur < best) { xtx = x; xty = y; xtz = z; vtx = vx; vty = vy; vtz = vz; best = cur; } // :225-227
      ++steps;
      if (esc) break;
      inside = cur_inside;
    }
    failed = esc ? 0u : 1u;
    st3(a.xt, i, xtx, xty, xtz); st3(a.vt, i, vtx, vty, vtz); a.dist2[i] = best;
    float* s = a.state;
    s[0 * a.n + i] = x;  s[1 * a.n + i] = y;  s[2 * a.n + i] = z;
    s[3 * a.n + i] = vx; s[4 * a.n + i] = vy; s[5 * a.n + i] = vz;
    s[6 * a.n + i] = __uint_as_float(steps);
  }
  block_stats(a.stats, steps, failed);
}

__global__ void __launch_bounds__(kBlock) k_target_b(TargetArgs a) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.n) return;
  const unsigned total = a.stats->iters;         // written by phase A (stream-ordered)
  const float* s = a.state;
  unsigned done = __float_as_uint(s[6 * a.n + i]);
  if (done >= total) return;
  float x = s[0 * a.n + i], y = s[1 * a.n + i], z = s[2 * a.n + i];
  const float vx = s[3 * a.n + i], vy = s[4 * a.n + i], vz = s[5 * a.n + i];
  Ray3 tg = ld3(a.target, i);
  float best = a.dist2[i];
  float bx = 0, by = 0, bz = 0; bool upd = false;
  const float ds = a.ds;
  for (unsigned k = done; k < total; ++k) {      // escaped ray: masked gathers => straight flight
    x = fmaf(ds, vx, x); y = fmaf(ds, vy, y); z = fmaf(ds, vz, z);
    float ex = x - tg.x, ey = y - tg.y, ez = z - tg.z;
    float cur = ex * ex + ey * ey + ez * ez;
    if (cur < best) { best = cur; bx = x; by = y; bz = z; upd = true; }
  }
  if (upd) { st3(a.xt, i, bx, by, bz); st3(a.vt, i, vx, vy, vz); a.dist2[i] = best; }
}

// ---------------------------------------------------------------------------------------------
// adjoint march: backtrace (MODE 0), backtrace_sdf (MODE 1); direct global atomics variant
// ---------------------------------------------------------------------------------------------
struct BackArgs {
  Vol vol;
  const float* sdf;
  const float* xt; const float* vt; const float* dx; const float* dv;
  float* grad;
  const uint32_t* perm;
  drrt_stats* stats;
  size_t n;
  float ds;
  float grad_scale;            // 1 (as written, Q3) or 1/h (DRRT_FLAG_CORRECTED_H)
  int max_steps;
};

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_backtrace_direct(BackArgs a) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned steps = 0;
  if (t < a.n) {
    const size_t i = a.perm ? (size_t)a.perm[t] : t;
    const Vol& V = a.vol;
    Ray3 p = ld3(a.xt, i), u = ld3(a.vt, i), gxv = ld3(a.dx, i), gvv = ld3(a.dv, i);
    float x = p.x, y = p.y, z = p.z, vx = u.x, vy = u.y, vz = u.z;
    const float ds = a.ds, inv_h = V.inv_h, inv_h2 = inv_h * inv_h;
    float lx = gxv.x, ly = gxv.y, lz = gxv.z;                               // :409
    float mx = fmaf(ds, gxv.x, gvv.x), my = fmaf(ds, gxv.y, gvv.y), mz = fmaf(ds, gxv.z, gvv.z); // :410
    bool active = !escaped(V, x, y, z, -vx, -vy, -vz);                      // :413-414
    bool outside = false;
    if (MODE == 1 && active) {                                              // :476-477
      Cell c = locate(V, x, y, z);
      outside = interp<false>(fetch(a.sdf, c), c.wx, c.wy, c.wz).n >= 0.f;
    }
    for (int it = 0; it < a.max_steps && active; ++it) {
      x = fmaf(-ds, vx, x); y = fmaf(-ds, vy, y); z = fmaf(-ds, vz, z);     // :420
      Cell c = locate(V, x, y, z);
      Sample s = interp<true>(fetch(V.data, c), c.wx, c.wy, c.wz);          // :421-422 (one fetch)
      const float n = s.n, gx = s.gx * inv_h, gy = s.gy * inv_h, gz = s.gz * inv_h;
      const float mdsn = -ds * n;
      vx = fmaf(mdsn, gx, vx); vy = fmaf(mdsn, gy, vy); vz = fmaf(mdsn, gz, vz);   // :423
      active = !escaped(V, x, y, z, -vx, -vy, -vz);                         // :425
      if (MODE == 1) {                                                      // :488-497
        bool now_out = interp<false>(fetch(a.sdf, c), c.wx, c.wy, c.wz).n >= 0.f;
        active = active & !((!outside) & now_out);
        outside = now_out;
      }
      if (!active) break;                                                   // :426-428
      ++steps;
      const float dn = mx * gx + my * gy + mz * gz;                         // :430
      const float nds = n * ds * a.grad_scale;
      Corners w = splat_weights(c.wx, c.wy, c.wz, dn * ds, nds * mx, nds * my, nds * mz); // :431-432
      float* g = a.grad + c.base;
      atomic_add_f32(g, w.c000);                    atomic_add_f32(g + c.ox, w.c100);
      atomic_add_f32(g + c.oy, w.c010);             atomic_add_f32(g + c.oy + c.ox, w.c110);
      atomic_add_f32(g + c.oz, w.c001);             atomic_add_f32(g + c.oz + c.ox, w.c101);
      atomic_add_f32(g + c.oz + c.oy, w.c011);      atomic_add_f32(g + c.oz + c.oy + c.ox, w.c111);
      // la += ds*(dn*grad n + n*H*mu), H = mixed partials / h^2 with zero diagonal (:434, Q10)
      const float hxy = s.hxy * inv_h2, hxz = s.hxz * inv_h2, hyz = s.hyz * inv_h2;
      const float hmx = hxy * my + hxz * mz, hmy = hxy * mx + hyz * mz, hmz = hxz * mx + hyz * my;
      lx = fmaf(ds, fmaf(dn, gx, n * hmx), lx);
      ly = fmaf(ds, fmaf(dn, gy, n * hmy), ly);
      lz = fmaf(ds, fmaf(dn, gz, n * hmz), lz);
      mx = fmaf(ds, lx, mx); my = fmaf(ds, ly, my); mz = fmaf(ds, lz, mz);  // :435
    }
  }
  block_stats(a.stats, steps, 0u);
}

// ---------------------------------------------------------------------------------------------
// cable (radial profile) variants, src/tracer.cpp:312-382 and :511-567
// The profile (<= a few hundred floats) lives in LDS; the adjoint accumulates into an LDS copy
// of the gradient profile (ds_add_f32) and flushes it once per block -- millions of rays would
// otherwise hammer <= 257 global addresses.
// ---------------------------------------------------------------------------------------------
constexpr int kCableMaxRes = 4096;   // profiles larger than this fall back to global memory

struct CableArgs {
  const float* rif; int rres; float radius, length, ds; int max_steps;
  const float* pos; const float* vel; const float* target;    // forward
  const float* dx; const float* dv;                           // adjoint (pos=xt, vel=vt)
  float* xt; float* vt; float* dist2; float* grad;
  drrt_stats* stats;
  size_t n;
};

__device__ __forceinline__ Cyl make_cyl(const CableArgs& a, const float* data) {
  Cyl C; C.data = data; C.rres = a.rres; C.radius = a.radius; C.length = a.length;
  C.h = a.radius / (float)(a.rres - 1); C.inv_h = 1.f / C.h; C.r2 = a.radius * a.radius;
  return C;
}

__global__ void __launch_bounds__(kBlock) k_trace_cable(CableArgs a) {
  extern __shared__ float s_prof[];
  const bool use_lds = a.rres <= kCableMaxRes;
  if (use_lds) {
    for (int k = threadIdx.x; k < a.rres; k += kBlock) s_prof[k] = a.rif[k];
    __syncthreads();
  }
  const Cyl C = make_cyl(a, use_lds ? s_prof : a.rif);
  unsigned steps_tot = 0, fail_tot = 0, steps_max = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (size_t)gridDim.x * kBlock) {
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i), tg = ld3(a.target, i);
    float x = p.x, y = p.y, z = p.z, vx = u.x, vy = u.y, vz = u.z;
    float xtx = x, xty = y, xtz = z, vtx = vx, vty = vy, vtz = vz;
    float ex = x - tg.x, ey = y - tg.y, ez = z - tg.z;
    float best = ex * ex + ey * ey + ez * ez;                               // :340
    bool inside = cyl_inbounds(C, x, y, z);                                 // :344
    bool esc = false;
    unsigned steps = 0;
    for (int it = 0; it < a.max_steps; ++it) {
      CylCell c = cyl_locate(C, x, z);                                      // :351 (unmasked gather)
      float v0 = C.data[c.i0], v1 = C.data[c.i1];
      float f = v0 * (1.f - c.w0) + v1 * c.w0;                              // :53
      float rx = (v1 - v0) * C.inv_h;                                       // :54
      float dsn = a.ds * f;
      vx = fmaf(dsn, rx * c.rhx, vx); vz = fmaf(dsn, rx * c.rhz, vz);       // :353 (y comp of grad = 0)
      x = fmaf(a.ds, vx, x); y = fmaf(a.ds, vy, y); z = fmaf(a.ds, vz, z);  // :354
      ex = x - tg.x; ey = y - tg.y; ez = z - tg.z;
      float cur = ex * ex + ey * ey + ez * ez;                              // :356
      bool cur_inside = cyl_inbounds(C, x, y, z);
      bool cross = inside & !cur_inside;
      esc = esc | cross | cyl_escaped(C, x, y, z, vx, vy, vz);              // :361-362
      if (cur < best) { xtx = x; xty = y; xtz = z; vtx = vx; vty = vy; vtz = vz; best = cur; } // :365-367
      ++steps;
      if (esc) break;          // state is frozen once !active (:353-354), so nothing changes later
      inside = cur_inside;
    }
    st3(a.xt, i, xtx, xty, xtz); st3(a.vt, i, vtx, vty, vtz); a.dist2[i] = best;
    steps_tot += steps; steps_max = max(steps_max, steps); fail_tot += esc ? 0u : 1u;
  }
  // block_stats takes (sum, fail) per thread and a max of the same quantity; feed the max separately
  if (a.stats) {
    unsigned wm = wave_max_u32(steps_max);
    if ((threadIdx.x & (kWave - 1)) == 0 && wm) atomicMax(&a.stats->iters, wm);
    unsigned ws = wave_sum_u32(steps_tot), wf = wave_sum_u32(fail_tot);
    if ((threadIdx.x & (kWave - 1)) == 0) {
      if (ws) atomicAdd(&a.stats->ray_steps, (unsigned long long)ws);
      if (wf) atomicAdd(&a.stats->n_failed, (unsigned long long)wf);
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_backtrace_cable(CableArgs a) {
  extern __shared__ float s_mem[];
  const bool use_lds = a.rres <= kCableMaxRes;
  float* s_prof = s_mem;
  float* s_grad = s_mem + (use_lds ? a.rres : 0);
  if (use_lds) {
    for (int k = threadIdx.x; k < a.rres; k += kBlock) { s_prof[k] = a.rif[k]; s_grad[k] = 0.f; }
    __syncthreads();
  }
  const Cyl C = make_cyl(a, use_lds ? s_prof : a.rif);
  float* acc = use_lds ? s_grad : a.grad;
  unsigned steps_tot = 0, steps_max = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (size_t)gridDim.x * kBlock) {
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i), gxv = ld3(a.dx, i), gvv = ld3(a.dv, i);
    float x = p.x, y = p.y, z = p.z, vx = u.x, vy = u.y, vz = u.z;
    const float ds = a.ds;
    float lx = gxv.x, ly = gxv.y, lz = gxv.z;                               // :536
    float mx = fmaf(ds, gxv.x, gvv.x), my = fmaf(ds, gxv.y, gvv.y), mz = fmaf(ds, gxv.z, gvv.z); // :537
    bool active = !cyl_escaped(C, x, y, z, -vx, -vy, -vz);                  // :540-541
    unsigned steps = 0;
    for (int it = 0; it < a.max_steps && active; ++it) {
      x = fmaf(-ds, vx, x); y = fmaf(-ds, vy, y); z = fmaf(-ds, vz, z);     // :547
      CylCell c = cyl_locate(C, x, z);
      float v0 = C.data[c.i0], v1 = C.data[c.i1];
      float w0 = c.w0, w1 = 1.f - c.w0;
      float n = v0 * w1 + v1 * w0;                                          // :53
      float rx = (v1 - v0) * C.inv_h;                                       // :54 / :88
      float gx = rx * c.rhx, gz = rx * c.rhz;                               // grad n (y comp 0)
      float mdsn = -ds * n;
      vx = fmaf(mdsn, gx, vx); vz = fmaf(mdsn, gz, vz);                     // :550
      active = !cyl_escaped(C, x, y, z, -vx, -vy, -vz);                     // :552
      if (!active) break;
      ++steps;
      float dn = mx * gx + mz * gz;                                         // :557
      // cylinder_volume::splat (:113-148): value taps val*w, gradient taps -+(grad.rhat)/h
      float val = dn * ds;
      float gv = (n * ds) * (mx * c.rhx + mz * c.rhz);                      // dot(dnx*ds, rhat), 0 if tiny
      float a0 = fmaf(val, w1, -gv * C.inv_h), a1 = fmaf(val, w0, gv * C.inv_h);
      if (use_lds) { atomicAdd(&acc[c.i0], a0); atomicAdd(&acc[c.i1], a1); }
      else { atomic_add_f32(&acc[c.i0], a0); atomic_add_f32(&acc[c.i1], a1); }
      // Hessian (:88-108): (I - rhat rhat^T)_{xz} * (n'/r), zero when r < eps
      float sH = c.tiny ? 0.f : rx / c.r;
      float h00 = (1.f - c.rhx * c.rhx) * sH, h02 = -(c.rhx * c.rhz) * sH, h22 = (1.f - c.rhz * c.rhz) * sH;
      float hmx = h00 * mx + h02 * mz, hmz = h02 * mx + h22 * mz;
      lx = fmaf(ds, fmaf(dn, gx, n * hmx), lx);                             // :561
      lz = fmaf(ds, fmaf(dn, gz, n * hmz), lz);
      mx = fmaf(ds, lx, mx); my = fmaf(ds, ly, my); mz = fmaf(ds, lz, mz);  // :562
    }
    steps_tot += steps; steps_max = max(steps_max, steps);
  }
  if (use_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < a.rres; k += kBlock) {
      float g = s_grad[k];
      if (g != 0.f) atomic_add_f32(&a.grad[k], g);
    }
  }
  if (a.stats) {
    unsigned wm = wave_max_u32(steps_max);
    unsigned ws = wave_sum_u32(steps_tot);
    if ((threadIdx.x & (kWave - 1)) == 0) {
      if (wm) atomicMax(&a.stats->iters, wm);
      if (ws) atomicAdd(&a.stats->ray_steps, (unsigned long long)ws);
    }
  }
}

}  // namespace drrt

// =============================================================================================
// host side: C ABI
// =============================================================================================
using namespace drrt;

// from drrt_sort.hip
namespace drrt {
size_t sort_workspace_bytes(size_t n);
hipError_t sort_rays_by_entry_voxel(const Vol& V, float h, size_t n, const float* pos, const float* vel,
                                    float dir_sign, void* ws, size_t ws_bytes, const uint32_t** perm_out,
                                    hipStream_t stream);
}

static thread_local char g_err[512] = "";

static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
static int fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return DRRT_ERR_HIP;
}

extern "C" const char* drrt_last_error(void) { return g_err; }
extern "C" const char* drrt_version(void) { return "drrt_hip 0.1 gfx950"; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t drrt_workspace_bytes(size_t n, unsigned flags) {
  size_t b = 0;
  if (flags & DRRT_FLAG_SORT_RAYS) b += align_up(sort_workspace_bytes(n), 256);
  b += align_up(n * 7 * sizeof(float), 256);       // trace_target state (cheap; always counted)
  return b;
}

// volume ctor checks: src/volume.cpp:31-38 (size) and :123-124 (width/height >= 2)
static int make_vol(const float* rif, long long nvox, const int res[3], float h, Vol* V) {
  if (!rif || !res) return fail(DRRT_ERR_ARG, "null rif/res pointer");
  if ((long long)res[0] * res[1] * res[2] != nvox || nvox <= 0)
    return fail(DRRT_ERR_RES_MISMATCH, "Resolution doesn't match data");
  if (!(res[0] == 1 && res[1] == 1 && res[2] == 1) && (res[0] < 2 || res[1] < 2))
    return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  if (nvox > 0x7fffffffLL) return fail(DRRT_ERR_ARG, "grid too large for int32 indexing");
  V->data = rif; V->W = res[0]; V->H = res[1]; V->D = res[2];
  V->sy = res[0]; V->sz = res[0] * res[1];
  V->inv_h = 1.0f / h;
  V->bx = (float)(res[0] - 1) * h; V->by = (float)(res[1] - 1) * h; V->bz = (float)(res[2] - 1) * h;
  return DRRT_OK;
}

static inline int max3(const int r[3]) { return r[0] > r[1] ? (r[0] > r[2] ? r[0] : r[2]) : (r[1] > r[2] ? r[1] : r[2]); }

// float expression truncated to int, exactly as written in the reference (Q5)
static inline int steps_fwd(float h, const int res[3], float ds)  { return (int)(4.0f * h * (float)max3(res) / ds); }
static inline int steps_sdf(float h, const int res[3], float ds)  { return (int)(2.0f * h * (float)max3(res) / ds); }
static inline int steps_adj(float h, const int res[3], float ds)  { return (int)(2.0f * h * (float)max3(res) / ds); }

static int zero_stats(drrt_stats* stats, hipStream_t s) {
  if (!stats) return DRRT_OK;
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(drrt_stats), s);
  return e == hipSuccess ? DRRT_OK : fail_hip(e, "hipMemsetAsync(stats)");
}

static int maybe_sort(const Vol& V, float h, size_t n, const float* pos, const float* vel, float dir_sign,
                      unsigned flags, void* ws, size_t ws_bytes, const uint32_t** perm, hipStream_t s) {
  *perm = nullptr;
  if (!(flags & DRRT_FLAG_SORT_RAYS) || n < 2) return DRRT_OK;
  if (!ws || ws_bytes < sort_workspace_bytes(n)) return fail(DRRT_ERR_ARG, "workspace too small for DRRT_FLAG_SORT_RAYS");
  hipError_t e = sort_rays_by_entry_voxel(V, h, n, pos, vel, dir_sign, ws, ws_bytes, perm, s);
  return e == hipSuccess ? DRRT_OK : fail_hip(e, "sort_rays_by_entry_voxel");
}

#define LAUNCH_CHECK(where)                                              \
  do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail_hip(e_, where); } while (0)

static inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

template <int MODE>
static int run_trace(const float* rif, const float* sdf, long long nvox, const int res[3], size_t n,
                     const float* pos, const float* vel, const float* pln_o, const float* pln_d,
                     float h, float ds, float* xt, float* vt, uint8_t* failmask, drrt_stats* stats,
                     void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  TraceArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  if (n == 0) return zero_stats(stats, s);
  if (!pos || !vel || !xt || !vt) return fail(DRRT_ERR_ARG, "null ray pointer");
  if (MODE == 1 && (!pln_o || !pln_d || !failmask)) return fail(DRRT_ERR_ARG, "null plane/failmask pointer");
  if (MODE == 2 && !sdf) return fail(DRRT_ERR_ARG, "null sdf pointer");
  if (n > 0xffffffffULL) return fail(DRRT_ERR_ARG, "too many rays for uint32 permutation");
  rc = zero_stats(stats, s); if (rc) return rc;
  rc = maybe_sort(a.vol, h, n, pos, vel, 1.f, flags, ws, ws_bytes, &a.perm, s); if (rc) return rc;
  a.sdf = sdf; a.pos = pos; a.vel = vel; a.pln_o = pln_o; a.pln_d = pln_d;
  a.xt = xt; a.vt = vt; a.failmask = failmask; a.stats = stats; a.n = n; a.ds = ds;
  a.max_steps = (MODE == 2) ? steps_sdf(h, res, ds) : steps_fwd(h, res, ds);
  hipLaunchKernelGGL(k_trace<MODE>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  LAUNCH_CHECK("k_trace");
  return DRRT_OK;
}

extern "C" int drrt_trace_f32(const float* rif, long long nvox, const int res[3], size_t n,
                              const float* pos, const float* vel, float h, float ds, float* xt, float* vt,
                              drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_trace<0>(rif, nullptr, nvox, res, n, pos, vel, nullptr, nullptr, h, ds, xt, vt, nullptr,
                      stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_trace_pln_f32(const float* rif, long long nvox, const int res[3], size_t n,
                                  const float* pos, const float* vel, const float* pln_o, const float* pln_d,
                                  float h, float ds, float* xt, float* vt, uint8_t* failmask,
                                  drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_trace<1>(rif, nullptr, nvox, res, n, pos, vel, pln_o, pln_d, h, ds, xt, vt, failmask,
                      stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_trace_sdf_f32(const float* rif, const float* sdf, long long nvox, const int res[3],
                                  size_t n, const float* pos, const float* vel, float h, float ds,
                                  float* xt, float* vt, drrt_stats* stats, void* ws, size_t ws_bytes,
                                  unsigned flags, void* stream) {
  return run_trace<2>(rif, sdf, nvox, res, n, pos, vel, nullptr, nullptr, h, ds, xt, vt, nullptr,
                      stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_trace_target_f32(const float* rif, long long nvox, const int res[3], size_t n,
                                     const float* pos, const float* vel, const float* target,
                                     float h, float ds, float* xt, float* vt, float* dist2,
                                     drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags,
                                     void* stream) {
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  TargetArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  if (n == 0) return zero_stats(stats, s);
  if (!pos || !vel || !target || !xt || !vt || !dist2) return fail(DRRT_ERR_ARG, "null ray pointer");
  if (n > 0xffffffffULL) return fail(DRRT_ERR_ARG, "too many rays for uint32 permutation");
  // the global iteration count lives in stats->iters: a stats block is mandatory here
  if (!stats) return fail(DRRT_ERR_ARG, "trace_target needs a stats block (global loop count)");
  const size_t state_bytes = align_up(n * 7 * sizeof(float), 256);
  const size_t sort_bytes = (flags & DRRT_FLAG_SORT_RAYS) ? align_up(sort_workspace_bytes(n), 256) : 0;
  if (!ws || ws_bytes < state_bytes + sort_bytes) return fail(DRRT_ERR_ARG, "workspace too small for trace_target");
  rc = zero_stats(stats, s); if (rc) return rc;
  a.state = (float*)ws;
  rc = maybe_sort(a.vol, h, n, pos, vel, 1.f, flags, (char*)ws + state_bytes, ws_bytes - state_bytes, &a.perm, s);
  if (rc) return rc;
  a.pos = pos; a.vel = vel; a.target = target; a.xt = xt; a.vt = vt; a.dist2 = dist2;
  a.stats = stats; a.n = n; a.ds = ds; a.max_steps = steps_fwd(h, res, ds);
  hipLaunchKernelGGL(k_target_a, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  LAUNCH_CHECK("k_target_a");
  hipLaunchKernelGGL(k_target_b, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  LAUNCH_CHECK("k_target_b");
  return DRRT_OK;
}

template <int MODE>
static int run_backtrace(const float* rif, const float* sdf, long long nvox, const int res[3], size_t n,
                         const float* xt, const float* vt, const float* dx, const float* dv,
                         float h, float ds, float* grad, drrt_stats* stats, void* ws, size_t ws_bytes,
                         unsigned flags, void* stream) {
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  BackArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  if (!grad) return fail(DRRT_ERR_ARG, "null grad pointer");
  if (MODE == 1 && !sdf) return fail(DRRT_ERR_ARG, "null sdf pointer");
  if (!(flags & DRRT_FLAG_NO_ZERO)) {                                        // src/tracer.cpp:401-403
    hipError_t e = hipMemsetAsync(grad, 0, (size_t)nvox * sizeof(float), s);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(grad)");
  }
  rc = zero_stats(stats, s); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!xt || !vt || !dx || !dv) return fail(DRRT_ERR_ARG, "null ray pointer");
  if (n > 0xffffffffULL) return fail(DRRT_ERR_ARG, "too many rays for uint32 permutation");
  rc = maybe_sort(a.vol, h, n, xt, vt, -1.f, flags, ws, ws_bytes, &a.perm, s); if (rc) return rc;
  a.sdf = sdf; a.xt = xt; a.vt = vt; a.dx = dx; a.dv = dv; a.grad = grad; a.stats = stats;
  a.n = n; a.ds = ds; a.max_steps = steps_adj(h, res, ds);
  a.grad_scale = (flags & DRRT_FLAG_CORRECTED_H) ? a.vol.inv_h : 1.0f;
  hipLaunchKernelGGL(k_backtrace_direct<MODE>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  LAUNCH_CHECK("k_backtrace");
  return DRRT_OK;
}

extern "C" int drrt_backtrace_f32(const float* rif, long long nvox, const int res[3], size_t n,
                                  const float* xt, const float* vt, const float* dx, const float* dv,
                                  float h, float ds, float* grad, drrt_stats* stats, void* ws,
                                  size_t ws_bytes, unsigned flags, void* stream) {
  return run_backtrace<0>(rif, nullptr, nvox, res, n, xt, vt, dx, dv, h, ds, grad, stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_backtrace_sdf_f32(const float* rif, const float* sdf, long long nvox, const int res[3],
                                      size_t n, const float* xt, const float* vt, const float* dx,
                                      const float* dv, float h, float ds, float* grad, drrt_stats* stats,
                                      void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_backtrace<1>(rif, sdf, nvox, res, n, xt, vt, dx, dv, h, ds, grad, stats, ws, ws_bytes, flags, stream);
}

static unsigned cable_grid(size_t n) {
  // grid-stride: at most 4 blocks per CU so that the per-block LDS gradient flush stays small
  unsigned want = grid_for(n);
  return want < 1024u ? want : 1024u;
}

extern "C" int drrt_trace_cable_f32(const float* rif, size_t rres, float radius, float length, size_t n,
                                    const float* pos, const float* vel, const float* target, float ds,
                                    float* xt, float* vt, float* dist2, drrt_stats* stats, void* ws,
                                    size_t ws_bytes, unsigned flags, void* stream) {
  (void)ws; (void)ws_bytes; (void)flags;
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  if (!rif) return fail(DRRT_ERR_ARG, "null rif pointer");
  if (rres < 2 || rres > 0x7fffffffULL) return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  int rc = zero_stats(stats, s); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!pos || !vel || !target || !xt || !vt || !dist2) return fail(DRRT_ERR_ARG, "null ray pointer");
  CableArgs a{};
  a.rif = rif; a.rres = (int)rres; a.radius = radius; a.length = length; a.ds = ds;
  a.max_steps = (int)(4.0f * length / ds);                                   // src/tracer.cpp:332
  a.pos = pos; a.vel = vel; a.target = target; a.xt = xt; a.vt = vt; a.dist2 = dist2;
  a.stats = stats; a.n = n;
  size_t lds = (a.rres <= kCableMaxRes) ? a.rres * sizeof(float) : 0;
  hipLaunchKernelGGL(k_trace_cable, dim3(cable_grid(n)), dim3(kBlock), lds, s, a);
  LAUNCH_CHECK("k_trace_cable");
  return DRRT_OK;
}

extern "C" int drrt_backtrace_cable_f32(const float* rif, size_t rres, float radius, float length, size_t n,
                                        const float* xt, const float* vt, const float* dx, const float* dv,
                                        float ds, float* grad, drrt_stats* stats, void* ws, size_t ws_bytes,
                                        unsigned flags, void* stream) {
  (void)ws; (void)ws_bytes;
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  if (!rif || !grad) return fail(DRRT_ERR_ARG, "null rif/grad pointer");
  if (rres < 2 || rres > 0x7fffffffULL) return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  if (!(flags & DRRT_FLAG_NO_ZERO)) {                                        // src/tracer.cpp:528-530
    hipError_t e = hipMemsetAsync(grad, 0, rres * sizeof(float), s);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(grad)");
  }
  int rc = zero_stats(stats, s); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!xt || !vt || !dx || !dv) return fail(DRRT_ERR_ARG, "null ray pointer");
  CableArgs a{};
  a.rif = rif; a.rres = (int)rres; a.radius = radius; a.length = length; a.ds = ds;
  a.max_steps = (int)(4.0f * length / ds);                                   // src/tracer.cpp:544
  a.pos = xt; a.vel = vt; a.dx = dx; a.dv = dv; a.grad = grad; a.stats = stats; a.n = n;
  size_t lds = (a.rres <= kCableMaxRes) ? 2 * a.rres * sizeof(float) : 0;
  hipLaunchKernelGGL(k_backtrace_cable, dim3(cable_grid(n)), dim3(kBlock), lds, s, a);
  LAUNCH_CHECK("k_backtrace_cable");
  return DRRT_OK;
}
