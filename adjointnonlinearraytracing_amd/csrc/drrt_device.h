// drrt_device.h -- device-side building blocks of the gfx950 eikonal ray-march kernels.
//
// What the reference computes (file:line = /root/reference/...):
//   volume::eval_grad  src/volume.cpp:101-181   trilinear n(p) and grad n(p)
//   volume::eval_hess  src/volume.cpp:40-99     mixed second partials (zero diagonal)
//   volume::splat      src/volume.cpp:182-244   adjoint of eval_grad w.r.t. the voxels
//   volume::inbounds / escaped  src/volume.cpp:246-271
//   cylinder_volume::*  src/cylinder_volume.cpp:26-170
//
// How it is computed here: the 8 taps are fetched once per ray-step and n, grad n and the
// three mixed partials are all derived from ONE factored lerp tree (differences first,
// ~27 flops instead of the ~150 of the expanded weight products); the 16 scatter_adds of
// volume::splat are fused into 8 corner contributions.  Differences-first also avoids the
// cancellation of the reference's (sum of 4 products) - (sum of 4 products) form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace drrt {

constexpr int kWave = 64;

struct Vol {
  const float* data;
  int W, H, D;          // res[0], res[1], res[2]  (x, y, z extents; x is memory-contiguous)
  int sy, sz;           // element strides of y and z: W, W*H
  float inv_h;          // 1/h   (reference: rcp(h_), src/volume.cpp:128)
  float bx, by, bz;     // (res-1)*h : upper bounds used by inbounds/escaped (src/volume.cpp:252-254)
};

struct Cell {
  int base;             // flat index of corner 000
  int ox, oy, oz;       // element offsets to the +x, +y, +z neighbours (0 where clamped)
  float wx, wy, wz;     // fractional weights w0 = pm - floor(pm)  (unclamped, Q11)
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// src/volume.cpp:128-141: pm = p*rcp(h); pos = floor2int(pm); w0 = pm - pos; clamp indices.
__device__ __forceinline__ Cell locate(const Vol& V, float px, float py, float pz) {
  Cell c;
  float fx = px * V.inv_h, fy = py * V.inv_h, fz = pz * V.inv_h;
  float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
  c.wx = fx - flx; c.wy = fy - fly; c.wz = fz - flz;
  int ix = (int)flx, iy = (int)fly, iz = (int)flz;       // v_cvt_i32_f32 saturates, NaN -> 0
  int x0 = clampi(ix, 0, V.W - 1), x1 = clampi(ix + 1, 0, V.W - 1);
  int y0 = clampi(iy, 0, V.H - 1), y1 = clampi(iy + 1, 0, V.H - 1);
  int z0 = clampi(iz, 0, V.D - 1), z1 = clampi(iz + 1, 0, V.D - 1);
  c.base = z0 * V.sz + y0 * V.sy + x0;
  c.ox = x1 - x0; c.oy = (y1 - y0) * V.sy; c.oz = (z1 - z0) * V.sz;
  return c;
}

struct Taps { float v000, v100, v010, v110, v001, v101, v011, v111; };

__device__ __forceinline__ Taps fetch(const float* __restrict__ d, const Cell& c) {
  Taps t;
  const float* p = d + c.base;
  t.v000 = p[0];            t.v100 = p[c.ox];
  t.v010 = p[c.oy];         t.v110 = p[c.oy + c.ox];
  t.v001 = p[c.oz];         t.v101 = p[c.oz + c.ox];
  t.v011 = p[c.oz + c.oy];  t.v111 = p[c.oz + c.oy + c.ox];
  return t;
}

// n and RAW gradient / mixed partials (not yet divided by h / h^2).
struct Sample { float n, gx, gy, gz, hxy, hxz, hyz; };

template <bool WITH_HESS>
__device__ __forceinline__ Sample interp(const Taps& t, float wx, float wy, float wz) {
  Sample s;
  // x-differences at the four (y,z) edges
  float d00 = t.v100 - t.v000, d10 = t.v110 - t.v010, d01 = t.v101 - t.v001, d11 = t.v111 - t.v011;
  // x-lerped values
  float c00 = fmaf(wx, d00, t.v000), c10 = fmaf(wx, d10, t.v010);
  float c01 = fmaf(wx, d01, t.v001), c11 = fmaf(wx, d11, t.v011);
  float e0 = c10 - c00, e1 = c11 - c01;                  // d/dy at z0, z1
  float l0 = fmaf(wy, e0, c00), l1 = fmaf(wy, e1, c01);  // (x,y)-lerped at z0, z1
  float dz = l1 - l0;
  s.gz = dz;
  s.n  = fmaf(wz, dz, l0);
  float eyz = e1 - e0;
  s.gy = fmaf(wz, eyz, e0);
  float dxy0 = d10 - d00, dxy1 = d11 - d01;              // d2/dxdy at z0, z1
  float gx0 = fmaf(wy, dxy0, d00), gx1 = fmaf(wy, dxy1, d01);
  float gxz = gx1 - gx0;
  s.gx = fmaf(wz, gxz, gx0);
  if (WITH_HESS) {
    s.hxy = fmaf(wz, dxy1 - dxy0, dxy0);                 // src/volume.cpp:79-81
    s.hxz = gxz;                                         // src/volume.cpp:82-84
    s.hyz = eyz;                                         // src/volume.cpp:85-87
  } else {
    s.hxy = s.hxz = s.hyz = 0.f;
  }
  return s;
}

// src/volume.cpp:246-256
__device__ __forceinline__ bool inbounds(const Vol& V, float px, float py, float pz) {
  return (px >= 0.f) & (py >= 0.f) & (pz >= 0.f) & (px < V.bx) & (py < V.by) & (pz < V.bz);
}
// src/volume.cpp:258-271
__device__ __forceinline__ bool escaped(const Vol& V, float px, float py, float pz,
                                        float vx, float vy, float vz) {
  bool ex = ((px < 0.f) & (vx < 0.f)) | ((px >= V.bx) & (vx > 0.f));
  bool ey = ((py < 0.f) & (vy < 0.f)) | ((py >= V.by) & (vy > 0.f));
  bool ez = ((pz < 0.f) & (vz < 0.f)) | ((pz >= V.bz) & (vz > 0.f));
  return ex | ey | ez;
}

// Fused volume::splat (src/volume.cpp:217-243): contribution of (val, grad) to the 8 corners.
//   corner(a,b,c) = val*X_a*Y_b*Z_c  +  gx*(+-)Y_b*Z_c + gy*(+-)X_a*Z_c + gz*(+-)X_a*Y_b
// with X_0 = 1-wx, X_1 = wx, sign = - for index 0, + for index 1.
struct Corners { float c000, c100, c010, c110, c001, c101, c011, c111; };

__device__ __forceinline__ Corners splat_weights(float wx, float wy, float wz, float val,
                                                 float gx, float gy, float gz) {
  float x1 = wx, x0 = 1.f - wx, y1 = wy, y0 = 1.f - wy, z1 = wz, z0 = 1.f - wz;
  float a0 = fmaf(val, x0, -gx), a1 = fmaf(val, x1, gx);          // val*X_a +- gx
  float yz00 = y0 * z0, yz10 = y1 * z0, yz01 = y0 * z1, yz11 = y1 * z1;
  float gyz0 = gy * z0, gyz1 = gy * z1, gzy0 = gz * y0, gzy1 = gz * y1;
  float b00 = -gyz0 - gzy0, b10 = gyz0 - gzy1, b01 = gzy0 - gyz1, b11 = gyz1 + gzy1;
  Corners c;
  c.c000 = fmaf(yz00, a0, x0 * b00);  c.c100 = fmaf(yz00, a1, x1 * b00);
  c.c010 = fmaf(yz10, a0, x0 * b10);  c.c110 = fmaf(yz10, a1, x1 * b10);
  c.c001 = fmaf(yz01, a0, x0 * b01);  c.c101 = fmaf(yz01, a1, x1 * b01);
  c.c011 = fmaf(yz11, a0, x0 * b11);  c.c111 = fmaf(yz11, a1, x1 * b11);
  return c;
}

// fp32 atomic add without return: one global_atomic_add_f32 on gfx950 (no CAS loop).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---- wave reductions (64 lanes) -------------------------------------------------------------
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor(v, o, kWave));
  return v;
}

// ---- cylinder (radial profile) volume, src/cylinder_volume.cpp ---------------------------------
struct Cyl {
  const float* data;    // may point to LDS
  int rres;
  float radius, length;
  float h, inv_h;       // h = radius/(rres-1)  (:42)
  float r2;             // radius^2
};

struct CylCell { int i0, i1; float w0, r, rhx, rhz; bool tiny; };

__device__ __forceinline__ CylCell cyl_locate(const Cyl& C, float px, float pz) {
  CylCell c;
  float xs = px - C.radius, zs = pz - C.radius;                 // :37-38 (y component zeroed)
  c.r = sqrtf(fmaf(xs, xs, zs * zs));                           // :41
  float rm = c.r * C.inv_h;                                     // :44 (r / h)
  int ir = (int)floorf(rm);
  c.i0 = clampi(ir, 0, C.rres - 1);                             // :45
  c.i1 = clampi(c.i0 + 1, 0, C.rres - 1);                       // :46
  c.w0 = rm - (float)c.i0;                                      // :48 (uses the CLAMPED idx0)
  c.tiny = c.r < 1e-6f;                                         // :15, :56
  float inv_r = c.tiny ? 0.f : 1.f / c.r;
  c.rhx = xs * inv_r; c.rhz = zs * inv_r;                       // normalize(xs), zeroed when tiny
  return c;
}

// src/cylinder_volume.cpp:150-156
__device__ __forceinline__ bool cyl_inbounds(const Cyl& C, float px, float py, float pz) {
  float xs = px - C.radius, zs = pz - C.radius;
  return ((xs * xs + zs * zs) < C.r2) & (py < C.length) & (py >= 0.f);
}
// src/cylinder_volume.cpp:158-170
__device__ __forceinline__ bool cyl_escaped(const Cyl& C, float px, float py, float pz,
                                            float vx, float vy, float vz) {
  float xs = px - C.radius, zs = pz - C.radius;
  bool esc_len = ((py < 0.f) & (vy < 0.f)) | ((py > C.length) & (vy > 0.f));
  bool out_r = (xs * xs + zs * zs) >= C.r2;
  bool esc_r = (xs * vx + zs * vz) > 0.f;
  return (out_r & esc_r) | esc_len;
}

}  // namespace drrt
