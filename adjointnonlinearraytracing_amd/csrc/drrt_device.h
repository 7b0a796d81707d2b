// drrt_device.h -- per-ray building blocks of the gfx950 eikonal ray-march kernels.
//
// What the reference computes (file:line = /root/reference/...):
//   volume::eval_grad  src/volume.cpp:101-181   trilinear n(p) and grad n(p)
//   volume::eval_hess  src/volume.cpp:40-99     mixed second partials (zero diagonal)
//   volume::splat      src/volume.cpp:182-244   adjoint of eval_grad w.r.t. the voxels
//   volume::inbounds / escaped  src/volume.cpp:246-271
//   cylinder_volume::*  src/cylinder_volume.cpp:26-170
//   one forward / adjoint march iteration: src/tracer.cpp:68-86, :420-435
//
// How it is computed here: the 8 taps are fetched once per ray-step and n, grad n and the
// three mixed partials all come from ONE factored lerp tree (differences first, ~27 flops
// instead of the ~150 of the expanded weight products; differences-first also avoids the
// cancellation of the reference's (sum of 4 products) - (sum of 4 products) form); the 16
// scatter_adds of volume::splat are fused into 8 corner contributions.
//
// ARITHMETIC CONTRACT ("factored fp32 spec"): every expression below is an explicit sequence of
// IEEE-754 binary32 operations (+, -, *, fmaf, floorf, sqrtf, /, conversions); the library is
// built with -ffp-contract=off so the compiler adds no fusion of its own.  The CPU oracle's
// `factored` arithmetic mode (oracle/drrt_oracle_impl.h) restates the same sequence in plain C,
// which makes forward trajectories and exit steps comparable BIT FOR BIT between the GPU and the
// oracle (only the order of the adjoint's atomic sums differs).  Functions that need nothing
// device-specific are __host__ __device__ so that tests/hostcheck can run this very code on the
// CPU; the package itself never does (no CPU compute path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRRT_HD __host__ __device__ __forceinline__

namespace drrt {

constexpr int kWave = 64;

struct Vol {
  const float* data;
  int W, H, D;          // res[0], res[1], res[2]  (x, y, z extents; x is memory-contiguous)
  int sy, sz;           // element strides of y and z: W, W*H
  float inv_h;          // 1.0f/h   (reference: rcp(h_), src/volume.cpp:128)
  float inv_h2;         // inv_h*inv_h
  float bx, by, bz;     // (float)(res-1)*h : bounds of inbounds/escaped (src/volume.cpp:252-254)
  unsigned lx, ly, lz;  // res-3 (0 when res < 4): a floor index i with 1 <= i <= res-3 is "strictly interior"
  const float* pair;    // optional (device only): the "pair copy" of the grid, 2 floats per voxel i: {n[i], n[i+sy]}
                        // (clamped at the far y face), built per call by k_build_pair; null = not in use
  float q_min, q_step, q_inv_step;   // 16-bit ray-state positions ("q16", include/drrt_hip.h): p = q_min + code * q_step
};

// ---- 16-bit ray state ("q16"): 6 bytes per 3-vector, like IEEE half, but with the precision where a march needs it ----
// position  unsigned 16-bit code over [-E/16, E + E/16], E = the largest box extent: step = 1.125 E / 65535 (h/228 on a
//           256^3 grid, against h/8 for an IEEE half near 1.0); positions outside the range saturate
// direction signed 16-bit fixed point, step 2^-14 (range [-2, 2): |v| = n stays below 2 for any index below 2)
// seeds     (dx, dv of the adjoint) stay IEEE half: they need relative, not absolute, precision
constexpr float kQ16VelStep = 1.0f / 16384.0f;
DRRT_HD float q16_pos_dec(const Vol& V, uint16_t c) { return fmaf((float)c, V.q_step, V.q_min); }
DRRT_HD uint16_t q16_pos_enc(const Vol& V, float x) {
  float t = (x - V.q_min) * V.q_inv_step;
  t = t > 0.f ? t : 0.f;                       // also NaN -> 0
  t = t < 65535.f ? t : 65535.f;
  return (uint16_t)(int)rintf(t);
}
DRRT_HD float q16_vel_dec(int16_t c) { return (float)c * kQ16VelStep; }
DRRT_HD int16_t q16_vel_enc(float v) {
  float t = v * 16384.0f;
  t = t > -32768.f ? t : -32768.f;             // also NaN -> -32768
  t = t < 32767.f ? t : 32767.f;
  return (int16_t)(int)rintf(t);
}

DRRT_HD void vol_finish(Vol& V, float h) {   // derived fields; data, W, H, D must be set
  V.sy = V.W; V.sz = V.W * V.H; V.pair = nullptr;
  V.inv_h = 1.0f / h; V.inv_h2 = V.inv_h * V.inv_h;
  V.bx = (float)(V.W - 1) * h; V.by = (float)(V.H - 1) * h; V.bz = (float)(V.D - 1) * h;
  V.lx = V.W >= 4 ? (unsigned)(V.W - 3) : 0u; V.ly = V.H >= 4 ? (unsigned)(V.H - 3) : 0u;
  V.lz = V.D >= 4 ? (unsigned)(V.D - 3) : 0u;
  const float ext = fmaxf(V.bx, fmaxf(V.by, V.bz));
  V.q_min = -ext * 0.0625f; V.q_step = (ext * 1.125f) / 65535.0f; V.q_inv_step = 65535.0f / (ext * 1.125f);
}

struct Cell {
  int base;             // flat index of corner 000
  int ix, iy, iz;       // (clamped) integer coordinates of corner 000
  int ox, oy, oz;       // element offsets to the +x, +y, +z neighbours (0 where clamped)
  float wx, wy, wz;     // fractional weights w0 = pm - floor(pm)  (unclamped, Q11)
  bool interior;        // every floor index in [1, res-3]: no clamp acts, the 8 taps are distinct, and
                        // the point is a whole cell away from every face, so whatever the rounding of
                        // p*inv_h: inbounds(p) is true and escaped(p, .) is false
};

DRRT_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// a*b + c for non-negative operands below 2^24 (grid coordinates and strides): v_mad_u32_u24 is a
// full-rate VALU op, the generic 32-bit integer multiply is quarter rate
DRRT_HD int mad24(int a, int b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(__umul24((unsigned)a, (unsigned)b) + (unsigned)c);
#else
  return a * b + c;
#endif
}

// float -> int with the saturating behaviour of v_cvt_i32_f32 (NaN -> 0), also on the host
DRRT_HD int f2i_sat(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)f;
#else
  if (!(f == f)) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
#endif
}

// src/volume.cpp:128-141: pm = p*rcp(h); pos = floor2int(pm); w0 = pm - pos; clamp indices.
DRRT_HD Cell locate(const Vol& V, float px, float py, float pz) {
  Cell c;
  float fx = px * V.inv_h, fy = py * V.inv_h, fz = pz * V.inv_h;
  float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
  c.wx = fx - flx; c.wy = fy - fly; c.wz = fz - flz;
  int ix = f2i_sat(flx), iy = f2i_sat(fly), iz = f2i_sat(flz);
  c.interior = (((unsigned)ix - 1u) < V.lx) & (((unsigned)iy - 1u) < V.ly) & (((unsigned)iz - 1u) < V.lz);   // wraps, no UB
  if (c.interior) {                      // fast path: the clamps below are no-ops here
    c.base = mad24(iz, V.sz, mad24(iy, V.sy, ix));
    c.ix = ix; c.iy = iy; c.iz = iz;
    c.ox = 1; c.oy = V.sy; c.oz = V.sz;
    return c;
  }
  // clamp(i + 1, 0, res - 1) written so that a saturated index (huge or non-finite coordinate) cannot overflow
  int x0 = clampi(ix, 0, V.W - 1), x1 = clampi(ix, -1, V.W - 2) + 1;
  int y0 = clampi(iy, 0, V.H - 1), y1 = clampi(iy, -1, V.H - 2) + 1;
  int z0 = clampi(iz, 0, V.D - 1), z1 = clampi(iz, -1, V.D - 2) + 1;
  c.base = mad24(z0, V.sz, mad24(y0, V.sy, x0));
  c.ix = x0; c.iy = y0; c.iz = z0;
  c.ox = x1 - x0; c.oy = (y1 != y0) ? V.sy : 0; c.oz = (z1 != z0) ? V.sz : 0;
  return c;
}

// The 8 taps of a cell, held as the four (x0, x0+1) PAIRS they are fetched as -- each pair is one register pair
// that an 8-byte load fills directly, so a lane can keep its taps across steps (fetch_reuse) without copies:
//   a = (v000, v100)  at (y0, z0)      b = (v010, v110)  at (y1, z0)
//   e = (v001, v101)  at (y0, z1)      f = (v011, v111)  at (y1, z1)
typedef float f2 __attribute__((ext_vector_type(2)));
struct Taps { f2 a, b, e, f; };

DRRT_HD Taps taps_zero() { Taps t; t.a = t.b = t.e = t.f = f2{0.f, 0.f}; return t; }

// One (x0, x0+1) pair: ONE 8-byte load on the device (global_load_dwordx2 needs only 4-byte alignment).
DRRT_HD f2 ld_pair(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float __attribute__((ext_vector_type(2), aligned(4))) f2u;
  const f2u a = *reinterpret_cast<const f2u*>(p);
  return f2{a.x, a.y};
#else
  return f2{p[0], p[1]};
#endif
}

DRRT_HD Taps fetch(const float* __restrict__ d, const Cell& c) {
  Taps t;
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(c.base >= 0 && c.base < (1 << 29));   // make_vol() rejects grids of 2^29 voxels or more
#endif
  const float* p = d + (unsigned)c.base;
  // The x-neighbour is the next float in memory: fetch each (x0, x0+1) pair with ONE 8-byte load.  The texture
  // addresser handles a wave's gather at a few lanes per clock, so 4 pair loads instead of 8 dword loads halve
  // the dominant cost of the march.  Cells clamped in x (ox == 0, only on the far x face) must not read p[1]:
  // that could run past the end of the grid allocation.
  if (c.ox == 1) {
    t.a = ld_pair(p); t.b = ld_pair(p + c.oy); t.e = ld_pair(p + c.oz); t.f = ld_pair(p + c.oz + c.oy);
    return t;
  }
  t.a = f2{p[0], p[c.ox]};                t.b = f2{p[c.oy], p[c.oy + c.ox]};
  t.e = f2{p[c.oz], p[c.oz + c.ox]};      t.f = f2{p[c.oz + c.oy], p[c.oz + c.oy + c.ox]};
  return t;
}

// The march's gather is bound by the texture addresser, and its cost is per gather INSTRUCTION: ~30-37 cycles of the
// CU's addresser for a divergent 64-lane load, the same with one active lane as with 64, and only ~15 % more for 16
// bytes per lane than for 8 (tools/chain_bench.hip, tools/gather_bench.hip).  With the pair copy of the grid --
// P[i] = {n[i], n[i+sy]}, the y-neighbour interleaved -- ONE 16-byte load at (x0, y0, z) returns the four taps of a
// z-face, {n(x0,y0), n(x0,y1), n(x0+1,y0), n(x0+1,y1)}, so a strictly interior cell is two loads instead of four
// (and the copy is 2x the grid, against 4x for a full (x,y) quad per voxel).  Same floats -> bit-identical results.
typedef float f4 __attribute__((ext_vector_type(4)));
DRRT_HD Taps taps_from_pair(f4 q0, f4 q1) {      // q0 = face z0, q1 = face z1 of the pair copy
  Taps t;
  t.a = f2{q0.x, q0.z}; t.b = f2{q0.y, q0.w};
  t.e = f2{q1.x, q1.z}; t.f = f2{q1.y, q1.w};
  return t;
}
#if defined(__HIPCC__)
__device__ __forceinline__ f4 ld_quad8(const float* p) {      // 16 bytes at 8-byte alignment: one global_load_dwordx4
  typedef float __attribute__((ext_vector_type(4), aligned(8))) f4u;
  const f4u a = *reinterpret_cast<const f4u*>(p);
  return f4{a.x, a.y, a.z, a.w};
}
#endif
DRRT_HD Taps fetch_vol(const Vol& V, const Cell& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (V.pair != nullptr && c.interior) {
    __builtin_assume(c.base >= 0 && c.base < (1 << 29));
    const float* p = V.pair + 2u * (unsigned)c.base;
    return taps_from_pair(ld_quad8(p), ld_quad8(p + 2u * (unsigned)V.sz));
  }
#endif
  return fetch(V.data, c);
}

// ---- tap reuse ------------------------------------------------------------------------------------
// With ds = h/step_res a ray samples the same cell ~step_res times in a row, and the cell it moves on to
// shares a face -- 4 of the 8 taps -- with the one it leaves.  The march's limiter is the texture addresser,
// whose cost is per lane-address (above), so the taps a lane already holds are kept in registers:
//   REUSE 1  same cell as the previous step: no load at all;
//   REUSE 2  additionally, a move across ONE y- or z-face keeps the two (x0, x0+1) pairs of the shared face and
//            loads only the two pairs of the new far face (an x-move would still need four pair loads: the
//            pairs straddle it).
// Only strictly interior cells are cached (regular strides, 8 distinct taps).  The floats are the ones a full
// fetch would return, so results are bit-identical (tests: hostcheck on the CPU, GPU parity vs the oracle).
struct TapCache { Taps t; int base; };     // base < 0: nothing cached

template <int REUSE>
DRRT_HD void fetch_reuse(const Vol& V, const Cell& c, TapCache& tc) {
  if (REUSE == 0 || V.pair != nullptr) { tc.t = fetch_vol(V, c); return; }
  if (!c.interior) { tc.t = fetch(V.data, c); tc.base = -1; return; }
  if (c.base == tc.base) return;
  Taps& t = tc.t;
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(c.base >= 0 && c.base < (1 << 29));
#endif
  const float* p = V.data + (unsigned)c.base;
  if (REUSE >= 2 && tc.base >= 0) {
    const int d = c.base - tc.base;
    if (d == V.sy)  { t.a = t.b; t.e = t.f; t.b = ld_pair(p + V.sy); t.f = ld_pair(p + V.sz + V.sy); tc.base = c.base; return; }   // +y
    if (d == -V.sy) { t.b = t.a; t.f = t.e; t.a = ld_pair(p);        t.e = ld_pair(p + V.sz);        tc.base = c.base; return; }   // -y
    if (d == V.sz)  { t.a = t.e; t.b = t.f; t.e = ld_pair(p + V.sz); t.f = ld_pair(p + V.sz + V.sy); tc.base = c.base; return; }   // +z
    if (d == -V.sz) { t.e = t.a; t.f = t.b; t.a = ld_pair(p);        t.b = ld_pair(p + V.sy);        tc.base = c.base; return; }   // -z
  }
  t.a = ld_pair(p); t.b = ld_pair(p + V.sy); t.e = ld_pair(p + V.sz); t.f = ld_pair(p + V.sz + V.sy);
  tc.base = c.base;
}

// n and RAW gradient / mixed partials (not yet multiplied by 1/h, 1/h^2).
struct Sample { float n, gx, gy, gz, hxy, hxz, hyz; };

template <bool WITH_HESS>
DRRT_HD Sample interp(const Taps& t, float wx, float wy, float wz) {
  Sample s;
  // x-differences at the four (y,z) edges
  float d00 = t.a.y - t.a.x, d10 = t.b.y - t.b.x, d01 = t.e.y - t.e.x, d11 = t.f.y - t.f.x;
  // x-lerped values
  float c00 = fmaf(wx, d00, t.a.x), c10 = fmaf(wx, d10, t.b.x);
  float c01 = fmaf(wx, d01, t.e.x), c11 = fmaf(wx, d11, t.f.x);
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
DRRT_HD bool inbounds(const Vol& V, float px, float py, float pz) {
  return (px >= 0.f) & (py >= 0.f) & (pz >= 0.f) & (px < V.bx) & (py < V.by) & (pz < V.bz);
}
// src/volume.cpp:258-271
DRRT_HD bool escaped(const Vol& V, float px, float py, float pz, float vx, float vy, float vz) {
  bool ex = ((px < 0.f) & (vx < 0.f)) | ((px >= V.bx) & (vx > 0.f));
  bool ey = ((py < 0.f) & (vy < 0.f)) | ((py >= V.by) & (vy > 0.f));
  bool ez = ((pz < 0.f) & (vz < 0.f)) | ((pz >= V.bz) & (vz > 0.f));
  return ex | ey | ez;
}

DRRT_HD float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

// Fused volume::splat (src/volume.cpp:217-243): contribution of (val, grad) to the 8 corners.
//   corner(a,b,c) = val*X_a*Y_b*Z_c  +  gx*(+-)Y_b*Z_c + gy*(+-)X_a*Z_c + gz*(+-)X_a*Y_b
// with X_0 = 1-wx, X_1 = wx, sign = - for index 0, + for index 1.
struct Corners { float c000, c100, c010, c110, c001, c101, c011, c111; };

DRRT_HD Corners splat_weights(float wx, float wy, float wz, float val, float gx, float gy, float gz) {
  float x1 = wx, x0 = 1.f - wx, y1 = wy, y0 = 1.f - wy, z1 = wz, z0 = 1.f - wz;
  float a0 = fmaf(val, x0, -gx), a1 = fmaf(val, x1, gx);          // val*X_a -+ gx
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

// The same 8 corner contributions as x-PAIRS (c_0bc, c_1bc), computed with packed fp32 operations (v_pk_mul_f32 /
// v_pk_fma_f32: two IEEE operations per instruction, the SAME operations in the same order as splat_weights, so the
// values are bit-identical): 13 packed + 7 scalar instructions instead of 42 scalar ones.
struct CornerPairs { f2 c00, c10, c01, c11; };    // (y0,z0) (y1,z0) (y0,z1) (y1,z1), each (x0, x1)

DRRT_HD f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

DRRT_HD CornerPairs splat_weights_pk(float wx, float wy, float wz, float val, float gx, float gy, float gz) {
  const float y0 = 1.f - wy, z0 = 1.f - wz;
  const f2 X = f2{1.f - wx, wx}, Y = f2{y0, wy};
  const f2 A = fma2(f2{val, val}, X, f2{-gx, gx});                       // val*X_a -+ gx
  const f2 yzl = Y * z0, yzh = Y * wz;                                   // (yz00, yz10), (yz01, yz11)
  const f2 gyz = f2{z0, wz} * gy, gzy = Y * gz;                          // (gyz0, gyz1), (gzy0, gzy1)
  const float b00 = -gyz.x - gzy.x, b10 = gyz.x - gzy.y, b01 = gzy.x - gyz.y, b11 = gyz.y + gzy.y;
  CornerPairs c;
  c.c00 = fma2(f2{yzl.x, yzl.x}, A, X * b00);
  c.c10 = fma2(f2{yzl.y, yzl.y}, A, X * b10);
  c.c01 = fma2(f2{yzh.x, yzh.x}, A, X * b01);
  c.c11 = fma2(f2{yzh.y, yzh.y}, A, X * b11);
  return c;
}

// ---------------------------------------------------------------------------------------------
// one forward march iteration (src/tracer.cpp:68-86; plane :144-145; sdf :287-288)
// MODE 0 = trace, 1 = trace_plane, 2 = trace_sdf, 3 = trace_target (closest-approach tracking)
// ---------------------------------------------------------------------------------------------
struct FwdState {
  float x, y, z, vx, vy, vz;         // marching state
  float xtx, xty, xtz, vtx, vty, vtz;  // recorded exit / closest-approach state
  float aux0, aux1, aux2, aux3, aux4, aux5;  // MODE 1: plane origin+normal; MODE 3: target (aux0..2), best dist2 (aux3)
  bool inside, esc;
};

DRRT_HD void fwd_init(const Vol& V, FwdState& s) {
  s.xtx = s.x; s.xty = s.y; s.xtz = s.z; s.vtx = s.vx; s.vty = s.vy; s.vtz = s.vz;   // :56-57
  s.inside = inbounds(V, s.x, s.y, s.z);                                              // :61
  s.esc = false;                                                                      // :62
}

// One forward iteration.  In: `c` = cell of the current position (always located, used only when the
// ray is inside), `t` = the 8 taps there (valid when s.inside).  Out: `c` = cell of the NEW position,
// reused by the next iteration -- and when that cell is strictly interior the box tests are skipped
// (inbounds is true and escaped is false there by construction, see Cell::interior).
template <int MODE>
DRRT_HD void fwd_step_c(const Vol& V, const float* __restrict__ sdf, float ds, FwdState& s, Cell& c, const Taps& t) {
  float n = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
  if (s.inside) {                                                           // masked gather (Q4)
    Sample q = interp<false>(t, c.wx, c.wy, c.wz);
    n = q.n; gx = q.gx * V.inv_h; gy = q.gy * V.inv_h; gz = q.gz * V.inv_h;
  }
  const float dsn = ds * n;
  s.vx = fmaf(dsn, gx, s.vx); s.vy = fmaf(dsn, gy, s.vy); s.vz = fmaf(dsn, gz, s.vz);   // :70
  s.x = fmaf(ds, s.vx, s.x); s.y = fmaf(ds, s.vy, s.y); s.z = fmaf(ds, s.vz, s.z);      // :71
  c = locate(V, s.x, s.y, s.z);
  bool cur_inside, esc_now = false;
  if (MODE == 2) {                                                          // :287-288
    float d = 0.f;
    if (s.inside) d = interp<false>(fetch(sdf, c), c.wx, c.wy, c.wz).n;
    cur_inside = d < 0.f;
    if (!c.interior) esc_now = escaped(V, s.x, s.y, s.z, s.vx, s.vy, s.vz);
  } else {
    cur_inside = true;
    if (!c.interior) {
      cur_inside = inbounds(V, s.x, s.y, s.z);                              // :73
      esc_now = escaped(V, s.x, s.y, s.z, s.vx, s.vy, s.vz);                // :76
    }
    if (MODE == 1) {                                                        // :144-145
      float d = dot3(s.x - s.aux0, s.y - s.aux1, s.z - s.aux2, s.aux3, s.aux4, s.aux5);
      cur_inside = cur_inside & !(d > 0.f);
    }
  }
  const bool cross = s.inside & !cur_inside;                                // :74
  s.esc = s.esc | cross | esc_now;                                          // :75-76
  if (MODE == 3) {                                                          // :216-227
    float ex = s.x - s.aux0, ey = s.y - s.aux1, ez = s.z - s.aux2;
    float cur = dot3(ex, ey, ez, ex, ey, ez);
    if (cur < s.aux3) { s.xtx = s.x; s.xty = s.y; s.xtz = s.z; s.vtx = s.vx; s.vty = s.vy; s.vtz = s.vz; s.aux3 = cur; }
  } else if (cross) {                                                       // :79-80
    s.xtx = s.x; s.xty = s.y; s.xtz = s.z; s.vtx = s.vx; s.vty = s.vy; s.vtz = s.vz;
  }
  s.inside = cur_inside;                                                    // :86
}

// trace_plane, for a ray that has just been flagged escaped: can it produce ANOTHER exit record later?
// The reference keeps marching escaped rays while its global loop runs (src/tracer.cpp:130-160) and records
// EVERY inside -> outside transition, so a ray that can come back into {in bounds, not past the plane} may
// overwrite its exit record later.  An escaped ray flies straight (its gathers are masked), hence it can
// NOT come back when it is outside the box moving away (volume::escaped), or past the plane and not
// approaching it.  Anything else -- in practice only a ray that STARTS past the plane and heads back
// through it -- is flagged and re-marched over the global loop count by ray_full<1>.
DRRT_HD bool plane_again(const Vol& V, const FwdState& s) {
  const float d = dot3(s.x - s.aux0, s.y - s.aux1, s.z - s.aux2, s.aux3, s.aux4, s.aux5);
  const float dv = dot3(s.vx, s.vy, s.vz, s.aux3, s.aux4, s.aux5);
  const bool gone = escaped(V, s.x, s.y, s.z, s.vx, s.vy, s.vz) | ((d > 0.f) & (dv >= 0.f));
  return !gone;
}

template <int MODE>
DRRT_HD void fwd_step(const Vol& V, const float* __restrict__ sdf, float ds, FwdState& s, Cell& c) {
  Taps t = taps_zero();
  if (s.inside) t = fetch_vol(V, c);
  fwd_step_c<MODE>(V, sdf, ds, s, c, t);
}

// The same with the taps the lane already holds kept across steps (fetch_reuse).  A ray that is not inside does
// not sample (Q4) and fwd_step_c ignores the taps then, so the cache is simply left alone.
template <int MODE, int REUSE>
DRRT_HD void fwd_step_r(const Vol& V, const float* __restrict__ sdf, float ds, FwdState& s, Cell& c, TapCache& tc) {
  if (s.inside) fetch_reuse<REUSE>(V, c, tc);
  fwd_step_c<MODE>(V, sdf, ds, s, c, tc.t);
}

// ---------------------------------------------------------------------------------------------
// one adjoint march iteration (src/tracer.cpp:420-435; sdf :488-497).  Returns true when the ray
// is still active after the step, in which case (c, w) is its contribution to dL/dn: add the
// 8 corner values `w` at the 8 taps of cell `c`.
// ---------------------------------------------------------------------------------------------
struct AdjState {
  float x, y, z, vx, vy, vz;
  float lx, ly, lz, mx, my, mz;      // lambda (dL/dx), mu (dL/dv)
  bool active, outside;
};

DRRT_HD void adj_init(const Vol& V, float ds, float dxx, float dxy, float dxz,
                      float dvx, float dvy, float dvz, AdjState& s) {
  s.lx = dxx; s.ly = dxy; s.lz = dxz;                                                   // :409
  s.mx = fmaf(ds, dxx, dvx); s.my = fmaf(ds, dxy, dvy); s.mz = fmaf(ds, dxz, dvz);      // :410
  s.active = !escaped(V, s.x, s.y, s.z, -s.vx, -s.vy, -s.vz);                           // :413-414
  s.outside = false;
}

// One adjoint iteration is split in two halves so that the windowed kernel can issue the NEXT sample's gather
// between them (software pipelining, k_backtrace_flat / k_backtrace_ring); adj_step below is the plain sequence.
struct AdjSample { float n, gx, gy, gz, hxy, hxz, hyz; };   // n, grad n (scaled by 1/h), mixed partials (raw)

// First half (src/tracer.cpp:421-425; sdf :488-497): sample the cell `c` the ray has just stepped into (taps `t`),
// update v, decide whether the ray is still active.  Everything on the march's critical path is here.
// MODE 1: `st` are the 8 sdf taps of the cell when the caller has gathered them ahead (`have_st`), else they are fetched here.
template <int MODE>   // 0 = backtrace, 1 = backtrace_sdf
DRRT_HD bool adj_sample_st(const Vol& V, const float* __restrict__ sdf, float ds, AdjState& s, const Cell& c, const Taps& t,
                           AdjSample& m, const Taps& st, bool have_st) {
  const Sample q = interp<true>(t, c.wx, c.wy, c.wz);                                  // :421-422
  m.n = q.n; m.gx = q.gx * V.inv_h; m.gy = q.gy * V.inv_h; m.gz = q.gz * V.inv_h;
  m.hxy = q.hxy; m.hxz = q.hxz; m.hyz = q.hyz;
  const float mdsn = -ds * m.n;
  s.vx = fmaf(mdsn, m.gx, s.vx); s.vy = fmaf(mdsn, m.gy, s.vy); s.vz = fmaf(mdsn, m.gz, s.vz); // :423
  bool active = true;
  if (!c.interior) active = !escaped(V, s.x, s.y, s.z, -s.vx, -s.vy, -s.vz);            // :425
  if (MODE == 1) {                                                                      // :488-497
    bool now_out = interp<false>(have_st ? st : fetch(sdf, c), c.wx, c.wy, c.wz).n >= 0.f;
    active = active & !((!s.outside) & now_out);
    s.outside = now_out;
  }
  s.active = active;
  return active;
}
template <int MODE>
DRRT_HD bool adj_sample(const Vol& V, const float* __restrict__ sdf, float ds, AdjState& s, const Cell& c, const Taps& t,
                        AdjSample& m) {
  return adj_sample_st<MODE>(V, sdf, ds, s, c, t, m, t, false);
}

// Second half (:430-435), for a ray that is still active: its contribution `w` to dL/dn at the 8 taps of `c`, then
// the lambda / mu recurrences.  Nothing here feeds the next sample's position.
DRRT_HD void adj_contrib(const Vol& V, float ds, float grad_scale, AdjState& s, const Cell& c, const AdjSample& m,
                         Corners& w) {
  const float n = m.n, gx = m.gx, gy = m.gy, gz = m.gz;
  const float dn = dot3(s.mx, s.my, s.mz, gx, gy, gz);                                  // :430
  const float nds = (n * ds) * grad_scale;
  w = splat_weights(c.wx, c.wy, c.wz, dn * ds, nds * s.mx, nds * s.my, nds * s.mz);     // :431-432
  // la += ds*(dn*grad n + n*H*mu), H = mixed partials / h^2, zero diagonal (:434, Q10)
  const float hxy = m.hxy * V.inv_h2, hxz = m.hxz * V.inv_h2, hyz = m.hyz * V.inv_h2;
  const float hmx = fmaf(hxz, s.mz, hxy * s.my);
  const float hmy = fmaf(hyz, s.mz, hxy * s.mx);
  const float hmz = fmaf(hyz, s.my, hxz * s.mx);
  s.lx = fmaf(ds, fmaf(dn, gx, n * hmx), s.lx);
  s.ly = fmaf(ds, fmaf(dn, gy, n * hmy), s.ly);
  s.lz = fmaf(ds, fmaf(dn, gz, n * hmz), s.lz);
  s.mx = fmaf(ds, s.lx, s.mx); s.my = fmaf(ds, s.ly, s.my); s.mz = fmaf(ds, s.lz, s.mz);   // :435
}

template <int MODE>   // 0 = backtrace, 1 = backtrace_sdf
DRRT_HD bool adj_step(const Vol& V, const float* __restrict__ sdf, float ds, float grad_scale,
                      AdjState& s, Cell& c, Corners& w) {
  s.x = fmaf(-ds, s.vx, s.x); s.y = fmaf(-ds, s.vy, s.y); s.z = fmaf(-ds, s.vz, s.z);   // :420
  c = locate(V, s.x, s.y, s.z);
  AdjSample m;
  if (!adj_sample<MODE>(V, sdf, ds, s, c, fetch_vol(V, c), m)) return false;            // :426-428
  adj_contrib(V, ds, grad_scale, s, c, m, w);
  return true;
}

// ---- cylinder (radial profile) volume, src/cylinder_volume.cpp ---------------------------------
struct Cyl {
  const float* data;    // may point to LDS
  int rres;
  float radius, length;
  float h, inv_h;       // h = radius/(rres-1) (:42), inv_h = 1/h
  float r2;             // radius*radius
};

DRRT_HD Cyl make_cyl(const float* data, int rres, float radius, float length) {
  Cyl C; C.data = data; C.rres = rres; C.radius = radius; C.length = length;
  C.h = radius / (float)(rres - 1); C.inv_h = 1.f / C.h; C.r2 = radius * radius;
  return C;
}

struct CylCell { int i0, i1; float w0, r, rhx, rhz; bool tiny; };

DRRT_HD CylCell cyl_locate(const Cyl& C, float px, float pz) {
  CylCell c;
  float xs = px - C.radius, zs = pz - C.radius;                 // :37-38 (y component zeroed)
  c.r = sqrtf(fmaf(xs, xs, zs * zs));                           // :41
  float rm = c.r * C.inv_h;                                     // :44 (r / h)
  int ir = f2i_sat(floorf(rm));
  c.i0 = clampi(ir, 0, C.rres - 1);                             // :45
  c.i1 = clampi(c.i0 + 1, 0, C.rres - 1);                       // :46
  c.w0 = rm - (float)c.i0;                                      // :48 (uses the CLAMPED idx0)
  c.tiny = c.r < 1e-6f;                                         // :15, :56
  float inv_r = c.tiny ? 0.f : 1.f / c.r;
  c.rhx = xs * inv_r; c.rhz = zs * inv_r;                       // normalize(xs), zeroed when tiny
  return c;
}

// src/cylinder_volume.cpp:150-156
DRRT_HD bool cyl_inbounds(const Cyl& C, float px, float py, float pz) {
  float xs = px - C.radius, zs = pz - C.radius;
  return (fmaf(xs, xs, zs * zs) < C.r2) & (py < C.length) & (py >= 0.f);
}
// src/cylinder_volume.cpp:158-170
DRRT_HD bool cyl_escaped(const Cyl& C, float px, float py, float pz, float vx, float vy, float vz) {
  float xs = px - C.radius, zs = pz - C.radius;
  bool esc_len = ((py < 0.f) & (vy < 0.f)) | ((py > C.length) & (vy > 0.f));
  bool out_r = fmaf(xs, xs, zs * zs) >= C.r2;
  bool esc_r = fmaf(xs, vx, zs * vz) > 0.f;
  return (out_r & esc_r) | esc_len;
}

// one forward cable iteration (src/tracer.cpp:351-373); target/best in aux0..3 like MODE 3
DRRT_HD void cable_fwd_step(const Cyl& C, float ds, FwdState& s) {
  CylCell c = cyl_locate(C, s.x, s.z);                                  // :351 (unmasked gather)
  float v0 = C.data[c.i0], v1 = C.data[c.i1];
  float f = fmaf(v1, c.w0, v0 * (1.f - c.w0));                          // :53
  float rx = (v1 - v0) * C.inv_h;                                       // :54
  float dsn = ds * f;
  s.vx = fmaf(dsn, rx * c.rhx, s.vx); s.vz = fmaf(dsn, rx * c.rhz, s.vz);   // :353 (grad_y = 0)
  s.x = fmaf(ds, s.vx, s.x); s.y = fmaf(ds, s.vy, s.y); s.z = fmaf(ds, s.vz, s.z);   // :354
  float ex = s.x - s.aux0, ey = s.y - s.aux1, ez = s.z - s.aux2;
  float cur = dot3(ex, ey, ez, ex, ey, ez);                             // :356
  bool cur_inside = cyl_inbounds(C, s.x, s.y, s.z);
  bool cross = s.inside & !cur_inside;
  s.esc = s.esc | cross | cyl_escaped(C, s.x, s.y, s.z, s.vx, s.vy, s.vz);   // :361-362
  if (cur < s.aux3) { s.xtx = s.x; s.xty = s.y; s.xtz = s.z; s.vtx = s.vx; s.vty = s.vy; s.vtz = s.vz; s.aux3 = cur; } // :365-367
  s.inside = cur_inside;
}

// one adjoint cable iteration (src/tracer.cpp:547-562); contribution: a0 at i0, a1 at i1
DRRT_HD bool cable_adj_step(const Cyl& C, float ds, AdjState& s, int& i0, int& i1, float& a0, float& a1) {
  s.x = fmaf(-ds, s.vx, s.x); s.y = fmaf(-ds, s.vy, s.y); s.z = fmaf(-ds, s.vz, s.z);     // :547
  CylCell c = cyl_locate(C, s.x, s.z);
  float v0 = C.data[c.i0], v1 = C.data[c.i1];
  float w0 = c.w0, w1 = 1.f - c.w0;
  float n = fmaf(v1, w0, v0 * w1);                                      // :53
  float rx = (v1 - v0) * C.inv_h;                                       // :54 / :88
  float gx = rx * c.rhx, gz = rx * c.rhz;                               // grad n (y comp 0)
  float mdsn = -ds * n;
  s.vx = fmaf(mdsn, gx, s.vx); s.vz = fmaf(mdsn, gz, s.vz);             // :550
  s.active = !cyl_escaped(C, s.x, s.y, s.z, -s.vx, -s.vy, -s.vz);       // :552
  if (!s.active) return false;
  float dn = fmaf(s.mz, gz, s.mx * gx);                                 // :557
  // cylinder_volume::splat (:113-148): value taps val*w, gradient taps -+(grad.rhat)/h
  float val = dn * ds;
  float gv = (n * ds) * fmaf(s.mz, c.rhz, s.mx * c.rhx);                // dot(dnx*ds, rhat); 0 if tiny
  float gvh = gv * C.inv_h;
  i0 = c.i0; i1 = c.i1;
  a0 = fmaf(val, w1, -gvh); a1 = fmaf(val, w0, gvh);
  // Hessian (:88-108): (I - rhat rhat^T)_{xz} * (n'/r), zero when r < eps
  float sH = c.tiny ? 0.f : rx / c.r;
  float h00 = (1.f - c.rhx * c.rhx) * sH, h02 = -(c.rhx * c.rhz) * sH, h22 = (1.f - c.rhz * c.rhz) * sH;
  float hmx = fmaf(h02, s.mz, h00 * s.mx), hmz = fmaf(h22, s.mz, h02 * s.mx);
  s.lx = fmaf(ds, fmaf(dn, gx, n * hmx), s.lx);                         // :561 (y row of H is 0)
  s.lz = fmaf(ds, fmaf(dn, gz, n * hmz), s.lz);
  s.mx = fmaf(ds, s.lx, s.mx); s.my = fmaf(ds, s.ly, s.my); s.mz = fmaf(ds, s.lz, s.mz);   // :562
  return true;
}

// ---------------------------------------------------------------------------------------------
// whole-ray drivers shared by the kernels (one ray per lane) and by tests/hostcheck
// ---------------------------------------------------------------------------------------------
struct RayOut { float xt[3], vt[3]; float dist2; bool esc, act, again; unsigned steps; };

// trace / trace_plane / trace_sdf for ONE ray, per-ray termination (see drrt_march.h header)
constexpr int kTapReuse = 1;      // product default of the forward marches (see fetch_reuse): measured on MI355X, 256^3 /
                                  // 1M rays: no reuse 1.47 ms, same-cell 1.31 ms, + shared-face 1.68 ms (its branches cost
                                  // more issue slots than the two pair loads they save)

template <int MODE, int REUSE = kTapReuse>
DRRT_HD RayOut trace_ray(const Vol& V, const float* __restrict__ sdf, float ds, int max_steps,
                         const float p[3], const float v[3], const float* pln_o, const float* pln_d) {
  FwdState s;
  s.x = p[0]; s.y = p[1]; s.z = p[2]; s.vx = v[0]; s.vy = v[1]; s.vz = v[2];
  s.aux0 = s.aux1 = s.aux2 = s.aux3 = s.aux4 = s.aux5 = 0.f;
  if (MODE == 1) {
    s.aux0 = pln_o[0]; s.aux1 = pln_o[1]; s.aux2 = pln_o[2];
    s.aux3 = pln_d[0]; s.aux4 = pln_d[1]; s.aux5 = pln_d[2];
  }
  fwd_init(V, s);
  Cell c = locate(V, s.x, s.y, s.z);
  bool act = true;
  if (MODE == 2) act = interp<false>(fetch(sdf, c), c.wx, c.wy, c.wz).n < 0.f;   // src/tracer.cpp:276-277
  unsigned steps = 0;
  TapCache tc;
  tc.base = -1;
  tc.t = taps_zero();
  for (int it = 0; it < max_steps; ++it) {
    fwd_step_r<MODE, REUSE>(V, sdf, ds, s, c, tc);
    ++steps;
    if (s.esc) break;                                                     // per-ray form of :82
  }
  act = act & !s.esc;                                                     // :77
  if (MODE != 2 && !s.esc) { s.xtx = s.x; s.xty = s.y; s.xtz = s.z; }     // :95 (vt stays, Q6)
  RayOut o;
  o.xt[0] = s.xtx; o.xt[1] = s.xty; o.xt[2] = s.xtz; o.vt[0] = s.vtx; o.vt[1] = s.vty; o.vt[2] = s.vtz;
  o.dist2 = 0.f; o.esc = s.esc; o.act = act; o.steps = steps; o.again = false;
  if (MODE == 1 && s.esc) o.again = plane_again(V, s);
  if (MODE == 2 && s.esc && s.inside) {
    // trace_sdf: the ray left the BOX (volume::escaped) while the clamped sdf sample still reads negative.  The
    // reference keeps marching it -- still "inside", still refracted by clamped samples -- and records the
    // moment the sdf turns non-negative, if its global loop lasts that long (:283-304).  Flag for ray_full<2>.
    // (Once the sdf sample is non-negative the later ones are masked to 0, so a ray can cross only once.)
    o.again = true;
  }
  return o;
}

// trace_plane / trace_sdf for ONE ray over exactly `total` iterations of the reference's global loop, no
// early termination (src/tracer.cpp:130-160, :283-304 as written): for the rays trace_ray flags with `again`.
template <int MODE>
DRRT_HD RayOut ray_full(const Vol& V, const float* __restrict__ sdf, float ds, unsigned total, const float p[3],
                        const float v[3], const float* pln_o, const float* pln_d) {
  FwdState s;
  s.x = p[0]; s.y = p[1]; s.z = p[2]; s.vx = v[0]; s.vy = v[1]; s.vz = v[2];
  s.aux0 = s.aux1 = s.aux2 = s.aux3 = s.aux4 = s.aux5 = 0.f;
  if (MODE == 1) {
    s.aux0 = pln_o[0]; s.aux1 = pln_o[1]; s.aux2 = pln_o[2];
    s.aux3 = pln_d[0]; s.aux4 = pln_d[1]; s.aux5 = pln_d[2];
  }
  fwd_init(V, s);
  Cell c = locate(V, s.x, s.y, s.z);
  for (unsigned it = 0; it < total; ++it) fwd_step<MODE>(V, sdf, ds, s, c);
  if (MODE != 2 && !s.esc) { s.xtx = s.x; s.xty = s.y; s.xtz = s.z; }
  RayOut o;
  o.xt[0] = s.xtx; o.xt[1] = s.xty; o.xt[2] = s.xtz; o.vt[0] = s.vtx; o.vt[1] = s.vty; o.vt[2] = s.vtz;
  o.dist2 = 0.f; o.esc = s.esc; o.act = !s.esc; o.steps = total; o.again = false;
  return o;
}

// trace_target phase A for ONE ray: march until escaped, track the closest approach;
// `cont` receives the marching state (x, v) for phase B.
template <int REUSE = kTapReuse>
DRRT_HD RayOut target_ray_a(const Vol& V, float ds, int max_steps, const float p[3], const float v[3],
                            const float tg[3], float cont[6]) {
  FwdState s;
  s.x = p[0]; s.y = p[1]; s.z = p[2]; s.vx = v[0]; s.vy = v[1]; s.vz = v[2];
  s.aux0 = tg[0]; s.aux1 = tg[1]; s.aux2 = tg[2]; s.aux4 = s.aux5 = 0.f;
  float ex = s.x - tg[0], ey = s.y - tg[1], ez = s.z - tg[2];
  s.aux3 = dot3(ex, ey, ez, ex, ey, ez);                                  // src/tracer.cpp:200
  fwd_init(V, s);
  Cell c = locate(V, s.x, s.y, s.z);
  unsigned steps = 0;
  TapCache tc;
  tc.base = -1;
  tc.t = taps_zero();
  for (int it = 0; it < max_steps; ++it) {
    fwd_step_r<3, REUSE>(V, nullptr, ds, s, c, tc);
    ++steps;
    if (s.esc) break;
  }
  RayOut o;
  o.xt[0] = s.xtx; o.xt[1] = s.xty; o.xt[2] = s.xtz; o.vt[0] = s.vtx; o.vt[1] = s.vty; o.vt[2] = s.vtz;
  o.dist2 = s.aux3; o.esc = s.esc; o.act = !s.esc; o.steps = steps;
  cont[0] = s.x; cont[1] = s.y; cont[2] = s.z; cont[3] = s.vx; cont[4] = s.vy; cont[5] = s.vz;
  return o;
}

// trace_target phase B for ONE ray: an escaped ray flies straight (its gathers are masked) for
// the remaining `total - done` iterations of the reference's global loop (:225-227 is not gated
// by `escaped`).  Returns true when the closest-approach record was improved.
DRRT_HD bool target_ray_b(float ds, unsigned done, unsigned total, const float cont[6], const float tg[3],
                          float& best, float xt[3], float vt[3]) {
  float x = cont[0], y = cont[1], z = cont[2];
  const float vx = cont[3], vy = cont[4], vz = cont[5];
  bool upd = false;
  for (unsigned k = done; k < total; ++k) {
    x = fmaf(ds, vx, x); y = fmaf(ds, vy, y); z = fmaf(ds, vz, z);
    float ex = x - tg[0], ey = y - tg[1], ez = z - tg[2];
    float cur = dot3(ex, ey, ez, ex, ey, ez);
    if (cur < best) { best = cur; xt[0] = x; xt[1] = y; xt[2] = z; upd = true; }
  }
  if (upd) { vt[0] = vx; vt[1] = vy; vt[2] = vz; }
  return upd;
}

// backtrace / backtrace_sdf for ONE ray; `sink(cell, corners)` receives every contribution.
template <int MODE, typename Sink>
DRRT_HD unsigned backtrace_ray(const Vol& V, const float* __restrict__ sdf, float ds, float grad_scale,
                               int max_steps, const float xt[3], const float vt[3], const float dx[3],
                               const float dv[3], Sink&& sink) {
  AdjState s;
  s.x = xt[0]; s.y = xt[1]; s.z = xt[2]; s.vx = vt[0]; s.vy = vt[1]; s.vz = vt[2];
  adj_init(V, ds, dx[0], dx[1], dx[2], dv[0], dv[1], dv[2], s);
  if (MODE == 1 && s.active) {                                            // src/tracer.cpp:476-477
    Cell c = locate(V, s.x, s.y, s.z);
    s.outside = interp<false>(fetch(sdf, c), c.wx, c.wy, c.wz).n >= 0.f;
  }
  unsigned steps = 0;
  for (int it = 0; it < max_steps && s.active; ++it) {
    Cell c; Corners w;
    if (!adj_step<MODE>(V, sdf, ds, grad_scale, s, c, w)) break;
    ++steps;
    sink(c, w);
  }
  return steps;
}

// trace_cable for ONE ray (src/tracer.cpp:312-382)
DRRT_HD RayOut cable_trace_ray(const Cyl& C, float ds, int max_steps, const float p[3], const float v[3],
                               const float tg[3]) {
  FwdState s;
  s.x = p[0]; s.y = p[1]; s.z = p[2]; s.vx = v[0]; s.vy = v[1]; s.vz = v[2];
  s.xtx = s.x; s.xty = s.y; s.xtz = s.z; s.vtx = s.vx; s.vty = s.vy; s.vtz = s.vz;
  s.aux0 = tg[0]; s.aux1 = tg[1]; s.aux2 = tg[2]; s.aux4 = s.aux5 = 0.f;
  float ex = s.x - tg[0], ey = s.y - tg[1], ez = s.z - tg[2];
  s.aux3 = dot3(ex, ey, ez, ex, ey, ez);                                  // :340
  s.inside = cyl_inbounds(C, s.x, s.y, s.z);                              // :344
  s.esc = false;
  unsigned steps = 0;
  for (int it = 0; it < max_steps; ++it) {
    cable_fwd_step(C, ds, s);
    ++steps;
    if (s.esc) break;          // state is frozen once !active (:353-354): nothing changes later
  }
  RayOut o;
  o.xt[0] = s.xtx; o.xt[1] = s.xty; o.xt[2] = s.xtz; o.vt[0] = s.vtx; o.vt[1] = s.vty; o.vt[2] = s.vtz;
  o.dist2 = s.aux3; o.esc = s.esc; o.act = !s.esc; o.steps = steps;
  return o;
}

// backtrace_cable for ONE ray (src/tracer.cpp:511-567); sink(i0, i1, a0, a1)
template <typename Sink>
DRRT_HD unsigned cable_backtrace_ray(const Cyl& C, float ds, int max_steps, const float xt[3], const float vt[3],
                                     const float dx[3], const float dv[3], Sink&& sink) {
  AdjState s;
  s.x = xt[0]; s.y = xt[1]; s.z = xt[2]; s.vx = vt[0]; s.vy = vt[1]; s.vz = vt[2];
  s.lx = dx[0]; s.ly = dx[1]; s.lz = dx[2];                               // :536
  s.mx = fmaf(ds, dx[0], dv[0]); s.my = fmaf(ds, dx[1], dv[1]); s.mz = fmaf(ds, dx[2], dv[2]);  // :537
  s.active = !cyl_escaped(C, s.x, s.y, s.z, -s.vx, -s.vy, -s.vz);         // :540-541
  s.outside = false;
  unsigned steps = 0;
  for (int it = 0; it < max_steps && s.active; ++it) {
    int i0, i1; float a0, a1;
    if (!cable_adj_step(C, ds, s, i0, i1, a0, a1)) break;
    ++steps;
    sink(i0, i1, a0, a1);
  }
  return steps;
}

#if defined(__HIPCC__)
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
#endif

}  // namespace drrt
