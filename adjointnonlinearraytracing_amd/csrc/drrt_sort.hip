// drrt_sort.hip -- locality sort of rays (device side, rocPRIM radix sort).
//
// Not part of the reference (enoki processes rays in caller order, array-at-a-time); this is the
// MI355X-side answer to SURVEY.md section 7 step 5: the 64 rays of a wave should stay spatially
// close for their WHOLE march, so that each of the 8 taps of a wave-step hits a handful of 128-B
// lines and the adjoint's scatter targets fall into the wave's LDS gradient window.
//
// Round 3 key ("light-field key", the default): a ray is the line it travels on, i.e. a direction and a 2-D offset in the
// plane perpendicular to it.
//   direction  octahedral map of the unit direction, cut into 31 x 31 cells (an ODD count: +-x, +-y, +z are cell CENTRES);
//              the cell index is the major part of the key, so a collimated view is one contiguous key range;
//   offset     the point of the line closest to the box centre, in the orthonormal frame (t1, t2) of the CELL's centre
//              direction (every ray of a cell uses the same frame), scaled by a power of two of the box extent and cut
//              into 2048 x 2048 cells; the minor part of the key is the HILBERT index of that cell.  10 + 22 bits: a 32-bit
//              key (the sort moves 8 instead of 12 bytes per ray and pass).
// Why Hilbert and not the Z-order interleave of rounds 1-2: 64 consecutive rays of a Z-order are an aligned 8 x 8 tile only
// when the sampling grid happens to be aligned with the key cells (the metric's pixel-aligned plane source); shift the
// same source by a third of a pixel and 11 % of the bundles straddle two far-apart tiles (measured: adjoint 4.8 -> 7.1 ms),
// and the views of core/source.py rand_rays_cube + random_rotate_ic, oblique to every axis, gave bundles whose box of grid
// cells was 12 cells wide in the median (82 % wider than the 9^3 window; adjoint 4.8 -> 20 ms: global-atomic fallback).  Any
// 64 consecutive cells of a Hilbert curve are a connected, compact patch; aligned 8 x 8 blocks are still contiguous, so the
// aligned source keeps its perfect tiles (tools/bundle_stats.py: offline comparison of the keys).  For the axis-aligned
// frame the power-of-two scale keeps pixel boundaries on key-cell boundaries.
//
// Rounds 1-2 key (DRRT_FLAG_CHORD_KEY, kept for A-B): 60-bit "light-field Z-order": the straight line through the ray
// (from where it meets the grid box along its march direction to where that line leaves the box again) is described by
// its two end points, each quantised to 10 bits per axis, and the 6 coordinates are bit-interleaved (Morton order in 6-D).
// For the adjoint the march direction is -vt and the "start" is the recorded exit sample xt.
// The permutation only changes the VISIT order; results are written back in the caller's ray order.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

#include "drrt_device.h"

namespace drrt {

constexpr int kKeyBitsPerAxis = 10;
constexpr int kKeyBits = 6 * kKeyBitsPerAxis;

// one ray component in storage format `io` (0 fp32, 1 IEEE half, 2 the 16-bit ray state of drrt_device.h)
__device__ __forceinline__ float ldr(const Vol& V, const void* p, size_t k, int io, bool is_pos) {
  if (io == 3) return is_pos ? q16_pos_dec(V, ((const uint16_t*)p)[k]) : ((const float*)p)[k];
  if (io == 2) return is_pos ? q16_pos_dec(V, ((const uint16_t*)p)[k]) : q16_vel_dec(((const int16_t*)p)[k]);
  return io ? __half2float(((const __half*)p)[k]) : ((const float*)p)[k];
}

__global__ void __launch_bounds__(256) k_chord_keys(Vol V, size_t n, const void* __restrict__ pos,
                                                    const void* __restrict__ vel, int io_half, float dir_sign,
                                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float p[3] = {ldr(V, pos, 3 * i, io_half, true), ldr(V, pos, 3 * i + 1, io_half, true), ldr(V, pos, 3 * i + 2, io_half, true)};
  float d[3] = {dir_sign * ldr(V, vel, 3 * i, io_half, false), dir_sign * ldr(V, vel, 3 * i + 1, io_half, false),
                dir_sign * ldr(V, vel, 3 * i + 2, io_half, false)};
  const float b[3] = {V.bx, V.by, V.bz};
  float tmin = 0.f, tmax = 3.0e38f;
  bool hit = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (fabsf(d[a]) > 1e-20f) {
      float inv = 1.f / d[a];
      float t1 = (0.f - p[a]) * inv, t2 = (b[a] - p[a]) * inv;
      tmin = fmaxf(tmin, fminf(t1, t2));
      tmax = fminf(tmax, fmaxf(t1, t2));
    } else if (p[a] < 0.f || p[a] > b[a]) {
      hit = false;
    }
  }
  hit = hit && tmax >= tmin;
  const float t0 = hit ? tmin : 0.f, t1 = hit ? tmax : 0.f;
  uint32_t q[6];
  const float scale = (float)(1 << kKeyBitsPerAxis);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float inv_b = b[a] > 0.f ? 1.f / b[a] : 0.f;
    float e0 = fminf(fmaxf(fmaf(t0, d[a], p[a]) * inv_b, 0.f), 0.99999f);
    float e1 = fminf(fmaxf(fmaf(t1, d[a], p[a]) * inv_b, 0.f), 0.99999f);
    q[a] = (uint32_t)(e0 * scale);
    q[3 + a] = (uint32_t)(e1 * scale);
  }
  uint64_t key = 0;
#pragma unroll
  for (int bit = 0; bit < kKeyBitsPerAxis; ++bit)
#pragma unroll
    for (int j = 0; j < 6; ++j)
      key |= (uint64_t)((q[j] >> bit) & 1u) << (6 * bit + (5 - j));
  keys[i] = key;
  idx[i] = (uint32_t)i;
}

// ---- light-field key ---------------------------------------------------------------------------------------------
constexpr int kDirHalf = 15;                         // direction cells per octahedral axis: 2 * kDirHalf + 1 = 31
constexpr int kPosBits = 11;                         // offset cells per axis: 2048
constexpr int kLfKeyBits = 2 * kPosBits + 10;        // 22 + 10 (31 * 31 = 961 direction cells < 2^10): 32 bits

__device__ __forceinline__ uint32_t hilbert2(uint32_t x, uint32_t y) {      // x, y < 2^kPosBits
  uint32_t d = 0;
#pragma unroll
  for (int b = kPosBits - 1; b >= 0; --b) {
    const uint32_t s = 1u << b, rx = (x >> b) & 1u, ry = (y >> b) & 1u;
    d += s * s * ((3u * rx) ^ ry);
    if (ry == 0u) {                                  // rotate / reflect the quadrant
      if (rx == 1u) { x = s - 1u - x; y = s - 1u - y; }
      const uint32_t t = x; x = y; y = t;
    }
    x &= s - 1u; y &= s - 1u;
  }
  return d;
}

__global__ void __launch_bounds__(256) k_lightfield_keys(Vol V, size_t n, const void* __restrict__ pos,
                                                         const void* __restrict__ vel, int io_half, float dir_sign,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float p[3] = {ldr(V, pos, 3 * i, io_half, true), ldr(V, pos, 3 * i + 1, io_half, true), ldr(V, pos, 3 * i + 2, io_half, true)};
  float d[3] = {dir_sign * ldr(V, vel, 3 * i, io_half, false), dir_sign * ldr(V, vel, 3 * i + 1, io_half, false),
                dir_sign * ldr(V, vel, 3 * i + 2, io_half, false)};
  idx[i] = (uint32_t)i;
  const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (!(len > 1e-30f) || !(len < 3.0e38f)) { keys[i] = 0; return; }       // a ray at rest (or non-finite): any place will do
  const float il = 1.f / len;
  d[0] *= il; d[1] *= il; d[2] *= il;
  // octahedral map -> cell (a, b) in [-kDirHalf, kDirHalf]^2
  const float l1 = fabsf(d[0]) + fabsf(d[1]) + fabsf(d[2]);
  float ox = d[0] / l1, oy = d[1] / l1;
  if (d[2] < 0.f) {
    const float fx = (1.f - fabsf(oy)) * (ox >= 0.f ? 1.f : -1.f), fy = (1.f - fabsf(ox)) * (oy >= 0.f ? 1.f : -1.f);
    ox = fx; oy = fy;
  }
  int a = (int)rintf(ox * (float)kDirHalf), b = (int)rintf(oy * (float)kDirHalf);
  a = a < -kDirHalf ? -kDirHalf : (a > kDirHalf ? kDirHalf : a);
  b = b < -kDirHalf ? -kDirHalf : (b > kDirHalf ? kDirHalf : b);
  // centre direction of the cell, and ITS frame: the coordinate axis least aligned with it, made orthogonal
  float cx = (float)a / (float)kDirHalf, cy = (float)b / (float)kDirHalf, cz = 1.f - fabsf(cx) - fabsf(cy);
  if (cz < 0.f) {
    const float fx = (1.f - fabsf(cy)) * (cx >= 0.f ? 1.f : -1.f), fy = (1.f - fabsf(cx)) * (cy >= 0.f ? 1.f : -1.f);
    cx = fx; cy = fy;
  }
  const float cl = 1.f / sqrtf(cx * cx + cy * cy + cz * cz);
  const float c[3] = {cx * cl, cy * cl, cz * cl};
  int ax = 0;
  if (fabsf(c[1]) < fabsf(c[ax])) ax = 1;
  if (fabsf(c[2]) < fabsf(c[ax])) ax = 2;
  float t1[3] = {-c[ax] * c[0], -c[ax] * c[1], -c[ax] * c[2]};
  t1[ax] += 1.f;
  const float tl = 1.f / sqrtf(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
  t1[0] *= tl; t1[1] *= tl; t1[2] *= tl;
  const float t2[3] = {c[1] * t1[2] - c[2] * t1[1], c[2] * t1[0] - c[0] * t1[2], c[0] * t1[1] - c[1] * t1[0]};
  // the point of the line closest to the box centre, in that frame; scale 1 / (2 E), E = the largest box extent
  const float w[3] = {p[0] - 0.5f * V.bx, p[1] - 0.5f * V.by, p[2] - 0.5f * V.bz};
  const float wd = w[0] * d[0] + w[1] * d[1] + w[2] * d[2];
  const float q[3] = {w[0] - wd * d[0], w[1] - wd * d[1], w[2] - wd * d[2]};
  const float ext = fmaxf(V.bx, fmaxf(V.by, V.bz));
  const float sc = ext > 0.f ? 0.5f / ext : 0.f;
  const float u = (q[0] * t1[0] + q[1] * t1[1] + q[2] * t1[2]) * sc + 0.5f;
  const float v = (q[0] * t2[0] + q[1] * t2[1] + q[2] * t2[2]) * sc + 0.5f;
  const float cells = (float)(1 << kPosBits);
  const uint32_t qu = (uint32_t)fminf(fmaxf(u * cells, 0.f), cells - 1.f);
  const uint32_t qv = (uint32_t)fminf(fmaxf(v * cells, 0.f), cells - 1.f);
  const uint32_t cell = (uint32_t)((a + kDirHalf) * (2 * kDirHalf + 1) + (b + kDirHalf));
  keys[i] = (cell << (2 * kPosBits)) | hilbert2(qu, qv);
}

static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

// (rocPRIM picks the algorithm by size: up to radix_sort_config<>::merge_sort_limit = 1M items a block sort + merge passes
// -- 21 launches of 5-9 us for the metric's 1 048 576 rays, 0.167 ms --, Onesweep above.  Lowering the limit so that 1M rays
// take Onesweep was measured, same box: 0.170 ms at 1M rays, and 0.137 against 0.057 ms for a 131 072-ray shard -- Onesweep's
// fixed cost is larger than the merge passes'.  The default stays.)
static size_t radix_temp_bytes(size_t n) {
  size_t t64 = 0, t32 = 0;          // both key widths share the buffers: the larger temporary storage
  hipError_t e = rocprim::radix_sort_pairs(nullptr, t64, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                          (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0u, (unsigned)kKeyBits,
                                          (hipStream_t)0);
  hipError_t e2 = rocprim::radix_sort_pairs(nullptr, t32, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                           (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0u, (unsigned)kLfKeyBits,
                                           (hipStream_t)0);
  size_t temp = t64 > t32 ? t64 : t32;
  if (e != hipSuccess || e2 != hipSuccess || temp == 0) {   // no device visible (CPU-only import): conservative bound
    (void)hipGetLastError();
    temp = n * 32 + (1u << 20);
  }
  return temp;
}

size_t sort_workspace_bytes(size_t n) {
  return 2 * al(n * sizeof(uint64_t)) + 2 * al(n * sizeof(uint32_t)) + al(radix_temp_bytes(n));
}

hipError_t sort_rays_by_entry_voxel(const Vol& V, float h, size_t n, const void* pos, const void* vel, int io_half,
                                    float dir_sign, void* ws, size_t ws_bytes, const uint32_t** perm_out,
                                    hipStream_t stream, bool chord_key) {
  (void)h;
  char* base = (char*)ws;
  const size_t k8 = al(n * sizeof(uint64_t)), k4 = al(n * sizeof(uint32_t));
  uint64_t* keys_in = (uint64_t*)(base);
  uint64_t* keys_out = (uint64_t*)(base + k8);
  uint32_t* idx_in = (uint32_t*)(base + 2 * k8);
  uint32_t* idx_out = (uint32_t*)(base + 2 * k8 + k4);
  void* temp = base + 2 * k8 + 2 * k4;
  size_t temp_bytes = ws_bytes - (2 * k8 + 2 * k4);
  hipError_t e;
  if (chord_key) {
    hipLaunchKernelGGL(k_chord_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, V, n, pos, vel,
                       io_half, dir_sign, keys_in, idx_in);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint64_t*)keys_in, keys_out,
                                  (const uint32_t*)idx_in, idx_out, n, 0u, (unsigned)kKeyBits, stream);
  } else {                                     // 32-bit keys in the same buffers
    hipLaunchKernelGGL(k_lightfield_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, V, n, pos, vel,
                       io_half, dir_sign, (uint32_t*)keys_in, idx_in);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)keys_in, (uint32_t*)keys_out,
                                  (const uint32_t*)idx_in, idx_out, n, 0u, (unsigned)kLfKeyBits, stream);
  }
  *perm_out = idx_out;
  return e;
}

}  // namespace drrt
