// drrt_sort.hip -- locality sort of rays by entry voxel (device side, rocPRIM radix sort).
//
// Not part of the reference (enoki processes rays in caller order, array-at-a-time); this is the
// MI355X-side answer to SURVEY.md section 7 step 5: the 64 rays of a wave should stay spatially
// close for their WHOLE march, so that each of the 8 taps of a wave-step hits a handful of 128-B
// lines and the adjoint's scatter targets fall into the wave's LDS gradient window.
//
// Key = 60-bit "light-field Z-order": the straight line through the ray (from where it meets the
// grid box along its march direction to where that line leaves the box again) is described by its
// two end points, each quantised to 10 bits per axis, and the 6 coordinates are bit-interleaved
// (Morton order in 6-D).  Consecutive keys are therefore close at BOTH ends of the chord:
//   * plane sources: both end points form 2-D patches  -> compact ray bundles;
//   * the adjoint of a focusing lens starts with thousands of rays in ONE voxel pointing in
//     different directions: the far end of the chord separates them by direction;
//   * point / cone sources: the near end is degenerate, the far end orders the fan.
// (A key on the start voxel alone -- the first version -- left 27 % of the adjoint's taps outside
// the LDS windows on the Luneburg benchmark: strips of rays on the side faces fan out.  Re-measured
// with the final kernels: forward 1.65 vs 1.53 ms, adjoint 6.39 vs 5.99 ms in favour of this key.)
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

static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

static size_t radix_temp_bytes(size_t n) {
  size_t temp = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, temp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                          (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0u, (unsigned)kKeyBits,
                                          (hipStream_t)0);
  if (e != hipSuccess || temp == 0) {   // no device visible (CPU-only import): conservative bound
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
                                    hipStream_t stream) {
  (void)h;
  char* base = (char*)ws;
  const size_t k8 = al(n * sizeof(uint64_t)), k4 = al(n * sizeof(uint32_t));
  uint64_t* keys_in = (uint64_t*)(base);
  uint64_t* keys_out = (uint64_t*)(base + k8);
  uint32_t* idx_in = (uint32_t*)(base + 2 * k8);
  uint32_t* idx_out = (uint32_t*)(base + 2 * k8 + k4);
  void* temp = base + 2 * k8 + 2 * k4;
  size_t temp_bytes = ws_bytes - (2 * k8 + 2 * k4);
  hipLaunchKernelGGL(k_chord_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, V, n, pos, vel,
                     io_half, dir_sign, keys_in, idx_in);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint64_t*)keys_in, keys_out,
                                (const uint32_t*)idx_in, idx_out, n, 0u, (unsigned)kKeyBits, stream);
  *perm_out = idx_out;
  return e;
}

}  // namespace drrt
