// drrt_sort.hip -- locality sort of rays by entry voxel (device side, rocPRIM radix sort).
//
// Not part of the reference (enoki processes rays in caller order, array-at-a-time); this is the
// MI355X-side answer to SURVEY.md section 7 step 5: lanes of a wave should touch neighbouring
// voxels so that each of the 8 taps of a wave-step hits a handful of 128-B lines and the adjoint's
// scatter targets coincide / are contiguous.  Key = flat voxel index (z*H + y)*W + x of the point
// where the ray first meets the grid box along its march direction (the ray's own position when it
// starts inside).  The permutation only changes the VISIT order; results are written back in the
// caller's ray order.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "drrt_device.h"

namespace drrt {

__global__ void __launch_bounds__(256) k_entry_keys(Vol V, size_t n, const float* __restrict__ pos,
                                                    const float* __restrict__ vel, float dir_sign,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  float d[3] = {dir_sign * vel[3 * i], dir_sign * vel[3 * i + 1], dir_sign * vel[3 * i + 2]};
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
  float t = (hit && tmax >= tmin) ? tmin : 0.f;
  const int r[3] = {V.W, V.H, V.D};
  int v[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float e = fmaf(t, d[a], p[a]);
    e = fminf(fmaxf(e, 0.f), b[a]);
    v[a] = clampi((int)floorf(e * V.inv_h), 0, r[a] - 1);
  }
  keys[i] = (uint32_t)((v[2] * V.H + v[1]) * V.W + v[0]);
  idx[i] = (uint32_t)i;
}

static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

static size_t cub_temp_bytes(size_t n) {
  size_t temp = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                                   (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 32,
                                                   (hipStream_t)0);
  if (e != hipSuccess || temp == 0) {   // no device visible (CPU-only import): conservative bound
    (void)hipGetLastError();
    temp = n * 16 + (1u << 20);
  }
  return temp;
}

size_t sort_workspace_bytes(size_t n) { return 4 * al(n * sizeof(uint32_t)) + al(cub_temp_bytes(n)); }

hipError_t sort_rays_by_entry_voxel(const Vol& V, float h, size_t n, const float* pos, const float* vel,
                                    float dir_sign, void* ws, size_t ws_bytes, const uint32_t** perm_out,
                                    hipStream_t stream) {
  (void)h;
  char* base = (char*)ws;
  const size_t arr = al(n * sizeof(uint32_t));
  uint32_t* keys_in = (uint32_t*)(base);
  uint32_t* keys_out = (uint32_t*)(base + arr);
  uint32_t* idx_in = (uint32_t*)(base + 2 * arr);
  uint32_t* idx_out = (uint32_t*)(base + 3 * arr);
  void* temp = base + 4 * arr;
  size_t temp_bytes = ws_bytes - 4 * arr;
  hipLaunchKernelGGL(k_entry_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, V, n, pos, vel,
                     dir_sign, keys_in, idx_in);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  long long nvox = (long long)V.W * V.H * V.D;
  int bits = 1;
  while (bits < 32 && (1LL << bits) < nvox) ++bits;
  e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const uint32_t*)keys_in, keys_out,
                                         (const uint32_t*)idx_in, idx_out, (int)n, 0, bits, stream);
  *perm_out = idx_out;
  return e;
}

}  // namespace drrt
