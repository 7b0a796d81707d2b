// Micro-benchmark (development tool): LDS accumulation with RANDOM slot addresses inside a per-wave window, as the
// sparse mode of k_backtrace_ring produces them (8 corner adds per cell hand-over, lanes in different cells).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/lds_random tools/lds_random_bench.hip && gpurun_out/lds_random
// Variants: ds_add_f64 (what the kernel does), ds_add_u32 / ds_add_u64, plain read-add-write of f64 (b64) and of f32 pairs
// (b64 = two fp32 slots), with all 64 lanes or a random half of them active.  16 waves per CU (4 blocks of 4) as in the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int kSlots = 1250;            // per wave
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, int iters, unsigned active_pct, int cellmode) {
  __shared__ double s[kSlots + 16];
  const int lane = threadIdx.x;
  for (int i = lane; i < kSlots + 16; i += 64) s[i] = 0.0;
  __syncthreads();
  unsigned rng = 1234567u * (blockIdx.x + 1) + 7919u * lane;
  double acc = 0.0;
  float* sf = reinterpret_cast<float*>(s);
  unsigned* su = reinterpret_cast<unsigned*>(s);
  unsigned long long* sl = reinterpret_cast<unsigned long long*>(s);
  for (int it = 0; it < iters; ++it) {
    rng = rng * 1664525u + 1013904223u;
    const bool on = (rng >> 8) % 100u < active_pct;
    // a "cell": base slot + the 8 corner offsets of a 10 x 10 x 12 window (x stride 1, y stride 10, z stride 100)
    const int base = cellmode ? (int)((rng >> 12) % (unsigned)(kSlots - 112)) : (lane * 19) % (kSlots - 112);
    if (on) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int a = base + (c & 1) + ((c >> 1) & 1) * 10 + (c >> 2) * 100;
        const float v = 1.0f + c;
        if (MODE == 0) atomicAdd(&s[a], (double)v);
        else if (MODE == 1) atomicAdd(&su[a], (unsigned)c + 1u);
        else if (MODE == 2) atomicAdd(&sl[a], (unsigned long long)c + 1ull);
        else if (MODE == 3) s[a] += (double)v;                       // plain RMW f64 (b64 read, b64 write)
        else if (MODE == 4) sf[a] += v;                              // plain RMW f32
      }
      if (MODE == 5) {                                               // plain RMW of f32 x-pairs: 4 x (read b64, 2 adds, write b64)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int a = (base & ~1) + (c & 1) * 10 + (c >> 1) * 100;
          float2 q = *reinterpret_cast<float2*>(&sf[a]);
          q.x += 1.0f; q.y += 2.0f;
          *reinterpret_cast<float2*>(&sf[a]) = q;
        }
      }
      if (MODE == 6) {                                               // plain RMW of f64 x-pairs: 4 x (read b128, 2 adds, write b128)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int a = (base & ~1) + (c & 1) * 10 + (c >> 1) * 100;
          double2 q = *reinterpret_cast<double2*>(&s[a]);
          q.x += 1.0; q.y += 2.0;
          *reinterpret_cast<double2*>(&s[a]) = q;
        }
      }
    }
  }
  __syncthreads();
  out[blockIdx.x * 64 + lane] = s[lane] + acc;
}

template <int MODE>
static void run(const char* name, unsigned pct, int cellmode) {
  double* d; hipMalloc(&d, 256 * 16 * 64 * sizeof(double));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4000, blocks = 256 * 16;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10, pct, cellmode);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, pct, cellmode);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: 16 waves x iters "hand-overs" of 8 corner adds
  const double handovers_per_cu = 16.0 * iters;
  printf("%-22s active %3u%% %-8s : %7.3f ms  %7.1f cycles per 8-corner hand-over per CU (at 2.4 GHz)\n", name, pct,
         cellmode ? "random" : "strided", ms, ms * 1e-3 * 2.4e9 / handovers_per_cu);
  hipFree(d);
}

int main() {
  for (int cm : {0, 1}) for (unsigned pct : {100u, 50u}) {
    run<0>("ds_add_f64 x8", pct, cm);
    run<1>("ds_add_u32 x8", pct, cm);
    run<2>("ds_add_u64 x8", pct, cm);
    run<3>("rmw f64 x8", pct, cm);
    run<4>("rmw f32 x8", pct, cm);
    run<5>("rmw f32 pairs x4", pct, cm);
    run<6>("rmw f64 pairs x4", pct, cm);
  }
  return 0;
}
