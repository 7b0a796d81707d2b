// Micro-benchmark (development tool): throughput of ds_add_f32 on gfx950 as a function of how many
// lanes of one instruction hit the same LDS address ("multiplicity").  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/lds_bench tools/lds_atomic_bench.hip && gpurun_out/lds_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>   // 0: ds_add_f32 (no return), 1: plain read-add-write (non-atomic), 2: ds_add_rtn
__global__ void __launch_bounds__(256) k(float* out, int iters, int mult, int spread) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // lanes [g*mult, (g+1)*mult) share an address; groups are `spread` dwords apart
  int addr = wid * 1024 + ((lane / mult) * spread) % 1024;
  float v = 1.0f + lane;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int a2 = wid * 1024 + ((addr + u * 37) & 1023);
      if (MODE == 0) atomicAdd(&s[a2], v);
      else if (MODE == 1) s[a2] += v;
      else acc += atomicAdd(&s[a2], v);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = s[threadIdx.x] + acc;
}

template <int MODE>
static void run(const char* name, int mult, int spread) {
  float* d; hipMalloc(&d, 2048 * 64 * sizeof(float));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, blocks = 256 * 4;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, mult, spread);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, mult, spread);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double lane_ops = (double)blocks * 256 * iters * 8;
  // per CU: blocks/256 CUs resident work; cycles per wave-instruction per CU at 2.4 GHz
  double wave_instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  printf("%-10s mult=%2d spread=%3d : %7.3f ms  %.3e lane-adds/s  %.1f cycles/wave-instr/CU\n", name, mult, spread, ms,
         lane_ops / (ms * 1e-3), ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
  hipFree(d);
}

template <typename T>
__global__ void __launch_bounds__(256) ki(T* out, int iters, int mult) {
  __shared__ T s[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) s[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int addr = (lane / mult);
  T v = (T)(1 + lane);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int a2 = wid * 512 + ((addr + u * 37) & 511);
      atomicAdd(&s[a2], v);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = s[threadIdx.x];
}

template <typename T>
static void runi(const char* name, int mult) {
  T* d; hipMalloc(&d, 2048 * 64 * sizeof(T));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, blocks = 256 * 4;
  hipLaunchKernelGGL(ki<T>, dim3(blocks), dim3(256), 0, 0, d, 10, mult);
  hipEventRecord(a);
  hipLaunchKernelGGL(ki<T>, dim3(blocks), dim3(256), 0, 0, d, iters, mult);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double lane_ops = (double)blocks * 256 * iters * 8;
  double wave_instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  printf("%-10s mult=%2d            : %7.3f ms  %.3e lane-adds/s  %.1f cycles/wave-instr/CU\n", name, mult, ms,
         lane_ops / (ms * 1e-3), ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
  hipFree(d);
}

int main() {
  for (int m : {1, 4, 16, 64}) runi<unsigned int>("ds_add_u32", m);
  for (int m : {1, 4, 16, 64}) runi<unsigned long long>("ds_add_u64", m);
  for (int m : {1, 16}) runi<double>("ds_add_f64", m);
  for (int m : {1, 2, 4, 8, 16, 32, 64}) run<0>("ds_add", m, 1);
  for (int m : {1, 4, 16}) run<0>("ds_add", m, 33);
  for (int m : {1, 4, 16, 64}) run<2>("ds_add_rtn", m, 1);
  for (int m : {1, 16}) run<1>("plain_rmw", m, 1);
  return 0;
}
