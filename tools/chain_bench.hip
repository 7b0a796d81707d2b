// chain_bench.hip -- development microbenchmark: how a wave's dependent gathers behave on gfx950.
// Every lane walks `steps` cells along y through a 256^3 fp32 grid (lanes Z-ordered in (x,z) like sorted ray bundles);
// the next cell index depends on the data just loaded (a serial chain, like the march).  G independent chains per lane,
// issued back to back.  Grid sizes: "lone" = one wave per SIMD (1024 waves of 64), "full" = 16384 waves.
// Build: hipcc -O3 --offload-arch=gfx950 tools/chain_bench.hip -o gpurun_out/chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int R = 256;
__device__ __forceinline__ unsigned compact(unsigned v) {
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu;
  v = (v | (v >> 4)) & 0x00ff00ffu; v = (v | (v >> 8)) & 0x0000ffffu; return v;
}
typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
template <int G, int LOADS, int ALU>   // LOADS: 4 = four dwordx2 per cell, 0 = no loads at all; ALU = dependent fmas per step
__global__ void __launch_bounds__(64) k_chain(const float* __restrict__ g, float* out, int steps, int lanes_mask) {
  const unsigned t = blockIdx.x * 64 + threadIdx.x;
  int base[G]; float acc[G];
#pragma unroll
  for (int c = 0; c < G; ++c) {
    const unsigned u = t * G + c;
    const int x = min((int)(compact(u) >> 2), R - 2), z = min((int)(compact(u >> 1) >> 2), R - 2);
    base[c] = (z * R) * R + x; acc[c] = 1.0f;
  }
  const bool on = (threadIdx.x & lanes_mask) == 0;     // lanes_mask = 0: all lanes load; 1: every other lane; 63: one lane
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int c = 0; c < G; ++c) {
      float v = acc[c];
      if (LOADS) {
        if (on) {
          const float* p = g + base[c];
          const f2 a = *(const f2*)p, b = *(const f2*)(p + R), e = *(const f2*)(p + R * R), f = *(const f2*)(p + R * R + R);
          v += (a.x + a.y) + (b.x + b.y) + (e.x + e.y) + (f.x + f.y);
        }
      }
#pragma unroll
      for (int k = 0; k < ALU; ++k) v = fmaf(v, 0.999f, 1e-3f);
      acc[c] = v;
      // next cell: one step in y every second iteration, plus a data-dependent zero (keeps the chain serial)
      base[c] += ((s & 1) ? R : 0) + (int)(v * 1e-30f);
      base[c] = min(base[c], R * R * R - R * R - R - 2);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < G; ++c) sum += acc[c];
  out[t] = sum;
}
template <int G, int LOADS, int ALU>
static void run(const char* name, const float* g, float* out, int waves, int lanes_mask) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int steps = 512;
  hipLaunchKernelGGL((k_chain<G, LOADS, ALU>), dim3(waves), dim3(64), 0, 0, g, out, 16, lanes_mask);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<G, LOADS, ALU>), dim3(waves), dim3(64), 0, 0, g, out, steps, lanes_mask);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves %6d mask %2d : %7.3f ms  = %7.1f ns per iteration (%d chain-steps)\n", name, waves, lanes_mask, ms,
         ms * 1e6 / steps, G);
}
int main() {
  const size_t nv = (size_t)R * R * R;
  float* g; float* out;
  hipMalloc(&g, nv * 4 + 4096); hipMalloc(&out, (1 << 21) * 4);
  std::vector<float> h(nv);
  for (size_t i = 0; i < nv; ++i) h[i] = 1.0f + (float)(i % 977) * 1e-4f;
  hipMemcpy(g, h.data(), nv * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    for (int waves : {1024, 4096, 16384}) {
      run<1, 0, 8>("G1 no loads, 8 dependent fma", g, out, waves, 0);
      run<1, 0, 64>("G1 no loads, 64 dependent fma", g, out, waves, 0);
      run<1, 4, 8>("G1 4 x dwordx2, 8 fma", g, out, waves, 0);
      run<2, 4, 8>("G2 4 x dwordx2, 8 fma", g, out, waves / 2, 0);
      run<4, 4, 8>("G4 4 x dwordx2, 8 fma", g, out, waves / 4, 0);
      run<1, 4, 8>("G1 4 x dwordx2, every 2nd lane", g, out, waves, 1);
      run<1, 4, 8>("G1 4 x dwordx2, every 4th lane", g, out, waves, 3);
      run<1, 4, 8>("G1 4 x dwordx2, one lane", g, out, waves, 63);
      run<1, 4, 64>("G1 4 x dwordx2, 64 fma", g, out, waves, 0);
    }
  }
  float o; hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost); printf("check %f\n", o);
  return 0;
}
