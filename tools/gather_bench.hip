// gather_bench.hip -- microbenchmark behind the "texture addresser bound" statement for the forward
// march: 1M threads walk 512 half-cell steps along y through a 256^3 grid, Z-ordered
// in (x,z) like the sorted ray bundles, and fetch the 8 corners of their cell either as
//   A: four 8-byte pair gathers from the plain fp32 grid (what k_trace does), or
//   B: two 16-byte gathers from a "quad" copy q[z][y][x] = {n(x,y,z), n(x+1,y,z), n(x,y+1,z), n(x+1,y+1,z)},
//   C: eight 4-byte gathers (the naive form).
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o gpurun_out/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int R = 256;

__device__ __forceinline__ unsigned compact(unsigned v) {   // even bits of v
  v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu;
  v = (v | (v >> 4)) & 0x00ff00ffu; v = (v | (v >> 8)) & 0x0000ffffu; return v;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_walk(const float* __restrict__ g, const float4* __restrict__ q, float* out, int steps) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  const unsigned ix = compact(t), iz = compact(t >> 1);       // 0..1023 each
  const int x = min((int)(ix >> 2), R - 2), z = min((int)(iz >> 2), R - 2);
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    const int y = min(s >> 1, R - 2);
    const int base = (z * R + y) * R + x;
    if (MODE == 0) {
      const float2 a = *(const float2*)(g + base), b = *(const float2*)(g + base + R);
      const float2 c = *(const float2*)(g + base + R * R), d = *(const float2*)(g + base + R * R + R);
      acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (d.x + d.y);
    } else if (MODE == 1) {
      const float4 a = q[base], b = q[base + R * R];
      acc += (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
    } else {
      acc += g[base] + g[base + 1] + g[base + R] + g[base + R + 1] + g[base + R * R] + g[base + R * R + 1] +
             g[base + R * R + R] + g[base + R * R + R + 1];
    }
    acc = acc * 0.999f;
  }
  out[t] = acc;
}

__global__ void k_quad(const float* __restrict__ g, float4* __restrict__ q) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int x = i % R, y = (i / R) % R;
  const int x1 = x + 1 < R ? 1 : 0, y1 = y + 1 < R ? R : 0;
  q[i] = make_float4(g[i], g[i + x1], g[i + y1], g[i + y1 + x1]);
}

int main() {
  const size_t nv = (size_t)R * R * R;
  float* g; float4* q; float* out;
  hipMalloc(&g, nv * 4 + 4096); hipMalloc(&q, nv * 16 + 4096); hipMalloc(&out, (1 << 20) * 4);
  std::vector<float> h(nv);
  for (size_t i = 0; i < nv; ++i) h[i] = 1.0f + (float)(i % 977) * 1e-4f;
  hipMemcpy(g, h.data(), nv * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k_quad, dim3(nv / 256), dim3(256), 0, 0, g, q); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("build quad copy: %.3f ms\n", ms);
    hipEventRecord(e0); hipLaunchKernelGGL(k_walk<0>, dim3(4096), dim3(256), 0, 0, g, q, out, 512); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("A 4 x dwordx2 : %.3f ms\n", ms);
    hipEventRecord(e0); hipLaunchKernelGGL(k_walk<1>, dim3(4096), dim3(256), 0, 0, g, q, out, 512); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("B 2 x dwordx4 : %.3f ms\n", ms);
    hipEventRecord(e0); hipLaunchKernelGGL(k_walk<2>, dim3(4096), dim3(256), 0, 0, g, q, out, 512); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("C 8 x dword   : %.3f ms\n", ms);
  }
  float o; hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost); printf("check %f\n", o);
  return 0;
}
