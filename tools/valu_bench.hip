// Micro-benchmark (development tool): issue cost of the VALU instruction kinds that dominate the march
// kernels, measured the way the kernels run them -- 256-thread blocks, several waves per SIMD, a mix of
// independent chains.  Prints SIMD cycles per wave-instruction (2.4 GHz assumed).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/valu_bench tools/valu_bench.hip && gpurun_out/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a0 = threadIdx.x * 0.001f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 1.0001f, c = 0.5f;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {          // v_fma_f32, 8 independent chains, 16 x 8 per iteration
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 1) {   // v_cmp + v_cndmask pairs
      REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");)
    } else if (KIND == 2) {   // v_mov_b32
      REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 3) {   // v_add_u32 / v_mul_u32_u24
      REP16(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 4) {   // v_floor_f32 + v_cvt_i32_f32
      REP16(asm volatile("v_floor_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_floor_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_floor_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_floor_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 6) {   // v_pk_fma_f32: two FMAs per lane per instruction
      float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));)
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if (KIND == 7) {   // v_pk_mul_f32 / v_pk_add_f32
      float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
      REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));)
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if (KIND == 8) {   // v_add_f32_dpp quad_perm (the pair / quad pre-reduction of the adjoint's emissions)
      REP16(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 9) {   // v_mov_b32_dpp quad_perm
      REP16(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 10) {  // v_cndmask_b32 with an SGPR-pair mask (VOP3) -- how the compiler writes per-lane selects
      unsigned long long msk = __ballot(threadIdx.x & 1);
      REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %10\n v_cndmask_b32_e64 %1, %1, %9, %10\n v_cndmask_b32_e64 %2, %2, %8, %10\n v_cndmask_b32_e64 %3, %3, %9, %10\n"
                         "v_cndmask_b32_e64 %4, %4, %8, %10\n v_cndmask_b32_e64 %5, %5, %9, %10\n v_cndmask_b32_e64 %6, %6, %8, %10\n v_cndmask_b32_e64 %7, %7, %9, %10\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(msk));)
    } else if (KIND == 11) {  // v_mov_b64 (accumulator pair copies)
      double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
      REP16(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
      a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
    } else if (KIND == 12) {  // v_cmp writing an SGPR pair + s_and (per-lane conditions)
      REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");)
    } else if (KIND == 13) {  // v_lshl_add_u64 (64-bit address arithmetic)
      unsigned long long u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, k = 12345;
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 2, %4\n v_lshl_add_u64 %1, %1, 2, %4\n v_lshl_add_u64 %2, %2, 2, %4\n v_lshl_add_u64 %3, %3, 2, %4\n"
                         "v_lshl_add_u64 %0, %0, 2, %4\n v_lshl_add_u64 %1, %1, 2, %4\n v_lshl_add_u64 %2, %2, 2, %4\n v_lshl_add_u64 %3, %3, 2, %4\n"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(k));)
      a0 += (float)(u0 + u1 + u2 + u3) * 1e-30f;
    } else if (KIND == 5) {   // v_cvt_f64_f32 (result discarded into a 64-bit temp) -- cost of the window conversions
      double d0, d1, d2, d3;
      REP16(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                         : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
      a4 += (float)(d0 + d1 + d2 + d3) * 1e-30f;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static void run(const char* name, int per_iter, int blocks_per_cu) {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  const int blocks = 256 * blocks_per_cu, iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per SIMD: blocks_per_cu waves (one wave of each block lands on each SIMD)
  double instr_per_simd = (double)blocks_per_cu * iters * per_iter;
  printf("%-28s waves/SIMD=%d : %.2f cycles per wave-instruction per SIMD\n", name, blocks_per_cu, ms * 1e-3 * 2.4e9 / instr_per_simd);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32", 128, w);
  for (int w : {2, 8}) run<1>("v_cmp_lt_f32+v_cndmask", 128, w);
  for (int w : {2, 8}) run<2>("v_mov_b32", 128, w);
  for (int w : {2, 8}) run<3>("v_mad_u32_u24/v_add_u32", 128, w);
  for (int w : {2, 8}) run<4>("v_floor_f32/v_cvt_i32_f32", 128, w);
  for (int w : {2, 8}) run<5>("v_cvt_f64_f32", 64, w);
  for (int w : {1, 2, 8}) run<6>("v_pk_fma_f32", 128, w);
  for (int w : {2, 8}) run<7>("v_pk_mul_f32/v_pk_add_f32", 128, w);
  for (int w : {2, 4, 8}) run<8>("v_add_f32_dpp quad_perm", 128, w);
  for (int w : {2, 4, 8}) run<9>("v_mov_b32_dpp quad_perm", 128, w);
  for (int w : {2, 4, 8}) run<10>("v_cndmask_b32_e64 (sgpr mask)", 128, w);
  for (int w : {2, 4, 8}) run<11>("v_mov_b64", 128, w);
  for (int w : {2, 4, 8}) run<12>("v_cmp_lt_f32 -> vcc", 128, w);
  for (int w : {2, 4, 8}) run<13>("v_lshl_add_u64", 128, w);
  for (int w : {4}) run<0>("v_fma_f32", 128, w);
  for (int w : {4}) run<5>("v_cvt_f64_f32", 64, w);
  return 0;
}
