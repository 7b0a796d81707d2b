// drrt_cable.hip -- gfx950 kernels of the radial-profile (fibre) march and its adjoint: Tracer::trace_cable,
// backtrace_cable (/root/reference/src/tracer.cpp:312-382, 511-567; src/cylinder_volume.cpp).
#include "drrt_march.h"

namespace drrt {

__device__ __forceinline__ void cable_stats(drrt_stats* stats, unsigned steps_tot, unsigned steps_max, unsigned fail_tot) {
  if (!stats) return;
  unsigned wm = wave_max_u32(steps_max), ws = wave_sum_u32(steps_tot), wf = wave_sum_u32(fail_tot);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    if (wm) atomicMax(&stats->iters, wm);
    if (ws) atomicAdd(&stats->ray_steps, (unsigned long long)ws);
    if (wf) atomicAdd(&stats->n_failed, (unsigned long long)wf);
  }
}

__global__ void __launch_bounds__(kBlock) k_trace_cable(CableArgs a) {
  extern __shared__ float s_prof[];
  const bool use_lds = a.rres <= kCableMaxRes;
  if (use_lds) {
    for (int k = threadIdx.x; k < a.rres; k += kBlock) s_prof[k] = a.rif[k];
    __syncthreads();
  }
  const Cyl C = make_cyl(use_lds ? s_prof : a.rif, a.rres, a.radius, a.length);
  unsigned steps_tot = 0, fail_tot = 0, steps_max = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (size_t)gridDim.x * kBlock) {
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i), tg = ld3(a.target, i);
    const float pp[3] = {p.x, p.y, p.z}, vv[3] = {u.x, u.y, u.z}, tt[3] = {tg.x, tg.y, tg.z};
    RayOut r = cable_trace_ray(C, a.ds, a.max_steps, pp, vv, tt);
    st3(a.xt, i, r.xt[0], r.xt[1], r.xt[2]); st3(a.vt, i, r.vt[0], r.vt[1], r.vt[2]); a.dist2[i] = r.dist2;
    steps_tot += r.steps; steps_max = max(steps_max, r.steps); fail_tot += r.esc ? 0u : 1u;
  }
  cable_stats(a.stats, steps_tot, steps_max, fail_tot);
}

__global__ void __launch_bounds__(kBlock) k_backtrace_cable(CableArgs a) {
  // LDS: the profile (floats) followed by the gradient accumulators (doubles: ds_add_f64 is ~25x
  // cheaper than ds_add_f32 on gfx950, tools/lds_atomic_bench.hip, and sums are more accurate)
  extern __shared__ double s_mem64[];
  const bool use_lds = a.rres <= kCableMaxRes;
  double* s_grad = s_mem64;
  float* s_prof = reinterpret_cast<float*>(s_mem64 + (use_lds ? a.rres : 0));
  if (use_lds) {
    for (int k = threadIdx.x; k < a.rres; k += kBlock) { s_prof[k] = a.rif[k]; s_grad[k] = 0.0; }
    __syncthreads();
  }
  const Cyl C = make_cyl(use_lds ? s_prof : a.rif, a.rres, a.radius, a.length);
  float* gacc = a.grad;
  unsigned steps_tot = 0, steps_max = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (size_t)gridDim.x * kBlock) {
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i), gxv = ld3(a.dx, i), gvv = ld3(a.dv, i);
    const float pp[3] = {p.x, p.y, p.z}, vv[3] = {u.x, u.y, u.z};
    const float dxx[3] = {gxv.x, gxv.y, gxv.z}, dvv[3] = {gvv.x, gvv.y, gvv.z};
    unsigned steps = cable_backtrace_ray(C, a.ds, a.max_steps, pp, vv, dxx, dvv,
      [s_grad, gacc, use_lds](int i0, int i1, float a0, float a1) {
        if (use_lds) {
          // Rays in source-pixel order sit at nearly the same radius as their neighbours: lanes of a pair / quad
          // then hit the same two bins, and a pair / quad pre-reduction (as in shift_emit4) halves the time.  For
          // rays in random order it would only cost instructions, so it runs when at least a quarter of the wave's
          // lanes have a matching partner (wave-uniform decision).
          const int key = i0 | (i1 << 16);
          const int k1 = __builtin_amdgcn_update_dpp(-1, key, 0xB1, 0xF, 0xF, false);
          const bool psame = k1 == key;
          if (__popcll(__ballot(psame)) >= 16) {
            const int k2 = __builtin_amdgcn_update_dpp(-1, key, 0x4E, 0xF, 0xF, false);
            const int k3 = __builtin_amdgcn_update_dpp(-1, key, 0x1B, 0xF, 0xF, false);
            const bool same = psame & (k2 == key) & (k3 == key);
            const float p0 = a0 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0xB1, 0xF, 0xF, false));
            const float p1 = a1 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0xB1, 0xF, 0xF, false));
            const float s0 = p0 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p0), 0x4E, 0xF, 0xF, false));
            const float s1 = p1 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p1), 0x4E, 0xF, 0xF, false));
            const unsigned ql = threadIdx.x & 3u;
            if (same ? ql == 0u : (psame ? (ql & 1u) == 0u : true)) {
              atomicAdd(&s_grad[i0], (double)(same ? s0 : (psame ? p0 : a0)));                   // ds_add_f64
              atomicAdd(&s_grad[i1], (double)(same ? s1 : (psame ? p1 : a1)));
            }
          } else {
            atomicAdd(&s_grad[i0], (double)a0); atomicAdd(&s_grad[i1], (double)a1);
          }
        } else { atomic_add_f32(&gacc[i0], a0); atomic_add_f32(&gacc[i1], a1); }
      });
    steps_tot += steps; steps_max = max(steps_max, steps);
  }
  if (use_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < a.rres; k += kBlock) {
      double g = s_grad[k];
      if (g != 0.0) atomic_add_f32(&a.grad[k], (float)g);
    }
  }
  cable_stats(a.stats, steps_tot, steps_max, 0u);
}

// ---- launchers ----------------------------------------------------------------------------------
static unsigned cable_grid(size_t n) {
  // grid-stride: at most 4 blocks per CU so that the per-block LDS gradient flush stays small
  const unsigned want = grid_for(n);
  return want < 1024u ? want : 1024u;
}
void launch_trace_cable(const CableArgs& a, hipStream_t s) {
  const size_t lds = (a.rres <= kCableMaxRes) ? a.rres * sizeof(float) : 0;
  hipLaunchKernelGGL(k_trace_cable, dim3(cable_grid(a.n)), dim3(kBlock), lds, s, a);
}
void launch_backtrace_cable(const CableArgs& a, hipStream_t s) {
  const size_t lds = (a.rres <= kCableMaxRes) ? a.rres * (sizeof(double) + sizeof(float)) : 0;
  hipLaunchKernelGGL(k_backtrace_cable, dim3(cable_grid(a.n)), dim3(kBlock), lds, s, a);
}

}  // namespace drrt
