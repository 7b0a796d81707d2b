/*
 * drrt_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU parity oracle for the eikonal ray-march hot path: a plain-C restatement of
 *   /root/reference/src/tracer.cpp, src/volume.cpp, src/cylinder_volume.cpp
 * (line citations inside drrt_oracle_impl.h).  Built as oracle/_build/libdrrt_oracle.so
 * by oracle/Makefile; loaded through ctypes by oracle/oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library -- as the checker / reported baseline, never as the product path.
 *
 * Parity status: "parity unpinned" against the reference's native enoki build (enoki is
 * an empty, un-vendored submodule; the reference ships no tests or golden vectors).
 * What is pinned: see tests/golden/README.md.
 *
 * Compile with -ffp-contract=off so that the only fused multiply-adds are the ones the
 * reference writes explicitly (fmadd / lerp).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#define EXPORT __attribute__((visibility("default")))

/* ---- optional per-ray trajectory signature of the adjoint march (tests only) --------------------
 * When a sink is set, backtrace (literal and factored, f32 and f64) folds the integer cell
 * (floor(p/h) per axis) of every CONTRIBUTING step of ray i into sig[i] (FNV-1a) and counts the steps
 * in nsteps[i].  Two arithmetics that give a ray the same signature walked it through the same cells in
 * the same number of steps: that is how tests separate "tie" rays (a sample within rounding of a cell
 * face lands in different cells under fp32 and fp64) from the rest.                        */
static unsigned long long* g_sig = NULL;
static int* g_sig_steps = NULL;
static inline void sig_visit(size_t i, int ix, int iy, int iz) {
  if (!g_sig) return;
  unsigned long long h = g_sig[i] ? g_sig[i] : 1469598103934665603ULL;
  const unsigned long long k = ((unsigned long long)(unsigned)ix << 42) ^ ((unsigned long long)(unsigned)iy << 21) ^
                               (unsigned long long)(unsigned)iz;
  h = (h ^ k) * 1099511628211ULL;
  g_sig[i] = h ? h : 1;
  g_sig_steps[i] += 1;
}

/* ---- float instantiation ---------------------------------------------------------- */
#define REAL float
#define FN(name) name##_f32
#define FLOOR floorf
#define SQRT sqrtf
#define FMA fmaf
#include "drrt_oracle_impl.h"
#undef REAL
#undef FN
#undef FLOOR
#undef SQRT
#undef FMA
#undef CYL_EPS

/* ---- double instantiation --------------------------------------------------------- */
#define REAL double
#define FN(name) name##_f64
#define FLOOR floor
#define SQRT sqrt
#define FMA fma
#include "drrt_oracle_impl.h"
#undef REAL
#undef FN
#undef FLOOR
#undef SQRT
#undef FMA
#undef CYL_EPS

/* arithmetic mode: 0 = literal (the reference's expression order), 1 = factored (the explicit
 * IEEE sequence the HIP kernels implement; see the second half of drrt_oracle_impl.h) */
EXPORT void oracle_set_trajectory_sink(unsigned long long* sig, int* nsteps) { g_sig = sig; g_sig_steps = nsteps; }
static int g_arith = 0;
EXPORT void oracle_set_arith(int mode) { g_arith = mode ? 1 : 0; }
EXPORT int oracle_get_arith(void) { return g_arith; }

#define DEFINE_API(REAL, SFX)                                                                  \
EXPORT int oracle_trace_##SFX(const REAL* rif, const int* res, long long nvox, size_t n,       \
    const REAL* pos, const REAL* vel, REAL h, REAL ds, REAL* xt, REAL* vt,                     \
    int* steps_out, long long* n_failed, int* iters) {                                         \
  if (g_arith) return trace_fact_##SFX(0, rif, NULL, res, nvox, n, pos, vel, NULL, NULL, h, ds,\
                                       xt, vt, NULL, NULL, steps_out, n_failed, iters);        \
  return trace_generic_##SFX(0, rif, NULL, res, nvox, n, pos, vel, NULL, NULL, h, ds, xt, vt,  \
                             NULL, steps_out, n_failed, iters);                                \
}                                                                                              \
EXPORT int oracle_trace_pln_##SFX(const REAL* rif, const int* res, long long nvox, size_t n,   \
    const REAL* pos, const REAL* vel, const REAL* pln_o, const REAL* pln_d, REAL h, REAL ds,   \
    REAL* xt, REAL* vt, unsigned char* failmask, int* steps_out, long long* n_failed,          \
    int* iters) {                                                                              \
  if (g_arith) return trace_fact_##SFX(1, rif, NULL, res, nvox, n, pos, vel, pln_o, pln_d, h,  \
                                       ds, xt, vt, NULL, failmask, steps_out, n_failed, iters);\
  return trace_generic_##SFX(1, rif, NULL, res, nvox, n, pos, vel, pln_o, pln_d, h, ds, xt, vt,\
                             failmask, steps_out, n_failed, iters);                            \
}                                                                                              \
EXPORT int oracle_trace_sdf_##SFX(const REAL* rif, const REAL* sdf, const int* res,            \
    long long nvox, size_t n, const REAL* pos, const REAL* vel, REAL h, REAL ds,               \
    REAL* xt, REAL* vt, int* steps_out, long long* n_failed, int* iters) {                     \
  if (g_arith) return trace_fact_##SFX(2, rif, sdf, res, nvox, n, pos, vel, NULL, NULL, h, ds, \
                                       xt, vt, NULL, NULL, steps_out, n_failed, iters);        \
  return trace_generic_##SFX(2, rif, sdf, res, nvox, n, pos, vel, NULL, NULL, h, ds, xt, vt,   \
                             NULL, steps_out, n_failed, iters);                                \
}                                                                                              \
EXPORT int oracle_trace_target_##SFX(const REAL* rif, const int* res, long long nvox, size_t n,\
    const REAL* pos, const REAL* vel, const REAL* target, REAL h, REAL ds,                     \
    REAL* xt, REAL* vt, REAL* dist2, long long* n_failed, int* iters) {                        \
  if (g_arith) return trace_fact_##SFX(3, rif, NULL, res, nvox, n, pos, vel, target, NULL, h,  \
                                       ds, xt, vt, dist2, NULL, NULL, n_failed, iters);        \
  return trace_target_impl_##SFX(rif, res, nvox, n, pos, vel, target, h, ds, xt, vt, dist2,    \
                                 n_failed, iters);                                             \
}                                                                                              \
EXPORT int oracle_trace_cable_##SFX(const REAL* rif, size_t rres, REAL radius, REAL length,    \
    size_t n, const REAL* pos, const REAL* vel, const REAL* target, REAL ds,                   \
    REAL* xt, REAL* vt, REAL* dist2, long long* n_failed, long long* steps_total) {            \
  if (g_arith) return trace_cable_fact_##SFX(rif, rres, radius, length, n, pos, vel, target,   \
                                             ds, xt, vt, dist2, n_failed, steps_total);        \
  return trace_cable_impl_##SFX(rif, rres, radius, length, n, pos, vel, target, ds, xt, vt,    \
                                dist2, n_failed, steps_total);                                 \
}                                                                                              \
EXPORT int oracle_backtrace_##SFX(const REAL* rif, const int* res, long long nvox, size_t n,   \
    const REAL* xt, const REAL* vt, const REAL* dx, const REAL* dv, REAL h, REAL ds,           \
    REAL grad_scale, REAL* grad, long long* steps_total) {                                     \
  if (g_arith) return backtrace_fact_##SFX(0, rif, NULL, res, nvox, n, xt, vt, dx, dv, h, ds,  \
                                           grad_scale, grad, steps_total);                     \
  return backtrace_generic_##SFX(0, rif, NULL, res, nvox, n, xt, vt, dx, dv, h, ds,            \
                                 grad_scale, grad, steps_total);                               \
}                                                                                              \
EXPORT int oracle_backtrace_sdf_##SFX(const REAL* rif, const REAL* sdf, const int* res,        \
    long long nvox, size_t n, const REAL* xt, const REAL* vt, const REAL* dx, const REAL* dv,  \
    REAL h, REAL ds, REAL grad_scale, REAL* grad, long long* steps_total) {                    \
  if (g_arith) return backtrace_fact_##SFX(1, rif, sdf, res, nvox, n, xt, vt, dx, dv, h, ds,   \
                                           grad_scale, grad, steps_total);                     \
  return backtrace_generic_##SFX(1, rif, sdf, res, nvox, n, xt, vt, dx, dv, h, ds,             \
                                 grad_scale, grad, steps_total);                               \
}                                                                                              \
EXPORT int oracle_backtrace_cable_##SFX(const REAL* rif, size_t rres, REAL radius, REAL length,\
    size_t n, const REAL* xt, const REAL* vt, const REAL* dx, const REAL* dv, REAL ds,         \
    REAL* grad, long long* steps_total) {                                                      \
  if (g_arith) return backtrace_cable_fact_##SFX(rif, rres, radius, length, n, xt, vt, dx, dv, \
                                                 ds, grad, steps_total);                       \
  return backtrace_cable_impl_##SFX(rif, rres, radius, length, n, xt, vt, dx, dv, ds, grad,    \
                                    steps_total);                                              \
}                                                                                              \
/* point-wise samplers, for pinning against core/grid.py / core/cable.py */                    \
EXPORT int oracle_eval_grad_##SFX(const REAL* data, const int* res, REAL h, size_t n,          \
    const REAL* pts, const unsigned char* mask, REAL* out_n, REAL* out_g) {                    \
  for (size_t i = 0; i < n; ++i)                                                               \
    vol_eval_grad_##SFX(data, res, h, pts + 3*i, mask ? mask[i] : 1, out_n + i, out_g + 3*i);  \
  return 0;                                                                                    \
}                                                                                              \
EXPORT int oracle_eval_hess_##SFX(const REAL* data, const int* res, REAL h, size_t n,          \
    const REAL* pts, const unsigned char* mask, REAL* out_h /* (n,3): xy,xz,yz */) {           \
  for (size_t i = 0; i < n; ++i)                                                               \
    vol_eval_hess_##SFX(data, res, h, pts + 3*i, mask ? mask[i] : 1, out_h + 3*i);             \
  return 0;                                                                                    \
}                                                                                              \
EXPORT int oracle_splat_##SFX(REAL* data, const int* res, REAL h, size_t n, const REAL* pts,   \
    const REAL* val, const REAL* grad, const unsigned char* mask, REAL grad_scale) {           \
  for (size_t i = 0; i < n; ++i)                                                               \
    vol_splat_##SFX(data, res, h, pts + 3*i, val[i], grad + 3*i, mask ? mask[i] : 1,           \
                    grad_scale);                                                               \
  return 0;                                                                                    \
}                                                                                              \
EXPORT int oracle_cyl_eval_grad_##SFX(const REAL* data, size_t rres, REAL radius, size_t n,    \
    const REAL* pts, REAL* out_n, REAL* out_g) {                                               \
  for (size_t i = 0; i < n; ++i)                                                               \
    cyl_eval_grad_##SFX(data, rres, radius, pts + 3*i, out_n + i, out_g + 3*i);                \
  return 0;                                                                                    \
}                                                                                              \
EXPORT int oracle_cyl_eval_hess_##SFX(const REAL* data, size_t rres, REAL radius, size_t n,    \
    const REAL* pts, REAL* out_h /* (n,4): H00,H02,H20,H22 */) {                               \
  for (size_t i = 0; i < n; ++i)                                                               \
    cyl_eval_hess_##SFX(data, rres, radius, pts + 3*i, out_h + 4*i);                           \
  return 0;                                                                                    \
}                                                                                              \
EXPORT int oracle_cyl_splat_##SFX(REAL* data, size_t rres, REAL radius, size_t n,              \
    const REAL* pts, const REAL* val, const REAL* grad, const unsigned char* mask) {           \
  for (size_t i = 0; i < n; ++i)                                                               \
    cyl_splat_##SFX(data, rres, radius, pts + 3*i, val[i], grad + 3*i, mask ? mask[i] : 1);    \
  return 0;                                                                                    \
}

DEFINE_API(float, f32)
DEFINE_API(double, f64)

/* ---- all-cores timing harness (bench.py cpu_baseline_allcores) -----------------------------------
 * Splits the rays into `nthreads` contiguous chunks; every OpenMP thread runs the SAME fp32 routines
 * (trace, then backtrace with dx = dv = 1 into a private grid) on its chunk; the private grids are
 * summed at the end.  Reports wall seconds of the two phases and the forward ray-step count.       */
#ifdef _OPENMP
#include <omp.h>
#endif
EXPORT int oracle_bench_allcores_f32(const float* rif, const int* res, long long nvox, size_t n,
    const float* pos, const float* vel, float h, float ds, int nthreads, float* grad_out,
    double* t_fwd, double* t_adj, long long* fwd_steps, int* threads_used,
    float* xt_out, float* vt_out, int* steps_out, long long* adj_steps) {
#ifndef _OPENMP
  (void)rif; (void)res; (void)nvox; (void)n; (void)pos; (void)vel; (void)h; (void)ds; (void)nthreads;
  (void)grad_out; (void)t_fwd; (void)t_adj; (void)fwd_steps; (void)threads_used;
  (void)xt_out; (void)vt_out; (void)steps_out; (void)adj_steps;
  return -9;
#else
  if (nthreads < 1) nthreads = omp_get_max_threads();
  float* xt = (float*)malloc(sizeof(float) * 3 * n);
  float* vt = (float*)malloc(sizeof(float) * 3 * n);
  float* ones = (float*)malloc(sizeof(float) * 3 * n);
  int* steps = (int*)malloc(sizeof(int) * (n ? n : 1));
  for (size_t i = 0; i < 3 * n; ++i) ones[i] = 1.0f;
  float* grads = (float*)calloc((size_t)nthreads * (size_t)nvox, sizeof(float));
  if (!xt || !vt || !ones || !steps || !grads) return -8;
  int rc_all = 0;
  double t0 = omp_get_wtime();
#pragma omp parallel num_threads(nthreads) reduction(|:rc_all)
  {
    int t = omp_get_thread_num(), T = omp_get_num_threads();
    size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
    long long nf; int it;
    rc_all |= oracle_trace_f32(rif, res, nvox, hi - lo, pos + 3 * lo, vel + 3 * lo, h, ds, xt + 3 * lo, vt + 3 * lo,
                               steps + lo, &nf, &it);
  }
  double t1 = omp_get_wtime();
  int used = 0;
  long long ast = 0;
#pragma omp parallel num_threads(nthreads) reduction(|:rc_all) reduction(+:ast)
  {
    int t = omp_get_thread_num(), T = omp_get_num_threads();
    size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
    long long st = 0;
    rc_all |= oracle_backtrace_f32(rif, res, nvox, hi - lo, xt + 3 * lo, vt + 3 * lo, ones + 3 * lo, ones + 3 * lo,
                                   h, ds, 1.0f, grads + (size_t)t * (size_t)nvox, &st);
    ast += st;
#pragma omp single
    used = T;
  }
  if (grad_out) {
#pragma omp parallel for num_threads(nthreads)
    for (long long k = 0; k < nvox; ++k) {
      float acc = 0.f;
      for (int t = 0; t < used; ++t) acc += grads[(size_t)t * (size_t)nvox + (size_t)k];
      grad_out[k] = acc;
    }
  }
  double t2 = omp_get_wtime();
  long long fs = 0;
  for (size_t i = 0; i < n; ++i) fs += steps[i];
  *t_fwd = t1 - t0; *t_adj = t2 - t1; *fwd_steps = fs; *threads_used = used;
  if (adj_steps) *adj_steps = ast;
  if (xt_out) memcpy(xt_out, xt, sizeof(float) * 3 * n);
  if (vt_out) memcpy(vt_out, vt, sizeof(float) * 3 * n);
  if (steps_out) memcpy(steps_out, steps, sizeof(int) * n);
  free(xt); free(vt); free(ones); free(steps); free(grads);
  return rc_all;
#endif
}
