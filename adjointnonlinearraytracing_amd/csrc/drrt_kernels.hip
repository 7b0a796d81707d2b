// drrt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the eikonal ray march and its adjoint,
// plus the C ABI declared in include/drrt_hip.h.
//
// Reference semantics: /root/reference/src/tracer.cpp (march loops :35-382, adjoint loops
// :384-567), src/volume.cpp, src/cylinder_volume.cpp.  Quirk numbers (Q1..Q16) refer to
// SURVEY.md section 8.1.
//
// Execution model (MI355X-first, not the reference's array-at-a-time enoki JIT):
//   * one ray per lane, wave64; the WHOLE march of a ray runs in registers inside one kernel
//     (the reference launches one fused kernel + one reduction + one host sync PER STEP and
//     streams x, v, xt, vt and masks through DRAM every step);
//   * per-ray termination: a ray stops as soon as it is flagged escaped.  This is exact for
//     trace / trace_plane / trace_sdf / backtrace*: an escaped ray flies straight outside the
//     convex box, can never produce another `cross`, and the adjoint masks every contribution
//     with `active` (proof sketch in DESIGN.md).  trace_target is the one variant whose result
//     depends on the GLOBAL loop count (its closest-approach update is not gated by `escaped`,
//     src/tracer.cpp:225-227), so it runs as two kernels around a device-side max reduction;
//   * rays are visited through an optional permutation (locality sort by entry voxel,
//     drrt_sort.hip) so that the 64 lanes of a wave touch a handful of 128-B lines per tap.
//
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/drrt_hip.h"
#include "drrt_device.h"

namespace drrt {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// stats: block-level reduction, then 3 atomics per block
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_stats(drrt_stats* stats, unsigned steps, unsigned failed) {
  if (!stats) return;
  __shared__ unsigned s_sum[kBlock / kWave], s_max[kBlock / kWave], s_fail[kBlock / kWave];
  unsigned ws = wave_sum_u32(steps), wm = wave_max_u32(steps), wf = wave_sum_u32(failed);
  int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) { s_sum[wid] = ws; s_max[wid] = wm; s_fail[wid] = wf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum = 0, fail = 0; unsigned mx = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) { sum += s_sum[w]; fail += s_fail[w]; mx = max(mx, s_max[w]); }
    if (sum)  atomicAdd(&stats->ray_steps, sum);
    if (fail) atomicAdd(&stats->n_failed, fail);
    if (mx)   atomicMax(&stats->iters, mx);
  }
}

struct Ray3 { float x, y, z; };
__device__ __forceinline__ Ray3 ld3(const float* p, size_t i) { return Ray3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(float* p, size_t i, float a, float b, float c) {
  p[3 * i] = a; p[3 * i + 1] = b; p[3 * i + 2] = c;
}

// ---------------------------------------------------------------------------------------------
// forward march: trace (MODE 0), trace_plane (MODE 1), trace_sdf (MODE 2)
// ---------------------------------------------------------------------------------------------
struct TraceArgs {
  Vol vol;
  const float* sdf;            // MODE 2
  const float* pos; const float* vel;
  const float* pln_o; const float* pln_d;   // MODE 1
  float* xt; float* vt; uint8_t* failmask;
  const uint32_t* perm;        // nullable: visit order
  drrt_stats* stats;
  size_t n;
  float ds;
  int max_steps;
};

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_trace(TraceArgs a) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned steps = 0, failed = 0;
  if (t < a.n) {
    const size_t i = a.perm ? (size_t)a.perm[t] : t;
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i);
    const float pp[3] = {p.x, p.y, p.z}, vv[3] = {u.x, u.y, u.z};
    float po[3] = {0.f, 0.f, 0.f}, pd[3] = {0.f, 0.f, 0.f};
    if (MODE == 1) {
      Ray3 o = ld3(a.pln_o, i), d = ld3(a.pln_d, i);
      po[0] = o.x; po[1] = o.y; po[2] = o.z; pd[0] = d.x; pd[1] = d.y; pd[2] = d.z;
    }
    RayOut r = trace_ray<MODE>(a.vol, a.sdf, a.ds, a.max_steps, pp, vv, po, pd);
    steps = r.steps; failed = r.act ? 1u : 0u;
    st3(a.xt, i, r.xt[0], r.xt[1], r.xt[2]);
    st3(a.vt, i, r.vt[0], r.vt[1], r.vt[2]);
    if (MODE == 1) a.failmask[i] = r.esc ? 0 : 1;                           // src/tracer.cpp:171
  }
  block_stats(a.stats, steps, failed);
}

// ---------------------------------------------------------------------------------------------
// trace_target (src/tracer.cpp:174-242): phase A marches until escaped, phase B continues the
// (now straight) flight up to the global iteration count, tracking the closest approach.
// ---------------------------------------------------------------------------------------------
struct TargetArgs {
  Vol vol;
  const float* pos; const float* vel; const float* target;
  float* xt; float* vt; float* dist2;
  float* state;                // workspace: n * 7 floats (x,v,steps) -- SoA
  const uint32_t* perm;
  drrt_stats* stats;
  size_t n;
  float ds;
  int max_steps;
};

__global__ void __launch_bounds__(kBlock) k_target_a(TargetArgs a) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned steps = 0, failed = 0;
  if (t < a.n) {
    const size_t i = a.perm ? (size_t)a.perm[t] : t;
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i), tg = ld3(a.target, i);
    const float pp[3] = {p.x, p.y, p.z}, vv[3] = {u.x, u.y, u.z}, tt[3] = {tg.x, tg.y, tg.z};
    float cont[6];
    RayOut r = target_ray_a(a.vol, a.ds, a.max_steps, pp, vv, tt, cont);
    steps = r.steps; failed = r.esc ? 0u : 1u;
    st3(a.xt, i, r.xt[0], r.xt[1], r.xt[2]); st3(a.vt, i, r.vt[0], r.vt[1], r.vt[2]); a.dist2[i] = r.dist2;
    float* w = a.state;
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k * a.n + i] = cont[k];
    w[6 * a.n + i] = __uint_as_float(steps);
  }
  block_stats(a.stats, steps, failed);
}

__global__ void __launch_bounds__(kBlock) k_target_b(TargetArgs a) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.n) return;
  const unsigned total = a.stats->iters;         // written by phase A (stream-ordered)
  const float* w = a.state;
  const unsigned done = __float_as_uint(w[6 * a.n + i]);
  if (done >= total) return;
  float cont[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) cont[k] = w[k * a.n + i];
  Ray3 tg = ld3(a.target, i);
  const float tt[3] = {tg.x, tg.y, tg.z};
  float best = a.dist2[i], xt[3], vt[3];
  if (target_ray_b(a.ds, done, total, cont, tt, best, xt, vt)) {
    st3(a.xt, i, xt[0], xt[1], xt[2]); st3(a.vt, i, vt[0], vt[1], vt[2]); a.dist2[i] = best;
  }
}

// ---------------------------------------------------------------------------------------------
// adjoint march: backtrace (MODE 0), backtrace_sdf (MODE 1); direct global atomics variant
// ---------------------------------------------------------------------------------------------
struct BackArgs {
  Vol vol;
  const float* sdf;
  const float* xt; const float* vt; const float* dx; const float* dv;
  float* grad;
  const uint32_t* perm;
  drrt_stats* stats;
  size_t n;
  float ds;
  float grad_scale;            // 1 (as written, Q3) or 1/h (DRRT_FLAG_CORRECTED_H)
  int max_steps;
};

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_backtrace_direct(BackArgs a) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned steps = 0;
  if (t < a.n) {
    const size_t i = a.perm ? (size_t)a.perm[t] : t;
    Ray3 p = ld3(a.xt, i), u = ld3(a.vt, i), gxv = ld3(a.dx, i), gvv = ld3(a.dv, i);
    const float pp[3] = {p.x, p.y, p.z}, vv[3] = {u.x, u.y, u.z};
    const float dxx[3] = {gxv.x, gxv.y, gxv.z}, dvv[3] = {gvv.x, gvv.y, gvv.z};
    float* grad = a.grad;
    steps = backtrace_ray<MODE>(a.vol, a.sdf, a.ds, a.grad_scale, a.max_steps, pp, vv, dxx, dvv,
      [grad](const Cell& c, const Corners& w) {
        float* g = grad + c.base;
        atomic_add_f32(g, w.c000);                    atomic_add_f32(g + c.ox, w.c100);
        atomic_add_f32(g + c.oy, w.c010);             atomic_add_f32(g + c.oy + c.ox, w.c110);
        atomic_add_f32(g + c.oz, w.c001);             atomic_add_f32(g + c.oz + c.ox, w.c101);
        atomic_add_f32(g + c.oz + c.oy, w.c011);      atomic_add_f32(g + c.oz + c.oy + c.ox, w.c111);
      });
  }
  block_stats(a.stats, steps, 0u);
}

// ---------------------------------------------------------------------------------------------
// cable (radial profile) variants, src/tracer.cpp:312-382 and :511-567
// The profile (<= a few hundred floats) lives in LDS; the adjoint accumulates into an LDS copy
// of the gradient profile (ds_add_f32) and flushes it once per block -- millions of rays would
// otherwise hammer <= 257 global addresses.
// ---------------------------------------------------------------------------------------------
constexpr int kCableMaxRes = 4096;   // profiles larger than this fall back to global memory

struct CableArgs {
  const float* rif; int rres; float radius, length, ds; int max_steps;
  const float* pos; const float* vel; const float* target;    // forward
  const float* dx; const float* dv;                           // adjoint (pos=xt, vel=vt)
  float* xt; float* vt; float* dist2; float* grad;
  drrt_stats* stats;
  size_t n;
};

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
  extern __shared__ float s_mem[];
  const bool use_lds = a.rres <= kCableMaxRes;
  float* s_prof = s_mem;
  float* s_grad = s_mem + (use_lds ? a.rres : 0);
  if (use_lds) {
    for (int k = threadIdx.x; k < a.rres; k += kBlock) { s_prof[k] = a.rif[k]; s_grad[k] = 0.f; }
    __syncthreads();
  }
  const Cyl C = make_cyl(use_lds ? s_prof : a.rif, a.rres, a.radius, a.length);
  float* acc = use_lds ? s_grad : a.grad;
  unsigned steps_tot = 0, steps_max = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (size_t)gridDim.x * kBlock) {
    Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i), gxv = ld3(a.dx, i), gvv = ld3(a.dv, i);
    const float pp[3] = {p.x, p.y, p.z}, vv[3] = {u.x, u.y, u.z};
    const float dxx[3] = {gxv.x, gxv.y, gxv.z}, dvv[3] = {gvv.x, gvv.y, gvv.z};
    unsigned steps = cable_backtrace_ray(C, a.ds, a.max_steps, pp, vv, dxx, dvv,
      [acc, use_lds](int i0, int i1, float a0, float a1) {
        if (use_lds) { atomicAdd(&acc[i0], a0); atomicAdd(&acc[i1], a1); }      // ds_add_f32
        else { atomic_add_f32(&acc[i0], a0); atomic_add_f32(&acc[i1], a1); }
      });
    steps_tot += steps; steps_max = max(steps_max, steps);
  }
  if (use_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < a.rres; k += kBlock) {
      float g = s_grad[k];
      if (g != 0.f) atomic_add_f32(&a.grad[k], g);
    }
  }
  cable_stats(a.stats, steps_tot, steps_max, 0u);
}

}  // namespace drrt

// =============================================================================================
// host side: C ABI
// =============================================================================================
using namespace drrt;

// from drrt_sort.hip
namespace drrt {
size_t sort_workspace_bytes(size_t n);
hipError_t sort_rays_by_entry_voxel(const Vol& V, float h, size_t n, const float* pos, const float* vel,
                                    float dir_sign, void* ws, size_t ws_bytes, const uint32_t** perm_out,
                                    hipStream_t stream);
}

static thread_local char g_err[512] = "";

static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
static int fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return DRRT_ERR_HIP;
}

extern "C" const char* drrt_last_error(void) { return g_err; }

// ---- optional per-kernel timing (bench / profiling aid; not thread-safe) --------------------
// Event pairs are recorded on the call's stream right around a kernel launch; nothing
// synchronises until drrt_profile_collect().
struct ProfRec { hipEvent_t a, b; int id; };
static ProfRec* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;

struct ProfScope {
  int slot; hipStream_t s;
  ProfScope(int id, hipStream_t st) : slot(-1), s(st) {
    if (g_prof && g_prof_n < g_prof_cap) {
      slot = g_prof_n++;
      g_prof[slot].id = id;
      (void)hipEventRecord(g_prof[slot].a, s);
    }
  }
  ~ProfScope() { if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, s); }
};

extern "C" void drrt_profile_end(void) {
  for (int i = 0; i < g_prof_cap; ++i) { (void)hipEventDestroy(g_prof[i].a); (void)hipEventDestroy(g_prof[i].b); }
  delete[] g_prof; g_prof = nullptr; g_prof_cap = g_prof_n = 0;
}

extern "C" int drrt_profile_begin(int capacity) {
  drrt_profile_end();
  if (capacity <= 0) return DRRT_OK;
  g_prof = new ProfRec[capacity];
  for (int i = 0; i < capacity; ++i) {
    hipError_t e = hipEventCreate(&g_prof[i].a);
    if (e == hipSuccess) e = hipEventCreate(&g_prof[i].b);
    if (e != hipSuccess) { g_prof_cap = i; drrt_profile_end(); return fail_hip(e, "hipEventCreate"); }
  }
  g_prof_cap = capacity; g_prof_n = 0;
  return DRRT_OK;
}

extern "C" int drrt_profile_collect(int* ids, float* ms, int max_out) {
  int n = g_prof_n < max_out ? g_prof_n : max_out;
  for (int i = 0; i < n; ++i) {
    (void)hipEventSynchronize(g_prof[i].b);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b);
    ids[i] = g_prof[i].id; ms[i] = t;
  }
  g_prof_n = 0;
  return n;
}
extern "C" const char* drrt_version(void) { return "drrt_hip 0.1 gfx950"; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t drrt_workspace_bytes(size_t n, unsigned flags) {
  size_t b = 0;
  if (flags & DRRT_FLAG_SORT_RAYS) b += align_up(sort_workspace_bytes(n), 256);
  b += align_up(n * 7 * sizeof(float), 256);       // trace_target state (cheap; always counted)
  return b;
}

// volume ctor checks: src/volume.cpp:31-38 (size) and :123-124 (width/height >= 2)
static int make_vol(const float* rif, long long nvox, const int res[3], float h, Vol* V) {
  if (!rif || !res) return fail(DRRT_ERR_ARG, "null rif/res pointer");
  if ((long long)res[0] * res[1] * res[2] != nvox || nvox <= 0)
    return fail(DRRT_ERR_RES_MISMATCH, "Resolution doesn't match data");
  if (!(res[0] == 1 && res[1] == 1 && res[2] == 1) && (res[0] < 2 || res[1] < 2))
    return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  if (nvox > 0x7fffffffLL) return fail(DRRT_ERR_ARG, "grid too large for int32 indexing");
  V->data = rif; V->W = res[0]; V->H = res[1]; V->D = res[2];
  V->sy = res[0]; V->sz = res[0] * res[1];
  V->inv_h = 1.0f / h; V->inv_h2 = V->inv_h * V->inv_h;
  V->bx = (float)(res[0] - 1) * h; V->by = (float)(res[1] - 1) * h; V->bz = (float)(res[2] - 1) * h;
  return DRRT_OK;
}

static inline int max3(const int r[3]) { return r[0] > r[1] ? (r[0] > r[2] ? r[0] : r[2]) : (r[1] > r[2] ? r[1] : r[2]); }

// float expression truncated to int, exactly as written in the reference (Q5)
static inline int steps_fwd(float h, const int res[3], float ds)  { return (int)(4.0f * h * (float)max3(res) / ds); }
static inline int steps_sdf(float h, const int res[3], float ds)  { return (int)(2.0f * h * (float)max3(res) / ds); }
static inline int steps_adj(float h, const int res[3], float ds)  { return (int)(2.0f * h * (float)max3(res) / ds); }

static int zero_stats(drrt_stats* stats, hipStream_t s) {
  if (!stats) return DRRT_OK;
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(drrt_stats), s);
  return e == hipSuccess ? DRRT_OK : fail_hip(e, "hipMemsetAsync(stats)");
}

static int maybe_sort(const Vol& V, float h, size_t n, const float* pos, const float* vel, float dir_sign,
                      unsigned flags, void* ws, size_t ws_bytes, const uint32_t** perm, hipStream_t s) {
  *perm = nullptr;
  if (!(flags & DRRT_FLAG_SORT_RAYS) || n < 2) return DRRT_OK;
  if (!ws || ws_bytes < sort_workspace_bytes(n)) return fail(DRRT_ERR_ARG, "workspace too small for DRRT_FLAG_SORT_RAYS");
  ProfScope prof(DRRT_PROF_SORT, s);
  hipError_t e = sort_rays_by_entry_voxel(V, h, n, pos, vel, dir_sign, ws, ws_bytes, perm, s);
  return e == hipSuccess ? DRRT_OK : fail_hip(e, "sort_rays_by_entry_voxel");
}

#define LAUNCH_CHECK(where)                                              \
  do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail_hip(e_, where); } while (0)

static inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

template <int MODE>
static int run_trace(const float* rif, const float* sdf, long long nvox, const int res[3], size_t n,
                     const float* pos, const float* vel, const float* pln_o, const float* pln_d,
                     float h, float ds, float* xt, float* vt, uint8_t* failmask, drrt_stats* stats,
                     void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  TraceArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  if (n == 0) return zero_stats(stats, s);
  if (!pos || !vel || !xt || !vt) return fail(DRRT_ERR_ARG, "null ray pointer");
  if (MODE == 1 && (!pln_o || !pln_d || !failmask)) return fail(DRRT_ERR_ARG, "null plane/failmask pointer");
  if (MODE == 2 && !sdf) return fail(DRRT_ERR_ARG, "null sdf pointer");
  if (n > 0xffffffffULL) return fail(DRRT_ERR_ARG, "too many rays for uint32 permutation");
  rc = zero_stats(stats, s); if (rc) return rc;
  rc = maybe_sort(a.vol, h, n, pos, vel, 1.f, flags, ws, ws_bytes, &a.perm, s); if (rc) return rc;
  a.sdf = sdf; a.pos = pos; a.vel = vel; a.pln_o = pln_o; a.pln_d = pln_d;
  a.xt = xt; a.vt = vt; a.failmask = failmask; a.stats = stats; a.n = n; a.ds = ds;
  a.max_steps = (MODE == 2) ? steps_sdf(h, res, ds) : steps_fwd(h, res, ds);
  {
    ProfScope prof(DRRT_PROF_TRACE, s);
    hipLaunchKernelGGL(k_trace<MODE>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  }
  LAUNCH_CHECK("k_trace");
  return DRRT_OK;
}

extern "C" int drrt_trace_f32(const float* rif, long long nvox, const int res[3], size_t n,
                              const float* pos, const float* vel, float h, float ds, float* xt, float* vt,
                              drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_trace<0>(rif, nullptr, nvox, res, n, pos, vel, nullptr, nullptr, h, ds, xt, vt, nullptr,
                      stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_trace_pln_f32(const float* rif, long long nvox, const int res[3], size_t n,
                                  const float* pos, const float* vel, const float* pln_o, const float* pln_d,
                                  float h, float ds, float* xt, float* vt, uint8_t* failmask,
                                  drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_trace<1>(rif, nullptr, nvox, res, n, pos, vel, pln_o, pln_d, h, ds, xt, vt, failmask,
                      stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_trace_sdf_f32(const float* rif, const float* sdf, long long nvox, const int res[3],
                                  size_t n, const float* pos, const float* vel, float h, float ds,
                                  float* xt, float* vt, drrt_stats* stats, void* ws, size_t ws_bytes,
                                  unsigned flags, void* stream) {
  return run_trace<2>(rif, sdf, nvox, res, n, pos, vel, nullptr, nullptr, h, ds, xt, vt, nullptr,
                      stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_trace_target_f32(const float* rif, long long nvox, const int res[3], size_t n,
                                     const float* pos, const float* vel, const float* target,
                                     float h, float ds, float* xt, float* vt, float* dist2,
                                     drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags,
                                     void* stream) {
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  TargetArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  if (n == 0) return zero_stats(stats, s);
  if (!pos || !vel || !target || !xt || !vt || !dist2) return fail(DRRT_ERR_ARG, "null ray pointer");
  if (n > 0xffffffffULL) return fail(DRRT_ERR_ARG, "too many rays for uint32 permutation");
  // the global iteration count lives in stats->iters: a stats block is mandatory here
  if (!stats) return fail(DRRT_ERR_ARG, "trace_target needs a stats block (global loop count)");
  const size_t state_bytes = align_up(n * 7 * sizeof(float), 256);
  const size_t sort_bytes = (flags & DRRT_FLAG_SORT_RAYS) ? align_up(sort_workspace_bytes(n), 256) : 0;
  if (!ws || ws_bytes < state_bytes + sort_bytes) return fail(DRRT_ERR_ARG, "workspace too small for trace_target");
  rc = zero_stats(stats, s); if (rc) return rc;
  a.state = (float*)ws;
  rc = maybe_sort(a.vol, h, n, pos, vel, 1.f, flags, (char*)ws + state_bytes, ws_bytes - state_bytes, &a.perm, s);
  if (rc) return rc;
  a.pos = pos; a.vel = vel; a.target = target; a.xt = xt; a.vt = vt; a.dist2 = dist2;
  a.stats = stats; a.n = n; a.ds = ds; a.max_steps = steps_fwd(h, res, ds);
  hipLaunchKernelGGL(k_target_a, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  LAUNCH_CHECK("k_target_a");
  hipLaunchKernelGGL(k_target_b, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  LAUNCH_CHECK("k_target_b");
  return DRRT_OK;
}

template <int MODE>
static int run_backtrace(const float* rif, const float* sdf, long long nvox, const int res[3], size_t n,
                         const float* xt, const float* vt, const float* dx, const float* dv,
                         float h, float ds, float* grad, drrt_stats* stats, void* ws, size_t ws_bytes,
                         unsigned flags, void* stream) {
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  BackArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  if (!grad) return fail(DRRT_ERR_ARG, "null grad pointer");
  if (MODE == 1 && !sdf) return fail(DRRT_ERR_ARG, "null sdf pointer");
  if (!(flags & DRRT_FLAG_NO_ZERO)) {                                        // src/tracer.cpp:401-403
    ProfScope prof(DRRT_PROF_ZERO, s);
    hipError_t e = hipMemsetAsync(grad, 0, (size_t)nvox * sizeof(float), s);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(grad)");
  }
  rc = zero_stats(stats, s); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!xt || !vt || !dx || !dv) return fail(DRRT_ERR_ARG, "null ray pointer");
  if (n > 0xffffffffULL) return fail(DRRT_ERR_ARG, "too many rays for uint32 permutation");
  rc = maybe_sort(a.vol, h, n, xt, vt, -1.f, flags, ws, ws_bytes, &a.perm, s); if (rc) return rc;
  a.sdf = sdf; a.xt = xt; a.vt = vt; a.dx = dx; a.dv = dv; a.grad = grad; a.stats = stats;
  a.n = n; a.ds = ds; a.max_steps = steps_adj(h, res, ds);
  a.grad_scale = (flags & DRRT_FLAG_CORRECTED_H) ? a.vol.inv_h : 1.0f;
  {
    ProfScope prof(DRRT_PROF_BACKTRACE, s);
    hipLaunchKernelGGL(k_backtrace_direct<MODE>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  }
  LAUNCH_CHECK("k_backtrace");
  return DRRT_OK;
}

extern "C" int drrt_backtrace_f32(const float* rif, long long nvox, const int res[3], size_t n,
                                  const float* xt, const float* vt, const float* dx, const float* dv,
                                  float h, float ds, float* grad, drrt_stats* stats, void* ws,
                                  size_t ws_bytes, unsigned flags, void* stream) {
  return run_backtrace<0>(rif, nullptr, nvox, res, n, xt, vt, dx, dv, h, ds, grad, stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_backtrace_sdf_f32(const float* rif, const float* sdf, long long nvox, const int res[3],
                                      size_t n, const float* xt, const float* vt, const float* dx,
                                      const float* dv, float h, float ds, float* grad, drrt_stats* stats,
                                      void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_backtrace<1>(rif, sdf, nvox, res, n, xt, vt, dx, dv, h, ds, grad, stats, ws, ws_bytes, flags, stream);
}

static unsigned cable_grid(size_t n) {
  // grid-stride: at most 4 blocks per CU so that the per-block LDS gradient flush stays small
  unsigned want = grid_for(n);
  return want < 1024u ? want : 1024u;
}

extern "C" int drrt_trace_cable_f32(const float* rif, size_t rres, float radius, float length, size_t n,
                                    const float* pos, const float* vel, const float* target, float ds,
                                    float* xt, float* vt, float* dist2, drrt_stats* stats, void* ws,
                                    size_t ws_bytes, unsigned flags, void* stream) {
  (void)ws; (void)ws_bytes; (void)flags;
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  if (!rif) return fail(DRRT_ERR_ARG, "null rif pointer");
  if (rres < 2 || rres > 0x7fffffffULL) return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  int rc = zero_stats(stats, s); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!pos || !vel || !target || !xt || !vt || !dist2) return fail(DRRT_ERR_ARG, "null ray pointer");
  CableArgs a{};
  a.rif = rif; a.rres = (int)rres; a.radius = radius; a.length = length; a.ds = ds;
  a.max_steps = (int)(4.0f * length / ds);                                   // src/tracer.cpp:332
  a.pos = pos; a.vel = vel; a.target = target; a.xt = xt; a.vt = vt; a.dist2 = dist2;
  a.stats = stats; a.n = n;
  size_t lds = (a.rres <= kCableMaxRes) ? a.rres * sizeof(float) : 0;
  hipLaunchKernelGGL(k_trace_cable, dim3(cable_grid(n)), dim3(kBlock), lds, s, a);
  LAUNCH_CHECK("k_trace_cable");
  return DRRT_OK;
}

extern "C" int drrt_backtrace_cable_f32(const float* rif, size_t rres, float radius, float length, size_t n,
                                        const float* xt, const float* vt, const float* dx, const float* dv,
                                        float ds, float* grad, drrt_stats* stats, void* ws, size_t ws_bytes,
                                        unsigned flags, void* stream) {
  (void)ws; (void)ws_bytes;
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  if (!rif || !grad) return fail(DRRT_ERR_ARG, "null rif/grad pointer");
  if (rres < 2 || rres > 0x7fffffffULL) return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  if (!(flags & DRRT_FLAG_NO_ZERO)) {                                        // src/tracer.cpp:528-530
    hipError_t e = hipMemsetAsync(grad, 0, rres * sizeof(float), s);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(grad)");
  }
  int rc = zero_stats(stats, s); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!xt || !vt || !dx || !dv) return fail(DRRT_ERR_ARG, "null ray pointer");
  CableArgs a{};
  a.rif = rif; a.rres = (int)rres; a.radius = radius; a.length = length; a.ds = ds;
  a.max_steps = (int)(4.0f * length / ds);                                   // src/tracer.cpp:544
  a.pos = xt; a.vel = vt; a.dx = dx; a.dv = dv; a.grad = grad; a.stats = stats; a.n = n;
  size_t lds = (a.rres <= kCableMaxRes) ? 2 * a.rres * sizeof(float) : 0;
  hipLaunchKernelGGL(k_backtrace_cable, dim3(cable_grid(n)), dim3(kBlock), lds, s, a);
  LAUNCH_CHECK("k_backtrace_cable");
  return DRRT_OK;
}
