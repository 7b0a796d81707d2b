// drrt_api.hip -- the C ABI of include/drrt_hip.h: argument checks (the reference's three error messages verbatim,
// src/volume.cpp:28,37,124), workspace layout, visit-order / step hand-over, per-kernel timing, and the launches of the
// kernels in drrt_forward.hip / drrt_adjoint_box.hip / drrt_adjoint_ring.hip / drrt_cable.hip.  Host code, plus the three
// small utility kernels that belong to no march (pair copy of the grid, q16 encode / decode).
#include "drrt_march.h"

using namespace drrt;

// from drrt_sort.hip
namespace drrt {
size_t sort_workspace_bytes(size_t n);
hipError_t sort_rays_by_entry_voxel(const Vol& V, float h, size_t n, const void* pos, const void* vel, int io_half,
                                    float dir_sign, void* ws, size_t ws_bytes, const uint32_t** perm_out,
                                    hipStream_t stream, bool chord_key);
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

namespace drrt { int sensor_fail(int code, const char* msg) { return fail(code, msg); } }   // used by drrt_sensor.hip

extern "C" const char* drrt_last_error(void) { return g_err; }

// ---- visit-order hand-over between paired calls (per host thread) ----------------------------
static thread_local const uint32_t* g_last_order = nullptr;   // order used by the last sorted call
static thread_local size_t g_last_order_n = 0;
static thread_local const uint32_t* g_hint_order = nullptr;   // order to use in the NEXT march call
static thread_local size_t g_hint_n = 0;

// Every march entry point takes (reads AND clears) the hint as its FIRST statement, so no return path --
// validation failure included -- can leave a stale device pointer armed for a later call.
static thread_local const uint32_t* g_last_steps = nullptr;   // per-ray iteration counts written by the last forward march
static thread_local size_t g_last_steps_n = 0;
static thread_local const uint32_t* g_hint_steps = nullptr;   // step hint for the NEXT adjoint call
static thread_local size_t g_hint_steps_n = 0;
struct OrderHint { const uint32_t* order; size_t n; const uint32_t* steps; size_t steps_n; };
static inline OrderHint take_hint() {
  OrderHint h{g_hint_order, g_hint_n, g_hint_steps, g_hint_steps_n};
  g_hint_order = nullptr; g_hint_n = 0;
  g_hint_steps = nullptr; g_hint_steps_n = 0;
  return h;
}

extern "C" const uint32_t* drrt_last_order(size_t* n_out) {
  if (n_out) *n_out = g_last_order_n;
  return g_last_order;
}
extern "C" void drrt_set_order_hint(const uint32_t* order, size_t n) { g_hint_order = order; g_hint_n = order ? n : 0; }
extern "C" size_t drrt_order_hint_pending(void) { return g_hint_order ? g_hint_n : (g_hint_steps ? g_hint_steps_n : 0); }
extern "C" const uint32_t* drrt_last_steps(size_t* n_out) {
  if (n_out) *n_out = g_last_steps_n;
  return g_last_steps;
}
extern "C" void drrt_set_step_hint(const uint32_t* steps, size_t n) { g_hint_steps = steps; g_hint_steps_n = steps ? n : 0; }
static thread_local const unsigned* g_last_counters = nullptr;   // bundle classification of the last adjoint call (device, in its workspace)
extern "C" const unsigned* drrt_last_bundle_counters(void) { return g_last_counters; }
extern "C" int drrt_ring_threshold_pct(void) { return DRRT_RING_MIN_NOFIT_PCT; }
extern "C" int drrt_ring_long_threshold_permille(void) { return DRRT_RING_MIN_LONG_PERMILLE; }
extern "C" int drrt_ring_direct_threshold_pct(void) { return DRRT_RING_DIRECT_MAX_PAIR_PCT; }

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
#ifndef DRRT_SRC_ID
#define DRRT_SRC_ID "unknown"
#endif
// "drrt_hip <abi> gfx950 src:<digest of the sources, csrc/Makefile>"
extern "C" const char* drrt_version(void) { return "drrt_hip 0.3 gfx950 src:" DRRT_SRC_ID; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t drrt_workspace_bytes(size_t n, unsigned flags) {
  size_t b = 0;
  if (flags & DRRT_FLAG_SORT_RAYS) b += align_up(sort_workspace_bytes(n), 256);
  b += align_up(n * 7 * sizeof(float), 256);       // trace_target state (cheap; always counted)
  return b;
}

// Workspace layout: [ sort buffers | trace_target state ][ pair copy of the grid, 8 B per voxel ][ 512 B counters ]
extern "C" size_t drrt_workspace_bytes_grid(size_t n, long long nvox, unsigned flags) {
  size_t b = drrt_workspace_bytes(n, flags);
  if ((flags & DRRT_FLAG_PAIR_GRID) && nvox > 0) b += (size_t)nvox * 2 * sizeof(float);
  return b + 512;
}

namespace drrt {
// pair[2 i] = n[i], pair[2 i + 1] = n[i + W] (the y-neighbour, clamped at the far y face -- those entries are never
// read: only strictly interior cells use the copy).  One thread per voxel, 4 + 8 B of traffic each.
__global__ void __launch_bounds__(256) k_build_pair(const float* __restrict__ g, float2* __restrict__ q, int W, int H,
                                                    unsigned nvox) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= nvox) return;
  const unsigned row = i / (unsigned)W, y = row % (unsigned)H;
  const unsigned y1 = (y + 1u < (unsigned)H) ? (unsigned)W : 0u;
  q[i] = make_float2(g[i], g[i + y1]);
}
}  // namespace drrt

// DRRT_FLAG_PAIR_GRID: place (and, unless DRRT_FLAG_PAIR_REUSE, build) the pair copy in the workspace.
static int maybe_pair(Vol& V, long long nvox, size_t n, unsigned flags, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!(flags & DRRT_FLAG_PAIR_GRID)) return DRRT_OK;
  const size_t off = drrt_workspace_bytes(n, flags), need = off + (size_t)nvox * 2 * sizeof(float) + 512;
  if (!ws || ws_bytes < need) return fail(DRRT_ERR_ARG, "workspace too small for DRRT_FLAG_PAIR_GRID (see drrt_workspace_bytes_grid)");
  if (((uintptr_t)ws + off) % 16 != 0) return fail(DRRT_ERR_ARG, "workspace must be 16-byte aligned for DRRT_FLAG_PAIR_GRID");
  float2* q = (float2*)((char*)ws + off);
  if (!(flags & DRRT_FLAG_PAIR_REUSE)) {
    ProfScope prof(DRRT_PROF_QUAD, s);
    hipLaunchKernelGGL(drrt::k_build_pair, dim3((unsigned)(((size_t)nvox + 255) / 256)), dim3(256), 0, s, V.data, q, V.W, V.H,
                       (unsigned)nvox);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "k_build_pair");
  }
  V.pair = (const float*)q;
  return DRRT_OK;
}

// volume ctor checks: src/volume.cpp:31-38 (size) and :123-124 (width/height >= 2)
static int check_steps(float h, float ds) {
  // the reference divides by ds and truncates to int (src/tracer.cpp:51): a non-positive or non-finite
  // step would make that undefined; refuse it instead
  if (!(h > 0.f) || !(ds > 0.f) || !(h < 3.0e38f) || !(ds < 3.0e38f))
    return fail(DRRT_ERR_ARG, "h and ds must be positive and finite");
  return DRRT_OK;
}

static int make_vol(const float* rif, long long nvox, const int res[3], float h, Vol* V) {
  if (!rif || !res) return fail(DRRT_ERR_ARG, "null rif/res pointer");
  if ((long long)res[0] * res[1] * res[2] != nvox || nvox <= 0)
    return fail(DRRT_ERR_RES_MISMATCH, "Resolution doesn't match data");
  if (!(res[0] == 1 && res[1] == 1 && res[2] == 1) && (res[0] < 2 || res[1] < 2))
    return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  if (nvox >= (1LL << 29)) return fail(DRRT_ERR_ARG, "grid too large (>= 2^29 voxels) for 32-bit byte offsets");
  if (res[0] >= (1 << 24) || res[1] >= (1 << 24) || res[2] >= (1 << 24) || (long long)res[0] * res[1] >= (1 << 24))
    return fail(DRRT_ERR_ARG, "grid extents too large for 24-bit index arithmetic");
  if (!(h > 0.f) || !(h < 3.0e38f)) return fail(DRRT_ERR_ARG, "h and ds must be positive and finite");
  V->data = rif; V->W = res[0]; V->H = res[1]; V->D = res[2];
  vol_finish(*V, h);
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

static int maybe_sort(const Vol& V, float h, size_t n, const void* pos, const void* vel, float dir_sign,
                      unsigned flags, void* ws, size_t ws_bytes, const uint32_t** perm, hipStream_t s,
                      OrderHint hint, int io_half = 0) {
  *perm = nullptr;
  // a hint from the caller (normally the paired forward call's order) replaces the sort; it was consumed
  // by this call at its entry (take_hint) whether or not it is usable.  Entries are range-checked on the
  // device (ray_index), so a wrong hint can leave rays unvisited but cannot make a kernel fault.
  if (hint.order && hint.n == n) { *perm = hint.order; return DRRT_OK; }
  if (!(flags & DRRT_FLAG_SORT_RAYS) || n < 2) return DRRT_OK;
  if (!ws || ws_bytes < sort_workspace_bytes(n)) return fail(DRRT_ERR_ARG, "workspace too small for DRRT_FLAG_SORT_RAYS");
  ProfScope prof(DRRT_PROF_SORT, s);
  hipError_t e = sort_rays_by_entry_voxel(V, h, n, pos, vel, io_half, dir_sign, ws, ws_bytes, perm, s,
                                          (flags & DRRT_FLAG_CHORD_KEY) != 0);
  if (e == hipSuccess) { g_last_order = *perm; g_last_order_n = n; }
  return e == hipSuccess ? DRRT_OK : fail_hip(e, "sort_rays_by_entry_voxel");
}

#define LAUNCH_CHECK(where)                                              \
  do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail_hip(e_, where); } while (0)

template <int MODE>
static int run_trace(const float* rif, const float* sdf, long long nvox, const int res[3], size_t n,
                     const void* pos, const void* vel, const float* pln_o, const float* pln_d,
                     float h, float ds, void* xt, void* vt, uint8_t* failmask, drrt_stats* stats,
                     void* ws, size_t ws_bytes, unsigned flags, void* stream, int io_half = 0) {
  const OrderHint hint = take_hint();
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  TraceArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  rc = check_steps(h, ds); if (rc) return rc;
  if (n == 0) return zero_stats(stats, s);
  if (!pos || !vel || !xt || !vt) return fail(DRRT_ERR_ARG, "null ray pointer");
  if (MODE == 1 && (!pln_o || !pln_d || !failmask)) return fail(DRRT_ERR_ARG, "null plane/failmask pointer");
  if ((MODE == 1 || MODE == 2) && !stats) {
    // the second pass needs the global loop count: library-owned block, one per device (allocated once).  It is
    // shared by every stream of that device: callers that run trace_pln / trace_sdf concurrently on several streams
    // of one device must pass their own stats block.
    constexpr int kMaxDev = 64;
    static drrt_stats* priv[kMaxDev] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail_hip(e, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDev) return fail(DRRT_ERR_ARG, "device ordinal out of range; pass a stats block");
    if (!priv[dev]) { e = hipMalloc((void**)&priv[dev], sizeof(drrt_stats)); if (e != hipSuccess) return fail_hip(e, "hipMalloc(stats)"); }
    stats = priv[dev];
  }
  if (MODE == 2 && !sdf) return fail(DRRT_ERR_ARG, "null sdf pointer");
  if (n > 0xffffffffULL) return fail(DRRT_ERR_ARG, "too many rays for uint32 permutation");
  rc = zero_stats(stats, s); if (rc) return rc;
  rc = maybe_sort(a.vol, h, n, pos, vel, 1.f, flags, ws, ws_bytes, &a.perm, s, hint, io_half); if (rc) return rc;
  rc = maybe_pair(a.vol, nvox, n, flags, ws, ws_bytes, s); if (rc) return rc;
  if (MODE == 2) {                      // n flag bytes, in the slack the workspace keeps after the sort buffers
    const size_t off = (flags & DRRT_FLAG_SORT_RAYS) ? align_up(sort_workspace_bytes(n), 256) : 0;
    if (!ws || ws_bytes < off + n) return fail(DRRT_ERR_ARG, "workspace too small for trace_sdf (see drrt_workspace_bytes)");
    a.again = (uint8_t*)ws + off;
  }
  g_last_steps = nullptr; g_last_steps_n = 0;
  if (MODE != 2) {
    // per-ray iteration counts for the paired adjoint (drrt_last_steps): n uint32 in the trace_target slot of the workspace
    const size_t off = (flags & DRRT_FLAG_SORT_RAYS) ? align_up(sort_workspace_bytes(n), 256) : 0;
    if (ws && ws_bytes >= off + n * sizeof(uint32_t)) {
      a.steps_out = (uint32_t*)((char*)ws + off);
      g_last_steps = a.steps_out; g_last_steps_n = n;
    }
  }
  a.io_half = io_half;
  a.sdf = sdf; a.pos = pos; a.vel = vel; a.pln_o = pln_o; a.pln_d = pln_d;
  a.xt = xt; a.vt = vt; a.failmask = failmask; a.stats = stats; a.n = n; a.ds = ds;
  a.max_steps = (MODE == 2) ? steps_sdf(h, res, ds) : steps_fwd(h, res, ds);
  a.xcd_order = (a.perm != nullptr && !(flags & DRRT_FLAG_DISPATCH_IN_ORDER)) ? 1 : 0;
  {
    ProfScope prof(DRRT_PROF_TRACE, s);
    launch_trace(MODE, a, s);
  }
  LAUNCH_CHECK("k_trace");
  if (MODE == 1 || MODE == 2) {
    launch_trace_again(MODE, a, s);
    LAUNCH_CHECK("k_trace_again");
  }
  return DRRT_OK;
}

extern "C" int drrt_trace_f32(const float* rif, long long nvox, const int res[3], size_t n,
                              const float* pos, const float* vel, float h, float ds, float* xt, float* vt,
                              drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_trace<0>(rif, nullptr, nvox, res, n, pos, vel, nullptr, nullptr, h, ds, xt, vt, nullptr,
                      stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_trace_f16io(const float* rif, long long nvox, const int res[3], size_t n,
                                const void* pos, const void* vel, float h, float ds, void* xt, void* vt,
                                drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_trace<0>(rif, nullptr, nvox, res, n, pos, vel, nullptr, nullptr, h, ds, xt, vt, nullptr,
                      stats, ws, ws_bytes, flags, stream, 1);
}

extern "C" int drrt_trace_q16io(const float* rif, long long nvox, const int res[3], size_t n,
                                const void* pos, const void* vel, float h, float ds, void* xt, void* vt,
                                drrt_stats* stats, void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_trace<0>(rif, nullptr, nvox, res, n, pos, vel, nullptr, nullptr, h, ds, xt, vt, nullptr,
                      stats, ws, ws_bytes, flags, stream, (flags & DRRT_FLAG_Q16_POS_ONLY) ? 3 : 2);
}

// ---- 16-bit ray state: encode / decode on the device (so that every binding rounds exactly as the kernels do) ----
namespace drrt {
__global__ void __launch_bounds__(256) k_q16_encode(Vol V, size_t n3, const float* __restrict__ pos, const float* __restrict__ vel,
                                                    uint16_t* __restrict__ pos_q, int16_t* __restrict__ vel_q) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n3) return;
  if (pos) pos_q[k] = q16_pos_enc(V, pos[k]);
  if (vel) vel_q[k] = q16_vel_enc(vel[k]);
}
__global__ void __launch_bounds__(256) k_q16_decode(Vol V, size_t n3, const uint16_t* __restrict__ pos_q, const int16_t* __restrict__ vel_q,
                                                    float* __restrict__ pos, float* __restrict__ vel) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n3) return;
  if (pos_q) pos[k] = q16_pos_dec(V, pos_q[k]);
  if (vel_q) vel[k] = q16_vel_dec(vel_q[k]);
}
}  // namespace drrt

static int q16_vol(const int res[3], float h, Vol* V) {
  if (!res) return fail(DRRT_ERR_ARG, "null res pointer");
  if (res[0] < 1 || res[1] < 1 || res[2] < 1) return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  if (!(h > 0.f) || !(h < 3.0e38f)) return fail(DRRT_ERR_ARG, "h and ds must be positive and finite");
  V->data = nullptr; V->W = res[0]; V->H = res[1]; V->D = res[2];
  vol_finish(*V, h);
  return DRRT_OK;
}

extern "C" int drrt_q16_params(const int res[3], float h, float out[3]) {
  g_err[0] = 0;
  Vol V; int rc = q16_vol(res, h, &V); if (rc) return rc;
  if (!out) return fail(DRRT_ERR_ARG, "null output pointer");
  out[0] = V.q_min; out[1] = V.q_step; out[2] = kQ16VelStep;
  return DRRT_OK;
}

extern "C" int drrt_q16_encode(const int res[3], float h, size_t n, const float* pos, const float* vel, void* pos_q,
                               void* vel_q, void* stream) {
  g_err[0] = 0;
  Vol V; int rc = q16_vol(res, h, &V); if (rc) return rc;
  if ((pos && !pos_q) || (vel && !vel_q)) return fail(DRRT_ERR_ARG, "null output pointer");
  if (n == 0 || (!pos && !vel)) return DRRT_OK;
  hipLaunchKernelGGL(k_q16_encode, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, V, 3 * n, pos, vel,
                     (uint16_t*)pos_q, (int16_t*)vel_q);
  LAUNCH_CHECK("k_q16_encode");
  return DRRT_OK;
}

extern "C" int drrt_q16_decode(const int res[3], float h, size_t n, const void* pos_q, const void* vel_q, float* pos,
                               float* vel, void* stream) {
  g_err[0] = 0;
  Vol V; int rc = q16_vol(res, h, &V); if (rc) return rc;
  if ((pos_q && !pos) || (vel_q && !vel)) return fail(DRRT_ERR_ARG, "null output pointer");
  if (n == 0 || (!pos_q && !vel_q)) return DRRT_OK;
  hipLaunchKernelGGL(k_q16_decode, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, V, 3 * n,
                     (const uint16_t*)pos_q, (const int16_t*)vel_q, pos, vel);
  LAUNCH_CHECK("k_q16_decode");
  return DRRT_OK;
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
  g_last_steps = nullptr; g_last_steps_n = 0;     // this forward march writes no iteration counts
  (void)take_hint();     // never honoured here: the phase-A state buffer at the start of the workspace would overlay
                         // an order that lives in the same workspace (drrt_last_order() of an earlier call)
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  TargetArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  rc = check_steps(h, ds); if (rc) return rc;
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
  rc = maybe_sort(a.vol, h, n, pos, vel, 1.f, flags, (char*)ws + state_bytes, ws_bytes - state_bytes, &a.perm, s,
                  OrderHint{nullptr, 0, nullptr, 0});
  if (rc) return rc;
  a.pos = pos; a.vel = vel; a.target = target; a.xt = xt; a.vt = vt; a.dist2 = dist2;
  a.stats = stats; a.n = n; a.ds = ds; a.max_steps = steps_fwd(h, res, ds);
  rc = maybe_pair(a.vol, nvox, n, flags, ws, ws_bytes, s); if (rc) return rc;     // lies behind [state | sort buffers]
  launch_target(a, s);
  LAUNCH_CHECK("k_target");
  return DRRT_OK;
}

// One chunk of a resumable adjoint march (drrt_backtrace_chunk_f32)
struct ChunkReq { void* state; size_t state_bytes; int it_begin, it_count; int* progress; };

namespace drrt {
__global__ void k_chunk_progress_init(int* p) {
  const int k = threadIdx.x;              // [0..11] mins, maxs, mins, maxs; [12] count; [13..18] mins, maxs; [19] spare
  if (k < 20) p[k] = (k == 12 || k == 19) ? 0 : ((((k < 12 ? k : k - 1) / 3) & 1) ? (int)0x80000000 : 0x7fffffff);
}
}  // namespace drrt

template <int MODE>
static int run_backtrace(const float* rif, const float* sdf, long long nvox, const int res[3], size_t n,
                         const void* xt, const void* vt, const void* dx, const void* dv,
                         float h, float ds, float* grad, drrt_stats* stats, void* ws, size_t ws_bytes,
                         unsigned flags, void* stream, int io_half = 0, const ChunkReq* ck = nullptr) {
  const OrderHint hint = take_hint();
  g_last_counters = nullptr;              // set again below when this call classifies its bundles
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  BackArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  rc = check_steps(h, ds); if (rc) return rc;
  if (!grad) return fail(DRRT_ERR_ARG, "null grad pointer");
  if (MODE == 1 && !sdf) return fail(DRRT_ERR_ARG, "null sdf pointer");
  const bool first_chunk = ck == nullptr || ck->it_begin == 0;
  if (ck != nullptr) {
    if (ck->it_begin < 0) return fail(DRRT_ERR_ARG, "chunk: it_begin must be >= 0");
    if (flags & DRRT_FLAG_DIRECT_ATOMICS) return fail(DRRT_ERR_ARG, "chunk: not available with DRRT_FLAG_DIRECT_ATOMICS");
    if (!ck->state || ck->state_bytes < drrt_backtrace_chunk_state_bytes(n))
      return fail(DRRT_ERR_ARG, "chunk: state buffer too small (see drrt_backtrace_chunk_state_bytes)");
    if (!first_chunk && (flags & DRRT_FLAG_SORT_RAYS) && !(hint.order && hint.n == n))
      return fail(DRRT_ERR_ARG, "chunk: a resumed chunk needs the visit order of its first chunk (drrt_set_order_hint)");
  }
  if (!(flags & DRRT_FLAG_NO_ZERO) && first_chunk) {                         // src/tracer.cpp:401-403
    ProfScope prof(DRRT_PROF_ZERO, s);
    hipError_t e = hipMemsetAsync(grad, 0, (size_t)nvox * sizeof(float), s);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(grad)");
  }
  if (first_chunk) { rc = zero_stats(stats, s); if (rc) return rc; }
  if (n == 0) return DRRT_OK;
  if (!xt || !vt || !dx || !dv) return fail(DRRT_ERR_ARG, "null ray pointer");
  if (n > 0xffffffffULL) return fail(DRRT_ERR_ARG, "too many rays for uint32 permutation");
  rc = maybe_sort(a.vol, h, n, xt, vt, -1.f, flags, ws, ws_bytes, &a.perm, s, hint, io_half); if (rc) return rc;
  rc = maybe_pair(a.vol, nvox, n, flags, ws, ws_bytes, s); if (rc) return rc;
  a.io_half = io_half;
  a.sdf = sdf; a.xt = xt; a.vt = vt; a.dx = dx; a.dv = dv; a.grad = grad; a.stats = stats;
  a.n = n; a.ds = ds; a.max_steps = steps_adj(h, res, ds);
  if (ck != nullptr) {                      // the iterations [it_begin, it_begin + it_count) of the march's max_steps
    const int left = a.max_steps - ck->it_begin;
    a.max_steps = ck->it_count < 0 ? left : (ck->it_count < left ? ck->it_count : left);
    if (a.max_steps < 0) a.max_steps = 0;
    a.chunk_state = (float*)ck->state; a.chunk_stride = (size_t)adj_grid_for(n) * kAdjBlock;
    a.chunk_resume = first_chunk ? 0 : 1; a.chunk_progress = ck->progress;
    if (ck->progress) { hipLaunchKernelGGL(drrt::k_chunk_progress_init, dim3(1), dim3(64), 0, s, ck->progress); LAUNCH_CHECK("k_chunk_progress_init"); }
  }
  a.grad_scale = (flags & DRRT_FLAG_CORRECTED_H) ? a.vol.inv_h : 1.0f;
  a.experiment = (int)((flags >> 8) & 0xffu);
  a.fsteps = (hint.steps && hint.steps_n == n) ? hint.steps : nullptr;
  a.xcd_order = (a.perm != nullptr && !(flags & DRRT_FLAG_DISPATCH_IN_ORDER)) ? 1 : 0;
  a.dbg = nullptr;
  if (flags & DRRT_FLAG_DEBUG_COUNTERS) {        // last 64 bytes of the workspace
    if (!ws || ws_bytes < 512) return fail(DRRT_ERR_ARG, "workspace too small for DRRT_FLAG_DEBUG_COUNTERS");
    a.dbg = (unsigned long long*)((char*)ws + ((ws_bytes - 512) & ~(size_t)7));
    hipError_t e = hipMemsetAsync(a.dbg, 0, 512, s);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(dbg)");
  }
  {
    ProfScope prof(DRRT_PROF_BACKTRACE, s);
    if (flags & DRRT_FLAG_DIRECT_ATOMICS)
      launch_backtrace_direct(MODE, a, s);
    else {
      const bool abl = a.experiment != 0 || a.dbg != nullptr;
      // Two kernels: k_backtrace_flat with its compile-time 9^3 box window for compact bundles, k_backtrace_ring (fitted ring
      // window, step hint) for the rest.  With a visit order the bundles are classified on the device and BOTH are launched;
      // the one the counters do not pick returns at once (no host round trip).  Needs the 512-byte counter block at the end
      // of a drrt_workspace_bytes_grid() workspace; without it, or without an order, the box-window kernel runs.
      // DRRT_FLAG_STATIC_WINDOW / DRRT_FLAG_RING_WINDOW force one of the two (A-B).
      a.select = nullptr;
      const size_t ctr_off = (drrt_workspace_bytes(n, flags) + ((flags & DRRT_FLAG_PAIR_GRID) ? (size_t)nvox * 2 * sizeof(float) : 0) + 7) & ~(size_t)7;
      const bool force_box = (flags & DRRT_FLAG_STATIC_WINDOW) != 0 || a.experiment == 7 || ck != nullptr;   // chunks: box-window kernel only
      const bool force_ring = (flags & DRRT_FLAG_RING_WINDOW) != 0 && ck == nullptr;
      if (!force_box && !force_ring && a.perm != nullptr && ws && ws_bytes >= ctr_off + 512) {
        a.select = (unsigned*)((char*)ws + ctr_off + 256);
        g_last_counters = a.select;
        hipError_t e = hipMemsetAsync(a.select, 0, 32, s);
        if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(select)");
        // [5] != 0 pins the general instantiation of the ring kernel: backtrace_sdf, the ablation / counter build and
        // DRRT_FLAG_RING_GENERAL (A-B) have no sparse-only one
        const bool sparse_ok = MODE == 0 && !abl && !(flags & DRRT_FLAG_RING_GENERAL);
        if (!sparse_ok) { e = hipMemsetAsync((char*)a.select + 20, 1, 1, s); if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(select)"); }
        launch_bundle_classify(a, s);
        if (sparse_ok) launch_backtrace_ring_sparse(a, s, 2);    // (the classification never picks the general one then)
        else launch_backtrace_ring(MODE, abl, a, s);
      }
      if (!force_ring) launch_backtrace_box(MODE, abl, a, s);
      if (force_ring && (flags & DRRT_FLAG_RING_SPARSE) && MODE == 0 && !abl)
        launch_backtrace_ring_sparse(a, s, (flags & DRRT_FLAG_RING_DIRECT) ? 1 : 0);
      else if (force_ring) launch_backtrace_ring(MODE, abl, a, s);
    }
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

extern "C" size_t drrt_backtrace_chunk_state_bytes(size_t n) { return (size_t)adj_grid_for(n) * kAdjBlock * 13 * sizeof(float); }
extern "C" int drrt_backtrace_max_steps(const int res[3], float h, float ds) {
  if (!res || !(h > 0.f) || !(ds > 0.f)) return -1;
  return steps_adj(h, res, ds);
}
extern "C" int drrt_backtrace_chunk_f32(const float* rif, long long nvox, const int res[3], size_t n,
                                        const float* xt, const float* vt, const float* dx, const float* dv,
                                        float h, float ds, float* grad, drrt_stats* stats, void* ws,
                                        size_t ws_bytes, unsigned flags, void* stream, void* state, size_t state_bytes,
                                        int it_begin, int it_count, int* progress) {
  const ChunkReq ck{state, state_bytes, it_begin, it_count, progress};
  return run_backtrace<0>(rif, nullptr, nvox, res, n, xt, vt, dx, dv, h, ds, grad, stats, ws, ws_bytes, flags, stream, 0, &ck);
}

extern "C" int drrt_backtrace_f16io(const float* rif, long long nvox, const int res[3], size_t n,
                                    const void* xt, const void* vt, const void* dx, const void* dv,
                                    float h, float ds, float* grad, drrt_stats* stats, void* ws,
                                    size_t ws_bytes, unsigned flags, void* stream) {
  return run_backtrace<0>(rif, nullptr, nvox, res, n, xt, vt, dx, dv, h, ds, grad, stats, ws, ws_bytes, flags, stream, 1);
}

extern "C" int drrt_backtrace_q16io(const float* rif, long long nvox, const int res[3], size_t n,
                                    const void* xt, const void* vt, const void* dx, const void* dv,
                                    float h, float ds, float* grad, drrt_stats* stats, void* ws,
                                    size_t ws_bytes, unsigned flags, void* stream) {
  return run_backtrace<0>(rif, nullptr, nvox, res, n, xt, vt, dx, dv, h, ds, grad, stats, ws, ws_bytes, flags, stream,
                          (flags & DRRT_FLAG_Q16_POS_ONLY) ? 3 : 2);
}

extern "C" int drrt_backtrace_sdf_f32(const float* rif, const float* sdf, long long nvox, const int res[3],
                                      size_t n, const float* xt, const float* vt, const float* dx,
                                      const float* dv, float h, float ds, float* grad, drrt_stats* stats,
                                      void* ws, size_t ws_bytes, unsigned flags, void* stream) {
  return run_backtrace<1>(rif, sdf, nvox, res, n, xt, vt, dx, dv, h, ds, grad, stats, ws, ws_bytes, flags, stream);
}

extern "C" int drrt_trace_cable_f32(const float* rif, size_t rres, float radius, float length, size_t n,
                                    const float* pos, const float* vel, const float* target, float ds,
                                    float* xt, float* vt, float* dist2, drrt_stats* stats, void* ws,
                                    size_t ws_bytes, unsigned flags, void* stream) {
  (void)ws; (void)ws_bytes; (void)flags;
  g_last_steps = nullptr; g_last_steps_n = 0;
  (void)take_hint();     // the cable kernels visit rays in caller order
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  if (!rif) return fail(DRRT_ERR_ARG, "null rif pointer");
  if (rres < 2 || rres > 0x7fffffffULL) return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  if (!(radius > 0.f) || !(length > 0.f) || !(ds > 0.f) || !(ds < 3.0e38f))
    return fail(DRRT_ERR_ARG, "radius, length and ds must be positive and finite");
  int rc = zero_stats(stats, s); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!pos || !vel || !target || !xt || !vt || !dist2) return fail(DRRT_ERR_ARG, "null ray pointer");
  CableArgs a{};
  a.rif = rif; a.rres = (int)rres; a.radius = radius; a.length = length; a.ds = ds;
  a.max_steps = (int)(4.0f * length / ds);                                   // src/tracer.cpp:332
  a.pos = pos; a.vel = vel; a.target = target; a.xt = xt; a.vt = vt; a.dist2 = dist2;
  a.stats = stats; a.n = n;
  launch_trace_cable(a, s);
  LAUNCH_CHECK("k_trace_cable");
  return DRRT_OK;
}

extern "C" int drrt_backtrace_cable_f32(const float* rif, size_t rres, float radius, float length, size_t n,
                                        const float* xt, const float* vt, const float* dx, const float* dv,
                                        float ds, float* grad, drrt_stats* stats, void* ws, size_t ws_bytes,
                                        unsigned flags, void* stream) {
  (void)ws; (void)ws_bytes;
  (void)take_hint();
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  if (!rif || !grad) return fail(DRRT_ERR_ARG, "null rif/grad pointer");
  if (rres < 2 || rres > 0x7fffffffULL) return fail(DRRT_ERR_BAD_RES, "volume: invalid resolution!");
  if (!(radius > 0.f) || !(length > 0.f) || !(ds > 0.f) || !(ds < 3.0e38f))
    return fail(DRRT_ERR_ARG, "radius, length and ds must be positive and finite");
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
  launch_backtrace_cable(a, s);
  LAUNCH_CHECK("k_backtrace_cable");
  return DRRT_OK;
}
