// drrt_march.h -- what the march kernels of the gfx950 library share: launch-side argument blocks, ray I/O in the three
// storage formats, the visit order and its XCD-aware block order, the tap gathers, per-block statistics, the LDS
// accumulator type of the gradient windows, and the launchers each translation unit exports to the C ABI (drrt_api.hip).
//
//   drrt_forward.hip        trace, trace_plane, trace_sdf, trace_target          (src/tracer.cpp:35-310)
//   drrt_adjoint_box.hip    backtrace / backtrace_sdf, compile-time box window   (src/tracer.cpp:384-509), bundle
//                           classification, the one-atomic-per-tap cross-check kernel
//   drrt_adjoint_ring.hip   backtrace / backtrace_sdf, fitted ring window (its own translation unit: it is built with a
//                           different instruction-scheduling strategy, csrc/Makefile)
//   drrt_cable.hip          trace_cable, backtrace_cable                         (src/tracer.cpp:312-382, 511-567)
//   drrt_api.hip            the C ABI of include/drrt_hip.h (host code only)
//
// Reference semantics: /root/reference/src/tracer.cpp, src/volume.cpp, src/cylinder_volume.cpp.  Quirk numbers
// (Q1..Q16) refer to SURVEY.md section 8.1.
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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include <hip/hip_fp16.h>

#include "../../include/drrt_hip.h"
#include "drrt_device.h"

namespace drrt {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// stats: block-level reduction, then 3 atomics per block
// ---------------------------------------------------------------------------------------------
template <int BLOCK = kBlock>
__device__ __forceinline__ void block_stats(drrt_stats* stats, unsigned steps, unsigned failed) {
  if (!stats) return;
  __shared__ unsigned s_sum[BLOCK / kWave], s_max[BLOCK / kWave], s_fail[BLOCK / kWave];
  unsigned ws = wave_sum_u32(steps), wm = wave_max_u32(steps), wf = wave_sum_u32(failed);
  int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) { s_sum[wid] = ws; s_max[wid] = wm; s_fail[wid] = wf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum = 0, fail = 0; unsigned mx = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / kWave; ++w) { sum += s_sum[w]; fail += s_fail[w]; mx = max(mx, s_max[w]); }
    if (sum)  atomicAdd(&stats->ray_steps, sum);
    if (fail) atomicAdd(&stats->n_failed, fail);
    if (mx)   atomicMax(&stats->iters, mx);
  }
}

// Ray visited by thread t: through the visit order when there is one.  An index outside [0, n) -- only possible
// with a corrupt caller-supplied order hint -- is skipped (that slot's outputs stay unwritten), never dereferenced.
__device__ __forceinline__ bool ray_index(const uint32_t* __restrict__ perm, size_t t, size_t n, size_t& i) {
  if (t >= n) return false;
  i = perm ? (size_t)perm[t] : t;
  return i < n;
}

// XCD-aware block order.  The dispatcher deals the blocks of a launch round-robin over the chip's 8 XCDs (observed, not
// promised: b % 8 labels the blocks that share an XCD), each with its own 4 MiB L2.  Consecutive blocks of the visit
// order are neighbours in space -- their bundles read overlapping cells of the grid -- so an XCD takes CONSECUTIVE blocks
// of the visit order instead of every 8th one: a border cell is then fetched into one L2, not eight.  Bijective for every
// block count; a pure speed choice (a different placement is slower, not wrong).  Measured on MI355X, same box, 256^3 /
// 1M rays (gpurun_out/xcd, xcd2, r3final), identity / runs of 16 / runs of 64 / one run per XCD:
//   forward march (metric)                    1.043 / 0.99-1.01 / 1.004 / 1.05-1.065 ms
//   box-window adjoint (metric)               4.66 / 4.68-4.78 / 4.71-4.73 / 4.68-4.73 ms
//   ring-window adjoint, six rotated views    8.92-9.09 / 8.74-8.85 / 9.11-9.35 / 8.62-8.84 ms
//   ring-window adjoint, 4 tomography views   5.30-5.32 / 5.23-5.25 / -- / 5.89-5.90 ms
// One run per XCD hands whole VIEWS to single XCDs -- views differ in length and cost, and the launch then waits for the
// XCD that drew the oblique ones -- so it is not used.  (Those rows were measured with 256-thread blocks everywhere.)  Since
// the adjoint kernels run one wave per block (kAdjBlock), consecutive WAVES would land on different XCDs in blockIdx order:
// runs of 16 one-wave blocks per XCD, same box: box adjoint 4.52-4.57 -> 4.49-4.50 ms (runs of 64: no change), ring adjoint on
// the six rotated views 8.64-8.66 (runs of 64) -> 8.47-8.63 ms.  Runs of 16 for all three march kernels.
enum { kXcdOff = 0, kXcdWhole = 1, kXcdRuns16 = 2 };
#ifndef DRRT_FLAT_XCD_MODE
#define DRRT_FLAT_XCD_MODE kXcdRuns16
#endif
#ifndef DRRT_RING_XCD_MODE
#define DRRT_RING_XCD_MODE kXcdRuns16
#endif
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb, int mode) {
  if (mode == kXcdRuns16) {            // groups of 8 * 16 consecutive blocks of the visit order: 16 for each XCD
    constexpr unsigned C = 16u, G = 8u * C;
    if (b >= nb / G * G) return b;
    const unsigned w = b % G;
    return b - w + (w & 7u) * C + (w >> 3);
  }
  if (mode == kXcdWhole) {             // one contiguous run per XCD
    const unsigned q = nb >> 3, r = nb & 7u, x = b & 7u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (b >> 3);
  }
  return b;
}

struct Ray3 { float x, y, z; };
// Ray arrays are (n,3) row-major in one of three storage formats (`io`): 0 = fp32; 1 = IEEE half (the *_f16io entry
// points); 2 = the 16-bit ray state "q16" of drrt_device.h (the *_q16io entry points: positions as box-relative
// unsigned codes, directions as 2^-14 fixed point, adjoint seeds as IEEE half) -- "fp16 ray state", config 5 of
// BASELINE.json.  Values are widened exactly on load; the march, the adjoint recurrences and the gradient
// accumulation are always fp32; outputs are rounded to the storage format once.
enum { RAY_POS = 0, RAY_VEL = 1, RAY_SEED = 2 };
__device__ __forceinline__ Ray3 ld3(const void* p, size_t i, int io = 0, const Vol* V = nullptr, int kind = RAY_SEED) {
  if (io == 2 && kind == RAY_POS) {
    const uint16_t* q = (const uint16_t*)p;
    return Ray3{q16_pos_dec(*V, q[3 * i]), q16_pos_dec(*V, q[3 * i + 1]), q16_pos_dec(*V, q[3 * i + 2])};
  }
  if (io == 3) {                       // q16 positions, everything else fp32
    if (kind == RAY_POS) {
      const uint16_t* q = (const uint16_t*)p;
      return Ray3{q16_pos_dec(*V, q[3 * i]), q16_pos_dec(*V, q[3 * i + 1]), q16_pos_dec(*V, q[3 * i + 2])};
    }
    const float* q = (const float*)p;
    return Ray3{q[3 * i], q[3 * i + 1], q[3 * i + 2]};
  }
  if (io == 2 && kind == RAY_VEL) {
    const int16_t* q = (const int16_t*)p;
    return Ray3{q16_vel_dec(q[3 * i]), q16_vel_dec(q[3 * i + 1]), q16_vel_dec(q[3 * i + 2])};
  }
  if (io) {
    const __half* q = (const __half*)p;
    return Ray3{__half2float(q[3 * i]), __half2float(q[3 * i + 1]), __half2float(q[3 * i + 2])};
  }
  const float* q = (const float*)p;
  return Ray3{q[3 * i], q[3 * i + 1], q[3 * i + 2]};
}
__device__ __forceinline__ void st3(void* p, size_t i, float a, float b, float c, int io = 0, const Vol* V = nullptr,
                                    int kind = RAY_SEED) {
  if ((io == 2 || io == 3) && kind == RAY_POS) {
    uint16_t* q = (uint16_t*)p;
    q[3 * i] = q16_pos_enc(*V, a); q[3 * i + 1] = q16_pos_enc(*V, b); q[3 * i + 2] = q16_pos_enc(*V, c);
  } else if (io == 3) {
    float* q = (float*)p;
    q[3 * i] = a; q[3 * i + 1] = b; q[3 * i + 2] = c;
  } else if (io == 2 && kind == RAY_VEL) {
    int16_t* q = (int16_t*)p;
    q[3 * i] = q16_vel_enc(a); q[3 * i + 1] = q16_vel_enc(b); q[3 * i + 2] = q16_vel_enc(c);
  } else if (io) {
    __half* q = (__half*)p;
    q[3 * i] = __float2half_rn(a); q[3 * i + 1] = __float2half_rn(b); q[3 * i + 2] = __float2half_rn(c);
  } else {
    float* q = (float*)p;
    q[3 * i] = a; q[3 * i + 1] = b; q[3 * i + 2] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// forward march: trace (MODE 0), trace_plane (MODE 1), trace_sdf (MODE 2)
// ---------------------------------------------------------------------------------------------
struct TraceArgs {
  Vol vol;
  const float* sdf;            // MODE 2
  const void* pos; const void* vel;         // fp32, or half when io_half
  const float* pln_o; const float* pln_d;   // MODE 1
  void* xt; void* vt; uint8_t* failmask;
  uint8_t* again;              // MODE 2: per-ray "re-march over the global loop count" flags (workspace)
  int io_half;
  const uint32_t* perm;        // nullable: visit order
  uint32_t* steps_out;         // nullable (workspace): per-ray number of march iterations, for the paired adjoint (step hint)
  drrt_stats* stats;
  size_t n;
  float ds;
  int max_steps;
  int xcd_order;               // 1: the launch's blocks take the visit order XCD by XCD (xcd_block)
};

// (int)floorf(f) in one instruction (v_cvt_flr_i32_f32: floor, then the saturating conversion of v_cvt_i32_f32)
__device__ __forceinline__ int cvt_floor_i32(float f) {
  int i;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(f));
  return i;
}

// the four rows of a cell's taps as wave-uniform base pointers + ONE 32-bit byte offset per lane
// (global_load ... saddr: one shift instead of four 64-bit address computations per gather)
struct TapRows { const char *d00, *d10, *d01, *d11; };
template <bool PAIR>
__device__ __forceinline__ TapRows tap_rows(const Vol& V) {
  TapRows R;
  if (PAIR) {                                   // pair copy: the z0 and the z1 face, 8 bytes per voxel
    R.d00 = (const char*)V.pair;
    R.d01 = R.d00 + 8u * (unsigned)V.sz;
    R.d10 = R.d11 = nullptr;
  } else {
    R.d00 = (const char*)V.data;
    R.d10 = R.d00 + 4u * (unsigned)V.sy;
    R.d01 = R.d00 + 4u * (unsigned)V.sz;
    R.d11 = R.d01 + 4u * (unsigned)V.sy;
  }
  return R;
}
// The 8 taps of a cell as the two 16-byte halves they are gathered as.  off = byte offset of corner 000 in the array
// gathered from (4 * flat index on the plain grid, 8 * flat index on the pair copy).
//   plain grid: q0 = (v000, v100 | v010, v110), q1 = (v001, v101 | v011, v111)   -- four 8-byte loads
//   pair copy : q0 = (v000, v010, v100, v110),  q1 = (v001, v011, v101, v111)    -- two 16-byte loads
template <bool PAIR>
__device__ __forceinline__ void gather_rows(const TapRows& R, unsigned off, f4& q0, f4& q1) {
  if (PAIR) {
    q0 = ld_quad8((const float*)(R.d00 + off)); q1 = ld_quad8((const float*)(R.d01 + off));
  } else {
    const f2 a = ld_pair((const float*)(R.d00 + off)), b = ld_pair((const float*)(R.d10 + off));
    const f2 e = ld_pair((const float*)(R.d01 + off)), f = ld_pair((const float*)(R.d11 + off));
    q0 = f4{a.x, a.y, b.x, b.y}; q1 = f4{e.x, e.y, f.x, f.y};
  }
}
template <bool PAIR>
__device__ __forceinline__ Taps taps_of(f4 q0, f4 q1) {
  if (PAIR) return taps_from_pair(q0, q1);
  Taps t;
  t.a = f2{q0.x, q0.y}; t.b = f2{q0.z, q0.w}; t.e = f2{q1.x, q1.y}; t.f = f2{q1.z, q1.w};
  return t;
}
template <bool PAIR>
__device__ __forceinline__ void taps_set(const Taps& t, f4& q0, f4& q1) {
  if (PAIR) { q0 = f4{t.a.x, t.b.x, t.a.y, t.b.y}; q1 = f4{t.e.x, t.f.x, t.e.y, t.f.y}; }
  else      { q0 = f4{t.a.x, t.a.y, t.b.x, t.b.y}; q1 = f4{t.e.x, t.e.y, t.f.x, t.f.y}; }
}
template <bool PAIR> __device__ __forceinline__ unsigned tap_offset(int base) { return (unsigned)base << (PAIR ? 3 : 2); }

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

// ---------------------------------------------------------------------------------------------
// adjoint march: backtrace (MODE 0), backtrace_sdf (MODE 1); direct global atomics variant
// ---------------------------------------------------------------------------------------------
struct BackArgs {
  Vol vol;
  const float* sdf;
  const void* xt; const void* vt; const void* dx; const void* dv;   // fp32, or half when io_half
  int io_half;
  float* grad;
  const uint32_t* perm;
  drrt_stats* stats;
  size_t n;
  float ds;
  float grad_scale;            // 1 (as written, Q3) or 1/h (DRRT_FLAG_CORRECTED_H)
  int max_steps;
  unsigned long long* dbg;     // nullable: [0] window flushes, [1] taps via LDS, [2] taps via global fallback
  int experiment;              // development ablations (0 = product behaviour)
  unsigned* select;            // nullable (k_backtrace_flat): [0] waves a fitted window would help, [1] waves classified
  const uint32_t* fsteps;      // nullable: per-ray iteration counts of the forward march that produced (xt, vt) (step hint)
  int xcd_order;               // 1: the launch's blocks take the visit order XCD by XCD (xcd_block)
  // resumable march (drrt_backtrace_chunk_f32; k_backtrace_flat<..., CHUNK = true>): nullable
  float* chunk_state;          // 13 words per visit slot, SoA with stride chunk_stride: x, v, lambda, mu, flags
  size_t chunk_stride;         // visit slots of the launch (grid * block)
  int chunk_resume;            // 0: start from (xt, vt, dx, dv); 1: continue from chunk_state
  int* chunk_progress;         // nullable: 13 ints, see drrt_backtrace_chunk_f32
};

// order-preserving integer key of a float (for atomicMin / atomicMax on floats of either sign); its own inverse
__host__ __device__ inline int float_order_key(float f) {
  int i;
  memcpy(&i, &f, sizeof(i));
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ inline float float_from_order_key(int k) {
  const int i = k >= 0 ? k : k ^ 0x7fffffff;
  float f;
  memcpy(&f, &i, sizeof(f));
  return f;
}

// Accumulators are DOUBLES: measured on gfx950 (tools/lds_atomic_bench.hip) ds_add_f32 costs ~193
// cycles per wave-instruction per CU even without address collisions (~3 cycles per lane), while
// ds_add_f64 costs ~8 (ds_add_u32 4.4); collisions add ~12 cycles per colliding lane for f64.  The
// window sums are therefore also more accurate than fp32 atomics; they are rounded to fp32 once,
// when the window is flushed into the fp32 grid.
typedef double win_t;
// k_backtrace_flat and k_backtrace_ring run ONE wave per block: nothing in them is shared between the waves of a block (each
// wave owns its window), and a block's LDS and wave slots come free only when its last wave has finished -- with four waves of
// different lengths per block that held resources idle.  Measured, same box, 256 / 128 / 64 threads per block (all kernels):
// box-window adjoint 4.63-4.65 / 4.61-4.67 / 4.54-4.58 ms, ring-window adjoint on the six rotated views 8.84-8.98 / 8.70-8.98 /
// 8.60-8.72 ms, forward march 0.99-1.00 / 1.02 / 1.02-1.03 ms (it keeps 256).
#ifndef DRRT_ADJ_BLOCK
#define DRRT_ADJ_BLOCK 64
#endif
constexpr int kAdjBlock = DRRT_ADJ_BLOCK;
constexpr int kAdjWavesPerBlock = kAdjBlock / kWave;
static inline unsigned adj_grid_for(size_t n) { return (unsigned)((n + kAdjBlock - 1) / kAdjBlock); }

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): LDS ops of this wave have completed
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- quad pre-reduction helpers (DPP; lanes outside the current branch read as -1 / 0) ----------------
__device__ __forceinline__ bool quad_same_key(int key) {
  const int k1 = __builtin_amdgcn_update_dpp(-1, key, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  const int k2 = __builtin_amdgcn_update_dpp(-1, key, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  const int k3 = __builtin_amdgcn_update_dpp(-1, key, 0x1B, 0xF, 0xF, false);   // quad_perm [3,2,1,0]
  return (k1 == key) & (k2 == key) & (k3 == key);
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  return v;
}

// How do the 64-ray bundles of this call sit at the start of the adjoint march?  One wave per bundle (the rays of 64
// consecutive visit slots), every 16th block of bundles:
//   [0] += 1 when the bounding box of the rays' start cells does not fit the default box window, [1] += 1 per bundle;
//   [2] += the lanes whose start cell lies more than kClassifyReach cells (on any axis) from the bundle's mean cell,
//          [3] += the lanes (diagnostic only: it does NOT predict which kernel is faster, see below);
//   [4] += the lanes whose pair partner (lane ^ 1) starts in the same cell (diagnostic);
//   [5]    set by the host: no sparse-only ring instantiation for this call;
//   [6] += 1 when the bundle's rays left the forward march >= kClassifyLongCells / (ds / h) iterations apart (step hint):
//          see bundles_long.
// k_backtrace_flat and k_backtrace_ring read the counters (bundles_want_ring): one of them runs.
// Calibration (tools/probe_classify.py, 256^3, 1M rays unless noted; share of bundles not fitting -> box / ring kernel ms):
//   metric 4 % -> 4.6 / 6.3; shifted plane 5 % -> 4.7 / 6.4; one view at 0 / 20 / 45 degrees through a weak lens 0 / 13 / 2 %
//   -> 3.6 / 4.5, 4.3 / 5.8, 4.1 / 5.7; the same views through the Luneburg ball 4 / 14 / 11 % -> 4.6 / 6.3, 8.2 / 7.1,
//   8.2 / 7.9; 527k rays ending on a sphere inside the lens, sorted by the adjoint itself 14 % -> 2.9 / 4.2; four sparse
//   tomography views 45 % -> 8.9 / 5.3; six rotated views 28 % -> 21.3 / 8.9 (38 % -> 25.2 / 18.5 when the adjoint sorts).
// Between 11 and 14 % the two kernels trade places by -34 ... +14 %; from 28 % on the ring kernel wins by 1.4-2.4x.  The
// threshold is a fifth of the bundles (rounds 2-3: an eighth, which sent the 13-14 % cases to the ring kernel: -34 / -42 %).
constexpr unsigned kClassifyStride = 16;     // every 16th block of bundles is looked at (contended atomics are the cost)
constexpr int kClassifyReach = 3;            // 9 slots = 8 cells: the mean cell +- 3 and the upper taps, one cell to spare
#ifndef DRRT_RING_MIN_NOFIT_PCT
#define DRRT_RING_MIN_NOFIT_PCT 20
#endif
// Second criterion (round 4, once the sparse-only ring instantiation had its fixed-point window): bundles whose rays left the
// FORWARD march many iterations apart.  k_backtrace_flat starts every ray at its own exit sample (it ignores the step hint),
// so such a bundle runs through the volume spread along its path by half a cell per iteration of difference, beyond any 9-slot
// window, while the start cells (all the first criterion looks at) still fit.  tools/probe_angle_sweep.py +
// probe_bundle_stats.py (256^3, ds = h / 2, one plane view of 512^2 x 4 samples at an angle about z; share of bundles whose
// iteration counts spread over >= 24 = 12 cells of travel -> box / sparse-only ring ms): through the Luneburg ball 0 deg 4.0 % -> 4.46 / 5.06, 1 deg 5.5 % ->
// 4.96 / 5.16, 2 deg 6.5 % -> 5.01 / 5.43, 5 deg 8.6 % -> 5.77 / 5.15, 10 deg 9.6 % -> 7.46 / 5.42, 20 deg 9.2 % -> 8.06 / 5.42,
// 30 deg 9.9 % -> 8.66 / 5.75, 45 deg 9.3 % -> 8.10 / 6.00 (shares not fitting: 4, 6, 7, 8, 11, 15, 16, 12 % -- they do not
// separate the two groups, nor does the mean or any capped mean of the start boxes' extents); through the weak medium, any
// angle, <= 5 % (the box window wins by 0-20 %); the metric's source 4.2 % -> 4.44 / 5.01, moved by a third of a pixel 5.3 %;
// six rotated views 27 % (ball), 10 % (weak).  From 7.5 % on the call goes to the ring kernel, whatever the first share says.
constexpr float kClassifyLongCells = 12.f;   // iterations * ds / h (24 iterations at the calibration's ds = h / 2)
#ifndef DRRT_RING_MIN_LONG_PERMILLE
#define DRRT_RING_MIN_LONG_PERMILLE 75
#endif
__device__ __forceinline__ bool bundles_long(const unsigned* __restrict__ sel) {
  return sel[6] * 1000u >= sel[1] * (unsigned)DRRT_RING_MIN_LONG_PERMILLE && sel[6] != 0u;
}
__device__ __forceinline__ bool bundles_want_ring(const unsigned* __restrict__ sel) {
  return (sel[0] * 100u >= sel[1] * (unsigned)DRRT_RING_MIN_NOFIT_PCT && sel[0] != 0u) || bundles_long(sel);
}
// ... and which instantiation of the ring kernel: the sparse-only one (compiled without the dense path; 2500-slot fixed-point
// window) unless sel[5] != 0 -- set by the host for backtrace_sdf, the ablation / counter build and DRRT_FLAG_RING_GENERAL,
// which only the general instantiation (per-wave dense / sparse rule, fp64 window of 1250 slots) serves.  Until the
// sparse-only window went to fixed point the choice was made per call from the density of the visit order (pair-sharing
// counters computed by the sort: the general instantiation was 13-34 % faster on dense 4-samples-per-pixel views); with 2500
// slots the sparse-only one wins on EVERY set the ring kernel is chosen for (tools/probe_ring_sets.py, round 4, general /
// sparse-only ms: six rotated views 8.4 / 7.1 (ball), 5.9 / 4.8 (weak medium); four views at 4 samples per pixel 5.0 / 4.3
// (blob), 7.4 / 6.4 (ball); one view at 45 degrees 7.2 / 6.0, at 20 degrees 6.5 / 5.4), and the counters, their kernel and
// their hand-over are gone.  (On compact dense sets -- the metric's -- the general instantiation equals the box window,
// 4.44 ms, and the sparse-only one takes 5.0: those never come here.)
__device__ __forceinline__ bool bundles_want_sparse(const unsigned* __restrict__ sel) { return sel[5] == 0u; }
// ... and of the two sparse-only instantiations the DIRECT one (every step's contributions straight to the window, see
// k_backtrace_ring) when few lanes share a cell: counter [4] / [3] = share of the sampled lanes whose pair partner starts in
// the same cell.  Six views through the weak medium 0.31 (2 lanes per cell: direct 6 % faster), through the Luneburg ball 0.56
// (11 per cell: equal), four views at 4 samples per pixel 0.49 (21 per cell inside the ball: direct 40 % slower).  The start
// cells say little about a lens set's inside (why this counter could not pick general / sparse-only in the first place), so the
// threshold sits well below the first set that loses.
#ifndef DRRT_RING_DIRECT_MAX_PAIR_PCT
#define DRRT_RING_DIRECT_MAX_PAIR_PCT 40
#endif
__device__ __forceinline__ bool bundles_want_direct(const unsigned* __restrict__ sel) {
  return sel[5] == 0u && sel[3] != 0u && sel[4] * 100u < sel[3] * (unsigned)DRRT_RING_DIRECT_MAX_PAIR_PCT;
}

// ---------------------------------------------------------------------------------------------
// cable (radial profile) variants, src/tracer.cpp:312-382 and :511-567
// The profile (<= a few hundred floats) lives in LDS; the adjoint accumulates into an LDS copy
// of the gradient profile (ds_add_f64) and flushes it once per block -- millions of rays would
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

// ---- launchers (host): one per kernel family, defined next to the kernels ------------------------------------------
static inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
// forward (drrt_forward.hip).  mode: 0 = trace, 1 = trace_plane, 2 = trace_sdf
void launch_trace(int mode, const TraceArgs& a, hipStream_t s);
void launch_trace_again(int mode, const TraceArgs& a, hipStream_t s);
void launch_target(const TargetArgs& a, hipStream_t s);                 // phase A + phase B
// adjoint.  mode: 0 = backtrace, 1 = backtrace_sdf; abl: the instantiation with ablation switches / debug counters
void launch_backtrace_direct(int mode, const BackArgs& a, hipStream_t s);
void launch_bundle_classify(const BackArgs& a, hipStream_t s);
void launch_backtrace_box(int mode, bool abl, const BackArgs& a, hipStream_t s);
void launch_backtrace_ring(int mode, bool abl, const BackArgs& a, hipStream_t s);
void launch_backtrace_ring_sparse(const BackArgs& a, hipStream_t s, int which);   // the sparse-only instantiations (backtrace)
// cable (drrt_cable.hip)
void launch_trace_cable(const CableArgs& a, hipStream_t s);
void launch_backtrace_cable(const CableArgs& a, hipStream_t s);

}  // namespace drrt
