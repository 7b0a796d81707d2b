// drrt_adjoint_ring.hip -- gfx950 kernel of the adjoint march Tracer::backtrace / backtrace_sdf
// (/root/reference/src/tracer.cpp:384-509) for ray bundles that do not sit in a compile-time box window (sparse views,
// views oblique to the grid -- the reference's own six randomly rotated views, core/source.py:398-412,555-563).
// Its own translation unit: csrc/Makefile builds it with -mllvm -amdgpu-sched-strategy=max-ilp.
#include "drrt_march.h"

namespace drrt {

// ---------------------------------------------------------------------------------------------
// k_backtrace_ring: the adjoint march of k_backtrace_flat (same per-ray arithmetic, same register accumulators and face
// carry-over) with a RING WINDOW: the wave's LDS window is addressed modulo its size on every axis (storage coordinate of
// voxel g: (b + g - o) mod n, invariant while the voxel is live), so the window FOLLOWS the rays -- when lanes step ahead
// of it the layers that every lane has left behind are flushed (one global atomic per touched voxel, each voxel once) and
// their storage is reused ahead; nothing else moves.  Dimensions are fitted to the bundle when the window is anchored.
// The kernel for ray sets whose 64-ray bundles do not sit in the compile-time window of k_backtrace_flat (sparse views,
// views oblique to the grid: k_bundle_classify decides per call, on the device).
//
// Why (round 3, measured on the reference's own ray distribution -- six plane views turned by a random rotation,
// core/source.py:398-412,555-563): the box window of k_backtrace_flat is flushed WHOLE and re-anchored whenever a lane
// leaves it.  A bundle that is oblique to the grid fills its bounding box in all three axes, so the window had room for
// 2-6 steps; every re-anchor re-flushed ~1000 slots, every voxel went to memory 3-4 times, lanes outside the window fell
// back to one global atomic per tap, and the memory-side atomic units (~2e10 requests/s) became the limiter.
//   * ring addressing: a voxel is flushed once; a lane that steps ahead is served before it has anything to emit;
//   * every slot address is computed modulo the window, so it is always INSIDE the wave's array: a leave across two or
//     three faces is handled as two or three one-face crossings in a row (the later ones hand over zeros for the corners
//     the earlier ones cleared -- wherever those land, adding 0.0 changes nothing) instead of handing over all eight
//     corners un-reduced: the per-axis blocks run anyway when a wave's lanes cross different faces, which is every step
//     for rays oblique to the grid;
//   * dense or sparse, wave by wave: while fewer than half of the lanes share their cell with their pair partner the wave
//     hands over all eight corners on every leave instead (no per-axis blocks at all; see `sparse` below);
//   * lanes far from the bundle (a bundle torn apart at the rim of a lens) do not take part in the window's decisions.
// ---------------------------------------------------------------------------------------------
#ifndef DRRT_RING_WAVES
#define DRRT_RING_WAVES 4           // waves per SIMD: 128 VGPRs (the kernel needs ~115; at the 96 of 5 waves it spills in the loop)
#endif
#ifndef DRRT_RING_CAP
#define DRRT_RING_CAP 1250          // slots per wave (10000 B): 4 blocks of 4 waves per CU fill the 160 KiB of LDS
#endif
#ifndef DRRT_RING_SLACK
#define DRRT_RING_SLACK 4           // slots of room along the dominant travel axis when the window is fitted (six rotated views,
                                    // same box: 4 -> 10.3 ms, 8 -> 10.5, 12 -> 10.9; 3 waves per SIMD with 1660 slots: 11.9)
#endif
#ifndef DRRT_RING_GROW
#define DRRT_RING_GROW 0
#endif
#ifndef DRRT_RING_SLACK_MIN
#define DRRT_RING_SLACK_MIN 2       // slots of room on every axis when the window is fitted
#endif
#ifndef DRRT_RING_SIMPLE
#define DRRT_RING_SIMPLE 0
#endif
#ifndef DRRT_RING_FLUSH_BATCH
#define DRRT_RING_FLUSH_BATCH 2     // LDS exchanges in flight per lane in a flush (4 costs ~10 more VGPRs at the kernel's pressure peak)
#endif
#ifndef DRRT_RING_DENSE_PCT
#define DRRT_RING_DENSE_PCT 50      // a wave's bundle counts as dense while at least this percentage of its lanes share their cell with
                                    // their pair partner (sampled every 16th iteration).  Same box, ring kernel forced, ms of the adjoint
                                    // on: six rotated views / metric / 4-view tomography set (tools/probe_views.py):
                                    //   never sparse 9.9 / 5.9 / 5.0;  50 % -> 9.0 / 5.9 / 5.2;  70 % -> 8.9 / 6.8 / 5.7;  always sparse 8.9 / 8.8 / 5.9
#endif
// Timing-only ablations of the PRODUCT instantiation (tools/build_variant.sh; the results of such a build are wrong by
// construction and it is never shipped): -DDRRT_RING_T_NO_LDS drops the window adds, -DDRRT_RING_T_U32 makes them 32-bit
// integer adds, -DDRRT_RING_T_NO_GLOBAL drops every global atomic, -DDRRT_RING_T_NO_FLUSH the flush loops.
#if defined(DRRT_RING_T_NO_LDS)
#define RING_ADD(q, v) ((void)0)
#elif defined(DRRT_RING_T_U32)
#define RING_ADD(q, v) atomicAdd(reinterpret_cast<unsigned*>(q), __float_as_uint(v))
#else
#define RING_ADD(q, v) atomicAdd((q), (win_t)(v))
#endif
#if defined(DRRT_RING_T_NO_GLOBAL)
#define RING_GADD(g, v) ((void)0)
#else
#define RING_GADD(g, v) atomic_add_f32((g), (v))
#endif
// Diagnostic build only (-DDRRT_RING_STAMPS, tools/ring_stamps.py; never in the product library): wave-level s_memtime
// brackets around the regions of an iteration, summed over the launch -- where a wave's TIME goes (issue + waiting), which
// the PMC instruction counts cannot say.  Stamp values go to a buffer of their own that nothing else reads.
#if defined(DRRT_RING_STAMPS)
// fixed-point window, per wave: [0] budget used up -> complete flush, [1] a hand-over left the range -> complete flush,
// [2] budget used up -> the window's largest slot looked at, nothing flushed, [3] re-scales; per lane: [4] hand-overs the
// guard sent to the grid, [5] hand-overs of the large class
__device__ unsigned long long g_ring_events[8];
#define QEVENT(k, n) { if (lane == 0 || (k) >= 4) atomicAdd(&g_ring_events[k], (unsigned long long)(n)); }
__device__ unsigned long long g_ring_stamps[8];
#define STAMP_DECL unsigned long long st_t = 0ull, st_acc[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; unsigned long long st_t0 = __builtin_amdgcn_s_memtime(); st_t = st_t0
#define STAMP(k) { const unsigned long long st_n = __builtin_amdgcn_s_memtime(); st_acc[k] += st_n - st_t; st_t = st_n; }
#define STAMP_END { const unsigned long long st_n = __builtin_amdgcn_s_memtime(); \
    if (lane == 0) { atomicAdd(&g_ring_stamps[0], st_n - st_t0); for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_ring_stamps[1 + k_], st_acc[k_]); atomicAdd(&g_ring_stamps[7], 1ull); } }
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_END
#define QEVENT(k, n)
#endif
struct Ring {                      // wave-uniform
  int nx, ny, nz;                  // slots per axis (>= 2)
  int sy, sz;                      // LDS strides of y and z in slots: nx, nx * ny
  int tot;                         // sparse-only instantiation: nx * ny * nz (the wrap strides sy - sz and sz - tot then cost a
                                   // subtraction each where -(n - 1) * stride costs a quarter-rate v_mul_lo_u32 per hand-over)
  int ox, oy, oz;                  // grid coordinates of the low corner of the live region [o, o + n)
  int bx, by, bz;                  // storage coordinates of that corner: voxel g lives at (b + g - o) mod n
};
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// wave-wide min / max without LDS traffic: DPP row shifts inside the four 16-lane rows, row broadcasts across them
__device__ __forceinline__ int wave_min_dpp(int v) {
  const int id = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xF, 0xF, false));    // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xF, 0xF, false));    // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x114, 0xF, 0xF, false));    // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x118, 0xF, 0xF, false));    // row_shr:8
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xA, 0xF, false));    // row_bcast:15 -> rows 1, 3
  v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xC, 0xF, false));    // row_bcast:31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_dpp(int v) { return -wave_min_dpp(-v); }

// Window slot of the regular cell (ix, iy, iz) (corner 000) and its storage coordinates, or -1 when the cell (it needs the
// slots g and g + 1 on every axis) is not inside the live region; `nearby`: within two cells of it on every axis.
__device__ __forceinline__ int ring_locate(const Ring& R, int ix, int iy, int iz, int& sx, int& sy, int& sz, bool& nearby) {
  const int rx = ix - R.ox, ry = iy - R.oy, rz = iz - R.oz;
  const bool in = ((unsigned)rx <= (unsigned)(R.nx - 2)) & ((unsigned)ry <= (unsigned)(R.ny - 2)) &
                  ((unsigned)rz <= (unsigned)(R.nz - 2));
  nearby = ((unsigned)(rx + 2) <= (unsigned)(R.nx + 2)) & ((unsigned)(ry + 2) <= (unsigned)(R.ny + 2)) &
           ((unsigned)(rz + 2) <= (unsigned)(R.nz + 2));
  sx = R.bx + rx; sx = sx >= R.nx ? sx - R.nx : sx;
  sy = R.by + ry; sy = sy >= R.ny ? sy - R.ny : sy;
  sz = R.bz + rz; sz = sz >= R.nz ? sz - R.nz : sz;
  return in ? mad24(sz, R.sz, mad24(sy, R.sy, sx)) : -1;
}

// Flush the k layers g0 .. g0 + k - 1 of axis A (grid coordinates; they must lie inside the live region) into the grid and
// leave their slots zeroed.  All 64 lanes.  The slots are enumerated x fastest so that the lanes of one atomic instruction
// cover runs of x-neighbours.
// WT = double (the general instantiation) or int (the sparse-only one: fixed point, value = slot * qinv).
template <int A, typename WT>
__device__ __forceinline__ void ring_flush(WT* win, const Ring& R, int g0, int k, float* __restrict__ grad, const Vol& V,
                                           int lane, bool no_global, float qinv = 1.0f) {
#if defined(DRRT_RING_T_NO_FLUSH)
  return;
#endif
  wave_lds_fence();
  const int e0 = A == 0 ? k : R.nx, e1 = A == 1 ? k : R.ny, e2 = A == 2 ? k : R.nz;
  const int e01 = e0 * e1, total = e01 * e2;
  // (v_rcp_f32, not an IEEE division -- ten instructions each, twice per flush: with e < 2500 the quotient (e + 0.5) / e01 keeps
  // 0.5 / e01 >= 2e-4 from the next integer, the reciprocal's error moves it by < 2500 * 2e-7.  Measured: nothing either way --
  // nor did 24-bit multiplies for the slot and voxel indices of the flush; the flush waits for LDS and atomics, not for VALU)
  const float inv0 = __builtin_amdgcn_rcpf((float)e0), inv01 = __builtin_amdgcn_rcpf((float)e01);
  const int r0 = g0 - (A == 0 ? R.ox : (A == 1 ? R.oy : R.oz));       // first layer, relative to the low corner
  constexpr int kBatch = DRRT_RING_FLUSH_BATCH;
#pragma unroll 1
  for (int e_base = 0; e_base < total; e_base += kWave * kBatch) {
    WT v[kBatch];
    unsigned g[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const int e = e_base + b * kWave + lane;
      const int jz = (int)(((float)e + 0.5f) * inv01), r = e - jz * e01;
      const int jy = (int)(((float)r + 0.5f) * inv0), jx = r - jy * e0;
      // on the flushed axis j counts layers from g0; on the others it IS the storage coordinate
      int sx, sy, sz, gx, gy, gz;
      if (A == 0) { const int rel = r0 + jx; sx = R.bx + rel; sx = sx >= R.nx ? sx - R.nx : sx; gx = g0 + jx; }
      else        { sx = jx; int rel = jx - R.bx; rel = rel < 0 ? rel + R.nx : rel; gx = R.ox + rel; }
      if (A == 1) { const int rel = r0 + jy; sy = R.by + rel; sy = sy >= R.ny ? sy - R.ny : sy; gy = g0 + jy; }
      else        { sy = jy; int rel = jy - R.by; rel = rel < 0 ? rel + R.ny : rel; gy = R.oy + rel; }
      if (A == 2) { const int rel = r0 + jz; sz = R.bz + rel; sz = sz >= R.nz ? sz - R.nz : sz; gz = g0 + jz; }
      else        { sz = jz; int rel = jz - R.bz; rel = rel < 0 ? rel + R.nz : rel; gz = R.oz + rel; }
      v[b] = (WT)0;
      g[b] = (unsigned)gz * (unsigned)V.sz + (unsigned)gy * (unsigned)V.sy + (unsigned)gx;
      // ds_wrxchg_rtn_b64: read the accumulated value and reset the slot in one LDS op
      if (e < total) v[b] = __hip_atomic_exchange(win + (sz * R.sz + sy * R.sy + sx), (WT)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b)
      if (v[b] != (WT)0 && !no_global) {
        const float fv = (std::is_same<WT, int>::value) ? (float)v[b] * qinv : (float)v[b];
        RING_GADD(grad + g[b], fv);
      }
  }
  wave_lds_fence();
}

// One lane crosses ONE face along axis A (storage coordinate sA, size nA, LDS stride SA; in-face axes P, Q) from the cell
// whose slot is `cur`: the four corners left behind (e0..e3 in (p, q) order) go to the window -- pair / quad DPP
// pre-reduced as in k_backtrace_flat while PRE -- and (cur, sA) move to the neighbour cell, modulo the window.
// Returns true when the new cell lies outside the live region on that axis.
template <bool ABL, typename WT>
__device__ __forceinline__ bool ring_cross(WT* win, int experiment, bool pre, int axis_id, bool fwd, int& cur, int& sA, int sP,
                                           int sQ, int nA, int nP, int nQ, int SA, int SP, int SQ, int gA_new, int oA,
                                           float e0, float e1, float e2, float e3, bool& matched,
                                           unsigned& ev_face, unsigned& ev_add, bool dbg) {
  // neighbour slots of the old cell: + stride, or back to storage layer 0 across the seam
  const int dA = sA == nA - 1 ? -(nA - 1) * SA : SA;
  const int dP = sP == nP - 1 ? -(nP - 1) * SP : SP;
  const int dQ = sQ == nQ - 1 ? -(nQ - 1) * SQ : SQ;
  const int qi = cur + (fwd ? 0 : dA);
  if (experiment != 3) {
    if (pre) {
      const int key = qi | (axis_id << 16);
      const int k1 = __builtin_amdgcn_update_dpp(-1, key, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
      const int k2 = __builtin_amdgcn_update_dpp(-1, key, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
      const int k3 = __builtin_amdgcn_update_dpp(-1, key, 0x1B, 0xF, 0xF, false);   // quad_perm [3,2,1,0]
      const bool psame = k1 == key;
      const bool same = psame & (k2 == key) & (k3 == key);
      matched |= psame;
      float q0 = e0, q1 = e1, q2 = e2, q3 = e3;
      q0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q0), 0xB1, 0xF, 0xF, false));
      q1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q1), 0xB1, 0xF, 0xF, false));
      q2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q2), 0xB1, 0xF, 0xF, false));
      q3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q3), 0xB1, 0xF, 0xF, false));
      const float s0 = q0 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q0), 0x4E, 0xF, 0xF, false));
      const float s1 = q1 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q1), 0x4E, 0xF, 0xF, false));
      const float s2 = q2 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q2), 0x4E, 0xF, 0xF, false));
      const float s3 = q3 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q3), 0x4E, 0xF, 0xF, false));
      const unsigned ql = threadIdx.x & 3u;
      const bool add = same ? ql == 0u : (psame ? (ql & 1u) == 0u : true);
      if (ABL && dbg) { ++ev_face; ev_add += add; }
      if (add) {
        WT* q = win + qi;
        RING_ADD(q, (same ? s0 : (psame ? q0 : e0)));      RING_ADD(q + dP, (same ? s1 : (psame ? q1 : e1)));
        RING_ADD(q + dQ, (same ? s2 : (psame ? q2 : e2))); RING_ADD(q + dQ + dP, (same ? s3 : (psame ? q3 : e3)));
      }
    } else {
      if (ABL && dbg) { ++ev_face; ++ev_add; }
      WT* q = win + qi;
      RING_ADD(q, e0); RING_ADD(q + dP, e1); RING_ADD(q + dQ, e2); RING_ADD(q + dQ + dP, e3);
    }
  }
  // the new cell: one step along A in storage
  int t = sA + (fwd ? 1 : -1);
  t = t == nA ? 0 : t;
  t = t < 0 ? nA - 1 : t;
  cur += (t - sA) * SA;
  sA = t;
  return (unsigned)(gA_new - oA) > (unsigned)(nA - 2);
}

// SPARSE: the instantiation without the dense path (no per-axis crossings, no pair-partner sampling): every leave hands over
//         all eight corners.  On ray sets like the reference's six rotated views the per-wave rule of the general
//         instantiation settles on "sparse" for three quarters of the lane-steps anyway; compiled without the dense code the
//         same march ran 4-9 % faster (six rotated views 8.49 -> 8.13 ms, weak medium 5.86 -> 5.31 ms, round 4) while dense
//         multi-sample views lost (one 45-degree view at 4 samples per pixel: 7.2 -> 9.6 ms) -- until its window went to
//         fixed point (below): since then it is the instantiation every call the classification sends to the ring kernel
//         takes (bundles_want_sparse, drrt_march.h), and the general one serves backtrace_sdf, the counter build and A-B.
#ifndef DRRT_RING_SPARSE_WAVES
#define DRRT_RING_SPARSE_WAVES DRRT_RING_WAVES      // (A-B: occupancy of the sparse-only instantiation)
#endif
#ifndef DRRT_RING_SPARSE_CAP
#define DRRT_RING_SPARSE_CAP (2 * DRRT_RING_CAP)      // 4-byte slots: the same 10 000 B per wave hold twice the window
#endif
// The sparse-only instantiation keeps its window in 32-bit FIXED POINT (slot = round(value * 2^e), one exponent per wave):
// `ds_add_u32` completes in a third of the time of `ds_add_f64` (tools/lds_random_bench.hip) and, more important, the same
// LDS holds twice the slots -- the window is twice as long along the rays' travel, slides half as often and fits bundles
// that a 1250-slot window could not hold (timing-only build with garbage sums, round 4: six rotated views 7.9 -> 6.5 ms,
// weak medium 5.3 -> 4.5 ms).  What makes it exact enough and safe:
//   scale     the lane accumulators already carry the factor 2^e (the splat weights are linear in their inputs and a power of
//             two is an exact factor), chosen so that the largest accumulator of the wave lies in [2^19, 2^20) when the
//             scale is set -- at the wave's first contributing step, and again whenever the window is EMPTY (a re-fit, a
//             forced flush) and the largest hand-over since the last such chance has left [2^17, 2^20);
//   rounding  v_cvt_rpi_i32_f32 (floor(x + 0.5)): half a unit = 2^-20..2^-21 of the wave's largest hand-over, unbiased.  In a
//             weak medium the gradient is a small difference of large hand-overs, which is what sets the bits needed: with
//             [2^15, 2^16) the six views through n = 1 + 3e-4 U came out 3.3e-5 from the oracle (bound 2e-5), with
//             [2^17, 2^18) 9e-6 and one configuration of the differential fuzz 2.1e-5, with [2^19, 2^20) 2.6e-6 (round 4);
//   guard     a hand-over whose largest value is not below 2^22 (or is not finite, or comes before the scale is set)
//             goes to the grid with fp32 atomics, unscaled, and asks for a re-scale;
//   overflow  a lane-emit adds less than 2^20 to any slot -- or less than 2^22, and then counts four times -- and after
//             kQBudget = 2024 counted lane-emits of the wave the window is LOOKED AT (ten 16-byte LDS reads per lane): while
//             the scale still fits, the count restarts from what the largest |slot| has really used (the count is a worst
//             case: every lane-emit on one slot, one sign, top of its class); otherwise -- and whenever a hand-over left
//             the range -- the whole window is flushed (and zeroed) and the scale re-chosen.  used + counted <= 2024 and
//             2024 * 2^20 < 2^31: no slot can overflow whatever the rays do.  Events per wave on the six rotated views
//             (tools/ring_stamps.py, round 4): 8.6 looks that kept the window, 2.0 + 0.2 complete flushes, 3.8 re-scales;
//             1.5e-4 of the lane hand-overs go to the grid through the guard.  (The looks replaced ~10 complete flushes per
//             wave and bought nothing measurable: 7.31-7.33 vs 7.32-7.35 ms.  Nor did a second, younger counter that takes
//             over once the window's rear has passed the place its front had reached -- a moving window renews its slots by
//             itself -- 7.34-7.44.  A timing-only build without any budget ran 6.95 against 7.25: that is the compiler's
//             arrangement of the loop without the block, not the flushes.)
//             Six rotated views, same box (gpurun_out/r4p), target range / budget, complete flushes only:
//             [2^17, 2^18) / 4072 -> 7.25-7.43 ms (fuzz fails), [2^18, 2^19) / 2024 -> 7.39-7.43, [2^19, 2^20) / 1000 -> 7.51-7.60,
//             [2^19, 2^20) with the small class ending at 2^20 / 2024 -> 7.41-7.45 (the default), [2^20, 2^21) / 1000 -> 7.5-7.6.
#ifndef DRRT_RING_QBITS
#define DRRT_RING_QBITS 19                   // the wave's largest accumulator is scaled into [2^QBITS, 2^(QBITS+1))
#endif
#ifndef DRRT_RING_QSMALL
#define DRRT_RING_QSMALL 1                   // the small class ends 2^QSMALL above the bottom of the target range (at its top)
#endif
constexpr float kQSmall = (float)(1u << (DRRT_RING_QBITS + DRRT_RING_QSMALL));       // below it a lane-emit counts once against the budget, else four times
constexpr float kQGuard = (float)(1u << (DRRT_RING_QBITS + DRRT_RING_QSMALL + 2));   // above it a hand-over goes to the grid
constexpr float kQLow = (float)(1u << (DRRT_RING_QBITS - 2));     // the scale is raised when the largest hand-over of a period stays below it
constexpr unsigned kQBudget = (1u << (31 - (DRRT_RING_QBITS + DRRT_RING_QSMALL))) - 24u;   // counted lane-emits between two complete flushes
// max(|a|, |b|, |c|) in one instruction (the compiler builds fabsf as v_max(|x|, |x|) first: 11 instructions for eight values)
__device__ __forceinline__ float max3abs(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max_raw(float a, float b) {   // one v_max_f32 (fmaxf quiets both operands first: three)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ int cvt_rpi_i32(float f) {          // floor(f + 0.5)
  int i;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(i) : "v"(f));
  return i;
}
// DIRECT (with SPARSE): every STEP's eight contributions go to the window at once -- no per-lane accumulators, no hand-over
//         on leaving the cell, at the ray's end or at the kernel's: 106 instead of 128 VGPRs, 2530 instead of 2944 static VALU
//         instructions, one inlined copy of the hand-over instead of four.  8 instead of ~5 LDS adds per lane-step, which is
//         free where few lanes share a cell (the six views through the weak medium, 2 lanes per cell: 4.84-4.88 -> 4.54-4.59 ms;
//         through the Luneburg ball, 11 per cell: 6.99-7.21 -> 6.94-7.05) and ruinous where many do (same-address LDS adds
//         serialise: four views at 4 samples per pixel through the ball, 21 lanes per cell: 6.4 -> 9.0 ms; one view at 45
//         degrees 6.0 -> 7.6) -- taken per CALL when few pair partners start in the same cell (bundles_want_direct).
template <bool ABL, bool PAIR, int MODE = 0, bool SPARSE = false, bool DIRECT = false>
__global__ void __launch_bounds__(kAdjBlock, SPARSE ? DRRT_RING_SPARSE_WAVES : DRRT_RING_WAVES) k_backtrace_ring(BackArgs a) {
  constexpr bool kDirect = SPARSE && DIRECT;
  constexpr int kRingCap = SPARSE ? DRRT_RING_SPARSE_CAP : DRRT_RING_CAP;
  using WT = typename std::conditional<SPARSE, int, win_t>::type;
  if (a.select != nullptr) {                                 // launched next to k_backtrace_flat: the bundle
    const bool want_fit = bundles_want_ring(a.select);                                // classification picks one of the three
    if (!want_fit || bundles_want_sparse(a.select) != SPARSE) return;
    if (SPARSE && bundles_want_direct(a.select) != DIRECT) return;
  }
  __shared__ WT s_win[kAdjWavesPerBlock][kRingCap];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  WT* win = s_win[wid];
  for (int k = lane; k < kRingCap; k += kWave) win[k] = (WT)0;
  wave_lds_fence();

  const Vol& V = a.vol;
  const size_t t = (size_t)xcd_block(blockIdx.x, gridDim.x, a.xcd_order ? DRRT_RING_XCD_MODE : kXcdOff) * kAdjBlock + threadIdx.x;
  AdjState s;
  s.x = s.y = s.z = s.vx = s.vy = s.vz = s.lx = s.ly = s.lz = s.mx = s.my = s.mz = 0.f;
  s.active = false; s.outside = false;
  size_t i;
  if (ray_index(a.perm, t, a.n, i)) {
    Ray3 p = ld3(a.xt, i, a.io_half, &a.vol, RAY_POS), u = ld3(a.vt, i, a.io_half, &a.vol, RAY_VEL), gxv = ld3(a.dx, i, a.io_half), gvv = ld3(a.dv, i, a.io_half);
    s.x = p.x; s.y = p.y; s.z = p.z; s.vx = u.x; s.vy = u.y; s.vz = u.z;
    adj_init(V, a.ds, gxv.x, gxv.y, gxv.z, gvv.x, gvv.y, gvv.z, s);
    if (MODE == 1 && s.active) {                                            // src/tracer.cpp:476-477
      const Cell c0 = locate(V, s.x, s.y, s.z);
      s.outside = interp<false>(fetch(a.sdf, c0), c0.wx, c0.wy, c0.wz).n >= 0.f;
    }
  }
  const int experiment = ABL ? a.experiment : 0;
  const bool dbg = ABL && a.dbg != nullptr;
  Ring R;                                                      // the wave's window (wave-uniform); nothing is inside yet
  R.nx = R.ny = R.nz = 3; R.sy = 3; R.sz = 9; R.tot = SPARSE ? 27 : 0; R.ox = R.oy = R.oz = -(1 << 28); R.bx = R.by = R.bz = 0;
  bool fitted = false;
  // the cell the ray stands on (see k_backtrace_flat) + its storage coordinates in the ring
  int base = 0, ix = 0, iy = 0, iz = 0;
  int sx = 0, sy = 0, sz = 0;
  float wx = 0.f, wy = 0.f, wz = 0.f;
  bool interior = false, regular = false;
  int lidx = -1;
  f4 q0 = f4{0.f, 0.f, 0.f, 0.f}, q1 = q0;             // the taps of cell `base` when it is strictly interior (gathered one step ahead)
  f2 p00 = f2{0.f, 0.f}, p10 = p00, p01 = p00, p11 = p00;
  bool miss = false;                                           // the lane wants the window to come to it
  bool have_taps = false;
  const TapRows Rw = tap_rows<PAIR>(V);
  auto step_locate = [&](int& nbase, bool& nregular) {
    s.x = fmaf(-a.ds, s.vx, s.x); s.y = fmaf(-a.ds, s.vy, s.y); s.z = fmaf(-a.ds, s.vz, s.z);
    const float fx = s.x * V.inv_h, fy = s.y * V.inv_h, fz = s.z * V.inv_h;
    ix = cvt_floor_i32(fx); iy = cvt_floor_i32(fy); iz = cvt_floor_i32(fz);
    const bool held = interior & have_taps;                  // the lane holds the taps of the (interior) cell `base`
    interior = (((unsigned)ix - 1u) < V.lx) & (((unsigned)iy - 1u) < V.ly) & (((unsigned)iz - 1u) < V.lz);
    if (interior) {
      wx = __builtin_amdgcn_fractf(fx); wy = __builtin_amdgcn_fractf(fy); wz = __builtin_amdgcn_fractf(fz);
      nbase = mad24(iz, V.sz, mad24(iy, V.sy, ix));
      nregular = true;
      if (!(held & (nbase == base))) {
        __builtin_assume(nbase >= 0 && nbase < (1 << 29));
        gather_rows<PAIR>(Rw, tap_offset<PAIR>(nbase), q0, q1);
      }
      have_taps = true;
    } else {
      const Cell cb = locate(V, s.x, s.y, s.z);
      wx = cb.wx; wy = cb.wy; wz = cb.wz; ix = cb.ix; iy = cb.iy; iz = cb.iz; nbase = cb.base;
      nregular = (cb.ox == 1) & (cb.oy == V.sy) & (cb.oz == V.sz);
    }
  };
  // Step hint (a.fsteps: the iteration count of the forward march that produced each exit ray): the rays of a wave
  // start the adjoint on the FORWARD march's clock -- a ray that left the volume d iterations before the last one of its
  // wave waits d iterations -- so that at every iteration the lanes stand where they stood together in the forward
  // march: the compact bundle the locality sort formed.  Without it every ray starts at its own exit sample and rays
  // that left through an oblique face (or at different times behind a lens) run through the volume spread ALONG their
  // common path (measured on the reference's six rotated views: bounding boxes of 17-21 cells instead of 8-10).  Only
  // the iteration at which a lane does its k-th step changes, not what it computes.  (k_backtrace_flat ignores the hint:
  // the bundles it is chosen for -- compact, leaving through one face together -- have nothing to gain, and it has no
  // register to spare.)
  unsigned steps = 0;                                        // contributing steps of the lane; while `pending`: its delay
  bool pending = false;
  int it_end = a.max_steps;
  if (a.fsteps != nullptr) {
    const unsigned K = s.active ? a.fsteps[i] : 0u;
    const unsigned kmax = wave_max_u32(K);
    const unsigned delay = s.active ? min(kmax - K, 96u) : 0u;
    it_end = a.max_steps + __builtin_amdgcn_readfirstlane((int)wave_max_u32(delay));
    pending = s.active & (delay > 0u);
    if (pending) { s.active = false; steps = delay; }
  }
  if (s.active) {
    int nbase; bool nregular;
    step_locate(nbase, nregular);
    base = nbase; regular = nregular;
    miss = regular;
  }
  bool dirty = false;
  unsigned n_flush = 0, n_slide = 0, n_fit = 0;
  // Dense or sparse bundle?  Wave by wave, re-decided every 32 iterations from how many lanes share their cell with their
  // pair partner.  Dense: faces carried over across crossings, pair / quad pre-reduction (many lanes add to the same slots).
  // Sparse: all eight corners handed over on every leave -- 8 instead of ~4 LDS adds per leave, but none of the per-axis
  // blocks, which for rays oblique to the grid all run on every step (measured on the six rotated views: 535 -> 382 VALU
  // instructions per wave-step, 10.5 -> 8.9 ms; on the metric's dense bundles the same choice costs 6.4 -> 8.8 ms).
  const bool pre = true;
  bool sparse = SPARSE || DRRT_RING_SIMPLE != 0;
  // fixed-point window (SPARSE): scale of the accumulators and of the window, wave-uniform; see above
  float qs = 1.0f, qinv = 1.0f;
  bool qset = false;                                         // the scale has been chosen
  bool qask = false;                                         // a hand-over left the range: re-scale at the next chance
  unsigned qbudget = 0u;                                     // lane-emits since the window was last flushed completely
  float pm_run = 0.f;                                        // per lane: its largest hand-over (scaled) since the scale was set
  bool qbig = false;                                         // per lane: its last hand-over was of the large class
  // choose the exponent so that `wm` (the wave's largest accumulator, in the CURRENT scale) maps into [2^19, 2^20)
  auto q_rescale = [&](float wm) {
    const int ex = ((__float_as_int(wm) >> 23) & 0xff) - 127;              // floor(log2(wm))
    int de = DRRT_RING_QBITS - ex;
    const int e_now = ((__float_as_int(qs) >> 23) & 0xff) - 127;
    de = max(-100 - e_now, min(100 - e_now, de));
    const float f = __int_as_float((uni(de) + 127) << 23);                // 2^de, exact
    p00 = f2{p00.x * f, p00.y * f}; p10 = f2{p10.x * f, p10.y * f}; p01 = f2{p01.x * f, p01.y * f}; p11 = f2{p11.x * f, p11.y * f};
    qs = __int_as_float((uni(e_now + de) + 127) << 23); qinv = __int_as_float((uni(-(e_now + de)) + 127) << 23);
    pm_run = 0.f; qset = true; qask = false;
    QEVENT(3, 1)
  };
  // the wave's largest |accumulator| / hand-over: the order of non-negative floats is the order of their bits
  // (a lane whose values are not finite does not take part: it goes to the grid on its own and must not keep its wave unscaled)
  auto q_wave_max = [&](float v) -> float {
    const float av = fabsf(v);
    return __int_as_float(wave_max_dpp(__float_as_int(av < 3.0e38f ? av : 0.f)));
  };
  auto q_lane_max = [&]() -> float {
    return fmaxf(fmaxf(fmaxf(fabsf(p00.x), fabsf(p00.y)), fmaxf(fabsf(p10.x), fabsf(p10.y))),
                 fmaxf(fmaxf(fabsf(p01.x), fabsf(p01.y)), fmaxf(fabsf(p11.x), fabsf(p11.y))));
  };
  // a chance to (re-)scale: the window is EMPTY.  Moves the scale only when the wave's largest value of the period since the
  // last chance has left [2^17, 2^21); the period's maximum starts again either way (magnitudes may also shrink).
  auto q_adapt = [&]() {
    const float wm = q_wave_max(fmaxf(pm_run, q_lane_max()));
    if ((wm > 0.f) & (wm < 3.0e38f) & (!qset | (wm >= kQSmall) | (wm < kQLow) | qask)) q_rescale(wm);
    pm_run = 0.f; qask = false; qbudget = 0u;
  };
  unsigned ev_face = 0, ev_add = 0, ev_glob = 0, ev_all8 = 0, ev_wsteps = 0, ev_multi = 0;
  unsigned ev_nofit = 0, ev_service = 0, ev_left = 0, ev_vol = 0, ev_all8g = 0, ev_nopre = 0;   // debug: see the end of the kernel

  // all 8 accumulated corners of the regular cell (window slot li with storage coordinates (csx, csy, csz), or straight to the grid)
  auto emit8 = [&](int li, int cbase, int csx, int csy, int csz) -> bool {
    // fixed point: the guard decides BEFORE the branch (folded into li: one level of divergent branching less than a test
    // inside the window path; six rotated views 7.28 / 7.06 -> 7.06 / 7.12 ms on a noisy box, weak medium 4.99 -> 4.93).
    // A hand-over out of range goes to the grid and asks for a re-scale; one that is not finite goes to the grid as it is
    // and asks for nothing.
    float pm = 0.f;
    if constexpr (SPARSE) {
      pm = max3abs(max3abs(p00.x, p00.y, p10.x), max3abs(p10.y, p01.x, p01.y), max3abs(p11.x, p11.y, 0.f));
      const bool okq = qset & (pm < kQGuard);
      if ((li >= 0) & !okq) { QEVENT(4, 1) }
      qask = qask | ((li >= 0) & !okq & (pm < 3.0e38f));
      li = okq ? li : -1;
    }
    if (li >= 0) {
      if (experiment != 3) {
        // the wrap strides -(ny - 1) * sy, -(nz - 1) * sz.  Sparse-only instantiation: two subtractions.  General one (it has
        // no register for R.tot): as SCALARS -- left to itself the compiler selects the factor per lane and multiplies with
        // the quarter-rate v_mul_lo_u32
        const int wY = SPARSE ? R.sy - R.sz : uni(-(R.ny - 1) * R.sy);
        const int wZ = SPARSE ? R.sz - R.tot : uni(-(R.nz - 1) * R.sz);
        const int dX = csx == R.nx - 1 ? -(R.nx - 1) : 1;
        const int dY = csy == R.ny - 1 ? wY : R.sy;
        const int dZ = csz == R.nz - 1 ? wZ : R.sz;
        WT* q = win + li;
        if constexpr (SPARSE) {
          atomicAdd(q, cvt_rpi_i32(p00.x));             atomicAdd(q + dX, cvt_rpi_i32(p00.y));
          atomicAdd(q + dY, cvt_rpi_i32(p10.x));        atomicAdd(q + dY + dX, cvt_rpi_i32(p10.y));
          atomicAdd(q + dZ, cvt_rpi_i32(p01.x));        atomicAdd(q + dZ + dX, cvt_rpi_i32(p01.y));
          atomicAdd(q + dZ + dY, cvt_rpi_i32(p11.x));   atomicAdd(q + dZ + dY + dX, cvt_rpi_i32(p11.y));
          pm_run = max_raw(pm_run, pm);
          qbig = pm >= kQSmall;
        } else {
          RING_ADD(q, p00.x);             RING_ADD(q + dX, p00.y);
          RING_ADD(q + dY, p10.x);        RING_ADD(q + dY + dX, p10.y);
          RING_ADD(q + dZ, p01.x);        RING_ADD(q + dZ + dX, p01.y);
          RING_ADD(q + dZ + dY, p11.x);   RING_ADD(q + dZ + dY + dX, p11.y);
        }
      }
      return true;
    }
    if (experiment != 2) {
      float* g = a.grad + cbase;
      const float u = SPARSE ? qinv : 1.0f;              // the accumulators of the sparse-only instantiation carry 2^e
      RING_GADD(g, p00.x * u);                RING_GADD(g + 1, p00.y * u);
      RING_GADD(g + V.sy, p10.x * u);         RING_GADD(g + V.sy + 1, p10.y * u);
      RING_GADD(g + V.sz, p01.x * u);         RING_GADD(g + V.sz + 1, p01.y * u);
      RING_GADD(g + V.sz + V.sy, p11.x * u);  RING_GADD(g + V.sz + V.sy + 1, p11.y * u);
    }
    return false;
  };
  // (Re-)place every lane in the window; -> lanes next to the window that it still does not hold.  Nobody keeps asking:
  // a lane the service could not bring in is placed again by the next service some OTHER lane asks for (a lane asks
  // when it steps out of the window, not while it stays outside), so a bundle that does not fit costs its outliers'
  // global atomics, not a futile service per step.
  auto place_all = [&](bool ok) -> bool {
    bool nearby = false;
    lidx = regular ? ring_locate(R, ix, iy, iz, sx, sy, sz, nearby) : -1;
    miss = false;
    return ok & (lidx < 0) & nearby;
  };

  STAMP_DECL;
  for (int it = 0; it < it_end; ++it) {
    if (!__any(s.active | pending)) break;                                    // wave-uniform exit
    if (a.fsteps != nullptr) {                                                // step hint (wave-uniform)
      if (pending & ((unsigned)it >= steps)) {
        pending = false; s.active = true; steps = 0u;
        int nbase; bool nregular;
        step_locate(nbase, nregular);
        base = nbase; regular = nregular;
        bool nearby = false;
        lidx = regular ? ring_locate(R, ix, iy, iz, sx, sy, sz, nearby) : -1;
        miss = regular & (lidx < 0) & (nearby | !fitted);
      }
      if (it >= a.max_steps) {
        if (s.active & (steps >= (unsigned)a.max_steps)) {
          if (!kDirect && regular && experiment != 1) { if (emit8(lidx, base, sx, sy, sz)) dirty = true; }
          s.active = false;
        }
        // `dirty` guards 64-lane cooperative flushes below: it has to be wave-uniform BEFORE the service of this very
        // iteration looks at it (a lone expiring lane must not enter a flush on its own)
        dirty = __ballot(dirty) != 0ull;
      }
    }
    if constexpr (SPARSE) {                                                   // fixed-point window: scale and budget (wave-uniform)
      const bool ask = __ballot(qask) != 0ull;
      if (!qset) {                                                            // the first steps: until something has been accumulated
        const float wm = q_wave_max(q_lane_max());
        if ((wm > 0.f) & (wm < 3.0e38f)) q_rescale(wm);
      } else if ((qbudget >= kQBudget) | ask) {                               // overflow budget used up, or a hand-over left the range                           // overflow budget used up, or a hand-over left the range
        // The budget is a worst case (every counted lane-emit on ONE slot, all of one sign, all at the top of their class).
        // Before flushing, LOOK: ten 16-byte LDS reads per lane give the window's largest |slot|; while the scale still fits
        // the period's hand-overs, the budget restarts from what that slot has really used and nothing is flushed.
        const float wm = q_wave_max(fmaxf(pm_run, q_lane_max()));
        bool keep = !ask & (wm >= kQLow) & (wm < kQSmall);
        if (keep) {
          wave_lds_fence();
          const int4* w4 = reinterpret_cast<const int4*>(win);
          int m = 0;
#pragma unroll 2
          for (int k = lane; k < kRingCap / 4; k += kWave) {
            const int4 t = w4[k];
            m = max(max(m, abs(t.x)), max(max(abs(t.y), abs(t.z)), abs(t.w)));
          }
          const unsigned used = ((unsigned)wave_max_dpp(m) >> (DRRT_RING_QBITS + DRRT_RING_QSMALL)) + 1u;
          keep = used < kQBudget / 2u;
          if (keep) { qbudget = used; pm_run = 0.f; QEVENT(2, 1) }
        }
        if (!keep) {
          if (dirty) { ring_flush<2>(win, R, R.oz, R.nz, a.grad, V, lane, experiment == 2, qinv); dirty = false; ++n_flush; }
          if (ask) { QEVENT(1, 1) } else { QEVENT(0, 1) }
          qask = ask;
          q_adapt();
        }
      }
    }
    // ---- lanes ahead of (or beside) the window: let it follow them (wave-uniform branch) ----
    const unsigned long long mm = __ballot(s.active & miss);
    STAMP(0)                                                                  // top of the iteration, step hint
    if (mm != 0ull) {
      const bool ok = s.active & regular;
      const int big = 1 << 28;
      bool settled = false;
      if (ABL && dbg) ++ev_service;
      if (fitted) {
        // Slide along every axis on which lanes are ahead of the window and none is at its rear (or the other way round):
        // the layers every lane has left are flushed and their storage is reused ahead.  Only the lanes in or next to the
        // window count (`nearby`): a lane far from the bundle neither asks for a slide nor holds one back.
        bool nearby = false;
        { int tx, ty, tz; (void)ring_locate(R, ix, iy, iz, tx, ty, tz, nearby); }
        const bool cnt = ok & nearby;
        {
          const int rel = ix - R.ox;
          const bool hi = __ballot(cnt & (rel > R.nx - 2)) != 0ull, lo = __ballot(cnt & (rel < 0)) != 0ull;
          if (hi != lo) {
            int k;
            if (hi) { k = min(wave_min_dpp(cnt ? rel : big), V.W - R.nx - R.ox); }
            else    { k = min((R.nx - 2) - wave_max_dpp(cnt ? rel : -big), R.ox); }
            k = uni(min(k, R.nx));
            if (k > 0) {
              if (hi) { ring_flush<0>(win, R, R.ox, k, a.grad, V, lane, experiment == 2, qinv); R.ox += k; R.bx += k; R.bx = R.bx >= R.nx ? R.bx - R.nx : R.bx; }
              else    { ring_flush<0>(win, R, R.ox + R.nx - k, k, a.grad, V, lane, experiment == 2, qinv); R.ox -= k; R.bx -= k; R.bx = R.bx < 0 ? R.bx + R.nx : R.bx; }
              ++n_slide;
            }
          }
        }
        {
          const int rel = iy - R.oy;
          const bool hi = __ballot(cnt & (rel > R.ny - 2)) != 0ull, lo = __ballot(cnt & (rel < 0)) != 0ull;
          if (hi != lo) {
            int k;
            if (hi) { k = min(wave_min_dpp(cnt ? rel : big), V.H - R.ny - R.oy); }
            else    { k = min((R.ny - 2) - wave_max_dpp(cnt ? rel : -big), R.oy); }
            k = uni(min(k, R.ny));
            if (k > 0) {
              if (hi) { ring_flush<1>(win, R, R.oy, k, a.grad, V, lane, experiment == 2, qinv); R.oy += k; R.by += k; R.by = R.by >= R.ny ? R.by - R.ny : R.by; }
              else    { ring_flush<1>(win, R, R.oy + R.ny - k, k, a.grad, V, lane, experiment == 2, qinv); R.oy -= k; R.by -= k; R.by = R.by < 0 ? R.by + R.ny : R.by; }
              ++n_slide;
            }
          }
        }
        {
          const int rel = iz - R.oz;
          const bool hi = __ballot(cnt & (rel > R.nz - 2)) != 0ull, lo = __ballot(cnt & (rel < 0)) != 0ull;
          if (hi != lo) {
            int k;
            if (hi) { k = min(wave_min_dpp(cnt ? rel : big), V.D - R.nz - R.oz); }
            else    { k = min((R.nz - 2) - wave_max_dpp(cnt ? rel : -big), R.oz); }
            k = uni(min(k, R.nz));
            if (k > 0) {
              if (hi) { ring_flush<2>(win, R, R.oz, k, a.grad, V, lane, experiment == 2, qinv); R.oz += k; R.bz += k; R.bz = R.bz >= R.nz ? R.bz - R.nz : R.bz; }
              else    { ring_flush<2>(win, R, R.oz + R.nz - k, k, a.grad, V, lane, experiment == 2, qinv); R.oz -= k; R.bz -= k; R.bz = R.bz < 0 ? R.bz + R.nz : R.bz; }
              ++n_slide;
            }
          }
        }
        settled = __ballot(place_all(ok)) == 0ull;
      }
      if (!settled) {
        // (re-)fit: the bounding box of the cells the rays stand on -- of the lanes next to the old window when there is
        // one, so that a bundle torn apart keeps a window for its larger part --, room ahead of them, within the capacity
        bool nearby = true;
        if (fitted) { int tx, ty, tz; (void)ring_locate(R, ix, iy, iz, tx, ty, tz, nearby); }
        bool cnt = ok & nearby;
        if (__ballot(cnt) == 0ull) cnt = ok;
        int x0 = wave_min_dpp(cnt ? ix : big), x1 = wave_max_dpp(cnt ? ix : -big);
        int y0 = wave_min_dpp(cnt ? iy : big), y1 = wave_max_dpp(cnt ? iy : -big);
        int z0 = wave_min_dpp(cnt ? iz : big), z1 = wave_max_dpp(cnt ? iz : -big);
        if (x1 >= x0) {
          // direction of travel (backwards along the ray) of the median lane
          const unsigned long long cm = __ballot(cnt);
          const int first = __ffsll((long long)cm) - 1, last = 63 - __clzll((long long)cm);
          int ref = (first + last) >> 1;
          if (!((cm >> ref) & 1ull)) ref = first;
          const float dx_ = -__shfl(s.vx, ref, kWave), dy_ = -__shfl(s.vy, ref, kWave), dz_ = -__shfl(s.vz, ref, kWave);
          const float inv_dm = __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(dx_), fabsf(dy_)), fmaxf(fabsf(dz_), 1e-30f)));
          int ex = x1 - x0 + 2, ey = y1 - y0 + 2, ez = z1 - z0 + 2;             // slots the cells need per axis
          const bool too_big = uni((int)(ex * ey * ez > kRingCap)) != 0;
          if (too_big && fitted) {
            // the lanes next to the window do not fit any window together: it stays where it is, they go to the grid
            if (ABL && dbg) ++ev_nofit;
          } else {
          if (too_big) {
            // no window holds them all: a cube of the capacity around the median lane's cell; the others go to the grid
            if (ABL && dbg) ++ev_nofit;
            const int rx = __shfl(ix, ref, kWave), ry = __shfl(iy, ref, kWave), rz = __shfl(iz, ref, kWave);
            const int half = 4;
            x0 = max(x0, rx - half); x1 = min(x1, rx + half); y0 = max(y0, ry - half); y1 = min(y1, ry + half);
            z0 = max(z0, rz - half); z1 = min(z1, rz + half);
            ex = x1 - x0 + 2; ey = y1 - y0 + 2; ez = z1 - z0 + 2;
          }
          if (dirty) { ring_flush<2>(win, R, R.oz, R.nz, a.grad, V, lane, experiment == 2, qinv); dirty = false; ++n_flush; }
          if constexpr (SPARSE) { if (qset) q_adapt(); }     // the window is empty: a chance to re-scale for free
          int nx = ex + DRRT_RING_SLACK_MIN + (int)((float)DRRT_RING_SLACK * fabsf(dx_) * inv_dm + 0.5f);
          int ny = ey + DRRT_RING_SLACK_MIN + (int)((float)DRRT_RING_SLACK * fabsf(dy_) * inv_dm + 0.5f);
          int nz = ez + DRRT_RING_SLACK_MIN + (int)((float)DRRT_RING_SLACK * fabsf(dz_) * inv_dm + 0.5f);
#if DRRT_RING_GROW > 0
          // spare capacity: room on every axis, so that a bundle that widens (the adjoint leaves a focus) is not re-fitted at once
          int grown = 0;
#pragma unroll 1
          for (; grown < DRRT_RING_GROW && (nx + 1) * (ny + 1) * (nz + 1) <= kRingCap; ++grown) { ++nx; ++ny; ++nz; }
          const int gh = grown >> 1;                       // half of it behind the rays
#else
          const int gh = 0;
#endif
          nx = uni(min(nx, V.W)); ny = uni(min(ny, V.H)); nz = uni(min(nz, V.D));
          ex = min(ex, nx); ey = min(ey, ny); ez = min(ez, nz);
#pragma unroll 1
          for (int guard = 0; guard < 64 && nx * ny * nz > kRingCap; ++guard) {
            if (nz - ez >= ny - ey && nz - ez >= nx - ex && nz > ez) --nz;        // give back the largest margin first
            else if (ny - ey >= nx - ex && ny > ey) --ny;
            else if (nx > ex) --nx;
            else break;
          }
          R.nx = nx; R.ny = ny; R.nz = nz; R.sy = nx; R.sz = nx * ny; R.tot = SPARSE ? nx * ny * nz : 0;
          // the spare slots lie ahead of the rays
          int ox = dx_ < 0.f ? x0 - (nx - ex) + gh : x0 - gh, oy = dy_ < 0.f ? y0 - (ny - ey) + gh : y0 - gh,
              oz = dz_ < 0.f ? z0 - (nz - ez) + gh : z0 - gh;
          ox = max(0, min(ox, V.W - nx)); oy = max(0, min(oy, V.H - ny)); oz = max(0, min(oz, V.D - nz));
          R.ox = uni(ox); R.oy = uni(oy); R.oz = uni(oz);
          R.bx = R.by = R.bz = 0;
          fitted = true; ++n_fit;
          if (ABL && dbg) ev_vol += (unsigned)(nx * ny * nz);
          const bool left = place_all(ok);
          if (ABL && dbg) ev_left += (unsigned)__popcll(__ballot(left));
          }
        }
      }
    }
    STAMP(1)                                                                  // window service
    if (!SPARSE && DRRT_RING_SIMPLE == 0 && (it & 15) == 15) {   // a sample of one iteration in 16 (scalar arithmetic only)
      const bool on = s.active & regular;
      const int pb = __builtin_amdgcn_update_dpp(-1, base, 0xB1, 0xF, 0xF, false);   // the pair partner's cell (quad_perm [1,0,3,2])
      const int lanes = __popcll(__ballot(on)), hits = __popcll(__ballot(on & (pb == base)));
      sparse = uni(hits * 100) < uni(lanes * DRRT_RING_DENSE_PCT);
    }
    bool used_lds = false;
    bool matched = false;                                    // (pair-partner match of this step's crossings: debug counters only)
    if (s.active) {
      if (!interior) taps_set<PAIR>(fetch(V.data, locate(V, s.x, s.y, s.z)), q0, q1);   // boundary cell (clamped neighbours): fetched here, not ahead
      Cell c;
      c.base = 0; c.ix = c.iy = c.iz = 0; c.ox = c.oy = c.oz = 0;
      c.wx = wx; c.wy = wy; c.wz = wz; c.interior = interior;
      if (MODE == 1) {
        if (interior) { c.base = base; c.ox = 1; c.oy = V.sy; c.oz = V.sz; }
        else c = locate(V, s.x, s.y, s.z);
      }
      const float px = s.x, py = s.y, pz = s.z;
      AdjSample m;
      if (!adj_sample<MODE>(V, a.sdf, a.ds, s, c, taps_of<PAIR>(q0, q1), m)) {
        // the ray has ended (:426-428): hand over what its cell has accumulated
        if (!kDirect && regular && experiment != 1) used_lds = emit8(lidx, base, sx, sy, sz);
      } else {
        ++steps;
        const float dn = dot3(s.mx, s.my, s.mz, m.gx, m.gy, m.gz);                            // :430
        const float nds = (m.n * a.ds) * a.grad_scale;
        if (regular) {
          const float u = SPARSE ? qs : 1.0f;              // (a power of two: an exact factor of every weight)
          const float ndq = nds * u;
          const CornerPairs cp = splat_weights_pk(wx, wy, wz, (dn * a.ds) * u, ndq * s.mx, ndq * s.my, ndq * s.mz);   // :431-432
          if (kDirect) { p00 = cp.c00; p10 = cp.c10; p01 = cp.c01; p11 = cp.c11; used_lds = emit8(lidx, base, sx, sy, sz); }
          else { p00 += cp.c00; p10 += cp.c10; p01 += cp.c01; p11 += cp.c11; }
        } else if (experiment != 2 && experiment != 1) {
          const Cell cb = locate(V, px, py, pz);
          const Corners w = splat_weights(cb.wx, cb.wy, cb.wz, dn * a.ds, nds * s.mx, nds * s.my, nds * s.mz);
          float* g = a.grad + cb.base;
          atomic_add_f32(g, w.c000);                     atomic_add_f32(g + cb.ox, w.c100);
          atomic_add_f32(g + cb.oy, w.c010);             atomic_add_f32(g + cb.oy + cb.ox, w.c110);
          atomic_add_f32(g + cb.oz, w.c001);             atomic_add_f32(g + cb.oz + cb.ox, w.c101);
          atomic_add_f32(g + cb.oz + cb.oy, w.c011);     atomic_add_f32(g + cb.oz + cb.oy + cb.ox, w.c111);
        }
        STAMP(2)                                             // sample (waits for the taps), weights, accumulate
        const int old_base = base, old_lidx = lidx;
        const bool old_regular = regular;
        int nbase; bool nregular;
        int dpack = -1;                                    // the move in cells, one VGPR: (ddx + 1) | (ddy + 1) << 2 | (ddz + 1) << 4, or -1
        if (SPARSE || sparse) {                            // (wave-uniform) a sparse bundle does not look at the move
          step_locate(nbase, nregular);
        } else {
          const int oix = ix, oiy = iy, oiz = iz;
          step_locate(nbase, nregular);
          const int ddx = ix - oix, ddy = iy - oiy, ddz = iz - oiz;
          const bool unit = ((unsigned)(ddx + 1) <= 2u) & ((unsigned)(ddy + 1) <= 2u) & ((unsigned)(ddz + 1) <= 2u);
          dpack = unit ? ((ddx + 1) | ((ddy + 1) << 2) | ((ddz + 1) << 4)) : -1;
        }
        // lambda / mu (:434-435)
        const float hxy = m.hxy * V.inv_h2, hxz = m.hxz * V.inv_h2, hyz = m.hyz * V.inv_h2;
        const float hmx = fmaf(hxz, s.mz, hxy * s.my);
        const float hmy = fmaf(hyz, s.mz, hxy * s.mx);
        const float hmz = fmaf(hyz, s.my, hxz * s.mx);
        s.lx = fmaf(a.ds, fmaf(dn, m.gx, m.n * hmx), s.lx);
        s.ly = fmaf(a.ds, fmaf(dn, m.gy, m.n * hmy), s.ly);
        s.lz = fmaf(a.ds, fmaf(dn, m.gz, m.n * hmz), s.lz);
        s.mx = fmaf(a.ds, s.lx, s.mx); s.my = fmaf(a.ds, s.ly, s.my); s.mz = fmaf(a.ds, s.lz, s.mz);
        STAMP(3)                                             // step, locate, gather issue, lambda / mu
        // ---- the ray leaves its cell ----
        if (nbase != old_base || !interior) {
          base = nbase; regular = nregular;
          if (nbase != old_base || regular != old_regular) {
            bool relocate = true;                        // the new cell still has to be placed in the window
            if (old_regular) {
              const bool unit = dpack >= 0;
              const int ddx = (dpack & 3) - 1, ddy = ((dpack >> 2) & 3) - 1, ddz = ((dpack >> 4) & 3) - 1;
              if (!SPARSE && !sparse && (regular & unit & (old_lidx >= 0) & (experiment != 1) & (experiment != 4))) {
                // one, two or three faces crossed: one crossing after the other (x, y, z), each emits the face left behind
                // and carries the shared one; the later ones hand over zeros where the earlier ones cleared
                if (ABL && dbg) {
                  const int nax = (__ballot(ddx != 0) != 0ull) + (__ballot(ddy != 0) != 0ull) + (__ballot(ddz != 0) != 0ull);
                  if (lane == __ffsll((long long)__ballot(true)) - 1) ev_multi += nax >= 2;
                  if (sparse) ++ev_nopre;
                }
                int cur = old_lidx;
                bool out = false;
                if (ddx != 0) {
                  const bool fwd = ddx > 0;
                  const float e0 = fwd ? p00.x : p00.y, e1 = fwd ? p10.x : p10.y, e2 = fwd ? p01.x : p01.y, e3 = fwd ? p11.x : p11.y;
                  p00 = fwd ? f2{p00.y, 0.f} : f2{0.f, p00.x}; p10 = fwd ? f2{p10.y, 0.f} : f2{0.f, p10.x};
                  p01 = fwd ? f2{p01.y, 0.f} : f2{0.f, p01.x}; p11 = fwd ? f2{p11.y, 0.f} : f2{0.f, p11.x};
                  out |= ring_cross<ABL, WT>(win, experiment, pre, 0, fwd, cur, sx, sy, sz, R.nx, R.ny, R.nz, 1, R.sy, R.sz, ix, R.ox,
                                         e0, e1, e2, e3, matched, ev_face, ev_add, dbg);
                }
                if (ddy != 0) {
                  const bool fwd = ddy > 0;
                  const f2 ea = fwd ? p00 : p10, eb = fwd ? p01 : p11;
                  const f2 ka = fwd ? p10 : p00, kb = fwd ? p11 : p01;
                  p00 = fwd ? ka : f2{0.f, 0.f}; p01 = fwd ? kb : f2{0.f, 0.f};
                  p10 = fwd ? f2{0.f, 0.f} : ka; p11 = fwd ? f2{0.f, 0.f} : kb;
                  out |= ring_cross<ABL, WT>(win, experiment, pre, 1, fwd, cur, sy, sx, sz, R.ny, R.nx, R.nz, R.sy, 1, R.sz, iy, R.oy,
                                         ea.x, ea.y, eb.x, eb.y, matched, ev_face, ev_add, dbg);
                }
                if (ddz != 0) {
                  const bool fwd = ddz > 0;
                  const f2 ea = fwd ? p00 : p01, eb = fwd ? p10 : p11;
                  const f2 ka = fwd ? p01 : p00, kb = fwd ? p11 : p10;
                  p00 = fwd ? ka : f2{0.f, 0.f}; p10 = fwd ? kb : f2{0.f, 0.f};
                  p01 = fwd ? f2{0.f, 0.f} : ka; p11 = fwd ? f2{0.f, 0.f} : kb;
                  out |= ring_cross<ABL, WT>(win, experiment, pre, 2, fwd, cur, sz, sx, sy, R.nz, R.nx, R.ny, R.sz, 1, R.sy, iz, R.oz,
                                         ea.x, ea.y, eb.x, eb.y, matched, ev_face, ev_add, dbg);
                }
                used_lds = true;
                lidx = out ? -1 : cur;                   // stepped ahead of the window by one cell: ask it to follow
                miss = out;
                relocate = false;
              } else {
                // out of a cell the window does not hold, into or out of a clamped cell, a jump over more than one cell: all eight
                if (ABL && dbg) { ++ev_all8; ev_all8g += old_lidx < 0; }
                if (!kDirect) {
                  if (experiment != 1) used_lds = emit8(old_lidx, old_base, sx, sy, sz);
                  p00 = p10 = p01 = p11 = f2{0.f, 0.f};
                }
              }
            }
            if (relocate) {
              bool nearby = false;
              lidx = regular ? ring_locate(R, ix, iy, iz, sx, sy, sz, nearby) : -1;
              miss = regular & (lidx < 0) & ((nearby & (old_lidx >= 0)) | !fitted);   // stepped out of the window: ask it to follow
            }
          }
        }
      }
    }
    STAMP(4)                                                 // leave: hand-over, new slot
    {
      const unsigned long long ul = __ballot(used_lds);
      dirty = dirty | (ul != 0ull);
      if constexpr (SPARSE) {
        QEVENT(5, qbig ? 1 : 0)
        qbudget += (unsigned)__popcll(ul) + 3u * (unsigned)__popcll(__ballot(used_lds & qbig));
        qbig = false;
      }
    }
    if (ABL && dbg) ev_wsteps += lane == 0;
  }
  STAMP(5)
  STAMP_END
  // rays still marching when max_steps ran out keep what their cell has accumulated: hand it over
  if (!kDirect && s.active && regular && experiment != 1) { if (emit8(lidx, base, sx, sy, sz)) dirty = true; }
  dirty = __ballot(dirty) != 0ull;
  if (dirty) { ring_flush<2>(win, R, R.oz, R.nz, a.grad, V, lane, experiment == 2, qinv); ++n_flush; }
  if (ABL && dbg) {
    if (lane == 0) { atomicAdd(&a.dbg[0], (unsigned long long)n_flush); atomicAdd(&a.dbg[1], (unsigned long long)n_slide);
                     atomicAdd(&a.dbg[2], (unsigned long long)n_fit); atomicAdd(&a.dbg[3], 1ull); }
    if (ev_face) atomicAdd(&a.dbg[4], (unsigned long long)ev_face);
    if (ev_add) atomicAdd(&a.dbg[5], (unsigned long long)ev_add);
    if (ev_glob) atomicAdd(&a.dbg[6], (unsigned long long)ev_glob);
    if (ev_all8) atomicAdd(&a.dbg[7], (unsigned long long)ev_all8);
    if (ev_wsteps) atomicAdd(&a.dbg[8], (unsigned long long)ev_wsteps);
    if (ev_multi) atomicAdd(&a.dbg[9], (unsigned long long)ev_multi);
    // ring: [10] fits around the median lane only (no window holds the bounding box), [11] service calls (per wave), [12] lanes
    // still asking after a service, [13] sum of the fitted window volumes, [14] all-eight hand-overs that went to the grid,
    // [15] crossings made without the pre-reduction
    if (lane == 0) { atomicAdd(&a.dbg[10], (unsigned long long)ev_nofit); atomicAdd(&a.dbg[11], (unsigned long long)ev_service);
                     atomicAdd(&a.dbg[12], (unsigned long long)ev_left); atomicAdd(&a.dbg[13], (unsigned long long)ev_vol); }
    if (ev_all8g) atomicAdd(&a.dbg[14], (unsigned long long)ev_all8g);
    if (ev_nopre) atomicAdd(&a.dbg[15], (unsigned long long)ev_nopre);
  }
  block_stats<kAdjBlock>(a.stats, steps, 0u);
}

// ---- launcher -----------------------------------------------------------------------------------
// backtrace only (MODE 0).  which: 0 = the accumulating instantiation, 1 = the direct one, 2 = both (classified call: one returns)
void launch_backtrace_ring_sparse(const BackArgs& a, hipStream_t s, int which) {
  const dim3 g(adj_grid_for(a.n)), b(kAdjBlock);
  const bool pair = a.vol.pair != nullptr;
  if (which != 1) {
    if (pair) hipLaunchKernelGGL((k_backtrace_ring<false, true, 0, true, false>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_ring<false, false, 0, true, false>), g, b, 0, s, a);
  }
  if (which != 0) {
    if (pair) hipLaunchKernelGGL((k_backtrace_ring<false, true, 0, true, true>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_ring<false, false, 0, true, true>), g, b, 0, s, a);
  }
}
void launch_backtrace_ring(int mode, bool abl, const BackArgs& a, hipStream_t s) {
  const dim3 g(adj_grid_for(a.n)), b(kAdjBlock);
  const bool pair = a.vol.pair != nullptr;
  if (mode == 1) {
    if (pair) hipLaunchKernelGGL((k_backtrace_ring<false, true, 1>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_ring<false, false, 1>), g, b, 0, s, a);
  } else if (abl) {
    if (pair) hipLaunchKernelGGL((k_backtrace_ring<true, true, 0>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_ring<true, false, 0>), g, b, 0, s, a);
  } else {
    if (pair) hipLaunchKernelGGL((k_backtrace_ring<false, true, 0>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_ring<false, false, 0>), g, b, 0, s, a);
  }
}

}  // namespace drrt

#if defined(DRRT_RING_STAMPS)
extern "C" __attribute__((visibility("default"))) int drrt_debug_ring_events(unsigned long long* out8, int reset) {
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(drrt::g_ring_events), sizeof(drrt::g_ring_events)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(drrt::g_ring_events), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" __attribute__((visibility("default"))) int drrt_debug_ring_stamps(unsigned long long* out8, int reset) {
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(drrt::g_ring_stamps), sizeof(drrt::g_ring_stamps)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(drrt::g_ring_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
