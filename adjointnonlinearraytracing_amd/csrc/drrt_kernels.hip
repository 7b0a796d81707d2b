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

template <int MODE, int REUSE = kTapReuse>
__global__ void __launch_bounds__(kBlock) k_trace(TraceArgs a) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned steps = 0, failed = 0;
  size_t i;
  if (ray_index(a.perm, t, a.n, i)) {
    Ray3 p = ld3(a.pos, i, a.io_half, &a.vol, RAY_POS), u = ld3(a.vel, i, a.io_half, &a.vol, RAY_VEL);
    const float pp[3] = {p.x, p.y, p.z}, vv[3] = {u.x, u.y, u.z};
    float po[3] = {0.f, 0.f, 0.f}, pd[3] = {0.f, 0.f, 0.f};
    if (MODE == 1) {
      Ray3 o = ld3(a.pln_o, i), d = ld3(a.pln_d, i);
      po[0] = o.x; po[1] = o.y; po[2] = o.z; pd[0] = d.x; pd[1] = d.y; pd[2] = d.z;
    }
    RayOut r = trace_ray<MODE, REUSE>(a.vol, a.sdf, a.ds, a.max_steps, pp, vv, po, pd);
    steps = r.steps; failed = r.act ? 1u : 0u;
    st3(a.xt, i, r.xt[0], r.xt[1], r.xt[2], a.io_half, &a.vol, RAY_POS);
    st3(a.vt, i, r.vt[0], r.vt[1], r.vt[2], a.io_half, &a.vol, RAY_VEL);
    if (MODE == 1) a.failmask[i] = (r.esc ? 0 : 1) | (r.again ? 2 : 0);     // src/tracer.cpp:171; bit 1: k_trace_again
    if (MODE == 2) a.again[i] = r.again ? 1 : 0;
  }
  block_stats(a.stats, steps, failed);
}

// ---------------------------------------------------------------------------------------------
// k_trace_flat: Tracer::trace (MODE 0) with the loop written out flat -- the same per-step arithmetic as trace_ray /
// fwd_step_c (drrt_device.h), but the cell is located IN PLACE (nothing but position, fractions, flat index and the
// 8 taps is carried from step to step), a strictly interior step touches no box test, no exit record and no clamp
// offsets, and the gather of the next cell is skipped while the ray stays in its cell.  Boundary cells (the
// outermost layer, where the clamps and the inbounds / escaped tests matter) take the generic path of drrt_device.h.
// Bit-identical to k_trace<0> (tests/test_gpu_parity.py).
//
// What bounds it (tools/chain_bench.hip, PMC): a wave issues a gather on nearly every step -- some lane always changes
// cell -- and a divergent 64-lane gather instruction occupies the CU's texture addresser for ~30-37 cycles whether one
// lane is active or all 64.  So the taps a lane keeps across steps save cache traffic but hardly any addresser time;
// what does is FEWER GATHER INSTRUCTIONS: with the pair copy of the grid (DRRT_FLAG_PAIR_GRID, template PAIR) a cell
// is two 16-byte gathers instead of four 8-byte ones (1.31 -> 1.06 ms on 256^3 / 1M rays).  Tried and dropped: two rays
// per lane (two gathers in flight per wave: 1.6-1.9 ms -- the addresser, not latency, is the limit), a branch-free
// interior path (1.38 ms), fewer VALU instructions alone (-20 %: no change).
// ---------------------------------------------------------------------------------------------
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

// One marching ray of k_trace_flat.  No exit-record registers: the march of a ray ends at the step that sets
// `escaped` (:75-76, per-ray form of :82), so the record (:79-80) is the final state when the ray CROSSED out of
// the box at that step, and the initial state (:56-57, re-read from the input) otherwise.
struct FlatRay {
  float x, y, z, vx, vy, vz, wx, wy, wz;
  unsigned off;            // BYTE offset (tap_offset) of the cell whose taps the lane holds, i.e. the load offset itself
  f4 q0, q1;               // the taps (gather_rows)
  unsigned steps;
  bool inside, interior, live, crossed;
};

template <bool PAIR>
__device__ __forceinline__ void flat_begin(const Vol& V, const TapRows& R, FlatRay& r, const Ray3& p, const Ray3& u) {
  r.x = p.x; r.y = p.y; r.z = p.z; r.vx = u.x; r.vy = u.y; r.vz = u.z;
  r.inside = inbounds(V, r.x, r.y, r.z);                                                    // :61
  r.live = true; r.crossed = false; r.steps = 0;                                            // :62
  const Cell c = locate(V, r.x, r.y, r.z);
  r.off = tap_offset<PAIR>(c.base); r.wx = c.wx; r.wy = c.wy; r.wz = c.wz; r.interior = c.interior;
  r.q0 = r.q1 = f4{0.f, 0.f, 0.f, 0.f};
  if (r.interior) gather_rows<PAIR>(R, r.off, r.q0, r.q1);
}

// trace_plane (MODE 1): the sensor plane of the ray; "inside" additionally means "not past the plane" (:144-145)
struct FlatPlane { float ox, oy, oz, dx, dy, dz; };
__device__ __forceinline__ bool flat_past_plane(const FlatPlane& P, const FlatRay& r) {
  return dot3(r.x - P.ox, r.y - P.oy, r.z - P.oz, P.dx, P.dy, P.dz) > 0.f;
}

// The box tests of a ray that has just stepped into a boundary cell (:73-76, :86): may end the ray.
template <int MODE>
__device__ __forceinline__ void flat_boundary(const Vol& V, const FlatPlane& P, int it, FlatRay& r) {
  bool cur_inside = inbounds(V, r.x, r.y, r.z);                                             // :73
  const bool esc_now = escaped(V, r.x, r.y, r.z, r.vx, r.vy, r.vz);                         // :76
  if (MODE == 1) cur_inside = cur_inside & !flat_past_plane(P, r);                          // :144-145
  const bool cross = r.inside & !cur_inside;                                                // :74
  r.inside = cur_inside;                                                                    // :86
  if (cross | esc_now) { r.crossed = cross; r.live = false; r.steps = (unsigned)it + 1u; }  // :75-76
}

// One iteration of one ray (any state), up to the point where the cell of the new position is known.  Returns true
// when the taps of that cell have to be gathered (the caller issues the gather).
// In-place locate(): the floor index by one conversion; the fractions by v_fract, == f - floor(f) bit for bit for the
// non-negative coordinates of an interior cell.  The fractions (and clamp offsets) of a BOUNDARY cell are not
// carried: they are re-derived by locate() when such a cell is sampled.
template <bool PAIR, int MODE>
__device__ __forceinline__ bool flat_advance(const Vol& V, const FlatPlane& P, float ds, int it, FlatRay& r) {
  if (r.inside) {                                                                           // masked gather (Q4)
    if (!r.interior) {                            // boundary cell: clamp offsets and fractions from the position, taps fetched here
      const Cell cb = locate(V, r.x, r.y, r.z);
      taps_set<PAIR>(fetch(V.data, cb), r.q0, r.q1);
      r.wx = cb.wx; r.wy = cb.wy; r.wz = cb.wz;
    }
    const Sample q = interp<false>(taps_of<PAIR>(r.q0, r.q1), r.wx, r.wy, r.wz);
    const float gx = q.gx * V.inv_h, gy = q.gy * V.inv_h, gz = q.gz * V.inv_h;
    const float dsn = ds * q.n;
    r.vx = fmaf(dsn, gx, r.vx); r.vy = fmaf(dsn, gy, r.vy); r.vz = fmaf(dsn, gz, r.vz);     // :70
  }
  r.x = fmaf(ds, r.vx, r.x); r.y = fmaf(ds, r.vy, r.y); r.z = fmaf(ds, r.vz, r.z);          // :71
  const float fx = r.x * V.inv_h, fy = r.y * V.inv_h, fz = r.z * V.inv_h;
  const int ix = cvt_floor_i32(fx), iy = cvt_floor_i32(fy), iz = cvt_floor_i32(fz);
  const bool was_interior = r.interior;
  r.interior = (((unsigned)ix - 1u) < V.lx) & (((unsigned)iy - 1u) < V.ly) & (((unsigned)iz - 1u) < V.lz);
  if (r.interior) {
    // strictly interior: in bounds, not escaped, nothing to record (Cell::interior) -- sample weights and taps only
    r.wx = __builtin_amdgcn_fractf(fx); r.wy = __builtin_amdgcn_fractf(fy); r.wz = __builtin_amdgcn_fractf(fz);
    const unsigned noff = tap_offset<PAIR>(mad24(iz, V.sz, mad24(iy, V.sy, ix)));
    if (MODE == 1) {                                                                        // the plane is the only test left
      const bool past = flat_past_plane(P, r);
      if (r.inside & past) { r.crossed = true; r.live = false; r.steps = (unsigned)it + 1u; }   // :74-75
      r.inside = !past;                                                                     // :86
    } else {
      r.inside = true;                                                                      // :73, :86
    }
    if (!(was_interior & (noff == r.off))) { r.off = noff; return true; }
  } else {
    flat_boundary<MODE>(V, P, it, r);
  }
  return false;
}

// PAIR: gather from the pair copy of the grid (two 16-byte loads per cell, see gather_rows).  MODE 0 = trace, 1 = trace_plane
// (same march; the ray also ends when it passes its sensor plane, and rays that could produce a later exit record in the
// reference's global loop are flagged for k_trace_again exactly as k_trace<1> does, see plane_again).
template <bool PAIR, int MODE>
__global__ void __launch_bounds__(kBlock) k_trace_flat(TraceArgs a) {
  const Vol& V = a.vol;
  const size_t t = (size_t)xcd_block(blockIdx.x, gridDim.x, a.xcd_order ? kXcdRuns16 : kXcdOff) * kBlock + threadIdx.x;
  const TapRows R = tap_rows<PAIR>(V);
  unsigned steps = 0, failed = 0;
  size_t i;
  if (ray_index(a.perm, t, a.n, i)) {
    FlatRay r;
    FlatPlane P{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
      const Ray3 p = ld3(a.pos, i, a.io_half, &a.vol, RAY_POS), u = ld3(a.vel, i, a.io_half, &a.vol, RAY_VEL);
      flat_begin<PAIR>(V, R, r, p, u);
      if (MODE == 1) {
        const Ray3 o = ld3(a.pln_o, i), d = ld3(a.pln_d, i);
        P.ox = o.x; P.oy = o.y; P.oz = o.z; P.dx = d.x; P.dy = d.y; P.dz = d.z;
      }
    }
    for (int it = 0; it < a.max_steps; ++it) {
      if (flat_advance<PAIR, MODE>(V, P, a.ds, it, r)) gather_rows<PAIR>(R, r.off, r.q0, r.q1);
      if (!r.live) break;
    }
    const bool esc = !r.live;
    if (!esc) { r.steps = a.max_steps > 0 ? (unsigned)a.max_steps : 0u; failed = 1u; }
    steps = r.steps;
    float xtx = r.x, xty = r.y, xtz = r.z, vtx = r.vx, vty = r.vy, vtz = r.vz;
    if (!(esc & r.crossed)) {
      // escaped without crossing out of the box (never entered it): the record is the initial state (:56-57);
      // ran out of steps: xt is the final position (:95), vt stays the initial direction (Q6)
      const Ray3 p = ld3(a.pos, i, a.io_half, &a.vol, RAY_POS), u = ld3(a.vel, i, a.io_half, &a.vol, RAY_VEL);
      if (esc) { xtx = p.x; xty = p.y; xtz = p.z; }
      vtx = u.x; vty = u.y; vtz = u.z;
    }
    st3(a.xt, i, xtx, xty, xtz, a.io_half, &a.vol, RAY_POS);
    st3(a.vt, i, vtx, vty, vtz, a.io_half, &a.vol, RAY_VEL);
    if (a.steps_out) a.steps_out[i] = steps;
    if (MODE == 1) {
      bool again = false;
      if (esc) {                                                                           // plane_again() on the final state
        const float d = dot3(r.x - P.ox, r.y - P.oy, r.z - P.oz, P.dx, P.dy, P.dz);
        const float dv = dot3(r.vx, r.vy, r.vz, P.dx, P.dy, P.dz);
        again = !(escaped(V, r.x, r.y, r.z, r.vx, r.vy, r.vz) | ((d > 0.f) & (dv >= 0.f)));
      }
      a.failmask[i] = (esc ? 0 : 1) | (again ? 2 : 0);                                     // src/tracer.cpp:171; bit 1: k_trace_again
    }
  }
  block_stats(a.stats, steps, failed);
}

// trace_plane / trace_sdf, second pass: rays flagged by the first pass (failmask bit 1 / `again` byte) are
// re-marched over the reference's GLOBAL loop count (stats->iters of the first pass), see trace_ray / ray_full.
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_trace_again(TraceArgs a) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.n) return;
  if (MODE == 1 ? !(a.failmask[i] & 2) : !a.again[i]) return;
  Ray3 p = ld3(a.pos, i, a.io_half, &a.vol, RAY_POS), u = ld3(a.vel, i, a.io_half, &a.vol, RAY_VEL);
  const float pp[3] = {p.x, p.y, p.z}, vv[3] = {u.x, u.y, u.z};
  float po[3] = {0.f, 0.f, 0.f}, pd[3] = {0.f, 0.f, 0.f};
  if (MODE == 1) {
    Ray3 o = ld3(a.pln_o, i), d = ld3(a.pln_d, i);
    po[0] = o.x; po[1] = o.y; po[2] = o.z; pd[0] = d.x; pd[1] = d.y; pd[2] = d.z;
  }
  RayOut r = ray_full<MODE>(a.vol, a.sdf, a.ds, a.stats->iters, pp, vv, po, pd);
  st3(a.xt, i, r.xt[0], r.xt[1], r.xt[2], a.io_half, &a.vol, RAY_POS);
  st3(a.vt, i, r.vt[0], r.vt[1], r.vt[2], a.io_half, &a.vol, RAY_VEL);
  if (MODE == 1) a.failmask[i] = r.esc ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// forward march with per-wave LDS bricks of the refractive-index grid (OPT-IN: DRRT_FLAG_LDS_BRICKS)
//
// Idea (north_star: "LDS-staged voxel bricks for the 8-tap stencil"): rays of a wave are spatially
// coherent (locality sort), so each wave stages the box of voxels it is walking through in LDS --
// one batched, independent fill instead of a dependent gather per step -- and reads its taps with
// ds_read2_b32.  Lanes whose cell is not in the brick gather from global memory as before; the taps
// are the same floats either way, so results are bit-identical (tests/test_gpu_parity.py).
// MEASURED (MI355X, 256^3 / 1M rays): 1.97 ms (2.4 ms with 10^3 bricks filled by dword loads), against
// 1.50 ms for the default kernel whose four 8-byte pair gathers per step are served by L1/L2.  The fill
// is cheap now (four 16-byte loads per lane every ~14 steps), but both kernels run at ~0.014 ms per
// VALU instruction per step, and the brick bookkeeping (membership test, ballots, re-anchoring) adds
// ~35 VALU instructions to the 107 of the plain march.  Kept as an option for grids that do not stay
// cache-resident; NOT the default.
// ---------------------------------------------------------------------------------------------
// Brick = 16 (x, the memory-contiguous axis) x 8 x 8 voxels: one row is a 64-byte segment, so the
// whole brick (4 KiB) is FOUR 16-byte loads per lane -- the wide, coalesced access the texture
// addresser is good at -- against 4 scattered 8-byte gathers per lane PER STEP without bricks.
constexpr int kBX = 16, kBY = 8, kBZ = 8;
constexpr int kBPX = 20;                                   // row pitch in floats (80 B: keeps 16-B alignment, skews banks)
constexpr int kBSY = kBPX, kBSZ = kBPX * kBY;
constexpr int kBFloats = kBSZ * kBZ;                       // 1280 floats = 5 KiB per wave

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): LDS ops of this wave have completed
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;     // 16-byte load, 4-byte alignment
typedef float __attribute__((ext_vector_type(4))) f4a;

__device__ __forceinline__ void nwin_fill(float* nw, int ox, int oy, int oz, const Vol& V, int lane) {
  wave_lds_sync();                           // earlier reads of the old brick are done
  f4a v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {              // quad q = lane + 64 j  ->  row q/4 = ly + 8 lz, x-quad q%4
    const int q = lane + kWave * j;
    const int lx = (q & 3) * 4, row = q >> 2, ly = row & 7, lz = row >> 3;
    const int gx = ox + lx, gy = oy + ly, gz = oz + lz;
    v[j] = f4a{0.f, 0.f, 0.f, 0.f};
    if ((gy < V.H) & (gz < V.D)) {
      const float* src = V.data + (unsigned)(gz * V.sz + gy * V.sy + gx);
      if (gx + 3 < V.W) {
        const f4u u = *reinterpret_cast<const f4u*>(src);
        v[j] = f4a{u.x, u.y, u.z, u.w};
      } else {                               // brick sticks out of a narrow grid: element-wise
        if (gx < V.W) v[j].x = src[0];
        if (gx + 1 < V.W) v[j].y = src[1];
        if (gx + 2 < V.W) v[j].z = src[2];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = lane + kWave * j;
    *reinterpret_cast<f4a*>(nw + (q >> 2) * kBPX + (q & 3) * 4) = v[j];        // ds_write_b128
  }
  wave_lds_sync();
}

__device__ __forceinline__ bool nwin_local(int wox, int woy, int woz, const Cell& c, int& lidx) {
  const int lx = c.ix - wox, ly = c.iy - woy, lz = c.iz - woz;
  lidx = lz * kBSZ + ly * kBSY + lx;
  return ((unsigned)lx < (unsigned)(kBX - 1)) & ((unsigned)ly < (unsigned)(kBY - 1)) &
         ((unsigned)lz < (unsigned)(kBZ - 1));
}

__device__ __forceinline__ Taps fetch_lds(const float* nw, int lidx) {
  const float* q = nw + lidx;
  Taps t;
  t.a = f2{q[0], q[1]};                       t.b = f2{q[kBSY], q[kBSY + 1]};
  t.e = f2{q[kBSZ], q[kBSZ + 1]};             t.f = f2{q[kBSZ + kBSY], q[kBSZ + kBSY + 1]};
  return t;
}

template <int MODE>     // 0 = trace, 1 = trace_plane
__global__ void __launch_bounds__(kBlock) k_trace_win(TraceArgs a) {
  __shared__ __attribute__((aligned(16))) float s_n[kBlock / kWave][kBFloats];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  float* nw = s_n[wid];
  const Vol& V = a.vol;
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  size_t i = 0;
  FwdState s;
  s.x = s.y = s.z = s.vx = s.vy = s.vz = 0.f;
  s.aux0 = s.aux1 = s.aux2 = s.aux3 = s.aux4 = s.aux5 = 0.f;
  bool live = ray_index(a.perm, t, a.n, i);
  const bool mine = live;
  if (live) {
    Ray3 p = ld3(a.pos, i, a.io_half, &a.vol, RAY_POS), u = ld3(a.vel, i, a.io_half, &a.vol, RAY_VEL);
    s.x = p.x; s.y = p.y; s.z = p.z; s.vx = u.x; s.vy = u.y; s.vz = u.z;
    if (MODE == 1) {
      Ray3 o = ld3(a.pln_o, i), d = ld3(a.pln_d, i);
      s.aux0 = o.x; s.aux1 = o.y; s.aux2 = o.z; s.aux3 = d.x; s.aux4 = d.y; s.aux5 = d.z;
    }
  }
  fwd_init(V, s);
  Cell c = locate(V, s.x, s.y, s.z);
  int wox = -(1 << 28), woy = -(1 << 28), woz = -(1 << 28);     // no brick yet
  int cooldown = 0;
  unsigned steps = 0;
  for (int it = 0; it < a.max_steps; ++it) {
    if (!__any(live)) break;                                                // wave-uniform exit
    const bool need = live & s.inside;                                      // lanes that gather this step
    const bool regular = (c.ox == 1) & (c.oy == V.sy) & (c.oz == V.sz);
    int lidx;
    bool inw = need & regular & nwin_local(wox, woy, woz, c, lidx);
    const unsigned long long cm = __ballot(need & regular);
    const unsigned long long mm = __ballot(need & regular & !inw);
    if (mm != 0ull && cooldown == 0) {
      // ---- move the brick ahead of the rays (wave-uniform branch) ----
      const int first = __ffsll((long long)cm) - 1, last = 63 - __clzll((long long)cm);
      int ref = (first + last) >> 1;
      if (!((cm >> ref) & 1ull)) ref = first;
      const int rx = __shfl(c.ix, ref, kWave), ry = __shfl(c.iy, ref, kWave), rz = __shfl(c.iz, ref, kWave);
      const float dx_ = __shfl(s.vx, ref, kWave), dy_ = __shfl(s.vy, ref, kWave), dz_ = __shfl(s.vz, ref, kWave);
      const float dm = fmaxf(fmaxf(fabsf(dx_), fabsf(dy_)), fmaxf(fabsf(dz_), 1e-30f));
      const float fx = 0.5f - 0.4f * (dx_ / dm), fy = 0.5f - 0.4f * (dy_ / dm), fz = 0.5f - 0.4f * (dz_ / dm);
      int ox = rx - (int)(fx * (float)(kBX - 2));
      int oy = ry - (int)(fy * (float)(kBY - 2));
      int oz = rz - (int)(fz * (float)(kBZ - 2));
      ox = max(0, min(ox, V.W - kBX)) & ~3;                                 // 16-byte aligned rows when W % 4 == 0
      oy = max(0, min(oy, V.H - kBY)); oz = max(0, min(oz, V.D - kBZ));
      wox = __builtin_amdgcn_readfirstlane(ox); woy = __builtin_amdgcn_readfirstlane(oy);
      woz = __builtin_amdgcn_readfirstlane(oz);
      nwin_fill(nw, wox, woy, woz, V, lane);
      inw = need & regular & nwin_local(wox, woy, woz, c, lidx);
      cooldown = (__ballot(need & regular & !inw) != 0ull) ? 4 : 0;
    } else if (cooldown > 0) {
      --cooldown;
    }
    if (live) {
      Taps tp = taps_zero();
      if (inw) tp = fetch_lds(nw, lidx);
      else if (need) tp = fetch(V.data, c);
      fwd_step_c<MODE>(V, nullptr, a.ds, s, c, tp);
      ++steps;
      if (s.esc) live = false;                                              // per-ray form of :82
    }
  }
  unsigned failed = 0;
  if (mine) {
    if (!s.esc) { s.xtx = s.x; s.xty = s.y; s.xtz = s.z; }                  // :95 (vt stays, Q6)
    failed = s.esc ? 0u : 1u;
    st3(a.xt, i, s.xtx, s.xty, s.xtz, a.io_half, &a.vol, RAY_POS);
    st3(a.vt, i, s.vtx, s.vty, s.vtz, a.io_half, &a.vol, RAY_VEL);
    // src/tracer.cpp:171; bit 1 = "may record a later exit": re-marched by k_trace_again (same test as trace_ray)
    if (MODE == 1) a.failmask[i] = (s.esc ? 0 : 1) | ((s.esc && plane_again(V, s)) ? 2 : 0);
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
  size_t i;
  if (ray_index(a.perm, t, a.n, i)) {
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

// trace_target phase A on the flat march (see k_trace_flat): the same loop with the closest-approach record of
// target_ray_a kept in registers (:216-227 -- updated on EVERY iteration, also the one that ends the march).
template <bool PAIR>
__global__ void __launch_bounds__(kBlock) k_target_a_flat(TargetArgs a) {
  const Vol& V = a.vol;
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const TapRows R = tap_rows<PAIR>(V);
  unsigned steps = 0, failed = 0;
  size_t i;
  if (ray_index(a.perm, t, a.n, i)) {
    const Ray3 p = ld3(a.pos, i), u = ld3(a.vel, i), tg = ld3(a.target, i);
    FlatRay r;
    const FlatPlane P{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    flat_begin<PAIR>(V, R, r, p, u);
    float bx = p.x, by = p.y, bz = p.z, bvx = u.x, bvy = u.y, bvz = u.z;          // :56-57
    float best;
    { const float ex = p.x - tg.x, ey = p.y - tg.y, ez = p.z - tg.z; best = dot3(ex, ey, ez, ex, ey, ez); }   // :200
    for (int it = 0; it < a.max_steps; ++it) {
      if (flat_advance<PAIR, 0>(V, P, a.ds, it, r)) gather_rows<PAIR>(R, r.off, r.q0, r.q1);
      const float ex = r.x - tg.x, ey = r.y - tg.y, ez = r.z - tg.z;
      const float cur = dot3(ex, ey, ez, ex, ey, ez);
      if (cur < best) { bx = r.x; by = r.y; bz = r.z; bvx = r.vx; bvy = r.vy; bvz = r.vz; best = cur; }
      if (!r.live) break;
    }
    const bool esc = !r.live;
    steps = esc ? r.steps : (a.max_steps > 0 ? (unsigned)a.max_steps : 0u);
    failed = esc ? 0u : 1u;
    st3(a.xt, i, bx, by, bz); st3(a.vt, i, bvx, bvy, bvz); a.dist2[i] = best;
    float* w = a.state;
    w[0 * a.n + i] = r.x; w[1 * a.n + i] = r.y; w[2 * a.n + i] = r.z;
    w[3 * a.n + i] = r.vx; w[4 * a.n + i] = r.vy; w[5 * a.n + i] = r.vz;
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
};

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_backtrace_direct(BackArgs a) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned steps = 0;
  size_t i;
  if (ray_index(a.perm, t, a.n, i)) {
    Ray3 p = ld3(a.xt, i, a.io_half, &a.vol, RAY_POS), u = ld3(a.vt, i, a.io_half, &a.vol, RAY_VEL), gxv = ld3(a.dx, i, a.io_half), gvv = ld3(a.dv, i, a.io_half);
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
// adjoint march with per-wave LDS gradient windows (the default adjoint kernel)
//
// Why: one global fp32 atomic per tap runs at the memory-side atomic rate, and that rate collapses
// when many lanes hit the same few addresses (measured on MI355X, Luneburg 256^3 / 1M rays: 237 ms
// for 4.1e9 lane-atomics -- every ray passes through the handful of voxels around the focus).
// Rays of a wave are spatially coherent (locality sort), so each wave keeps a small box of
// gradient voxels ("window") in LDS and accumulates there (ds_add_f64, fed by per-lane register
// accumulators that emit one cell face at a time, see below); the window is flushed to the global
// grid -- one fp32 atomic per touched voxel, contiguous in x -- only when the rays walk out of it.  Lanes whose cell falls outside the window (incoherent wave, clamped
// boundary cell) fall back to direct global atomics, so the result never depends on the window.
//
//   window      kWinX x kWinY x kWinZ voxels of double accumulators, row pitch kWinPX (measured, final
//               kernels: edge 9 beats 7, 8, 10, 12; padding the pitch or not is within 1 %)
//   ablations   BackArgs::experiment (bits 8..15 of `flags`, development only): 1 = no accumulation
//               at all, 2 = no global atomics, 3 = no LDS adds, 4 = hand over all 8 corners on every
//               leave, 5 = never flush, 6 = no DPP pre-reduction, 7 = the box-window kernel, nothing ablated
//   anchor      around the cell of the wave's median contributing lane, shifted towards its
//               direction of travel (most of the window lies ahead of the rays)
//   re-anchor   as soon as a contributing lane misses the window (wave-uniform decision); if lanes
//               still miss afterwards the window stays put for 4 steps (no thrashing)
//   sync        none across waves: every wave owns its window; all control flow around the
//               cooperative flush is wave-uniform (ballot / readlane values)
// ---------------------------------------------------------------------------------------------
// Accumulators are DOUBLES: measured on gfx950 (tools/lds_atomic_bench.hip) ds_add_f32 costs ~193
// cycles per wave-instruction per CU even without address collisions (~3 cycles per lane), while
// ds_add_f64 costs ~8 (ds_add_u32 4.4); collisions add ~12 cycles per colliding lane for f64.  The
// window sums are therefore also more accurate than fp32 atomics; they are rounded to fp32 once,
// when the window is flushed into the fp32 grid.
typedef double win_t;
#ifndef DRRT_WIN
#define DRRT_WIN 9     // measured on MI355X (256^3 / 1M rays, same box): 7 -> 6.75 ms, 8 -> 6.08, 9 -> 4.94, 10 -> 5.23
#endif
#ifndef DRRT_WIN_PAD
#define DRRT_WIN_PAD 1
#endif
constexpr int kWinX = DRRT_WIN, kWinY = DRRT_WIN, kWinZ = DRRT_WIN;
constexpr int kWinPX = DRRT_WIN + DRRT_WIN_PAD;           // row pitch
constexpr int kWinSY = kWinPX, kWinSZ = kWinPX * kWinY;   // LDS strides of y and z
constexpr int kWinFloats = kWinSZ * kWinZ;                // 810 slots = 6.3 KiB per wave (9^3 window, pitch 10)
constexpr int kWavesPerBlock = kBlock / kWave;
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

// Flush the wave's window into the global grid and leave it zeroed.  Called with all 64 lanes.
// The window's rows (a row = the kWinX slots of one (ly, lz)) are contiguous in LDS at pitch kWinPX, so the flush walks
// them linearly: a pass covers 64 / kWinX rows (lane -> row lane / kWinX, slot lane % kWinX), the LDS address advances
// by a constant and the grid address by a constant plus a wrap from one z-slice to the next.  Passes go in batches of
// four: the four LDS exchanges are issued before the first result is used, so a flush costs ceil(rows / 7 / 4) LDS
// round trips (3 for a 9^3 window) instead of one per pass.
__device__ __forceinline__ void win_flush(win_t* win, int ox, int oy, int oz, float* __restrict__ grad,
                                          const Vol& V, int lane, bool no_global = false) {
  constexpr int kRowsPerPass = kWave / kWinX, kRowsTotal = kWinY * kWinZ;
  constexpr int kPasses = (kRowsTotal + kRowsPerPass - 1) / kRowsPerPass, kBatch = 4;
  static_assert(kWinX <= 16 && kRowsPerPass <= kWinY, "win_flush: one pass must not span more than two z-slices");
  wave_lds_fence();
  const int rsub = lane / kWinX, lx = lane - rsub * kWinX;
  const bool lane_ok = rsub < kRowsPerPass;
  int r = rsub, ly = rsub;                                               // row index, its ly (lz = 0: kRowsPerPass <= kWinY)
  win_t* wl = win + rsub * kWinPX + lx;
  unsigned go = (unsigned)oz * (unsigned)V.sz + (unsigned)(oy + rsub) * (unsigned)V.sy + (unsigned)(ox + lx);
  const unsigned step_y = (unsigned)kRowsPerPass * (unsigned)V.sy, wrap = (unsigned)V.sz - (unsigned)kWinY * (unsigned)V.sy;
#pragma unroll 1
  for (int p0 = 0; p0 < kPasses; p0 += kBatch) {
    win_t v[kBatch];
    unsigned g[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      v[b] = (win_t)0; g[b] = go;
      // ds_wrxchg_rtn_b64: read the accumulated value and reset the slot in one LDS op
      if (lane_ok & (r < kRowsTotal)) v[b] = __hip_atomic_exchange(wl, (win_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      r += kRowsPerPass; wl += kRowsPerPass * kWinPX; ly += kRowsPerPass; go += step_y;
      if (ly >= kWinY) { ly -= kWinY; go += wrap; }
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b)
      if (v[b] != (win_t)0 && !no_global) atomic_add_f32(grad + g[b], (float)v[b]);
  }
  wave_lds_fence();
}

// Per-lane register accumulators for the 8 corners of the cell the ray is currently in.  A ray
// stays in one cell for ~h/ds steps and neighbouring cells share a face, so the lane sums its own
// contributions in registers and hands them to the LDS window only when the ray leaves the cell:
// a move across ONE face emits the 4 corners left behind and carries the 4 shared ones over.
// This cuts the ds_add_f32 traffic ~4x, and -- just as important -- same-address collisions inside
// one LDS atomic instruction (rays of a wave share voxels; colliding lanes are serialised by the
// LDS: measured 18 ms of a 25 ms adjoint) because lanes cross faces at different steps.
// (kept as plain scalars, not a struct: hipcc otherwise parks the aggregate in scratch memory)

struct WinCtx {
  win_t* win; float* grad; int wox, woy, woz; int sy, sz; int experiment;
};

__device__ __forceinline__ bool win_local(const WinCtx& W, int cx, int cy, int cz, int& lidx) {
  const int lx = cx - W.wox, ly = cy - W.woy, lz = cz - W.woz;
  lidx = lz * kWinSZ + ly * kWinSY + lx;
  return ((unsigned)lx < (unsigned)(kWinX - 1)) & ((unsigned)ly < (unsigned)(kWinY - 1)) &
         ((unsigned)lz < (unsigned)(kWinZ - 1));
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
// add all 8 accumulated corners of the lane's cell; returns true when the LDS window took them
__device__ __forceinline__ bool emit8(const WinCtx& W, int cx, int cy, int cz, int base,
                                      float a000, float a100, float a010, float a110,
                                      float a001, float a101, float a011, float a111) {
  int lidx;
  const bool inw = win_local(W, cx, cy, cz, lidx);
  if (inw) {
    if (W.experiment != 3) {
      win_t* q = W.win + lidx;
      atomicAdd(q, (win_t)a000);                        atomicAdd(q + 1, (win_t)a100);
      atomicAdd(q + kWinSY, (win_t)a010);               atomicAdd(q + kWinSY + 1, (win_t)a110);
      atomicAdd(q + kWinSZ, (win_t)a001);               atomicAdd(q + kWinSZ + 1, (win_t)a101);
      atomicAdd(q + kWinSZ + kWinSY, (win_t)a011);      atomicAdd(q + kWinSZ + kWinSY + 1, (win_t)a111);
    }
  } else if (W.experiment != 2) {
    float* g = W.grad + base;
    atomic_add_f32(g, a000);                     atomic_add_f32(g + 1, a100);
    atomic_add_f32(g + W.sy, a010);              atomic_add_f32(g + W.sy + 1, a110);
    atomic_add_f32(g + W.sz, a001);              atomic_add_f32(g + W.sz + 1, a101);
    atomic_add_f32(g + W.sz + W.sy, a011);       atomic_add_f32(g + W.sz + W.sy + 1, a111);
  }
  return inw;
}

// The ray moved from cell A to the face-adjacent cell (axis 0/1/2, dir +1/-1): emit the face of A
// that is left behind, carry the shared face over, zero the new far face.  Returns true when the
// LDS window took the emitted corners.
__device__ __forceinline__ bool shift_emit4(const WinCtx& W, int cx, int cy, int cz, int base,
                                            float& a000, float& a100, float& a010, float& a110,
                                            float& a001, float& a101, float& a011, float& a111,
                                            int axis, int dir) {
  const bool ax = axis == 0, ay = axis == 1;
  // the two faces perpendicular to `axis` (lo: corner bit 0, hi: corner bit 1), in a fixed (p, q) order
  const float lo0 = ax ? a000 : (ay ? a000 : a000), hi0 = ax ? a100 : (ay ? a010 : a001);
  const float lo1 = ax ? a010 : (ay ? a100 : a100), hi1 = ax ? a110 : (ay ? a110 : a101);
  const float lo2 = ax ? a001 : (ay ? a001 : a010), hi2 = ax ? a101 : (ay ? a011 : a011);
  const float lo3 = ax ? a011 : (ay ? a101 : a110), hi3 = ax ? a111 : (ay ? a111 : a111);
  const bool fwd = dir > 0;
  const float e0 = fwd ? lo0 : hi0, e1 = fwd ? lo1 : hi1, e2 = fwd ? lo2 : hi2, e3 = fwd ? lo3 : hi3;   // emitted
  const float k0 = fwd ? hi0 : lo0, k1 = fwd ? hi1 : lo1, k2 = fwd ? hi2 : lo2, k3 = fwd ? hi3 : lo3;   // carried
  // LDS / global strides of the in-face directions p (elements 0->1) and q (0->2), and of the axis itself
  const int lp = ax ? kWinSY : 1, lq = (ax | ay) ? kWinSZ : kWinSY, la = ax ? 1 : (ay ? kWinSY : kWinSZ);
  const int gp = ax ? W.sy : 1, gq = (ax | ay) ? W.sz : W.sy, ga = ax ? 1 : (ay ? W.sy : W.sz);
  int lidx;
  const bool inw = win_local(W, cx, cy, cz, lidx);
  if (inw) {
    if (W.experiment != 3) {
      const int qi = lidx + (fwd ? 0 : la);
      // Pair / quad pre-reduction: sorted rays put the 4 lanes of a quad in the same cell, crossing the same face in
      // the same step, most of the time -- then the lanes' values go to the same four LDS slots.  One lane adds the
      // quad's sums (or one lane per pair the pair's sums) instead of all lanes colliding on each slot.  (DPP reads
      // of lanes that are not in this branch return -1 / 0, so a partly active quad or pair does not qualify.)
      const int key = qi | (axis << 16);
      const int k1 = __builtin_amdgcn_update_dpp(-1, key, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
      const int k2 = __builtin_amdgcn_update_dpp(-1, key, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
      const int k3 = __builtin_amdgcn_update_dpp(-1, key, 0x1B, 0xF, 0xF, false);   // quad_perm [3,2,1,0]
      const bool psame = k1 == key;                                  // my pair partner goes to the same slots
      const bool same = psame & (k2 == key) & (k3 == key);           // the whole quad does
      float p0 = e0, p1 = e1, p2 = e2, p3 = e3;                      // pair sums, then quad sums
      p0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p0), 0xB1, 0xF, 0xF, false));
      p1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p1), 0xB1, 0xF, 0xF, false));
      p2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p2), 0xB1, 0xF, 0xF, false));
      p3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p3), 0xB1, 0xF, 0xF, false));
      const float s0 = p0 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p0), 0x4E, 0xF, 0xF, false));
      const float s1 = p1 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p1), 0x4E, 0xF, 0xF, false));
      const float s2 = p2 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p2), 0x4E, 0xF, 0xF, false));
      const float s3 = p3 + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p3), 0x4E, 0xF, 0xF, false));
      const unsigned ql = threadIdx.x & 3u;
      const bool add = same ? ql == 0u : (psame ? (ql & 1u) == 0u : true);
      if (add) {
        win_t* q = W.win + qi;
        atomicAdd(q, (win_t)(same ? s0 : (psame ? p0 : e0)));      atomicAdd(q + lp, (win_t)(same ? s1 : (psame ? p1 : e1)));
        atomicAdd(q + lq, (win_t)(same ? s2 : (psame ? p2 : e2))); atomicAdd(q + lq + lp, (win_t)(same ? s3 : (psame ? p3 : e3)));
      }
    }
  } else if (W.experiment != 2) {
    float* g = W.grad + base + (fwd ? 0 : ga);
    atomic_add_f32(g, e0); atomic_add_f32(g + gp, e1); atomic_add_f32(g + gq, e2); atomic_add_f32(g + gq + gp, e3);
  }
  // new cell: the carried face becomes the near face (lo when moving forward, hi when moving backward)
  const float n_lo0 = fwd ? k0 : 0.f, n_lo1 = fwd ? k1 : 0.f, n_lo2 = fwd ? k2 : 0.f, n_lo3 = fwd ? k3 : 0.f;
  const float n_hi0 = fwd ? 0.f : k0, n_hi1 = fwd ? 0.f : k1, n_hi2 = fwd ? 0.f : k2, n_hi3 = fwd ? 0.f : k3;
  // scatter the faces back (inverse of the gather above)
  a000 = n_lo0;
  a100 = ax ? n_hi0 : n_lo1;                       // x: hi0 | y: lo1 | z: lo1
  a010 = ax ? n_lo1 : (ay ? n_hi0 : n_lo2);        // x: lo1 | y: hi0 | z: lo2
  a110 = ax ? n_hi1 : (ay ? n_hi1 : n_lo3);        // x: hi1 | y: hi1 | z: lo3
  a001 = ax ? n_lo2 : (ay ? n_lo2 : n_hi0);        // x: lo2 | y: lo2 | z: hi0
  a101 = ax ? n_hi2 : (ay ? n_lo3 : n_hi1);        // x: hi2 | y: lo3 | z: hi1
  a011 = ax ? n_lo3 : (ay ? n_hi2 : n_hi2);        // x: lo3 | y: hi2 | z: hi2
  a111 = n_hi3;
  return inw;
}

// ABL = true compiles the development ablations and debug counters in (DRRT_FLAG_DEBUG_COUNTERS or ablation
// bits set); the product instantiation has neither in its loop.
// The ONE gather site of the pipelined loop: the four pair loads of a strictly interior cell, unless the lane
// already holds that cell's taps.  (A single load site with loop-carried destination registers keeps the
// compiler's s_waitcnt at the first USE of the taps, i.e. at the top of the next iteration.)
__device__ __forceinline__ void prefetch_taps(const Vol& V, const Cell& cn, TapCache& tc) {
  if (!cn.interior) { tc.base = -1; return; }
  if (cn.base == tc.base) return;
  __builtin_assume(cn.base >= 0 && cn.base < (1 << 29));
  // the four tap rows as wave-uniform base pointers + ONE 32-bit byte offset per lane (global_load ... saddr)
  const char* const d00 = (const char*)V.data;
  const char* const d10 = d00 + 4u * (unsigned)V.sy;
  const char* const d01 = d00 + 4u * (unsigned)V.sz;
  const char* const d11 = d01 + 4u * (unsigned)V.sy;
  const unsigned off = (unsigned)cn.base << 2;
  tc.t.a = ld_pair((const float*)(d00 + off)); tc.t.b = ld_pair((const float*)(d10 + off));
  tc.t.e = ld_pair((const float*)(d01 + off)); tc.t.f = ld_pair((const float*)(d11 + off));
  tc.base = cn.base;
}

// PIPE = true (the product default): the loop is software-pipelined.  The march's critical path is
//   taps(x_k) -> n, grad n -> v_k -> x_{k+1} -> taps(x_{k+1}) -> ...
// while the gradient bookkeeping of step k (8 splat weights, lambda / mu, register accumulation, face emission,
// LDS adds) only CONSUMES (n, grad n, H) of step k.  So each iteration samples, updates v, steps to x_{k+1} and
// issues THAT cell's gather first -- skipping it when the ray stays in its cell (fetch_reuse) -- and only then does
// the bookkeeping of step k, under the gather's latency.  Same per-ray arithmetic (adj_sample / adj_contrib), same
// emission order, so the results equal the unpipelined loop's.
template <int MODE, bool ABL = false, bool PIPE = true>
__global__ void __launch_bounds__(kBlock) k_backtrace_win(BackArgs a) {
  __shared__ win_t s_win[kWavesPerBlock][kWinFloats];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  win_t* win = s_win[wid];
  for (int k = lane; k < kWinFloats; k += kWave) win[k] = (win_t)0;
  wave_lds_fence();

  const Vol& V = a.vol;
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  AdjState s;
  s.active = false;
  size_t i;
  if (ray_index(a.perm, t, a.n, i)) {
    Ray3 p = ld3(a.xt, i, a.io_half, &a.vol, RAY_POS), u = ld3(a.vt, i, a.io_half, &a.vol, RAY_VEL), gxv = ld3(a.dx, i, a.io_half), gvv = ld3(a.dv, i, a.io_half);
    s.x = p.x; s.y = p.y; s.z = p.z; s.vx = u.x; s.vy = u.y; s.vz = u.z;
    adj_init(V, a.ds, gxv.x, gxv.y, gxv.z, gvv.x, gvv.y, gvv.z, s);
    if (MODE == 1 && s.active) {                                            // src/tracer.cpp:476-477
      Cell c = locate(V, s.x, s.y, s.z);
      s.outside = interp<false>(fetch(a.sdf, c), c.wx, c.wy, c.wz).n >= 0.f;
    }
  }
  WinCtx W;
  W.win = win; W.grad = a.grad; W.sy = V.sy; W.sz = V.sz; W.experiment = ABL ? a.experiment : 0;
  // window origin (wave-uniform).  Start far away: nothing is "in the window" before the first anchor.
  W.wox = W.woy = W.woz = -(1 << 28);
  bool acc_valid = false;
  int acx = 0, acy = 0, acz = 0, abase = 0;
  float a000 = 0.f, a100 = 0.f, a010 = 0.f, a110 = 0.f, a001 = 0.f, a101 = 0.f, a011 = 0.f, a111 = 0.f;
  bool dirty = false;
  int cooldown = 0;
  unsigned steps = 0;
  unsigned n_flush = 0, n_lds = 0, n_glb = 0;
  const int experiment = ABL ? a.experiment : 0;
  unsigned long long* const dbg = ABL ? a.dbg : nullptr;

  // PIPE: cell of the sample the ray stands on + its taps (gather already issued)
  Cell cn;
  cn.base = 0; cn.ix = cn.iy = cn.iz = 0; cn.ox = cn.oy = cn.oz = 0; cn.wx = cn.wy = cn.wz = 0.f; cn.interior = false;
  TapCache tc;
  tc.base = -1; tc.t = taps_zero();
  if (PIPE && s.active) {
    s.x = fmaf(-a.ds, s.vx, s.x); s.y = fmaf(-a.ds, s.vy, s.y); s.z = fmaf(-a.ds, s.vz, s.z);   // :420, first sample
    cn = locate(V, s.x, s.y, s.z);
    prefetch_taps(V, cn, tc);
  }

  for (int it = 0; it < a.max_steps; ++it) {
    if (!__any(s.active | acc_valid)) break;                                  // wave-uniform exit
    Cell c; Corners w;
    c.base = 0; c.ix = c.iy = c.iz = 0; c.ox = c.oy = c.oz = 0;
    bool contrib = false;
    if (PIPE) {
      AdjSample m;
      if (s.active) {
        c = cn;
        if (!c.interior) tc.t = fetch(V.data, c);      // boundary cell (clamped neighbours): fetched here, not ahead
        contrib = adj_sample<MODE>(V, a.sdf, a.ds, s, c, tc.t, m);
        if (contrib) {
          ++steps;
          // step to the next sample and issue its gather now; nothing below depends on it (it + 1 == max_steps: the
          // extra sample is located and fetched but never used -- locate() clamps, so the addresses are valid)
          s.x = fmaf(-a.ds, s.vx, s.x); s.y = fmaf(-a.ds, s.vy, s.y); s.z = fmaf(-a.ds, s.vz, s.z);   // :420
          cn = locate(V, s.x, s.y, s.z);
          prefetch_taps(V, cn, tc);
          adj_contrib(V, a.ds, a.grad_scale, s, c, m, w);
        }
      }
    } else if (s.active) {
      contrib = adj_step<MODE>(V, a.sdf, a.ds, a.grad_scale, s, c, w);
      if (contrib) ++steps;
    }
    // the 8 taps are distinct voxels with the regular strides (no clamped neighbour)
    const bool regular = (c.ox == 1) & (c.oy == V.sy) & (c.oz == V.sz);
    int lidx_cur;
    bool inw = contrib & regular & win_local(W, c.ix, c.iy, c.iz, lidx_cur);
    const unsigned long long cm = __ballot(contrib & regular);
    const unsigned long long mm = __ballot(contrib & regular & !inw);
    if (mm != 0ull && cooldown == 0) {
      // ---- re-anchor around the cells the rays are in NOW (wave-uniform branch) ----
      if (dirty) { win_flush(win, W.wox, W.woy, W.woz, a.grad, V, lane, experiment == 2); dirty = false; ++n_flush; }
      // reference lane: the middle of the contributing lanes (sorted rays => spatial median-ish)
      const int first = __ffsll((long long)cm) - 1, last = 63 - __clzll((long long)cm);
      int ref = (first + last) >> 1;
      if (!((cm >> ref) & 1ull)) ref = first;
      const int rx = __shfl(c.ix, ref, kWave), ry = __shfl(c.iy, ref, kWave), rz = __shfl(c.iz, ref, kWave);
      // backward direction of travel of the reference lane: d = -v
      const float dx_ = -__shfl(s.vx, ref, kWave), dy_ = -__shfl(s.vy, ref, kWave), dz_ = -__shfl(s.vz, ref, kWave);
      const float dm = fmaxf(fmaxf(fabsf(dx_), fabsf(dy_)), fmaxf(fabsf(dz_), 1e-30f));
      // fraction of the window kept BEHIND the reference cell: 0.1 when moving fast along +axis, 0.9 along -axis
      // (at least one cell behind: the cell being left still has to emit its far face)
      const float fx = 0.5f - 0.35f * (dx_ / dm), fy = 0.5f - 0.35f * (dy_ / dm), fz = 0.5f - 0.35f * (dz_ / dm);
      int ox = rx - (int)(fx * (float)(kWinX - 2));
      int oy = ry - (int)(fy * (float)(kWinY - 2));
      int oz = rz - (int)(fz * (float)(kWinZ - 2));
      ox = max(0, min(ox, V.W - kWinX)); oy = max(0, min(oy, V.H - kWinY)); oz = max(0, min(oz, V.D - kWinZ));
      W.wox = __builtin_amdgcn_readfirstlane(ox); W.woy = __builtin_amdgcn_readfirstlane(oy);
      W.woz = __builtin_amdgcn_readfirstlane(oz);
      inw = contrib & regular & win_local(W, c.ix, c.iy, c.iz, lidx_cur);
      // lanes still outside (incoherent wave) use global atomics; do not thrash: leave the window
      // where it is for a few steps before trying again
      const unsigned long long mm2 = __ballot(contrib & regular & !inw);
      cooldown = (mm2 != 0ull) ? 4 : 0;
    } else if (cooldown > 0) {
      --cooldown;
    }
    bool used_lds = false;
    if (contrib) {
      if (ABL && dbg) { if (inw) ++n_lds; else ++n_glb; }
      if (!regular) {
        // clamped boundary cell: taps coincide; bypass accumulators and window
        if (acc_valid) { used_lds |= emit8(W, acx, acy, acz, abase, a000, a100, a010, a110, a001, a101, a011, a111); acc_valid = false; }
        if (experiment != 2 && experiment != 1) {
          float* g = a.grad + c.base;
          atomic_add_f32(g, w.c000);                    atomic_add_f32(g + c.ox, w.c100);
          atomic_add_f32(g + c.oy, w.c010);             atomic_add_f32(g + c.oy + c.ox, w.c110);
          atomic_add_f32(g + c.oz, w.c001);             atomic_add_f32(g + c.oz + c.ox, w.c101);
          atomic_add_f32(g + c.oz + c.oy, w.c011);      atomic_add_f32(g + c.oz + c.oy + c.ox, w.c111);
        }
      } else {
        if (acc_valid && c.base != abase) {
          const int sx = c.ix - acx, sy = c.iy - acy, sz = c.iz - acz;
          const int nchg = (sx != 0) + (sy != 0) + (sz != 0);
          const int ssum = sx + sy + sz;
          if (nchg == 1 && (ssum == 1 || ssum == -1)) {
            if (experiment != 1) used_lds |= shift_emit4(W, acx, acy, acz, abase, a000, a100, a010, a110, a001, a101, a011, a111,
                                                             sx != 0 ? 0 : (sy != 0 ? 1 : 2), ssum);
          } else {
            if (experiment != 1) used_lds |= emit8(W, acx, acy, acz, abase, a000, a100, a010, a110, a001, a101, a011, a111);
            a000 = a100 = a010 = a110 = a001 = a101 = a011 = a111 = 0.f;
          }
        } else if (!acc_valid) {
          a000 = a100 = a010 = a110 = a001 = a101 = a011 = a111 = 0.f;
        }
        acc_valid = true; acx = c.ix; acy = c.iy; acz = c.iz; abase = c.base;
        a000 += w.c000; a100 += w.c100; a010 += w.c010; a110 += w.c110;
        a001 += w.c001; a101 += w.c101; a011 += w.c011; a111 += w.c111;
      }
    } else if (acc_valid) {
      // the ray has ended: hand over what is left
      if (experiment != 1) used_lds |= emit8(W, acx, acy, acz, abase, a000, a100, a010, a110, a001, a101, a011, a111);
      acc_valid = false;
    }
    dirty = dirty | (__ballot(used_lds) != 0ull);
  }
  if (acc_valid && experiment != 1) { if (emit8(W, acx, acy, acz, abase, a000, a100, a010, a110, a001, a101, a011, a111)) dirty = true; }
  dirty = __ballot(dirty) != 0ull;
  if (dirty) { win_flush(win, W.wox, W.woy, W.woz, a.grad, V, lane, experiment == 2); ++n_flush; }
  if (ABL && dbg) {
    unsigned f = lane == 0 ? n_flush : 0u;
    unsigned l = wave_sum_u32(n_lds), g = wave_sum_u32(n_glb);
    if (lane == 0) {
      atomicAdd(&dbg[0], (unsigned long long)f); atomicAdd(&dbg[1], (unsigned long long)l);
      atomicAdd(&dbg[2], (unsigned long long)g);
    }
  }
  block_stats(a.stats, steps, 0u);
}

// ---------------------------------------------------------------------------------------------
// k_backtrace_flat: the same adjoint march and the same window scheme as k_backtrace_win<0>, reorganised around ONE
// invariant -- the lane's register accumulators always belong to the cell the ray stands on -- so that the hot loop
// carries no second cell, no "accumulator cell" bookkeeping and no per-step window test:
//
//   top      taps(x_k) arrive (gather issued one iteration earlier) -> n, grad n, H -> v_k -> still active?
//            8 splat weights of step k (they need the in-cell fractions of cell k) -> accumulators += weights
//   step     x_{k+1} = x_k - ds v_k ; locate its cell IN PLACE (the fractions of cell k are dead by now) ;
//            issue its gather unless the ray stays in its cell
//   then     lambda / mu recurrences (under the gather)
//   leave    only if the cell changed: the ray LEAVES cell k -- a move across one face hands the four corners left
//            behind to the LDS window (quad pre-reduced, as before) and carries the shared four; anything else hands
//            over all eight -- and the window index of the new cell is computed once, here (a miss votes for a
//            re-anchor).  A ray that ends hands over all eight.
// Per-ray arithmetic is adj_sample / adj_contrib of drrt_device.h (bit-identical contributions); only the order in
// which contributions reach the window differs from k_backtrace_win, i.e. the usual fp32 summation-order noise.
// The default kernel of drrt_backtrace_f32 / _f16io (measured on MI355X, 256^3 / 1M rays, same box: 5.18-5.24 ms against
// 5.41-5.50 ms for k_backtrace_win<0> with the pipelined loop and 5.45 ms without); backtrace_sdf, the quad-grid option
// and DRRT_FLAG_LEGACY_ADJOINT keep k_backtrace_win.
// ---------------------------------------------------------------------------------------------
// ---- window of k_backtrace_flat --------------------------------------------------------------------------------
// (Bundles that do not sit in the compile-time window -- sparse views, views oblique to the grid -- are the ring kernel's.)
struct WinOrg { int ox, oy, oz; };   // corner 000 of the wave's window, in voxels (far away = nothing is inside); wave-uniform
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ int win_index(int wox, int woy, int woz, int cx, int cy, int cz) {
  const int lx = cx - wox, ly = cy - woy, lz = cz - woz;
  const bool in = ((unsigned)lx < (unsigned)(kWinX - 1)) & ((unsigned)ly < (unsigned)(kWinY - 1)) &
                  ((unsigned)lz < (unsigned)(kWinZ - 1));
  // v_mad_u32_u24 by hand: with constant strides the compiler turns the 24-bit multiply into the quarter-rate
  // v_mul_lo_u32 (lx, ly, lz are small and non-negative whenever the result is used)
  int r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(ly), "s"(kWinSY), "v"(lx));
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(lz), "s"(kWinSZ), "v"(r));
  return in ? r : -1;
}

// all 8 accumulated corners of the regular cell `base` (window slot lidx, or -1: straight to the grid)
template <bool PRE = true>
__device__ __forceinline__ bool flat_emit8(win_t* win, int wsy, int wsz, float* grad, int sy, int sz, int lidx, int base,
                                           f2 p00, f2 p10, f2 p01, f2 p11) {
  if (lidx >= 0) {
    // PRE: quad pre-reduction -- when the 4 lanes of a quad hand over the same cell, one lane adds the quad's sums
    const bool same = PRE ? quad_same_key(lidx) : false;
    float v[8] = {p00.x, p00.y, p10.x, p10.y, p01.x, p01.y, p11.x, p11.y};
    if (PRE) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float qs = quad_sum(v[k]); v[k] = same ? qs : v[k]; }
    }
    if (!same || (threadIdx.x & 3u) == 0u) {
      win_t* q = win + lidx;
      atomicAdd(q, (win_t)v[0]);                 atomicAdd(q + 1, (win_t)v[1]);
      atomicAdd(q + wsy, (win_t)v[2]);           atomicAdd(q + wsy + 1, (win_t)v[3]);
      atomicAdd(q + wsz, (win_t)v[4]);           atomicAdd(q + wsz + 1, (win_t)v[5]);
      atomicAdd(q + wsz + wsy, (win_t)v[6]);     atomicAdd(q + wsz + wsy + 1, (win_t)v[7]);
    }
    return true;
  }
  float* g = grad + base;
  atomic_add_f32(g, p00.x);            atomic_add_f32(g + 1, p00.y);
  atomic_add_f32(g + sy, p10.x);       atomic_add_f32(g + sy + 1, p10.y);
  atomic_add_f32(g + sz, p01.x);       atomic_add_f32(g + sz + 1, p01.y);
  atomic_add_f32(g + sz + sy, p11.x);  atomic_add_f32(g + sz + sy + 1, p11.y);
  return false;
}

// How do the 64-ray bundles of this call sit at the start of the adjoint march?  One wave per bundle (the rays of 64
// consecutive visit slots), every 16th block of bundles:
//   [0] += 1 when the bounding box of the rays' start cells does not fit the default box window, [1] += 1 per bundle;
//   [2] += the lanes whose start cell lies more than kClassifyReach cells (on any axis) from the bundle's mean cell,
//          [3] += the lanes (diagnostic only: it does NOT predict which kernel is faster, see below).
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
__device__ __forceinline__ bool bundles_want_ring(const unsigned* __restrict__ sel) {
  return sel[0] * 100u >= sel[1] * (unsigned)DRRT_RING_MIN_NOFIT_PCT && sel[0] != 0u;
}
__global__ void __launch_bounds__(kBlock) k_bundle_classify(BackArgs a) {
  __shared__ unsigned s_cnt[4];                  // the block's four counters: one set of global atomics per block, not per wave
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const Vol& V = a.vol;
  const size_t t = (size_t)blockIdx.x * kClassifyStride * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  size_t i;
  bool ok = false;
  int cx = 0, cy = 0, cz = 0;
  if (ray_index(a.perm, t, a.n, i)) {
    const Ray3 p = ld3(a.xt, i, a.io_half, &a.vol, RAY_POS);
    const Cell c = locate(V, p.x, p.y, p.z);
    cx = c.ix; cy = c.iy; cz = c.iz; ok = true;
  }
  const int big = 1 << 28;
  const int x0 = wave_min_i32(ok ? cx : big), x1 = wave_max_i32(ok ? cx : -big);
  const int y0 = wave_min_i32(ok ? cy : big), y1 = wave_max_i32(ok ? cy : -big);
  const int z0 = wave_min_i32(ok ? cz : big), z1 = wave_max_i32(ok ? cz : -big);
  const unsigned lanes = (unsigned)__popcll(__ballot(ok));
  // mean cell of the bundle (rounded); cells are < 2^24 per axis, 64 of them fit an int
  const float inv = 1.0f / (float)max(lanes, 1u);
  const int mx = (int)floorf((float)(int)wave_sum_u32(ok ? (unsigned)cx : 0u) * inv + 0.5f);
  const int my = (int)floorf((float)(int)wave_sum_u32(ok ? (unsigned)cy : 0u) * inv + 0.5f);
  const int mz = (int)floorf((float)(int)wave_sum_u32(ok ? (unsigned)cz : 0u) * inv + 0.5f);
  const bool far = ok && (abs(cx - mx) > kClassifyReach || abs(cy - my) > kClassifyReach || abs(cz - mz) > kClassifyReach);
  const unsigned outside = (unsigned)__popcll(__ballot(far));
  if (lane == 0 && lanes != 0u) {
    const int ex = x1 - x0 + 2, ey = y1 - y0 + 2, ez = z1 - z0 + 2;           // slots per axis
    const bool dflt = (ex <= kWinX - 2) & (ey <= kWinY - 2) & (ez <= kWinZ - 2);   // two slots of room for the placement
    if (!dflt) atomicAdd(&s_cnt[0], 1u);
    atomicAdd(&s_cnt[1], 1u);
    if (outside) atomicAdd(&s_cnt[2], outside);
    atomicAdd(&s_cnt[3], lanes);
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_cnt[threadIdx.x] != 0u) atomicAdd(&a.select[threadIdx.x], s_cnt[threadIdx.x]);
}

#ifndef DRRT_ANCHOR_SHIFT
#define DRRT_ANCHOR_SHIFT 0.35f   // how far the window is pushed towards the direction of travel when it is anchored (0.5 = all of it ahead);
                                  // measured 256^3 / 1M rays, same box: 0.30 -> 4.85 ms, 0.35 -> 4.85, 0.40 -> 5.78 (trailing lanes miss), 0.45 -> 5.90
#endif
#ifndef DRRT_ADJ_WAVES
#define DRRT_ADJ_WAVES 5     // 5 waves per SIMD: caps the kernel at 96 VGPRs (it sits right at that edge); LDS allows 5 blocks per CU too
                             // (4 -> 4.89 ms, 5 -> 4.85, 6 -> 5.19 on the final kernel)
#endif
// PAIR: gather from the pair copy of the grid (two 16-byte loads per cell, see gather_rows).
// When the call has a visit order the host launches this kernel AND k_backtrace_ring, and each returns at once unless
// a.select picks it (k_bundle_classify decides on the device, from how the 64-ray bundles sit at their start, without a
// host round trip).  (Round 2's run-time-sized box windows -- a DYN instantiation of this kernel -- were replaced by the
// ring kernel in round 3 and are gone.)
// MODE: 0 = backtrace, 1 = backtrace_sdf (the ray also ends where the sdf sample turns non-negative, :488-497; the sdf
//       taps ride along with the grid's taps).
template <bool ABL, bool PAIR, int MODE = 0>
__global__ void __launch_bounds__(kAdjBlock, DRRT_ADJ_WAVES) k_backtrace_flat(BackArgs a) {
  if (a.select != nullptr && bundles_want_ring(a.select)) return;
  constexpr int kSlots = kWinFloats;
  __shared__ win_t s_win[kAdjWavesPerBlock][kSlots];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  win_t* win = s_win[wid];
  for (int k = lane; k < kSlots; k += kWave) win[k] = (win_t)0;
  wave_lds_fence();

  const Vol& V = a.vol;
  const size_t t = (size_t)xcd_block(blockIdx.x, gridDim.x, a.xcd_order ? DRRT_FLAT_XCD_MODE : kXcdOff) * kAdjBlock + threadIdx.x;
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
  WinOrg W;                                                    // the wave's window (wave-uniform)
  W.ox = W.oy = W.oz = -(1 << 28);                             // far away = nothing is inside
  // the cell the ray stands on, located IN PLACE: flat index and coordinates of corner 000, in-cell fractions, strictly
  // interior / regular (no clamped neighbour), window slot.  A boundary cell's clamp offsets are not carried: the
  // boundary branch at the top of the step re-derives them from the position (locate()).
  int base = 0, ix = 0, iy = 0, iz = 0;
  float wx = 0.f, wy = 0.f, wz = 0.f;
  bool interior = false, regular = false;
  int lidx = -1;
  f4 q0 = f4{0.f, 0.f, 0.f, 0.f}, q1 = q0;                   // the taps, as gathered (gather_rows)
  int tbase = -1;                                            // cell whose taps the lane holds (-1: none)
  Taps st;                                                   // MODE 1: the sdf taps of that cell, gathered with them
  st.a = st.b = st.e = st.f = f2{0.f, 0.f};
  f2 p00 = f2{0.f, 0.f}, p10 = p00, p01 = p00, p11 = p00;   // accumulators of the cell: x-pairs at (y0,z0) (y1,z0) (y0,z1) (y1,z1)
  bool miss = false;                                         // the cell just entered lies outside the window
  const TapRows R = tap_rows<PAIR>(V);                       // wave-uniform row pointers + ONE 32-bit byte offset per lane
  // step to the next sample (:420), locate its cell in place and issue its gather unless the lane holds those taps
  auto step_locate = [&](int& nbase, bool& nregular) {
    s.x = fmaf(-a.ds, s.vx, s.x); s.y = fmaf(-a.ds, s.vy, s.y); s.z = fmaf(-a.ds, s.vz, s.z);
    const float fx = s.x * V.inv_h, fy = s.y * V.inv_h, fz = s.z * V.inv_h;
    ix = cvt_floor_i32(fx); iy = cvt_floor_i32(fy); iz = cvt_floor_i32(fz);
    interior = (((unsigned)ix - 1u) < V.lx) & (((unsigned)iy - 1u) < V.ly) & (((unsigned)iz - 1u) < V.lz);
    if (interior) {
      // v_fract == f - floor(f) bit for bit for the non-negative coordinates of an interior cell
      wx = __builtin_amdgcn_fractf(fx); wy = __builtin_amdgcn_fractf(fy); wz = __builtin_amdgcn_fractf(fz);
      nbase = mad24(iz, V.sz, mad24(iy, V.sy, ix));
      nregular = true;
      if (nbase != tbase) {
        __builtin_assume(nbase >= 0 && nbase < (1 << 29));
        gather_rows<PAIR>(R, tap_offset<PAIR>(nbase), q0, q1);
        if (MODE == 1) {                                   // the sdf taps of the same cell ride along (src/tracer.cpp:488-497 reads them every step)
          const float* sp = a.sdf + (unsigned)nbase;
          st.a = ld_pair(sp); st.b = ld_pair(sp + V.sy); st.e = ld_pair(sp + V.sz); st.f = ld_pair(sp + V.sz + V.sy);
        }
        tbase = nbase;
      }
    } else {
      const Cell cb = locate(V, s.x, s.y, s.z);
      wx = cb.wx; wy = cb.wy; wz = cb.wz; ix = cb.ix; iy = cb.iy; iz = cb.iz; nbase = cb.base;
      nregular = (cb.ox == 1) & (cb.oy == V.sy) & (cb.oz == V.sz);
      tbase = -1;
    }
  };
  if (s.active) {
    int nbase; bool nregular;
    step_locate(nbase, nregular);                            // first sample
    base = nbase; regular = nregular;
    miss = regular;                                          // no window yet
  }
  bool dirty = false;
  int cooldown = 0;
  unsigned steps = 0;
  unsigned n_flush = 0;
  // event counters of the debug instantiation (a.dbg): [4] one-face leaves handed to the window, [5] of those, lanes that
  // issued the LDS adds after the pair / quad pre-reduction, [6] one-face leaves that went to global atomics (cell outside
  // the window), [7] leaves that handed over all eight corners, [8] wave-steps, [9] wave-steps with leaves across >= 2 axes
  unsigned ev_face = 0, ev_add = 0, ev_glob = 0, ev_all8 = 0, ev_wsteps = 0, ev_multi = 0;

#define WSY kWinSY
#define WSZ kWinSZ
#define WIN_INDEX(cx, cy, cz) win_index(W.ox, W.oy, W.oz, cx, cy, cz)
    for (int it = 0; it < a.max_steps; ++it) {
      if (!__any(s.active)) break;                                              // wave-uniform exit
#if defined(DRRT_PAD_VALU)
      { float pa_ = 1.f, pb_ = 2.f, pc_ = 3.f, pd_ = 4.f;   // sensitivity experiment: 4 * DRRT_PAD_VALU extra v_fma per step
#pragma unroll
        for (int k_ = 0; k_ < DRRT_PAD_VALU; ++k_)
          asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3"
                       : "+v"(pa_), "+v"(pb_), "+v"(pc_), "+v"(pd_)); }
#endif
#if defined(DRRT_PAD_LDS)
#pragma unroll
      for (int k_ = 0; k_ < DRRT_PAD_LDS; ++k_) atomicAdd(win + 64 * k_ + lane, (win_t)0);   // conflict-free ds_add_f64 of 0.0
#endif
#if defined(DRRT_PAD_SALU)
#pragma unroll
      for (int k_ = 0; k_ < DRRT_PAD_SALU; ++k_) asm volatile("s_nop 0");
#endif
      // ---- (re-)anchor the window around the cells the rays stand on (wave-uniform branch) ----
      const unsigned long long mm = __ballot(s.active & miss);
      if (mm != 0ull && cooldown == 0) {
        if (dirty) {
          win_flush(win, W.ox, W.oy, W.oz, a.grad, V, lane, experiment == 2);
          dirty = false; ++n_flush;
        }
        const bool ok = s.active & regular;
        const unsigned long long cm = __ballot(ok);
        const int first = __ffsll((long long)cm) - 1, last = 63 - __clzll((long long)cm);
        int ref = (first + last) >> 1;
        if (!((cm >> ref) & 1ull)) ref = first;
        const int rx = __shfl(ix, ref, kWave), ry = __shfl(iy, ref, kWave), rz = __shfl(iz, ref, kWave);
        // backward direction of travel of the reference lane
        const float dx_ = -__shfl(s.vx, ref, kWave), dy_ = -__shfl(s.vy, ref, kWave), dz_ = -__shfl(s.vz, ref, kWave);
        const float inv_dm = __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(dx_), fabsf(dy_)), fmaxf(fabsf(dz_), 1e-30f)));   // placement only
        {
          // default: a kWin^3 window around the median lane's cell, shifted towards the direction of travel
          const float fx = 0.5f - DRRT_ANCHOR_SHIFT * (dx_ * inv_dm), fy = 0.5f - DRRT_ANCHOR_SHIFT * (dy_ * inv_dm), fz = 0.5f - DRRT_ANCHOR_SHIFT * (dz_ * inv_dm);
          int ox = rx - (int)(fx * (float)(kWinX - 2));
          int oy = ry - (int)(fy * (float)(kWinY - 2));
          int oz = rz - (int)(fz * (float)(kWinZ - 2));
          ox = max(0, min(ox, V.W - kWinX)); oy = max(0, min(oy, V.H - kWinY)); oz = max(0, min(oz, V.D - kWinZ));
          W.ox = __builtin_amdgcn_readfirstlane(ox); W.oy = __builtin_amdgcn_readfirstlane(oy);
          W.oz = __builtin_amdgcn_readfirstlane(oz);
          lidx = regular ? WIN_INDEX(ix, iy, iz) : -1;                          // every lane's cell, in the new window
          miss = ok & (lidx < 0);
        }
        cooldown = (__ballot(miss) != 0ull) ? 4 : 0;                            // incoherent wave: do not thrash
      } else if (cooldown > 0) {
        --cooldown;
      }
      bool used_lds = false;
      if (s.active) {
        if (!interior) taps_set<PAIR>(fetch(V.data, locate(V, s.x, s.y, s.z)), q0, q1);   // boundary cell (clamped neighbours): fetched here, not ahead
        Cell c;                                              // what adj_sample reads of the cell: fractions, interior
        c.base = 0; c.ix = c.iy = c.iz = 0; c.ox = c.oy = c.oz = 0;
        c.wx = wx; c.wy = wy; c.wz = wz; c.interior = interior;
        if (MODE == 1 && !interior) c = locate(V, s.x, s.y, s.z);   // boundary cell: adj_sample<1> gathers its sdf taps itself
        const float px = s.x, py = s.y, pz = s.z;            // position of this sample (the clamped splat re-locates it)
        AdjSample m;
        if (!adj_sample_st<MODE>(V, a.sdf, a.ds, s, c, taps_of<PAIR>(q0, q1), m, st, MODE == 1 && interior)) {
          // the ray has ended (:426-428): it contributes nothing here; hand over what its cell has accumulated
          if (regular && experiment != 1) used_lds = flat_emit8<true>(win, WSY, WSZ, a.grad, V.sy, V.sz, lidx, base, p00, p10, p01, p11);
        } else {
          ++steps;
          {
            // first half of adj_contrib: the 8 splat weights (they need the in-cell fractions of THIS cell)
            const float dn = dot3(s.mx, s.my, s.mz, m.gx, m.gy, m.gz);                            // :430
            const float nds = (m.n * a.ds) * a.grad_scale;
            if (regular) {
              const CornerPairs cp = splat_weights_pk(wx, wy, wz, dn * a.ds, nds * s.mx, nds * s.my, nds * s.mz);   // :431-432
              p00 += cp.c00; p10 += cp.c10; p01 += cp.c01; p11 += cp.c11;
            } else if (experiment != 2 && experiment != 1) {
              // clamped boundary cell: taps coincide; straight to the grid
              const Cell cb = locate(V, px, py, pz);
              const Corners w = splat_weights(cb.wx, cb.wy, cb.wz, dn * a.ds, nds * s.mx, nds * s.my, nds * s.mz);
              float* g = a.grad + cb.base;
              atomic_add_f32(g, w.c000);                     atomic_add_f32(g + cb.ox, w.c100);
              atomic_add_f32(g + cb.oy, w.c010);             atomic_add_f32(g + cb.oy + cb.ox, w.c110);
              atomic_add_f32(g + cb.oz, w.c001);             atomic_add_f32(g + cb.oz + cb.ox, w.c101);
              atomic_add_f32(g + cb.oz + cb.oy, w.c011);     atomic_add_f32(g + cb.oz + cb.oy + cb.ox, w.c111);
            }
            // step to the next sample and issue its gather (it + 1 == max_steps: located and fetched, never used)
            const int old_base = base, old_lidx = lidx;
            const bool old_regular = regular;
            int nbase; bool nregular;
            step_locate(nbase, nregular);
            // second half of adj_contrib: lambda / mu (:434-435)
            const float hxy = m.hxy * V.inv_h2, hxz = m.hxz * V.inv_h2, hyz = m.hyz * V.inv_h2;
            const float hmx = fmaf(hxz, s.mz, hxy * s.my);
            const float hmy = fmaf(hyz, s.mz, hxy * s.mx);
            const float hmz = fmaf(hyz, s.my, hxz * s.mx);
            s.lx = fmaf(a.ds, fmaf(dn, m.gx, m.n * hmx), s.lx);
            s.ly = fmaf(a.ds, fmaf(dn, m.gy, m.n * hmy), s.ly);
            s.lz = fmaf(a.ds, fmaf(dn, m.gz, m.n * hmz), s.lz);
            s.mx = fmaf(a.ds, s.lx, s.mx); s.my = fmaf(a.ds, s.ly, s.my); s.mz = fmaf(a.ds, s.lz, s.mz);
            // ---- the ray leaves its cell ----
            if (nbase != old_base || !interior) {
              base = nbase; regular = nregular;
              const int d = nbase - old_base;
              if (d != 0 || regular != old_regular) {
                const bool ax = (d == 1) | (d == -1), ay = (d == V.sy) | (d == -V.sy), az = (d == V.sz) | (d == -V.sz);
                if (old_regular) {
                  if (regular & (ax | ay | az) & (experiment != 1) & (experiment != 4)) {
                    // one face crossed: emit the face left behind, carry the shared one
                    const bool fwd = d > 0;
                    if (ABL && a.dbg) {
                      const int nax = (__ballot(ax) != 0ull) + (__ballot(ay) != 0ull) + (__ballot(az) != 0ull);
                      if (lane == __ffsll((long long)__ballot(true)) - 1) ev_multi += nax >= 2;
                    }
                    // emitted corners e0..e3 and carried ones, in (p, q) in-face order; LDS / grid strides of p, q and of the axis
                    float e0, e1, e2, e3;
                    int lp, lq, la, gp, gq, ga;
                    if (ay) {
                      const f2 ea = fwd ? p00 : p10, eb = fwd ? p01 : p11;
                      e0 = ea.x; e1 = ea.y; e2 = eb.x; e3 = eb.y;
                      const f2 ka = fwd ? p10 : p00, kb = fwd ? p11 : p01;
                      p00 = fwd ? ka : f2{0.f, 0.f}; p01 = fwd ? kb : f2{0.f, 0.f};
                      p10 = fwd ? f2{0.f, 0.f} : ka; p11 = fwd ? f2{0.f, 0.f} : kb;
                      lp = 1; lq = WSZ; la = WSY; gp = 1; gq = V.sz; ga = V.sy;
                    } else if (az) {
                      const f2 ea = fwd ? p00 : p01, eb = fwd ? p10 : p11;
                      e0 = ea.x; e1 = ea.y; e2 = eb.x; e3 = eb.y;
                      const f2 ka = fwd ? p01 : p00, kb = fwd ? p11 : p10;
                      p00 = fwd ? ka : f2{0.f, 0.f}; p10 = fwd ? kb : f2{0.f, 0.f};
                      p01 = fwd ? f2{0.f, 0.f} : ka; p11 = fwd ? f2{0.f, 0.f} : kb;
                      lp = 1; lq = WSY; la = WSZ; gp = 1; gq = V.sy; ga = V.sz;
                    } else {
                      e0 = fwd ? p00.x : p00.y; e1 = fwd ? p10.x : p10.y; e2 = fwd ? p01.x : p01.y; e3 = fwd ? p11.x : p11.y;
                      p00 = fwd ? f2{p00.y, 0.f} : f2{0.f, p00.x}; p10 = fwd ? f2{p10.y, 0.f} : f2{0.f, p10.x};
                      p01 = fwd ? f2{p01.y, 0.f} : f2{0.f, p01.x}; p11 = fwd ? f2{p11.y, 0.f} : f2{0.f, p11.x};
                      lp = WSY; lq = WSZ; la = 1; gp = V.sy; gq = V.sz; ga = 1;
                    }
                    if (old_lidx >= 0) {
                      if (experiment != 3) {
                        const int qi = old_lidx + (fwd ? 0 : la);
                        // pair / quad pre-reduction (see shift_emit4): lanes of a quad that go to the same four slots
                        const int key = qi | ((ay ? 1 : (az ? 2 : 0)) << 16);
                        const int k1 = __builtin_amdgcn_update_dpp(-1, key, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
                        const int k2 = __builtin_amdgcn_update_dpp(-1, key, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
                        const int k3 = __builtin_amdgcn_update_dpp(-1, key, 0x1B, 0xF, 0xF, false);   // quad_perm [3,2,1,0]
                        const bool psame = k1 == key;
                        const bool same = psame & (k2 == key) & (k3 == key);
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
                        if (ABL && a.dbg) { ++ev_face; ev_add += add; }
                        if (experiment == 6) {            // ablation: no pre-reduction, every lane adds its own values
                          win_t* q = win + qi;
                          atomicAdd(q, (win_t)e0); atomicAdd(q + lp, (win_t)e1); atomicAdd(q + lq, (win_t)e2); atomicAdd(q + lq + lp, (win_t)e3);
                        } else if (add) {
                          win_t* q = win + qi;
                          atomicAdd(q, (win_t)(same ? s0 : (psame ? q0 : e0)));      atomicAdd(q + lp, (win_t)(same ? s1 : (psame ? q1 : e1)));
                          atomicAdd(q + lq, (win_t)(same ? s2 : (psame ? q2 : e2))); atomicAdd(q + lq + lp, (win_t)(same ? s3 : (psame ? q3 : e3)));
                        }
                      }
                      used_lds = experiment != 5;         // ablation 5: never flush (until the end)
                    } else if (experiment != 2) {
                      if (ABL && a.dbg) ++ev_glob;
                      float* g = a.grad + old_base + (fwd ? 0 : ga);
                      atomic_add_f32(g, e0); atomic_add_f32(g + gp, e1); atomic_add_f32(g + gq, e2); atomic_add_f32(g + gq + gp, e3);
                    }
                  } else {
                    // jump over more than one face, or into a clamped cell: hand over all eight
                    if (ABL && a.dbg) ++ev_all8;
                    // (no quad pre-reduction here: two-face crossings are rarely shared by a quad -- 4.87 -> 4.81 ms without it)
                    if (experiment != 1) used_lds = flat_emit8<false>(win, WSY, WSZ, a.grad, V.sy, V.sz, old_lidx, old_base, p00, p10, p01, p11);
                    p00 = p10 = p01 = p11 = f2{0.f, 0.f};
                  }
                }
                lidx = regular ? WIN_INDEX(ix, iy, iz) : -1;
                miss = regular & (lidx < 0);
              }
            }
          }
        }
      }
      dirty = dirty | (__ballot(used_lds) != 0ull);
      if (ABL && a.dbg) ev_wsteps += lane == 0;
    }
  // rays still marching when max_steps ran out keep what their cell has accumulated: hand it over
  if (s.active && regular && experiment != 1) { if (flat_emit8(win, WSY, WSZ, a.grad, V.sy, V.sz, lidx, base, p00, p10, p01, p11)) dirty = true; }
#undef WSY
#undef WSZ
#undef WIN_INDEX
  dirty = __ballot(dirty) != 0ull;
  if (dirty) {
    win_flush(win, W.ox, W.oy, W.oz, a.grad, V, lane, experiment == 2);
    ++n_flush;
  }
  if (ABL && a.dbg) {
    if (lane == 0) atomicAdd(&a.dbg[0], (unsigned long long)n_flush);
    if (ev_face) atomicAdd(&a.dbg[4], (unsigned long long)ev_face);
    if (ev_add) atomicAdd(&a.dbg[5], (unsigned long long)ev_add);
    if (ev_glob) atomicAdd(&a.dbg[6], (unsigned long long)ev_glob);
    if (ev_all8) atomicAdd(&a.dbg[7], (unsigned long long)ev_all8);
    if (ev_wsteps) atomicAdd(&a.dbg[8], (unsigned long long)ev_wsteps);
    if (ev_multi) atomicAdd(&a.dbg[9], (unsigned long long)ev_multi);
  }
  block_stats<kAdjBlock>(a.stats, steps, 0u);
}

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
constexpr int kRingCap = DRRT_RING_CAP;
struct Ring {                      // wave-uniform
  int nx, ny, nz;                  // slots per axis (>= 2)
  int sy, sz;                      // LDS strides of y and z in slots: nx, nx * ny
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
template <int A>
__device__ __forceinline__ void ring_flush(win_t* win, const Ring& R, int g0, int k, float* __restrict__ grad, const Vol& V,
                                           int lane, bool no_global) {
  wave_lds_fence();
  const int e0 = A == 0 ? k : R.nx, e1 = A == 1 ? k : R.ny, e2 = A == 2 ? k : R.nz;
  const int e01 = e0 * e1, total = e01 * e2;
  const float inv0 = 1.0f / (float)e0, inv01 = 1.0f / (float)e01;
  const int r0 = g0 - (A == 0 ? R.ox : (A == 1 ? R.oy : R.oz));       // first layer, relative to the low corner
  constexpr int kBatch = DRRT_RING_FLUSH_BATCH;
#pragma unroll 1
  for (int e_base = 0; e_base < total; e_base += kWave * kBatch) {
    win_t v[kBatch];
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
      v[b] = (win_t)0;
      g[b] = (unsigned)gz * (unsigned)V.sz + (unsigned)gy * (unsigned)V.sy + (unsigned)gx;
      // ds_wrxchg_rtn_b64: read the accumulated value and reset the slot in one LDS op
      if (e < total) v[b] = __hip_atomic_exchange(win + (sz * R.sz + sy * R.sy + sx), (win_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b)
      if (v[b] != (win_t)0 && !no_global) atomic_add_f32(grad + g[b], (float)v[b]);
  }
  wave_lds_fence();
}

// One lane crosses ONE face along axis A (storage coordinate sA, size nA, LDS stride SA; in-face axes P, Q) from the cell
// whose slot is `cur`: the four corners left behind (e0..e3 in (p, q) order) go to the window -- pair / quad DPP
// pre-reduced as in k_backtrace_flat while PRE -- and (cur, sA) move to the neighbour cell, modulo the window.
// Returns true when the new cell lies outside the live region on that axis.
template <bool ABL>
__device__ __forceinline__ bool ring_cross(win_t* win, int experiment, bool pre, int axis_id, bool fwd, int& cur, int& sA, int sP,
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
        win_t* q = win + qi;
        atomicAdd(q, (win_t)(same ? s0 : (psame ? q0 : e0)));      atomicAdd(q + dP, (win_t)(same ? s1 : (psame ? q1 : e1)));
        atomicAdd(q + dQ, (win_t)(same ? s2 : (psame ? q2 : e2))); atomicAdd(q + dQ + dP, (win_t)(same ? s3 : (psame ? q3 : e3)));
      }
    } else {
      if (ABL && dbg) { ++ev_face; ++ev_add; }
      win_t* q = win + qi;
      atomicAdd(q, (win_t)e0); atomicAdd(q + dP, (win_t)e1); atomicAdd(q + dQ, (win_t)e2); atomicAdd(q + dQ + dP, (win_t)e3);
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

template <bool ABL, bool PAIR, int MODE = 0>
__global__ void __launch_bounds__(kAdjBlock, DRRT_RING_WAVES) k_backtrace_ring(BackArgs a) {
  if (a.select != nullptr) {                                 // launched next to k_backtrace_flat: the bundle
    const bool want_fit = bundles_want_ring(a.select);                                // classification picks one of the two
    if (!want_fit) return;
  }
  __shared__ win_t s_win[kAdjWavesPerBlock][kRingCap];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  win_t* win = s_win[wid];
  for (int k = lane; k < kRingCap; k += kWave) win[k] = (win_t)0;
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
  R.nx = R.ny = R.nz = 3; R.sy = 3; R.sz = 9; R.ox = R.oy = R.oz = -(1 << 28); R.bx = R.by = R.bz = 0;
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
  bool sparse = DRRT_RING_SIMPLE != 0;
  unsigned ev_face = 0, ev_add = 0, ev_glob = 0, ev_all8 = 0, ev_wsteps = 0, ev_multi = 0;
  unsigned ev_nofit = 0, ev_service = 0, ev_left = 0, ev_vol = 0, ev_all8g = 0, ev_nopre = 0;   // debug: see the end of the kernel

  // all 8 accumulated corners of the regular cell (window slot li with storage coordinates (csx, csy, csz), or straight to the grid)
  auto emit8 = [&](int li, int cbase, int csx, int csy, int csz) -> bool {
    if (li >= 0) {
      if (experiment != 3) {
        const int dX = csx == R.nx - 1 ? -(R.nx - 1) : 1;
        const int dY = csy == R.ny - 1 ? -(R.ny - 1) * R.sy : R.sy;
        const int dZ = csz == R.nz - 1 ? -(R.nz - 1) * R.sz : R.sz;
        win_t* q = win + li;
        atomicAdd(q, (win_t)p00.x);             atomicAdd(q + dX, (win_t)p00.y);
        atomicAdd(q + dY, (win_t)p10.x);        atomicAdd(q + dY + dX, (win_t)p10.y);
        atomicAdd(q + dZ, (win_t)p01.x);        atomicAdd(q + dZ + dX, (win_t)p01.y);
        atomicAdd(q + dZ + dY, (win_t)p11.x);   atomicAdd(q + dZ + dY + dX, (win_t)p11.y);
      }
      return true;
    }
    if (experiment != 2) {
      float* g = a.grad + cbase;
      atomic_add_f32(g, p00.x);                atomic_add_f32(g + 1, p00.y);
      atomic_add_f32(g + V.sy, p10.x);         atomic_add_f32(g + V.sy + 1, p10.y);
      atomic_add_f32(g + V.sz, p01.x);         atomic_add_f32(g + V.sz + 1, p01.y);
      atomic_add_f32(g + V.sz + V.sy, p11.x);  atomic_add_f32(g + V.sz + V.sy + 1, p11.y);
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
          if (regular && experiment != 1) { if (emit8(lidx, base, sx, sy, sz)) dirty = true; }
          s.active = false;
        }
      }
    }
    // ---- lanes ahead of (or beside) the window: let it follow them (wave-uniform branch) ----
    const unsigned long long mm = __ballot(s.active & miss);
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
              if (hi) { ring_flush<0>(win, R, R.ox, k, a.grad, V, lane, experiment == 2); R.ox += k; R.bx += k; R.bx = R.bx >= R.nx ? R.bx - R.nx : R.bx; }
              else    { ring_flush<0>(win, R, R.ox + R.nx - k, k, a.grad, V, lane, experiment == 2); R.ox -= k; R.bx -= k; R.bx = R.bx < 0 ? R.bx + R.nx : R.bx; }
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
              if (hi) { ring_flush<1>(win, R, R.oy, k, a.grad, V, lane, experiment == 2); R.oy += k; R.by += k; R.by = R.by >= R.ny ? R.by - R.ny : R.by; }
              else    { ring_flush<1>(win, R, R.oy + R.ny - k, k, a.grad, V, lane, experiment == 2); R.oy -= k; R.by -= k; R.by = R.by < 0 ? R.by + R.ny : R.by; }
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
              if (hi) { ring_flush<2>(win, R, R.oz, k, a.grad, V, lane, experiment == 2); R.oz += k; R.bz += k; R.bz = R.bz >= R.nz ? R.bz - R.nz : R.bz; }
              else    { ring_flush<2>(win, R, R.oz + R.nz - k, k, a.grad, V, lane, experiment == 2); R.oz -= k; R.bz -= k; R.bz = R.bz < 0 ? R.bz + R.nz : R.bz; }
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
          if (dirty) { ring_flush<2>(win, R, R.oz, R.nz, a.grad, V, lane, experiment == 2); dirty = false; ++n_flush; }
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
          R.nx = nx; R.ny = ny; R.nz = nz; R.sy = nx; R.sz = nx * ny;
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
    if (DRRT_RING_SIMPLE == 0 && (it & 15) == 15) {          // a sample of one iteration in 16 (scalar arithmetic only)
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
        if (regular && experiment != 1) used_lds = emit8(lidx, base, sx, sy, sz);
      } else {
        ++steps;
        const float dn = dot3(s.mx, s.my, s.mz, m.gx, m.gy, m.gz);                            // :430
        const float nds = (m.n * a.ds) * a.grad_scale;
        if (regular) {
          const CornerPairs cp = splat_weights_pk(wx, wy, wz, dn * a.ds, nds * s.mx, nds * s.my, nds * s.mz);   // :431-432
          p00 += cp.c00; p10 += cp.c10; p01 += cp.c01; p11 += cp.c11;
        } else if (experiment != 2 && experiment != 1) {
          const Cell cb = locate(V, px, py, pz);
          const Corners w = splat_weights(cb.wx, cb.wy, cb.wz, dn * a.ds, nds * s.mx, nds * s.my, nds * s.mz);
          float* g = a.grad + cb.base;
          atomic_add_f32(g, w.c000);                     atomic_add_f32(g + cb.ox, w.c100);
          atomic_add_f32(g + cb.oy, w.c010);             atomic_add_f32(g + cb.oy + cb.ox, w.c110);
          atomic_add_f32(g + cb.oz, w.c001);             atomic_add_f32(g + cb.oz + cb.ox, w.c101);
          atomic_add_f32(g + cb.oz + cb.oy, w.c011);     atomic_add_f32(g + cb.oz + cb.oy + cb.ox, w.c111);
        }
        const int old_base = base, old_lidx = lidx;
        const bool old_regular = regular;
        int nbase; bool nregular;
        int dpack = -1;                                    // the move in cells, one VGPR: (ddx + 1) | (ddy + 1) << 2 | (ddz + 1) << 4, or -1
        if (sparse) {                                      // (wave-uniform) a sparse bundle does not look at the move
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
        // ---- the ray leaves its cell ----
        if (nbase != old_base || !interior) {
          base = nbase; regular = nregular;
          if (nbase != old_base || regular != old_regular) {
            bool relocate = true;                        // the new cell still has to be placed in the window
            if (old_regular) {
              const bool unit = dpack >= 0;
              const int ddx = (dpack & 3) - 1, ddy = ((dpack >> 2) & 3) - 1, ddz = ((dpack >> 4) & 3) - 1;
              if (!sparse && (regular & unit & (old_lidx >= 0) & (experiment != 1) & (experiment != 4))) {
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
                  out |= ring_cross<ABL>(win, experiment, pre, 0, fwd, cur, sx, sy, sz, R.nx, R.ny, R.nz, 1, R.sy, R.sz, ix, R.ox,
                                         e0, e1, e2, e3, matched, ev_face, ev_add, dbg);
                }
                if (ddy != 0) {
                  const bool fwd = ddy > 0;
                  const f2 ea = fwd ? p00 : p10, eb = fwd ? p01 : p11;
                  const f2 ka = fwd ? p10 : p00, kb = fwd ? p11 : p01;
                  p00 = fwd ? ka : f2{0.f, 0.f}; p01 = fwd ? kb : f2{0.f, 0.f};
                  p10 = fwd ? f2{0.f, 0.f} : ka; p11 = fwd ? f2{0.f, 0.f} : kb;
                  out |= ring_cross<ABL>(win, experiment, pre, 1, fwd, cur, sy, sx, sz, R.ny, R.nx, R.nz, R.sy, 1, R.sz, iy, R.oy,
                                         ea.x, ea.y, eb.x, eb.y, matched, ev_face, ev_add, dbg);
                }
                if (ddz != 0) {
                  const bool fwd = ddz > 0;
                  const f2 ea = fwd ? p00 : p01, eb = fwd ? p10 : p11;
                  const f2 ka = fwd ? p01 : p00, kb = fwd ? p11 : p10;
                  p00 = fwd ? ka : f2{0.f, 0.f}; p10 = fwd ? kb : f2{0.f, 0.f};
                  p01 = fwd ? f2{0.f, 0.f} : ka; p11 = fwd ? f2{0.f, 0.f} : kb;
                  out |= ring_cross<ABL>(win, experiment, pre, 2, fwd, cur, sz, sx, sy, R.nz, R.nx, R.ny, R.sz, 1, R.sy, iz, R.oz,
                                         ea.x, ea.y, eb.x, eb.y, matched, ev_face, ev_add, dbg);
                }
                used_lds = true;
                lidx = out ? -1 : cur;                   // stepped ahead of the window by one cell: ask it to follow
                miss = out;
                relocate = false;
              } else {
                // out of a cell the window does not hold, into or out of a clamped cell, a jump over more than one cell: all eight
                if (ABL && dbg) { ++ev_all8; ev_all8g += old_lidx < 0; }
                if (experiment != 1) used_lds = emit8(old_lidx, old_base, sx, sy, sz);
                p00 = p10 = p01 = p11 = f2{0.f, 0.f};
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
    dirty = dirty | (__ballot(used_lds) != 0ull);
    if (ABL && dbg) ev_wsteps += lane == 0;
  }
  // rays still marching when max_steps ran out keep what their cell has accumulated: hand it over
  if (s.active && regular && experiment != 1) { if (emit8(lidx, base, sx, sy, sz)) dirty = true; }
  dirty = __ballot(dirty) != 0ull;
  if (dirty) { ring_flush<2>(win, R, R.oz, R.nz, a.grad, V, lane, experiment == 2); ++n_flush; }
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

}  // namespace drrt

// =============================================================================================
// host side: C ABI
// =============================================================================================
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

static inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

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
  if (!(flags & DRRT_FLAG_LDS_BRICKS)) { rc = maybe_pair(a.vol, nvox, n, flags, ws, ws_bytes, s); if (rc) return rc; }
  if (MODE == 2) {                      // n flag bytes, in the slack the workspace keeps after the sort buffers
    const size_t off = (flags & DRRT_FLAG_SORT_RAYS) ? align_up(sort_workspace_bytes(n), 256) : 0;
    if (!ws || ws_bytes < off + n) return fail(DRRT_ERR_ARG, "workspace too small for trace_sdf (see drrt_workspace_bytes)");
    a.again = (uint8_t*)ws + off;
  }
  g_last_steps = nullptr; g_last_steps_n = 0;
  if (MODE != 2 && !(flags & DRRT_FLAG_LDS_BRICKS) && !(flags & DRRT_FLAG_TAP_REUSE_MASK) && !(flags & DRRT_FLAG_LEGACY_FORWARD)) {
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
    if (MODE == 2 || !(flags & DRRT_FLAG_LDS_BRICKS)) {
      const unsigned reuse = (flags & DRRT_FLAG_TAP_REUSE_MASK);
      if ((MODE == 0 || MODE == 1) && reuse == 0 && !(flags & DRRT_FLAG_LEGACY_FORWARD)) {
        constexpr int FM = MODE == 1 ? 1 : 0;
        if (a.vol.pair != nullptr) hipLaunchKernelGGL((k_trace_flat<true, FM>), dim3(grid_for(n)), dim3(kBlock), 0, s, a);
        else                       hipLaunchKernelGGL((k_trace_flat<false, FM>), dim3(grid_for(n)), dim3(kBlock), 0, s, a);
      }
      else if (reuse == DRRT_FLAG_TAP_REUSE_OFF)
        hipLaunchKernelGGL((k_trace<MODE, 0>), dim3(grid_for(n)), dim3(kBlock), 0, s, a);
      else if (reuse == DRRT_FLAG_TAP_REUSE_FACE)
        hipLaunchKernelGGL((k_trace<MODE, 2>), dim3(grid_for(n)), dim3(kBlock), 0, s, a);
      else
        hipLaunchKernelGGL((k_trace<MODE, 1>), dim3(grid_for(n)), dim3(kBlock), 0, s, a);
    }
    else
      hipLaunchKernelGGL(k_trace_win<(MODE == 2 ? 0 : MODE)>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  }
  LAUNCH_CHECK("k_trace");
  if (MODE == 1 || MODE == 2) {
    hipLaunchKernelGGL(k_trace_again<(MODE == 2 ? 2 : 1)>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
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
  if (flags & DRRT_FLAG_LEGACY_FORWARD) hipLaunchKernelGGL(k_target_a, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  else if (a.vol.pair != nullptr)       hipLaunchKernelGGL(k_target_a_flat<true>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  else                                  hipLaunchKernelGGL(k_target_a_flat<false>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  LAUNCH_CHECK("k_target_a");
  hipLaunchKernelGGL(k_target_b, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
  LAUNCH_CHECK("k_target_b");
  return DRRT_OK;
}

template <int MODE>
static int run_backtrace(const float* rif, const float* sdf, long long nvox, const int res[3], size_t n,
                         const void* xt, const void* vt, const void* dx, const void* dv,
                         float h, float ds, float* grad, drrt_stats* stats, void* ws, size_t ws_bytes,
                         unsigned flags, void* stream, int io_half = 0) {
  const OrderHint hint = take_hint();
  g_last_counters = nullptr;              // set again below when this call classifies its bundles
  g_err[0] = 0;
  hipStream_t s = (hipStream_t)stream;
  BackArgs a{};
  int rc = make_vol(rif, nvox, res, h, &a.vol); if (rc) return rc;
  rc = check_steps(h, ds); if (rc) return rc;
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
  rc = maybe_sort(a.vol, h, n, xt, vt, -1.f, flags, ws, ws_bytes, &a.perm, s, hint, io_half); if (rc) return rc;
  rc = maybe_pair(a.vol, nvox, n, flags, ws, ws_bytes, s); if (rc) return rc;
  a.io_half = io_half;
  a.sdf = sdf; a.xt = xt; a.vt = vt; a.dx = dx; a.dv = dv; a.grad = grad; a.stats = stats;
  a.n = n; a.ds = ds; a.max_steps = steps_adj(h, res, ds);
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
      hipLaunchKernelGGL(k_backtrace_direct<MODE>, dim3(grid_for(n)), dim3(kBlock), 0, s, a);
    else if (!(flags & DRRT_FLAG_LEGACY_ADJOINT)) {
      const bool abl = a.experiment != 0 || a.dbg != nullptr, pair = a.vol.pair != nullptr;
      const dim3 g(adj_grid_for(n));
      // Two kernels: k_backtrace_flat with its compile-time 9^3 box window for compact bundles, k_backtrace_ring (fitted ring
      // window, step hint) for the rest.  With a visit order the bundles are classified on the device and BOTH are launched;
      // the one the counters do not pick returns at once (no host round trip).  Needs the 512-byte counter block at the end
      // of a drrt_workspace_bytes_grid() workspace; without it, or without an order, the box-window kernel runs.
      // DRRT_FLAG_STATIC_WINDOW / DRRT_FLAG_RING_WINDOW force one of the two (A-B).
      a.select = nullptr;
      const size_t ctr_off = (drrt_workspace_bytes(n, flags) + ((flags & DRRT_FLAG_PAIR_GRID) ? (size_t)nvox * 2 * sizeof(float) : 0) + 7) & ~(size_t)7;
      const bool force_box = (flags & DRRT_FLAG_STATIC_WINDOW) != 0 || a.experiment == 7, force_ring = (flags & DRRT_FLAG_RING_WINDOW) != 0;
      if (!force_box && !force_ring && a.perm != nullptr && ws && ws_bytes >= ctr_off + 512) {
        a.select = (unsigned*)((char*)ws + ctr_off + 256);
        g_last_counters = a.select;
        hipError_t e = hipMemsetAsync(a.select, 0, 16, s);
        if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(select)");
        hipLaunchKernelGGL(k_bundle_classify, dim3((grid_for(n) + kClassifyStride - 1) / kClassifyStride), dim3(kBlock), 0, s, a);
      }
      if (!force_ring) {
        if (abl && MODE == 0) {
          if (pair) hipLaunchKernelGGL((k_backtrace_flat<true, true, 0>), g, dim3(kAdjBlock), 0, s, a);
          else      hipLaunchKernelGGL((k_backtrace_flat<true, false, 0>), g, dim3(kAdjBlock), 0, s, a);
        } else {    /* the ablation / counter instantiation exists for backtrace only */
          if (pair) hipLaunchKernelGGL((k_backtrace_flat<false, true, MODE>), g, dim3(kAdjBlock), 0, s, a);
          else      hipLaunchKernelGGL((k_backtrace_flat<false, false, MODE>), g, dim3(kAdjBlock), 0, s, a);
        }
      }
      if (force_ring || a.select != nullptr) {
        if (abl && MODE == 0) {
          if (pair) hipLaunchKernelGGL((k_backtrace_ring<true, true, 0>), g, dim3(kAdjBlock), 0, s, a);
          else      hipLaunchKernelGGL((k_backtrace_ring<true, false, 0>), g, dim3(kAdjBlock), 0, s, a);
        } else {
          if (pair) hipLaunchKernelGGL((k_backtrace_ring<false, true, MODE>), g, dim3(kAdjBlock), 0, s, a);
          else      hipLaunchKernelGGL((k_backtrace_ring<false, false, MODE>), g, dim3(kAdjBlock), 0, s, a);
        }
      }
    }
    else if (a.experiment != 0 || a.dbg != nullptr)
      hipLaunchKernelGGL((k_backtrace_win<MODE, true, true>), dim3(grid_for(n)), dim3(kBlock), 0, s, a);
    else if (flags & DRRT_FLAG_NO_PIPELINE)
      hipLaunchKernelGGL((k_backtrace_win<MODE, false, false>), dim3(grid_for(n)), dim3(kBlock), 0, s, a);
    else
      hipLaunchKernelGGL((k_backtrace_win<MODE, false, true>), dim3(grid_for(n)), dim3(kBlock), 0, s, a);
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
  size_t lds = (a.rres <= kCableMaxRes) ? a.rres * (sizeof(double) + sizeof(float)) : 0;
  hipLaunchKernelGGL(k_backtrace_cable, dim3(cable_grid(n)), dim3(kBlock), lds, s, a);
  LAUNCH_CHECK("k_backtrace_cable");
  return DRRT_OK;
}
