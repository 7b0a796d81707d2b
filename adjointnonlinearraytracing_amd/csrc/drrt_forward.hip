// drrt_forward.hip -- gfx950 kernels of the forward eikonal march: Tracer::trace, trace_plane, trace_sdf, trace_target
// (/root/reference/src/tracer.cpp:35-310).  Shared pieces: drrt_march.h; per-ray arithmetic: drrt_device.h.
#include "drrt_march.h"

namespace drrt {

// ---------------------------------------------------------------------------------------------
// generic per-ray march (trace_ray of drrt_device.h): the kernel of trace_sdf (MODE 2)
// ---------------------------------------------------------------------------------------------
template <int MODE, int REUSE = kTapReuse>     // instantiated for MODE 2 (trace_sdf) only: trace / trace_plane run k_trace_flat
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
// Bit-identical to the generic per-ray march trace_ray<0> of drrt_device.h (tests/test_hostcheck.py, tests/test_gpu_parity.py).
//
// What bounds it (tools/chain_bench.hip, PMC): a wave issues a gather on nearly every step -- some lane always changes
// cell -- and a divergent 64-lane gather instruction occupies the CU's texture addresser for ~30-37 cycles whether one
// lane is active or all 64.  So the taps a lane keeps across steps save cache traffic but hardly any addresser time;
// what does is FEWER GATHER INSTRUCTIONS: with the pair copy of the grid (DRRT_FLAG_PAIR_GRID, template PAIR) a cell
// is two 16-byte gathers instead of four 8-byte ones (1.31 -> 1.06 ms on 256^3 / 1M rays).  Tried and dropped: two rays
// per lane (two gathers in flight per wave: 1.6-1.9 ms -- the addresser, not latency, is the limit), a branch-free
// interior path (1.38 ms), fewer VALU instructions alone (-20 %: no change).
// ---------------------------------------------------------------------------------------------
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
// reference's global loop are flagged for k_trace_again exactly as trace_ray<1> of drrt_device.h does, see plane_again).
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

// ---- launchers ----------------------------------------------------------------------------------
void launch_trace(int mode, const TraceArgs& a, hipStream_t s) {
  const dim3 g(grid_for(a.n)), b(kBlock);
  const bool pair = a.vol.pair != nullptr;
  if (mode == 2)      hipLaunchKernelGGL((k_trace<2>), g, b, 0, s, a);
  else if (mode == 1) { if (pair) hipLaunchKernelGGL((k_trace_flat<true, 1>), g, b, 0, s, a); else hipLaunchKernelGGL((k_trace_flat<false, 1>), g, b, 0, s, a); }
  else                { if (pair) hipLaunchKernelGGL((k_trace_flat<true, 0>), g, b, 0, s, a); else hipLaunchKernelGGL((k_trace_flat<false, 0>), g, b, 0, s, a); }
}
void launch_trace_again(int mode, const TraceArgs& a, hipStream_t s) {
  const dim3 g(grid_for(a.n)), b(kBlock);
  if (mode == 2) hipLaunchKernelGGL(k_trace_again<2>, g, b, 0, s, a);
  else           hipLaunchKernelGGL(k_trace_again<1>, g, b, 0, s, a);
}
void launch_target(const TargetArgs& a, hipStream_t s) {
  const dim3 g(grid_for(a.n)), b(kBlock);
  if (a.vol.pair != nullptr) hipLaunchKernelGGL(k_target_a_flat<true>, g, b, 0, s, a);
  else                       hipLaunchKernelGGL(k_target_a_flat<false>, g, b, 0, s, a);
  hipLaunchKernelGGL(k_target_b, g, b, 0, s, a);
}

}  // namespace drrt
