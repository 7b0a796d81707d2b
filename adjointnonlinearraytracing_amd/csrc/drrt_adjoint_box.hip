// drrt_adjoint_box.hip -- gfx950 kernels of the adjoint march Tracer::backtrace / backtrace_sdf
// (/root/reference/src/tracer.cpp:384-509) for compact ray bundles: k_backtrace_flat with its compile-time box window,
// the bundle classification that chooses between it and the ring-window kernel (drrt_adjoint_ring.hip), and the
// one-atomic-per-tap kernel every windowed variant is cross-checked against.
#include "drrt_march.h"

namespace drrt {

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
// adjoint march with per-wave LDS gradient windows
//
// Why: one global fp32 atomic per tap runs at the memory-side atomic rate, and that rate collapses
// when many lanes hit the same few addresses (measured on MI355X, Luneburg 256^3 / 1M rays: 237 ms
// for 4.1e9 lane-atomics -- every ray passes through the handful of voxels around the focus).
// Rays of a wave are spatially coherent (locality sort), so each wave keeps a small box of
// gradient voxels ("window") in LDS and accumulates there (ds_add_f64, fed by per-lane register
// accumulators that emit one cell face at a time, see below); the window is flushed to the global
// grid -- one fp32 atomic per touched voxel, contiguous in x -- only when the rays walk out of it.  Lanes whose cell
// falls outside the window (incoherent wave, clamped boundary cell) fall back to direct global atomics, so the result
// never depends on the window.
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


// ---------------------------------------------------------------------------------------------
// k_backtrace_flat: the adjoint march organised around ONE invariant -- the lane's register accumulators always belong to
// the cell the ray stands on -- so that the hot loop carries no second cell, no "accumulator cell" bookkeeping and no
// per-step window test:
//
//   top      taps(x_k) arrive (gather issued one iteration earlier) -> n, grad n, H -> v_k -> still active?
//            8 splat weights of step k (they need the in-cell fractions of cell k) -> accumulators += weights
//   step     x_{k+1} = x_k - ds v_k ; locate its cell IN PLACE (the fractions of cell k are dead by now) ;
//            issue its gather unless the ray stays in its cell
//   then     lambda / mu recurrences (under the gather)
//   leave    only if the cell changed: the ray LEAVES cell k -- a move across one face hands the four corners left
//            behind to the LDS window (pair / quad DPP pre-reduced) and carries the shared four; anything else hands
//            over all eight -- and the window index of the new cell is computed once, here (a miss votes for a
//            re-anchor).  A ray that ends hands over all eight.
// Per-ray arithmetic is adj_sample / adj_contrib of drrt_device.h (the two halves of adj_step, which the one-atomic-per-tap
// kernel above runs as is): bit-identical contributions, only the order in which they reach the grid differs, i.e. the
// usual fp32 summation-order noise.  (Rounds 1-2 had a predecessor, k_backtrace_win, with a second "accumulator cell" per
// lane: 5.45 ms against 4.5 ms for this kernel on 256^3 / 1M rays; removed in round 4, NOTES.md.)
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

__global__ void __launch_bounds__(kBlock) k_bundle_classify(BackArgs a) {
  __shared__ unsigned s_cnt[7];                  // the block's counters ([5] is the host's): one set of global atomics per block, not per wave
  if (threadIdx.x < 7) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const Vol& V = a.vol;
  const size_t t = (size_t)blockIdx.x * kClassifyStride * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  size_t i;
  bool ok = false;
  int cx = 0, cy = 0, cz = 0, cbase = -1;
  unsigned fk = 0u;                               // the forward march's iteration count of the ray (step hint), 0 without
  if (ray_index(a.perm, t, a.n, i)) {
    const Ray3 p = ld3(a.xt, i, a.io_half, &a.vol, RAY_POS);
    const Cell c = locate(V, p.x, p.y, p.z);
    cx = c.ix; cy = c.iy; cz = c.iz; cbase = c.base; ok = true;
    if (a.fsteps != nullptr) fk = a.fsteps[i];
  }
  const int pbase = __builtin_amdgcn_update_dpp(-2, cbase, 0xB1, 0xF, 0xF, false);      // the pair partner's cell (quad_perm [1,0,3,2])
  const unsigned paired = (unsigned)__popcll(__ballot(ok & (pbase == cbase)));
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
    if (paired) atomicAdd(&s_cnt[4], paired);
  }
  // by how many iterations the bundle's rays left the forward march apart -- as a length: iterations * ds, in cells
  const int fki = (int)min(fk, 1u << 30);
  const int kspread = wave_max_i32(ok ? fki : 0) - wave_min_i32(ok ? fki : big);
  const bool long_ = (float)kspread * a.ds * V.inv_h >= kClassifyLongCells;
  if (lane == 0 && lanes != 0u && a.fsteps != nullptr && long_) atomicAdd(&s_cnt[6], 1u);
  __syncthreads();
  if (threadIdx.x < 7 && s_cnt[threadIdx.x] != 0u) atomicAdd(&a.select[threadIdx.x], s_cnt[threadIdx.x]);
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
// CHUNK: the resumable instantiation behind drrt_backtrace_chunk_f32 (depth-chunked adjoint for the multi-GPU slab reduce,
//       DESIGN.md section 7): the launch marches a.max_steps iterations of every ray, starting from the exit rays or from
//       the state the previous chunk left in a.chunk_state (12 floats + a flag word per visit slot, SoA), hands every
//       accumulator over and flushes every window at its end, saves the state, and reduces the bounding box of the positions
//       and velocities of the rays that are still marching into a.chunk_progress.  Same arithmetic, same contributions;
//       only the order in which they reach the grid differs from the one-launch march.
template <bool ABL, bool PAIR, int MODE = 0, bool CHUNK = false>
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
  const bool resume = CHUNK && a.chunk_resume != 0;
  if (resume) {                                                // continue where the previous chunk stopped
    const size_t S = a.chunk_stride;
    const float* cs = a.chunk_state + t;
    const unsigned fl = __float_as_uint(cs[12 * S]);
    s.x = cs[0]; s.y = cs[S]; s.z = cs[2 * S]; s.vx = cs[3 * S]; s.vy = cs[4 * S]; s.vz = cs[5 * S];
    s.lx = cs[6 * S]; s.ly = cs[7 * S]; s.lz = cs[8 * S]; s.mx = cs[9 * S]; s.my = cs[10 * S]; s.mz = cs[11 * S];
    s.active = (fl & 1u) != 0u; s.outside = (fl & 2u) != 0u;
  } else if (ray_index(a.perm, t, a.n, i)) {
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
  auto step_locate = [&](int& nbase, bool& nregular, bool move = true) {      // move = false: locate only (a resumed ray)
    if (move) { s.x = fmaf(-a.ds, s.vx, s.x); s.y = fmaf(-a.ds, s.vy, s.y); s.z = fmaf(-a.ds, s.vz, s.z); }
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
    if (resume) step_locate(nbase, nregular, false);         // the saved position IS the next sample
    else        step_locate(nbase, nregular);                // first sample
    base = nbase; regular = nregular;
    miss = regular;                                          // no window yet
  }
  bool dirty = false;
  int cooldown = 0;
  unsigned steps = 0;
  unsigned n_flush = 0;
  // CHUNK: bounding box of the samples this launch contributes at (which voxels it can have touched)
  float bx0 = 3.0e38f, by0 = 3.0e38f, bz0 = 3.0e38f, bx1 = -3.0e38f, by1 = -3.0e38f, bz1 = -3.0e38f;
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
          if (CHUNK) { bx0 = fminf(bx0, px); by0 = fminf(by0, py); bz0 = fminf(bz0, pz); bx1 = fmaxf(bx1, px); by1 = fmaxf(by1, py); bz1 = fmaxf(bz1, pz); }
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
  if (CHUNK) {
    // the state the next chunk resumes from (position = the next sample, already stepped to), and where the rays that are
    // still marching stand and head: bounding boxes of their positions [0..5] and velocities [6..11] as order-preserving
    // integer keys (min: 0..2 / 6..8, max: 3..5 / 9..11), [12] their number; [13..18] bounding box of the samples taken
    const size_t S = a.chunk_stride;
    float* cs = a.chunk_state + t;
    cs[0] = s.x; cs[S] = s.y; cs[2 * S] = s.z; cs[3 * S] = s.vx; cs[4 * S] = s.vy; cs[5 * S] = s.vz;
    cs[6 * S] = s.lx; cs[7 * S] = s.ly; cs[8 * S] = s.lz; cs[9 * S] = s.mx; cs[10 * S] = s.my; cs[11 * S] = s.mz;
    cs[12 * S] = __uint_as_float((s.active ? 1u : 0u) | (s.outside ? 2u : 0u));
    if (a.chunk_progress != nullptr && __any(s.active)) {
      const float vals[6] = {s.x, s.y, s.z, s.vx, s.vy, s.vz};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int key = float_order_key(vals[k]);
        const int lo = wave_min_i32(s.active ? key : 0x7fffffff), hi = wave_max_i32(s.active ? key : (int)0x80000000);
        if (lane == 0) { atomicMin(&a.chunk_progress[(k / 3) * 6 + (k % 3)], lo); atomicMax(&a.chunk_progress[(k / 3) * 6 + 3 + (k % 3)], hi); }
      }
      const unsigned cnt = (unsigned)__popcll(__ballot(s.active));
      if (lane == 0) atomicAdd((unsigned*)&a.chunk_progress[12], cnt);
    }
    if (a.chunk_progress != nullptr && __any(steps != 0u)) {       // [13..15] min, [16..18] max of the samples of THIS chunk
      const float lo[3] = {bx0, by0, bz0}, hi[3] = {bx1, by1, bz1};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int l = wave_min_i32(float_order_key(lo[k])), h = wave_max_i32(float_order_key(hi[k]));
        if (lane == 0) { atomicMin(&a.chunk_progress[13 + k], l); atomicMax(&a.chunk_progress[16 + k], h); }
      }
    }
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

// ---- launchers ----------------------------------------------------------------------------------
void launch_backtrace_direct(int mode, const BackArgs& a, hipStream_t s) {
  const dim3 g(grid_for(a.n)), b(kBlock);
  if (mode == 1) hipLaunchKernelGGL(k_backtrace_direct<1>, g, b, 0, s, a);
  else           hipLaunchKernelGGL(k_backtrace_direct<0>, g, b, 0, s, a);
}
void launch_bundle_classify(const BackArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_bundle_classify, dim3((grid_for(a.n) + kClassifyStride - 1) / kClassifyStride), dim3(kBlock), 0, s, a);
}
void launch_backtrace_box(int mode, bool abl, const BackArgs& a, hipStream_t s) {
  const dim3 g(adj_grid_for(a.n)), b(kAdjBlock);
  const bool pair = a.vol.pair != nullptr;
  if (a.chunk_state != nullptr) {      /* resumable march (drrt_backtrace_chunk_f32): backtrace only */
    if (pair) hipLaunchKernelGGL((k_backtrace_flat<false, true, 0, true>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_flat<false, false, 0, true>), g, b, 0, s, a);
    return;
  }
  if (mode == 1) {          /* the ablation / counter instantiation exists for backtrace only */
    if (pair) hipLaunchKernelGGL((k_backtrace_flat<false, true, 1>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_flat<false, false, 1>), g, b, 0, s, a);
  } else if (abl) {
    if (pair) hipLaunchKernelGGL((k_backtrace_flat<true, true, 0>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_flat<true, false, 0>), g, b, 0, s, a);
  } else {
    if (pair) hipLaunchKernelGGL((k_backtrace_flat<false, true, 0>), g, b, 0, s, a);
    else      hipLaunchKernelGGL((k_backtrace_flat<false, false, 0>), g, b, 0, s, a);
  }
}

}  // namespace drrt
