/*
 * drrt_hip.h -- C ABI of the MI355X-native eikonal ray-march library (libdrrt_hip.so).
 *
 * Drop-in boundary for the hot path of ArjunTeh/AdjointNonlinearRayTracing: one entry
 * point per method of the reference's `drrt.TracerC` pybind class
 * (reference: src/drrt.cpp:47-58, declarations include/tracer.h:15-89).  Plain pointers and
 * sizes only -- no torch / enoki / pybind types.
 *
 * Conventions shared by all entry points
 *   - every array pointer is a DEVICE pointer (HIP, gfx950) unless stated otherwise;
 *   - ray arrays (pos, vel, xt, vt, dx, dv, pln_o, pln_d, target) are (n,3) row-major fp32,
 *     exactly the torch tensors the reference's core/tracer.py hands to enoki
 *     (core/tracer.py:300-301); outputs have the same layout;
 *   - `rif` / `sdf` / `grad` are flat fp32[nvox], C-order flatten of the torch (D,H,W) tensor
 *     (core/tracer.py:299); `res` is a HOST pointer to 3 ints = tuple(rif.shape), used as
 *     (width,height,depth) with flat index (z*height + y)*width + x (src/volume.cpp:134-141);
 *   - `h`, `ds` are fp32 scalars (include/tracer.h:20-21 narrows python floats to float);
 *   - calls are ASYNCHRONOUS on `stream` (a hipStream_t; NULL = the null stream) and never
 *     allocate, free or synchronise; scratch memory comes from the caller-provided workspace;
 *   - return value: DRRT_OK or a negative DRRT_ERR_*; drrt_last_error() returns the message
 *     of the last failing call on this host thread.  The three messages of the reference
 *     (src/volume.cpp:28,37,115,124) are reproduced verbatim.
 *
 * Statistics: `stats` (nullable) is a DEVICE pointer to one drrt_stats that the call zeroes and
 * fills; read it back after synchronising the stream.  `n_failed > 0` is the condition under
 * which the reference prints "failed to exit all rays" (src/tracer.cpp:89-90).
 */
#ifndef DRRT_HIP_H
#define DRRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define DRRT_API __attribute__((visibility("default")))
#else
#define DRRT_API
#endif

#define DRRT_OK                0
#define DRRT_ERR_RES_MISMATCH (-1)  /* "Resolution doesn't match data"  (src/volume.cpp:28,37)  */
#define DRRT_ERR_BAD_RES      (-2)  /* "volume: invalid resolution!"     (src/volume.cpp:124)    */
#define DRRT_ERR_ARG          (-3)  /* null pointer / workspace too small / bad flag             */
#define DRRT_ERR_HIP          (-4)  /* a HIP runtime call failed (message carries hipGetErrorString) */

/* flags (bit-or).  Bits 32u, 0x10000u..0x100000u carried A-B switches for kernels that were removed in round 4 (LDS
 * bricks, tap-reuse variants, the round-1 forward / adjoint kernels); they are ignored now. */
#define DRRT_FLAG_NONE         0u
#define DRRT_FLAG_SORT_RAYS    1u   /* locality-sort rays by entry voxel before marching (results
                                       are written back in the caller's ray order)                */
#define DRRT_FLAG_CORRECTED_H  2u   /* adjoint only: divide the gradient-splat term by h (exact
                                       discrete adjoint).  Default = as written in the reference,
                                       which omits it (src/volume.cpp:227-243 vs :178)            */
#define DRRT_FLAG_NO_ZERO      4u   /* adjoint only: accumulate into `grad` without zeroing it
                                       first (the reference always starts from zeros,
                                       src/tracer.cpp:401-403)                                    */
#define DRRT_FLAG_DIRECT_ATOMICS 8u /* adjoint only: bypass the LDS gradient windows and issue
                                       one global atomic per tap (debug / A-B measurement)        */

#define DRRT_FLAG_PAIR_GRID    64u  /* trace / trace_pln / trace_sdf / backtrace*: build the "pair copy" of the grid in the
                                       workspace (8 B per voxel: each voxel interleaved with its +y neighbour) and
                                       fetch the 8 corners of a strictly interior cell with two 16-byte loads instead
                                       of four 8-byte ones -- the march is bound by the number of gather instructions
                                       (DESIGN.md 5.1).  Bit-identical results.  Needs a 16-byte aligned workspace of
                                       drrt_workspace_bytes_grid() bytes.  Pays off when the call does well over ~4
                                       ray-steps per voxel (the copy costs 12 B of traffic per voxel)              */
#define DRRT_FLAG_PAIR_REUSE  128u  /* with PAIR_GRID: the workspace still holds the pair copy built by the previous
                                       call (same rif contents, same n, same flags & SORT_RAYS, same workspace) --
                                       e.g. the adjoint paired with its forward: skip the rebuild                  */
#define DRRT_FLAG_QUAD_GRID  DRRT_FLAG_PAIR_GRID    /* round-1 names (the copy then held an (x,y) quad per voxel)   */
#define DRRT_FLAG_QUAD_REUSE DRRT_FLAG_PAIR_REUSE
#define DRRT_FLAG_Q16_POS_ONLY 0x200000u  /* drrt_trace_q16io / drrt_backtrace_q16io: only the POSITION arrays are q16 codes;
                                             directions and adjoint seeds are fp32 arrays (18 B per exit ray instead of 12) */
#define DRRT_FLAG_CHORD_KEY 0x800000u      /* with SORT_RAYS (A-B measurement; same results): the rounds-1/2 sort key (6-D Morton interleave of the
                                              chord end points) instead of the light-field key (direction cell + Hilbert index of the
                                              transverse offset, csrc/drrt_sort.hip) */
/* Which adjoint kernel runs (same results up to fp32 summation order).  Default: with a visit order and a
 * drrt_workspace_bytes_grid() workspace the 64-ray bundles of the call are classified on the device and either the box-window
 * kernel (compact bundles) or the ring-window kernel (sparse views, views oblique to the grid) runs; the two flags force one. */
#define DRRT_FLAG_RING_WINDOW 0x1000000u   /* backtrace, backtrace_sdf: always the ring-window kernel (k_backtrace_ring: the wave's LDS window is
                                              addressed modulo its size and follows the rays; each voxel is flushed once; honours the step hint) */
#define DRRT_FLAG_RING_SPARSE  0x4000000u  /* with RING_WINDOW (A-B): the sparse-only instantiation of the ring-window kernel (every cell
                                              leave hands over all eight corners; chosen by itself for sparse ray sets) */
#define DRRT_FLAG_RING_GENERAL 0x8000000u  /* backtrace (A-B): never the sparse-only instantiation (the per-wave dense / sparse rule only) */
#define DRRT_FLAG_RING_DIRECT  0x10000000u /* with RING_WINDOW | RING_SPARSE (A-B): the DIRECT sparse-only instantiation (every step's eight
                                              contributions straight to the window; chosen by itself when few lanes share a cell) */
#define DRRT_FLAG_STATIC_WINDOW 0x400000u  /* backtrace, backtrace_sdf: always the box-window kernel (k_backtrace_flat, compile-time 9^3 gradient
                                              windows), no per-call bundle classification */
#define DRRT_FLAG_DISPATCH_IN_ORDER 0x2000000u /* trace, trace_pln, backtrace, backtrace_sdf (A-B measurement; forward bit-identical, adjoint the same
                                         up to fp32 summation order): block b of a launch marches block b of the visit order.  Default
                                         with a visit order: each of the chip's 8 XCDs (own L2; blocks are dealt to them round-robin)
                                         takes a contiguous run of the visit order, so neighbouring bundles share one L2 */
#define DRRT_FLAG_DEBUG_COUNTERS 16u /* adjoint only (development aid): uint64 counters are
                                       written to the last 512 bytes of the workspace:
                                       [0] LDS-window flushes, [1] ray-steps accumulated through
                                       the LDS window, [2] ray-steps that fell back to global atomics
                                       ([1], [2]: unused since round 4), [3] waves of the ring-window kernel;
                                       k_backtrace_flat events: [4] one-face cell leaves handed to the
                                       window, [5] lanes of those that issued the LDS adds after the
                                       pair / quad pre-reduction, [6] one-face leaves sent to global
                                       atomics (cell outside the window), [7] leaves that handed over
                                       all eight corners, [8] wave-steps, [9] wave-steps whose one-face
                                       leaves cross two or three different axes.
                                       Bits 8..15 of `flags` select development ablations (0 = product) */

/* Limits (violations are refused with DRRT_ERR_ARG and a message, never truncated silently):
 *   grid      fewer than 2^29 voxels (2 GiB of fp32), each extent and res[0]*res[1] below 2^24
 *             (e.g. up to 812^3, or 4095 x 4095 x 32); res[0], res[1] >= 2 as in the reference (src/volume.cpp:123)
 *   rays      n < 2^32 per call
 *   h, ds     positive and finite
 * Non-finite ray components are tolerated: such rays march until max_steps and do not affect other rays'
 * forward results (their adjoint contributions are non-finite, as in any IEEE implementation).
 * Thread-safety: the visit-order hand-over (drrt_last_order / drrt_set_order_hint) and drrt_last_error() are
 * PER HOST THREAD (thread_local): a hint set on one thread is only seen by march calls of that thread.  The
 * profiling aid (drrt_profile_*) is process-global and not thread-safe.  trace_pln / trace_sdf called with
 * stats == NULL use one library-owned 24-byte stats block per device, shared by all streams of that device:
 * pass your own stats block when running them concurrently on several streams of one device.        */

typedef struct drrt_stats {
  unsigned long long ray_steps;  /* sum over rays of march iterations executed while the ray was live */
  unsigned long long n_failed;   /* rays still live after max_steps                                  */
  unsigned int       iters;      /* max over rays of iterations = the reference's global loop count  */
  unsigned int       reserved;
} drrt_stats;

/* Bytes of device scratch a call over `n` rays may need (0 when flags need none). */
DRRT_API size_t drrt_workspace_bytes(size_t n, unsigned flags);
/* Same, for a call on a grid of `nvox` voxels: adds the pair copy when DRRT_FLAG_PAIR_GRID is set and the
 * 512-byte counter block.  Layout: [sort buffers | trace_target state][pair copy][counters].          */
DRRT_API size_t drrt_workspace_bytes_grid(size_t n, long long nvox, unsigned flags);

/* Message of the last error on this thread ("" if none). */
DRRT_API const char* drrt_last_error(void);

/* Library / build identification: "drrt_hip <abi> gfx950 src:<12 hex digits>", the digest being that of the sources
 * the library was built from (csrc/Makefile); profiles/ *_pmc.json record it so that a stale profile is detectable. */
DRRT_API const char* drrt_version(void);

/* ---- visit order hand-over (optimisation hint; with a valid permutation results do not depend on it) ---
 * A sorted call (DRRT_FLAG_SORT_RAYS) leaves the permutation it used -- n uint32 ray indices, in
 * visit order -- inside the caller's workspace; drrt_last_order() returns that device pointer
 * (valid until the workspace is overwritten) and its length.
 * drrt_set_order_hint(order, n) makes the NEXT march call on this host thread visit its rays in
 * `order` instead of sorting.  That call consumes the hint as its first action -- also when it then fails
 * validation or ignores the hint -- so a hint never outlives one call.  It is ignored when its ray count
 * differs, by drrt_trace_target_f32 (whose state buffer would overlay an order living in the same workspace)
 * and by the cable calls.  `order` must stay valid and unmodified until the call's kernels have finished;
 * entries are range-checked on the device: an entry >= n leaves that slot's ray unvisited (its outputs
 * unwritten) but is never dereferenced.  With a valid permutation results do not depend on the hint.
 * Intended pairing: the adjoint of a forward march (core/tracer.py:294-335 couples them through
 * ctx.outx/ctx.outv) reuses the forward's order -- rays that entered the grid together stay
 * together through any smooth medium, which keeps each wave inside its LDS gradient window even
 * where the exit rays alone (a focus, a caustic) say nothing about the bundle they came from.   */
DRRT_API const uint32_t* drrt_last_order(size_t* n_out);
DRRT_API void drrt_set_order_hint(const uint32_t* order, size_t n);
/* Length of the hint currently armed on this host thread -- the order hint's, or the step hint's (below) when only
 * that one is armed; 0 = neither: lets a binding assert that no hint survives a call. */
DRRT_API size_t drrt_order_hint_pending(void);
/* ---- step hint (optimisation hint of the same kind; results do not depend on it) ---------------------------------
 * A forward march (drrt_trace_f32 / _f16io / _q16io / drrt_trace_pln_f32 called with a workspace) leaves the number of
 * march iterations of every ray -- n uint32, caller ray order -- in the workspace; drrt_last_steps() returns that
 * device pointer (valid until the workspace is overwritten; NULL when the last forward call on this thread wrote none).
 * drrt_set_step_hint(steps, n) hands them to the NEXT drrt_backtrace_* call on this host thread (consumed at its entry,
 * like the order hint; ignored when n differs): the rays of a wave then start the adjoint on the forward march's clock
 * (a ray that left d iterations before the last one of its wave starts d iterations later, d <= 96), so a bundle that
 * was compact in the forward march is compact in the adjoint too, whatever the exit face.  Only WHEN a lane does its
 * k-th step changes; every ray still gets its max_steps iterations (src/tracer.cpp:417).  Wrong counts cost speed only. */
DRRT_API const uint32_t* drrt_last_steps(size_t* n_out);
DRRT_API void drrt_set_step_hint(const uint32_t* steps, size_t n);

/* Which adjoint kernel ran, and why: a DEVICE pointer to the four counters of the bundle classification of the last
 * drrt_backtrace_* / drrt_backtrace_sdf_f32 call on this host thread that classified its bundles (it lies in that call's
 * workspace: valid while the workspace is, read it after synchronising the stream), or NULL when that call did not (no visit
 * order, a forced kernel, a workspace without the counter block).  Over every 16th block of 64-ray bundles:
 *   [0] bundles whose start cells do not fit the box window, [1] bundles looked at,
 *   [2] lanes whose start cell lies more than 3 cells from their bundle's mean cell, [3] lanes looked at (diagnostic),
 *   [4] lanes whose pair partner (lane ^ 1) starts in the same cell (diagnostic), [5] non-zero when the call pinned the
 *   general instantiation of the ring kernel, [6] bundles whose rays' forward iteration counts (step hint) spread over 12
 *   cells of travel or more (spread * ds / h >= 12: 24 iterations at ds = h / 2; 0 without a hint) (8 ints in all; [7] unused).
 * The ring-window kernel runs when [0] * 100 >= [1] * drrt_ring_threshold_pct() (the library's compile-time threshold, 20 in
 * the product build) or when [6] * 1000 >= [1] * drrt_ring_long_threshold_permille() (75: the box-window kernel does not use
 * the step hint, so rays that left the forward march that far apart run spread along their path beyond its window); its
 * sparse-only instantiation (fixed-point window) unless [5] is set -- the one that hands over per step ("direct") when
 * [4] * 100 < [3] * drrt_ring_direct_threshold_pct() (40: few lanes share a cell), else the one that hands over per cell.
 * Calibration:
 * csrc/drrt_march.h, bundles_want_ring / bundles_long / bundles_want_sparse.  The counters describe the START cells of the
 * bundles (the exit rays as given), not where the step hint's delays put the lanes later on. */
DRRT_API const unsigned* drrt_last_bundle_counters(void);
DRRT_API int drrt_ring_threshold_pct(void);
DRRT_API int drrt_ring_long_threshold_permille(void);   /* ring kernel also when counters [6] * 1000 >= [1] * this */
DRRT_API int drrt_ring_direct_threshold_pct(void);      /* its direct sparse-only instantiation when counters [4] * 100 < [3] * this */

/* ---- 16-bit ray state "q16" (BASELINE.json config 5: "fp16 ray state + fp32 adjoint accumulate") -----------------
 * The reference is fp32-only (include/types.h:36-46).  IEEE half keeps 11 significant bits wherever the value is:
 * a position near 1.0 is rounded to 2^-11 = 1/8 voxel of a 256^3 grid, which changes the adjoint trajectories
 * enough to move the gradient by 26 % rel-L2 (tests/test_baseline_configs.py, round 1).  The q16 formats spend the
 * same 6 bytes per 3-vector where a march needs them:
 *   positions   unsigned 16-bit codes c: p = q_min + c * q_step, over [-E/16, E + E/16] with E = max((res-1)*h);
 *               q_step = 1.125 E / 65535 (= h/228 at 256^3); positions outside the range SATURATE
 *   directions  signed 16-bit fixed point c: v = c * 2^-14 (range [-2, 2): |v| = n < 2); saturating
 *   seeds       dx, dv of the adjoint: IEEE half (relative precision is what a seed needs; scale them into range)
 * drrt_trace_q16io / drrt_backtrace_q16io widen exactly on load, march / accumulate in fp32 and round outputs once:
 *   drrt_trace_q16io(encode(x), encode(v)) == encode(drrt_trace_f32(decode(encode(x)), decode(encode(v)))) bit for bit.
 * drrt_q16_encode / drrt_q16_decode are the device-side conversions (either array may be NULL);
 * drrt_q16_params returns {q_min, q_step, 2^-14} for a grid (host pointers).
 * Tolerance (stated, no reference counterpart; tests/test_baseline_configs.py, 256^3 / 512 steps / 512^2 sensor / 262k
 * rays): the gradient splat of a sample jumps by one voxel when the sample changes cell, and a perturbation of d voxels
 * of the exit state makes a fraction ~d of the samples do so, so two SPARSE gradient grids (a handful of samples per
 * voxel) differ by ~sqrt(d) rel-L2 whatever the storage format: IEEE half (d ~ 1/8) 0.26, q16 (d ~ 2e-3) 0.055 raw and
 * 0.013 after a 3^3 box filter (the scale an optimiser sees); q16 positions with fp32 directions
 * (DRRT_FLAG_Q16_POS_ONLY) 0.050 / 0.012.  Tested bounds: 0.1 raw, 2e-2 filtered.                                      */
DRRT_API int drrt_q16_params(const int res[3], float h, float out[3]);
DRRT_API int drrt_q16_encode(const int res[3], float h, size_t n, const float* pos, const float* vel,
                    void* pos_q, void* vel_q, void* stream);
DRRT_API int drrt_q16_decode(const int res[3], float h, size_t n, const void* pos_q, const void* vel_q,
                    float* pos, float* vel, void* stream);
DRRT_API int drrt_trace_q16io(const float* rif, long long nvox, const int res[3], size_t n,
                     const void* pos_q, const void* vel_q, float h, float ds,
                     void* xt_q, void* vt_q,
                     drrt_stats* stats, void* workspace, size_t workspace_bytes,
                     unsigned flags, void* stream);
DRRT_API int drrt_backtrace_q16io(const float* rif, long long nvox, const int res[3], size_t n,
                         const void* xt_q, const void* vt_q, const void* dx_h, const void* dv_h,
                         float h, float ds, float* grad,
                         drrt_stats* stats, void* workspace, size_t workspace_bytes,
                         unsigned flags, void* stream);

/* ---- forward marches ------------------------------------------------------------------- */

/* Tracer<false,true>::trace  -- src/tracer.cpp:35-100, bound as TracerC.trace (src/drrt.cpp:51) */
DRRT_API int drrt_trace_f32(const float* rif, long long nvox, const int res[3], size_t n,
                   const float* pos, const float* vel, float h, float ds,
                   float* xt, float* vt,
                   drrt_stats* stats, void* workspace, size_t workspace_bytes,
                   unsigned flags, void* stream);

/* fp16 ray-state variant of drrt_trace_f32 (BASELINE.json config 5: "fp16 ray state + fp32 adjoint
 * accumulate"; the reference is fp32-only, include/types.h:36-46).  pos, vel, xt, vt are (n,3) IEEE
 * half; values are widened exactly on load, the march is fp32, outputs are rounded to half once:
 * result == half(drrt_trace_f32(float(pos), float(vel))) bit for bit.                            */
DRRT_API int drrt_trace_f16io(const float* rif, long long nvox, const int res[3], size_t n,
                     const void* pos_h, const void* vel_h, float h, float ds,
                     void* xt_h, void* vt_h,
                     drrt_stats* stats, void* workspace, size_t workspace_bytes,
                     unsigned flags, void* stream);

/* Tracer::trace_plane -- src/tracer.cpp:102-172, bound as TracerC.trace_pln (src/drrt.cpp:52).
 * failmask[i] = 1 for rays that never got flagged escaped (uint8, n entries).                 */
DRRT_API int drrt_trace_pln_f32(const float* rif, long long nvox, const int res[3], size_t n,
                       const float* pos, const float* vel,
                       const float* pln_o, const float* pln_d, float h, float ds,
                       float* xt, float* vt, uint8_t* failmask,
                       drrt_stats* stats, void* workspace, size_t workspace_bytes,
                       unsigned flags, void* stream);

/* Tracer::trace_target -- src/tracer.cpp:174-242, TracerC.trace_target (src/drrt.cpp:54).
 * Records the state at closest approach to target[i]; dist2 = squared distance there.          */
DRRT_API int drrt_trace_target_f32(const float* rif, long long nvox, const int res[3], size_t n,
                          const float* pos, const float* vel, const float* target,
                          float h, float ds, float* xt, float* vt, float* dist2,
                          drrt_stats* stats, void* workspace, size_t workspace_bytes,
                          unsigned flags, void* stream);

/* Tracer::trace_sdf -- src/tracer.cpp:244-310, TracerC.trace_sdf (src/drrt.cpp:53).
 * Always needs a workspace of drrt_workspace_bytes(n, flags) bytes (n flag bytes for the second pass that
 * reproduces the reference's global-loop behaviour of rays leaving the box while still sdf-inside).  */
DRRT_API int drrt_trace_sdf_f32(const float* rif, const float* sdf, long long nvox, const int res[3],
                       size_t n, const float* pos, const float* vel, float h, float ds,
                       float* xt, float* vt,
                       drrt_stats* stats, void* workspace, size_t workspace_bytes,
                       unsigned flags, void* stream);

/* Tracer::trace_cable -- src/tracer.cpp:312-382, TracerC.trace_cable (src/drrt.cpp:55).
 * rif = radial profile fp32[rres] (src/cylinder_volume.cpp).                                    */
DRRT_API int drrt_trace_cable_f32(const float* rif, size_t rres, float radius, float length, size_t n,
                         const float* pos, const float* vel, const float* target, float ds,
                         float* xt, float* vt, float* dist2,
                         drrt_stats* stats, void* workspace, size_t workspace_bytes,
                         unsigned flags, void* stream);

/* ---- adjoint marches: accumulate dL/dn into `grad` --------------------------------------- */

/* Tracer::backtrace -- src/tracer.cpp:384-440, TracerC.backtrace (src/drrt.cpp:56).
 * grad: fp32[nvox]; zeroed by the call unless DRRT_FLAG_NO_ZERO.                                 */
DRRT_API int drrt_backtrace_f32(const float* rif, long long nvox, const int res[3], size_t n,
                       const float* xt, const float* vt, const float* dx, const float* dv,
                       float h, float ds, float* grad,
                       drrt_stats* stats, void* workspace, size_t workspace_bytes,
                       unsigned flags, void* stream);

/* fp16 ray-state variant of drrt_backtrace_f32: xt, vt, dx, dv are (n,3) IEEE half, the adjoint
 * recurrences and the accumulation into `grad` stay fp32.                                        */
DRRT_API int drrt_backtrace_f16io(const float* rif, long long nvox, const int res[3], size_t n,
                         const void* xt_h, const void* vt_h, const void* dx_h, const void* dv_h,
                         float h, float ds, float* grad,
                         drrt_stats* stats, void* workspace, size_t workspace_bytes,
                         unsigned flags, void* stream);

/* Tracer::backtrace in depth chunks (not in the reference; for the multi-GPU path): the iterations
 * [it_begin, it_begin + it_count) of the adjoint march's max_steps (drrt_backtrace_max_steps(); it_count < 0 = all that are
 * left), same arguments as drrt_backtrace_f32.  it_begin == 0 starts from (xt, vt, dx, dv) and zeroes `grad` (unless
 * DRRT_FLAG_NO_ZERO) and `stats`; it_begin > 0 continues from `state` (drrt_backtrace_chunk_state_bytes(n) bytes, written
 * by the previous chunk; opaque) and accumulates into `grad` and `stats`.  After the last chunk `grad` holds what
 * drrt_backtrace_f32 computes -- the same per-ray contributions, summed in another order.  Why: with rays that travel on
 * one clock (a plane source) the part of the grid every ray has left behind is FINAL after a chunk, so a rank can start
 * reducing that slab with the other ranks while its next chunk marches (adjointnonlinearraytracing_amd/dist.py).
 *   visit order  every chunk of a march must visit the rays in the SAME order (the state is stored per visit slot): with
 *                DRRT_FLAG_SORT_RAYS hand the order of the first chunk (drrt_last_order(), or the paired forward's) to
 *                every later chunk with drrt_set_order_hint(); a resumed chunk without it is refused.
 *   progress     nullable DEVICE pointer to 20 ints, written by the chunk: where the rays that are still marching stand
 *                and head -- [0..2] min and [3..5] max of their positions (x, y, z: the NEXT sample of each ray), [6..8] min
 *                and [9..11] max of their velocities, as order-preserving keys (key = bits >= 0 ? bits : bits ^ 0x7fffffff;
 *                the map is its own inverse), [12] their number -- and [13..15] min, [16..18] max of the positions of
 *                the samples THIS chunk contributed at (a sample at p touches the voxels floor(p / h) and floor(p / h) + 1
 *                per axis), [19] unused.  The march moves a ray by -ds * v per iteration: a half space that lies behind
 *                every marching ray and that no ray heads back into is final; the sample box of the NEXT chunk tells
 *                afterwards whether that held.
 * Runs the box-window kernel (k_backtrace_flat) whatever the bundles look like; fp32 ray state only.                  */
DRRT_API size_t drrt_backtrace_chunk_state_bytes(size_t n);
DRRT_API int drrt_backtrace_max_steps(const int res[3], float h, float ds);      /* src/tracer.cpp:417 (Q5); < 0: bad arguments */
DRRT_API int drrt_backtrace_chunk_f32(const float* rif, long long nvox, const int res[3], size_t n,
                         const float* xt, const float* vt, const float* dx, const float* dv,
                         float h, float ds, float* grad,
                         drrt_stats* stats, void* workspace, size_t workspace_bytes,
                         unsigned flags, void* stream,
                         void* state, size_t state_bytes, int it_begin, int it_count, int* progress);

/* Tracer::backtrace_sdf -- src/tracer.cpp:443-509, TracerC.backtrace_sdf (src/drrt.cpp:57).     */
DRRT_API int drrt_backtrace_sdf_f32(const float* rif, const float* sdf, long long nvox, const int res[3],
                           size_t n, const float* xt, const float* vt,
                           const float* dx, const float* dv, float h, float ds, float* grad,
                           drrt_stats* stats, void* workspace, size_t workspace_bytes,
                           unsigned flags, void* stream);

/* Tracer::backtrace_cable -- src/tracer.cpp:511-567, TracerC.backtrace_cable (src/drrt.cpp:58).
 * grad: fp32[rres].                                                                             */
DRRT_API int drrt_backtrace_cable_f32(const float* rif, size_t rres, float radius, float length, size_t n,
                             const float* xt, const float* vt, const float* dx, const float* dv,
                             float ds, float* grad,
                             drrt_stats* stats, void* workspace, size_t workspace_bytes,
                             unsigned flags, void* stream);

/* ---- sensor image splat (SURVEY.md 8.8 "next" row 1; the reference does this in torch) ----------
 * Forward: core/sensor.py:5-28 generate_sensor = trace_rays_to_plane (:195-202) + sensor frame
 * (t1 = n x t2, t2; get_tan_vecs :219-231 is evaluated by the caller) + foreshortening |v.n| +
 * Grid.Splat(average=False) with 4x4 tent taps (core/grid.py:37-64,77-81,133-151).
 *   x, v      (n,3) fp32 rays (normally the exit rays of drrt_trace_f32)
 *   e         per-ray energy fp32[n], or NULL to use e_scalar for every ray
 *   plane_p/n, t1, t2   HOST pointers to 3 floats each (one sensor plane per call, as in the reference)
 *   image     fp32[res*res], row-major image[ia*res + ib] with ia along t1; zeroed unless DRRT_FLAG_NO_ZERO
 * Backward: gradient of sum(grad_image * image) w.r.t. (x, v) -- what torch.autograd yields through
 * the reference's generate_sensor; grad_x, grad_v are (n,3) fp32 outputs (the seed of backtrace).  */
DRRT_API int drrt_sensor_splat_f32(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                          const float plane_p[3], const float plane_n[3], const float t1[3], const float t2[3],
                          int res, float span, float* image, unsigned flags, void* stream);
DRRT_API int drrt_sensor_splat_bwd_f32(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                              const float plane_p[3], const float plane_n[3], const float t1[3], const float t2[3],
                              int res, float span, const float* grad_image, float* grad_x, float* grad_v,
                              void* stream);

/* Far-field sensor: core/sensor.py:31-53 generate_inf_sensor (called at core/image_opt.py:116).  The image is a
 * histogram of the NORMALISED ray directions in the sensor frame: coordinates (vhat.t1, vhat.t2) + ang_cut with
 * ang_cut = sin(0.5 * deg2rad(angle_span)) computed by the caller (sensor.py:38), cell size 2*ang_cut/res, weight e
 * (no foreshortening), the same Grid.Splat.  Positions do not enter; the backward writes grad_x = 0.            */
DRRT_API int drrt_sensor_far_splat_f32(size_t n, const float* v, const float* e, float e_scalar, const float t1[3],
                              const float t2[3], int res, float ang_cut, float* image, unsigned flags, void* stream);
DRRT_API int drrt_sensor_far_splat_bwd_f32(size_t n, const float* v, const float* e, float e_scalar, const float t1[3],
                                  const float t2[3], int res, float ang_cut, const float* grad_image,
                                  float* grad_x, float* grad_v, void* stream);

/* core/sensor.py:102-138 get_sdf_vals_near (mode 0) / get_sdf_vals_far (mode 1): Grid(tex, h).Get at the rays' sensor
 * coordinates -- the 4x4 radial-tent interpolant of core/grid.py:100-124, tap indices clipped to the (res, res) texture
 * (edge extrapolation off the texture, as in the reference) -- and its analytic backward w.r.t. the rays.  mode 0: rays ->
 * plane -> sensor frame, cell size span / res; mode 1: coordinates v . T + span / 2 from the direction as it is (pass
 * span = 2 * ang_cut).  tex, f_out, grad_f: device fp32; the plane and the tangents are host float[3].              */
DRRT_API int drrt_sensor_tex_get_f32(size_t n, const float* x, const float* v, const float plane_p[3], const float plane_n[3],
                            const float t1[3], const float t2[3], const float* tex, int res, float span, int mode,
                            float* f_out, void* stream);
DRRT_API int drrt_sensor_tex_get_bwd_f32(size_t n, const float* x, const float* v, const float plane_p[3],
                                const float plane_n[3], const float t1[3], const float t2[3], const float* tex, int res,
                                float span, int mode, const float* grad_f, float* grad_x, float* grad_v, void* stream);

/* The same six sensor operators with the sensor frame in DEVICE memory: frame12 = 12 fp32 values (plane_p, plane_n, t1, t2;
 * the far-field entries read t1, t2 only).  The reference keeps its planes as torch tensors on the device (core/image_opt.py
 * :88-119 slices them from the ray generator's output), so a caller of the host-pointer entries above has to copy them back
 * first -- a device synchronisation per call.  These entries need none; results are identical.                            */
DRRT_API int drrt_sensor_splat_dframe_f32(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                                 const float* frame12, int res, float span, float* image, unsigned flags, void* stream);
DRRT_API int drrt_sensor_splat_dframe_bwd_f32(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                                     const float* frame12, int res, float span, const float* grad_image, float* grad_x,
                                     float* grad_v, void* stream);
DRRT_API int drrt_sensor_far_splat_dframe_f32(size_t n, const float* v, const float* e, float e_scalar, const float* frame12,
                                     int res, float ang_cut, float* image, unsigned flags, void* stream);
DRRT_API int drrt_sensor_far_splat_dframe_bwd_f32(size_t n, const float* v, const float* e, float e_scalar,
                                         const float* frame12, int res, float ang_cut, const float* grad_image,
                                         float* grad_x, float* grad_v, void* stream);
DRRT_API int drrt_sensor_tex_get_dframe_f32(size_t n, const float* x, const float* v, const float* frame12, const float* tex,
                                   int res, float span, int mode, float* f_out, void* stream);
DRRT_API int drrt_sensor_tex_get_dframe_bwd_f32(size_t n, const float* x, const float* v, const float* frame12,
                                       const float* tex, int res, float span, int mode, const float* grad_f,
                                       float* grad_x, float* grad_v, void* stream);

/* core/sensor.py:195-202 trace_rays_to_plane, the statement right after the march in every experiment: x_out = x + t v with
 * t = n.(p - x) / n.v (v passes through unchanged), and its analytic backward w.r.t. the rays (grad_x, grad_v receive the
 * part of the gradient that flows through x_out).  plane_stride 3 = one (p, n) per ray, 0 = one plane for all rays.
 * All pointers are device pointers to fp32 (n,3) row-major arrays (planes: (n,3) or (1,3)).                        */
DRRT_API int drrt_rays_to_plane_f32(size_t n, const float* x, const float* v, const float* plane_p, const float* plane_n,
                           int plane_stride, float* x_out, void* stream);
DRRT_API int drrt_rays_to_plane_bwd_f32(size_t n, const float* x, const float* v, const float* plane_p, const float* plane_n,
                               int plane_stride, const float* grad_x_out, float* grad_x, float* grad_v, void* stream);

/* ---- multires up-sampling (SURVEY.md 8.8 "next" row 3) --------------------------------------------
 * core/optimizer.py:7-10 upres_scene / core/grid.py:318-330 upres_volume: trilinear resampling of a
 * cubic (R,R,R) fp32 volume at linspace(0,1,S) per axis, evaluated in float64 like the reference and
 * rounded to fp32.  Shapes are HOST pointers to 3 ints in torch order.                            */
DRRT_API int drrt_upres_volume_f32(const float* src, const int src_shape[3], float* dst, const int dst_shape[3],
                          void* stream);

/* core/optimizer.py:57-69, the tail of one multires_opt iteration in ONE pass over the volume: the boundary-gradient
 * mask `n.grad[mask] = 0` (mask = outermost voxel layer, :54-55), `opto.step()` of torch.optim.Adam (amsgrad = maximize
 * = False; formula of torch/optim/adam.py) and `n.clamp_(min=1)`.  All four arrays are DEVICE fp32 arrays of
 * shape[0]*shape[1]*shape[2] elements (torch order, last axis fastest), updated in place (grad: zeroed on the boundary
 * layer, like the reference).  `step` is the step count AFTER this update's increment (1 for the first call); lr, betas,
 * eps, weight_decay are the param group's values.                                                                  */
#define DRRT_ADAM_MASK_BOUNDARY 1u  /* treat the gradient of the outermost voxel layer as 0 (and zero it in `grad`) */
#define DRRT_ADAM_CLAMP_MIN     2u  /* clamp the updated parameter from below at clamp_min                          */
DRRT_API int drrt_adam_step_f32(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const int shape[3],
                       double step, double lr, double beta1, double beta2, double eps, double weight_decay,
                       double clamp_min, unsigned flags, void* stream);

/* ---- ray generation (SURVEY.md 8.8 "next" row 2) ---------------------------------------------------
 * kind 0: core/source.py:54-69 plane_source3_rand + :275-293 rotate_pts_to_source;
 * kind 1: :72-104 point_source3_rand -- for n_views views in one call (what :352-357 rand_rays_in_sphere,
 * :360-365 rand_ptrays_in_sphere and :398-412 rand_rays_cube concatenate), optionally followed by
 * :555-563 random_rotate_ic.  All pointers are DEVICE pointers except ic_rot.
 *   u          fp32[n_views][2*spp][p0][p1] uniforms in [0,1) (the reference's torch.rand draws)
 *   view_rot   fp32[n_views][9] row-major rotate_ray3 matrix of each view (:303-312)
 *   width, sensor_dist, span   python scalars of the reference (doubles, rounded to fp32 where the
 *              reference's tensors are fp32)
 *   circle     keep only points with r < width/2 (order-preserving compaction, :277-280 / :81-92)
 *   independent  kind 0 only: :60-63 instead of the jittered pixel grid
 *   ic_rot     HOST pointer to 9 floats (row-major M) or NULL: x' = M (x - span/2) + span/2 etc.
 *   x, v       fp32[cap][3], planes fp32[cap][3][3] with cap = n_views*spp*p0*p1; kept rays are
 *              written densely in the reference's order (view, sample, pixel row, pixel column)
 *   view_counts  int32[n_views+1]: exclusive prefix of kept rays per view, [n_views] = total
 *   workspace  drrt_gen_workspace_bytes(...) bytes of device scratch                                */
#define DRRT_SOURCE_PLANE 0
#define DRRT_SOURCE_POINT 1
DRRT_API size_t drrt_gen_workspace_bytes(int n_views, int spp, int p0, int p1);
DRRT_API int drrt_gen_rays_f32(int kind, const float* u, const float* view_rot, int n_views, int spp, int p0, int p1,
                      double width, double sensor_dist, int circle, int independent,
                      const float* ic_rot, double span, float* x, float* v, float* planes,
                      int* view_counts, void* workspace, size_t workspace_bytes, void* stream);
/* Cone source: core/source.py:186-203 cone_source3_rand (the fibre experiment's source, core/fiber_opt.py:131) --
 * spp*p0*p1 rays per view from the point R (0, -width/2, 0) + width/2 with directions drawn by hatbox_sample
 * (:531-545) in a cone of full angle cone_angle about R e_y.  `u` holds, per view, the two uniform draws of
 * hatbox_sample: u[view][0][c] for z, u[view][1][c] for theta, c < spp*p0*p1.  cone_cos = cos(cone_angle / 2) as the
 * reference evaluates it (fp32, :533-534).  No disc mask: every candidate is a ray.                              */
DRRT_API int drrt_gen_cone_rays_f32(const float* u, const float* view_rot, int n_views, int spp, int p0, int p1,
                           double width, double sensor_dist, double cone_cos, const float* ic_rot_host, double span,
                           float* x, float* v, float* planes, int* view_counts, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- profiling aid (bench.py; no counterpart in the reference) --------------------------------
 * After drrt_profile_begin(capacity) every march call records a HIP event pair on its stream
 * around each kernel it launches (no synchronisation).  drrt_profile_collect() waits for the
 * recorded events, writes up to max_out (kernel id, milliseconds) pairs in launch order, resets
 * the log and returns the count.  Single-threaded use only.                                    */
#define DRRT_PROF_TRACE      1   /* forward march kernel                 */
#define DRRT_PROF_BACKTRACE  2   /* adjoint march kernel                 */
#define DRRT_PROF_SORT       3   /* entry-voxel keys + radix sort        */
#define DRRT_PROF_ZERO       4   /* zero-fill of the gradient grid       */
#define DRRT_PROF_QUAD       5   /* build of the pair copy of the grid   */
DRRT_API int  drrt_profile_begin(int capacity);
DRRT_API int  drrt_profile_collect(int* kernel_ids, float* ms, int max_out);
DRRT_API void drrt_profile_end(void);

#ifdef __cplusplus
}
#endif
#endif /* DRRT_HIP_H */
