// drrt_sensor.hip -- sensor image splat and its backward ("next" row 1):
// the step that follows the march in the reference's image experiments.
//
// Reference semantics (all torch, /root/reference/core):
//   sensor.py:195-202  trace_rays_to_plane   t = n.(p - x) / n.v ;  x' = x + t v
//   sensor.py:5-28     generate_sensor       2-D coords of x' in the sensor frame (t1 = n x t2, t2),
//                                            foreshortening fs = |v.n|, Grid.Splat(xn, fs*e, average=False)
//   sensor.py:31-53    generate_inf_sensor   far-field ("infinite distance") sensor: 2-D coords of the NORMALISED
//                                            direction in the sensor frame + ang_cut, ang_cut = sin(angle_span/2),
//                                            cell size 2*ang_cut/res, weight e (no foreshortening), same Splat
//   grid.py:37-64      Grid.index_values     u = xn/h - 0.5, 4x4 taps around floor(u), r = |u - idx|
//   grid.py:77-81      rbf_tent              w = max(sqrt(2) - r, 0)
//   grid.py:133-151    Grid.Splat            image[idx] += w/sum(w) * f  for taps inside the image
//                                            (normalised over ALL 16 taps, in or out)
// The reference builds ~20 (N,16)-sized temporaries per call and scatters with
// index_put_(accumulate=True); its backward is torch.autograd through all of that.  Here both
// directions are one fused kernel each: ray -> plane -> frame -> 16 taps -> fp32 atomics, and the
// analytic backward (gather 16 taps of dL/dimage, chain through the tent weights, the frame and the
// plane intersection) producing (grad_x, grad_v) directly -- the seed of Tracer::backtrace.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/drrt_hip.h"

namespace drrt {

struct SensorArgs {
  const float* x; const float* v; const float* e;   // e nullable -> e_scalar
  float e_scalar;
  float p[3], n[3], t1[3], t2[3];
  const float* frame_dev;      // nullable: 12 device floats (p, n, t1, t2) read by the kernel INSTEAD of the four above --
                               // the *_dframe entries: a caller whose plane lives on the device needs no host copy / sync
  int res; float span, inv_hs, half_span;
  int far;                     // 1: generate_inf_sensor (coordinates from the direction only; span = 2*ang_cut);
                               // 2: get_sdf_vals_far (coordinates from the UN-normalised direction, sensor.py:134)
  float* image;                // forward out (res*res)
  const float* grad_image;     // backward in (tex_get: the texture being sampled)
  const float* grad_f;         // tex_get backward in: dL/df per ray
  float* f_out;                // tex_get forward out: one value per ray
  float* grad_x; float* grad_v;
  size_t n_rays;
};

__device__ __forceinline__ void sensor_frame(SensorArgs& a) {       // wave-uniform scalar loads
  if (a.frame_dev != nullptr) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.p[k] = a.frame_dev[k]; a.n[k] = a.frame_dev[3 + k]; a.t1[k] = a.frame_dev[6 + k]; a.t2[k] = a.frame_dev[9 + k]; }
  }
}

struct SensorRay {
  float den, t, F, u[2];
  int i1[2];
  bool ok;
};

__device__ __forceinline__ SensorRay sensor_locate(const SensorArgs& a, size_t i, float x[3], float v[3]) {
  SensorRay r;
  x[0] = a.x[3 * i]; x[1] = a.x[3 * i + 1]; x[2] = a.x[3 * i + 2];
  v[0] = a.v[3 * i]; v[1] = a.v[3 * i + 1]; v[2] = a.v[3 * i + 2];
  if (a.far) {                                                          // sensor.py:36-47
    const float nv = a.far == 2 ? 1.f : sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float inv = 1.f / nv;
    const float h0 = v[0] * inv, h1 = v[1] * inv, h2 = v[2] * inv;      // v / norm(v)
    const float xa = h0 * a.t1[0] + h1 * a.t1[1] + h2 * a.t1[2] + a.half_span;   // + ang_cut
    const float xb = h0 * a.t2[0] + h1 * a.t2[1] + h2 * a.t2[2] + a.half_span;
    r.den = nv; r.t = 0.f;
    r.u[0] = xa * a.inv_hs - 0.5f; r.u[1] = xb * a.inv_hs - 0.5f;
    const float f0 = floorf(r.u[0]), f1 = floorf(r.u[1]);
    r.ok = (f0 >= -3.f) & (f0 <= (float)(a.res + 1)) & (f1 >= -3.f) & (f1 <= (float)(a.res + 1));
    r.i1[0] = r.ok ? (int)f0 : 0; r.i1[1] = r.ok ? (int)f1 : 0;
    r.F = a.e ? a.e[i] : a.e_scalar;                                    // :49 (no foreshortening)
    return r;
  }
  r.den = v[0] * a.n[0] + v[1] * a.n[1] + v[2] * a.n[2];
  const float num = (a.p[0] - x[0]) * a.n[0] + (a.p[1] - x[1]) * a.n[1] + (a.p[2] - x[2]) * a.n[2];
  r.t = num / r.den;                                                  // sensor.py:199-200
  const float q0 = x[0] + r.t * v[0] - a.p[0], q1 = x[1] + r.t * v[1] - a.p[1], q2 = x[2] + r.t * v[2] - a.p[2];
  const float xa = q0 * a.t1[0] + q1 * a.t1[1] + q2 * a.t1[2] + a.half_span;    // sensor.py:22-23
  const float xb = q0 * a.t2[0] + q1 * a.t2[1] + q2 * a.t2[2] + a.half_span;
  r.u[0] = xa * a.inv_hs - 0.5f; r.u[1] = xb * a.inv_hs - 0.5f;       // grid.py:38
  const float f0 = floorf(r.u[0]), f1 = floorf(r.u[1]);
  // rays that miss the image by more than the tap footprint (or are NaN) contribute nothing
  r.ok = (f0 >= -3.f) & (f0 <= (float)(a.res + 1)) & (f1 >= -3.f) & (f1 <= (float)(a.res + 1));
  r.i1[0] = r.ok ? (int)f0 : 0; r.i1[1] = r.ok ? (int)f1 : 0;
  r.F = fabsf(r.den) * (a.e ? a.e[i] : a.e_scalar);                   // sensor.py:18-19
  return r;
}

// Forward splat.  Caustic / focused images put most rays on a few pixels, and same-address global
// atomics are serialised at the memory side (measured: 5.4 ms for 1M Luneburg-focused rays onto 512^2,
// 0.5 ms for spread-out rays).  Rays of a block are neighbours in the source and therefore land close
// together, so each block accumulates into an LDS tile of the image (doubles, ds_add_f64) anchored at
// the block's smallest hit pixel and flushes it with one global atomic per touched pixel; taps
// outside the tile go to global memory directly.
constexpr int kTile = 48;                       // tile edge in pixels (48*48 doubles = 18 KiB)
constexpr int kVoteCell = 32, kVote = 40;       // anchor vote: 40x40 coarse cells of 32 pixels (images up to 1277^2; larger ones clamp)

__global__ void __launch_bounds__(256) k_sensor_splat(SensorArgs a) {
  sensor_frame(a);
  __shared__ double s_tile[kTile * kTile];
  __shared__ unsigned s_vote[kVote * kVote];
  __shared__ int s_min[2];
  __shared__ unsigned s_best;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (int k = threadIdx.x; k < kTile * kTile; k += 256) s_tile[k] = 0.0;
  for (int k = threadIdx.x; k < kVote * kVote; k += 256) s_vote[k] = 0u;
  if (threadIdx.x < 2) s_min[threadIdx.x] = 0x7fffffff;
  if (threadIdx.x == 0) s_best = 0u;
  __syncthreads();
  float x[3], v[3];
  SensorRay r;
  r.ok = false; r.i1[0] = r.i1[1] = 0; r.u[0] = r.u[1] = 0.f; r.F = 0.f; r.den = 1.f; r.t = 0.f;
  if (i < a.n_rays) r = sensor_locate(a, i, x, v);
  // Tile anchor.  If all the block's hits fit one tile, anchor at the smallest hit pixel.  Otherwise (a focused
  // bundle plus stray rays: the min corner would leave the FOCUS outside the tile, on the slow same-address
  // global atomics) the rays vote on a coarse kVoteCell-pixel grid and the tile is centred on the winning cell.
  int ca = 0, cb = 0;
  if (r.ok) {
    atomicMin(&s_min[0], r.i1[0] - 1); atomicMin(&s_min[1], r.i1[1] - 1);
    ca = min(max((r.i1[0] + 3) / kVoteCell, 0), kVote - 1); cb = min(max((r.i1[1] + 3) / kVoteCell, 0), kVote - 1);
    atomicAdd(&s_vote[ca * kVote + cb], 1u);
  }
  __syncthreads();
  {
    // argmax over the votes: pack (count, cell) so that atomicMax picks the fullest cell
    unsigned best = 0u;
    for (int k = threadIdx.x; k < kVote * kVote; k += 256) best = max(best, (s_vote[k] << 12) | (unsigned)k);
    if (best >> 12) atomicMax(&s_best, best);
  }
  __syncthreads();
  int oa = s_min[0], ob = s_min[1];             // tile origin (block-uniform)
  {
    const int wa = (int)((s_best & 0xfffu) / kVote), wb = (int)((s_best & 0xfffu) % kVote);
    const int va = wa * kVoteCell - 3 - (kTile - kVoteCell) / 2, vb = wb * kVoteCell - 3 - (kTile - kVoteCell) / 2;
    // keep the min-corner anchor when it already covers the winning cell entirely
    if (va + (kTile - kVoteCell) / 2 + kVoteCell + 3 > oa + kTile || vb + (kTile - kVoteCell) / 2 + kVoteCell + 3 > ob + kTile) {
      oa = va; ob = vb;
    }
  }
  if (r.ok) {
    float w[16], wsum = 0.f;
#pragma unroll
    for (int ja = 0; ja < 4; ++ja) {
      const float da = r.u[0] - (float)(r.i1[0] - 1 + ja);
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const float db = r.u[1] - (float)(r.i1[1] - 1 + jb);
        const float ww = fmaxf(1.41421356237f - sqrtf(da * da + db * db), 0.f);   // grid.py:79
        w[ja * 4 + jb] = ww; wsum += ww;
      }
    }
    const float scale = r.F / wsum;                                     // grid.py:145 (all 16 taps)
    const int la = r.i1[0] - 1 - oa, lb = r.i1[1] - 1 - ob;             // tile coords of the first tap
    const bool in_tile = (la >= 0) & (lb >= 0) & (la + 3 < kTile) & (lb + 3 < kTile);
#pragma unroll
    for (int ja = 0; ja < 4; ++ja) {
      const int ia = r.i1[0] - 1 + ja;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const int ib = r.i1[1] - 1 + jb;
        const float c = w[ja * 4 + jb] * scale;
        if (((unsigned)ia < (unsigned)a.res) & ((unsigned)ib < (unsigned)a.res) & (c != 0.f)) {   // grid.py:140
          if (in_tile) atomicAdd(&s_tile[(la + ja) * kTile + (lb + jb)], (double)c);
          else unsafeAtomicAdd(a.image + (size_t)ia * a.res + ib, c);
        }
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kTile * kTile; k += 256) {
    const double g = s_tile[k];
    if (g != 0.0) {
      const int ia = oa + k / kTile, ib = ob + k % kTile;     // only in-image taps were accumulated
      unsafeAtomicAdd(a.image + (size_t)ia * a.res + ib, (float)g);
    }
  }
}

__global__ void __launch_bounds__(256) k_sensor_splat_bwd(SensorArgs a) {
  sensor_frame(a);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_rays) return;
  float x[3], v[3];
  const SensorRay r = sensor_locate(a, i, x, v);
  float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
  if (r.ok) {
    float W = 0.f, gw = 0.f, gda = 0.f, gdb = 0.f, sda = 0.f, sdb = 0.f;
#pragma unroll
    for (int ja = 0; ja < 4; ++ja) {
      const int ia = r.i1[0] - 1 + ja;
      const float da = r.u[0] - (float)ia;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const int ib = r.i1[1] - 1 + jb;
        const float db = r.u[1] - (float)ib;
        const float rr = sqrtf(da * da + db * db);
        const float ww = fmaxf(1.41421356237f - rr, 0.f);
        const bool valid = ((unsigned)ia < (unsigned)a.res) & ((unsigned)ib < (unsigned)a.res);
        const float g = valid ? a.grad_image[(size_t)ia * a.res + ib] : 0.f;
        const bool live = (ww > 0.f) & (rr > 0.f);
        const float inv_r = live ? 1.f / rr : 0.f;
        const float dwa = -da * inv_r, dwb = -db * inv_r;             // d w / d u
        W += ww; gw += g * ww; gda += g * dwa; gdb += g * dwb; sda += dwa; sdb += dwb;
      }
    }
    const float G = gw / W;
    const float k = r.F / W * a.inv_hs;
    const float ga = k * (gda - G * sda), gb = k * (gdb - G * sdb);    // dL/d xn
    const float gp0 = ga * a.t1[0] + gb * a.t2[0], gp1 = ga * a.t1[1] + gb * a.t2[1], gp2 = ga * a.t1[2] + gb * a.t2[2];
    if (a.far) {
      // coordinates depend on v only, through vhat = v/|v|:  d vhat / d v = (I - vhat vhat^T) / |v|
      const float inv = 1.f / r.den;
      const float h0 = v[0] * inv, h1 = v[1] * inv, h2 = v[2] * inv;
      const float hg = h0 * gp0 + h1 * gp1 + h2 * gp2;
      gv[0] = (gp0 - h0 * hg) * inv; gv[1] = (gp1 - h1 * hg) * inv; gv[2] = (gp2 - h2 * hg) * inv;
      a.grad_x[3 * i] = 0.f; a.grad_x[3 * i + 1] = 0.f; a.grad_x[3 * i + 2] = 0.f;
      a.grad_v[3 * i] = gv[0]; a.grad_v[3 * i + 1] = gv[1]; a.grad_v[3 * i + 2] = gv[2];
      return;
    }
    const float vg = (v[0] * gp0 + v[1] * gp1 + v[2] * gp2) / r.den;
    gx[0] = gp0 - a.n[0] * vg; gx[1] = gp1 - a.n[1] * vg; gx[2] = gp2 - a.n[2] * vg;     // (I - v n^T/den)^T
    const float ef = G * (a.e ? a.e[i] : a.e_scalar) * (r.den > 0.f ? 1.f : (r.den < 0.f ? -1.f : 0.f));
    gv[0] = r.t * gx[0] + ef * a.n[0]; gv[1] = r.t * gx[1] + ef * a.n[1]; gv[2] = r.t * gx[2] + ef * a.n[2];
  }
  a.grad_x[3 * i] = gx[0]; a.grad_x[3 * i + 1] = gx[1]; a.grad_x[3 * i + 2] = gx[2];
  a.grad_v[3 * i] = gv[0]; a.grad_v[3 * i + 1] = gv[1]; a.grad_v[3 * i + 2] = gv[2];
}

// ---- texture lookups at the sensor: core/sensor.py:102-138 get_sdf_vals_near / get_sdf_vals_far ---------------------
// Grid(d_tex, h).Get(xn) (core/grid.py:100-124) at the rays' sensor coordinates: the same 4x4 radial tent taps as the
// splat, used as an interpolant f = sum_i w_i f_i / sum_i w_i with tap indices CLIPPED to the texture (grid.py:57),
// not masked (a point off the texture is extrapolated from the edge texels).  The backward is the interpolant's analytic derivative (what Get returns as its second value),
// chained through the sensor frame and the plane intersection to (dL/dx, dL/dv).
struct TexTaps { float W, fw, fda, fdb, sda, sdb; };
// The taps follow the point (their texels are clipped, not the taps), so a point anywhere off the texture still has its
// 16 taps around it -- unlike the splat, which drops rays that miss the image: re-derive the tap origin without that test.
__device__ __forceinline__ void tex_origin(SensorRay& r) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float f = fminf(fmaxf(floorf(r.u[k]), -1048576.f), 1048576.f);     // NaN -> a bound; the weights are NaN/0 then
    r.i1[k] = (int)f;
  }
  r.ok = true;
}
__device__ __forceinline__ TexTaps tex_taps(const SensorArgs& a, const SensorRay& r) {
  TexTaps t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ja = 0; ja < 4; ++ja) {
    const int ia = r.i1[0] - 1 + ja;
    const float da = r.u[0] - (float)ia;
    const int ca = min(max(ia, 0), a.res - 1);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      const int ib = r.i1[1] - 1 + jb;
      const float db = r.u[1] - (float)ib;
      const int cb = min(max(ib, 0), a.res - 1);
      const float rr = sqrtf(da * da + db * db);
      const float ww = fmaxf(1.41421356237f - rr, 0.f);
      const float f = a.grad_image[(size_t)ca * a.res + cb];
      const bool live = (ww > 0.f) & (rr > 0.f);
      const float inv_r = live ? 1.f / rr : 0.f;
      const float dwa = -da * inv_r, dwb = -db * inv_r;               // d w / d u
      t.W += ww; t.fw += f * ww; t.fda += f * dwa; t.fdb += f * dwb; t.sda += dwa; t.sdb += dwb;
    }
  }
  return t;
}

__global__ void __launch_bounds__(256) k_sensor_tex_get(SensorArgs a) {
  sensor_frame(a);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_rays) return;
  float x[3], v[3];
  SensorRay r = sensor_locate(a, i, x, v);
  tex_origin(r);
  const TexTaps t = tex_taps(a, r);
  a.f_out[i] = t.fw / t.W;                                            // NaN coordinates: all weights 0 -> 0/0 = NaN
}

__global__ void __launch_bounds__(256) k_sensor_tex_get_bwd(SensorArgs a) {
  sensor_frame(a);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_rays) return;
  float x[3], v[3];
  SensorRay r = sensor_locate(a, i, x, v);
  tex_origin(r);
  float gx[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
  {
    const TexTaps t = tex_taps(a, r);
    const float f = t.fw / t.W;
    const float k = a.grad_f[i] / t.W * a.inv_hs;
    const float ga = k * (t.fda - f * t.sda), gb = k * (t.fdb - f * t.sdb);        // dL/d xn
    const float gp0 = ga * a.t1[0] + gb * a.t2[0], gp1 = ga * a.t1[1] + gb * a.t2[1], gp2 = ga * a.t1[2] + gb * a.t2[2];
    if (a.far) {                                    // far == 2: xn = v . T + ang_cut
      gv[0] = gp0; gv[1] = gp1; gv[2] = gp2;
    } else {
      const float vg = (v[0] * gp0 + v[1] * gp1 + v[2] * gp2) / r.den;
      gx[0] = gp0 - a.n[0] * vg; gx[1] = gp1 - a.n[1] * vg; gx[2] = gp2 - a.n[2] * vg;
      gv[0] = r.t * gx[0]; gv[1] = r.t * gx[1]; gv[2] = r.t * gx[2];
    }
  }
  a.grad_x[3 * i] = gx[0]; a.grad_x[3 * i + 1] = gx[1]; a.grad_x[3 * i + 2] = gx[2];
  a.grad_v[3 * i] = gv[0]; a.grad_v[3 * i + 1] = gv[1]; a.grad_v[3 * i + 2] = gv[2];
}

int sensor_fail(int code, const char* msg);   // drrt_kernels.hip

static const float kZero3[3] = {0.f, 0.f, 0.f};

static int fill_args(SensorArgs& a, size_t n, const float* x, const float* v, const float* e, float e_scalar,
                     const float p[3], const float nrm[3], const float t1[3], const float t2[3], int res, float span) {
  if (!x || !v || !p || !nrm || !t1 || !t2) return sensor_fail(DRRT_ERR_ARG, "null pointer");
  a.frame_dev = nullptr;
  if (res < 1 || res > 32768 || !(span > 0.f)) return sensor_fail(DRRT_ERR_ARG, "bad sensor resolution / span");
  a.x = x; a.v = v; a.e = e; a.e_scalar = e_scalar; a.n_rays = n;
  for (int k = 0; k < 3; ++k) { a.p[k] = p[k]; a.n[k] = nrm[k]; a.t1[k] = t1[k]; a.t2[k] = t2[k]; }
  a.res = res; a.span = span; a.inv_hs = 1.0f / (span / (float)res); a.half_span = span / 2;
  a.far = 0;
  return DRRT_OK;
}

}  // namespace drrt

using namespace drrt;

static int splat_fwd(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                     const float plane_p[3], const float plane_n[3], const float t1[3], const float t2[3],
                     const float* frame_dev, int res, float span, float* image, unsigned flags, void* stream) {
  SensorArgs a{};
  int rc = fill_args(a, n, x, v, e, e_scalar, plane_p, plane_n, t1, t2, res, span); if (rc) return rc;
  a.frame_dev = frame_dev;
  if (!image) return sensor_fail(DRRT_ERR_ARG, "null image pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & DRRT_FLAG_NO_ZERO)) {
    hipError_t e_ = hipMemsetAsync(image, 0, (size_t)res * res * sizeof(float), s);
    if (e_ != hipSuccess) return sensor_fail(DRRT_ERR_HIP, hipGetErrorString(e_));
  }
  if (n == 0) return DRRT_OK;
  a.image = image;
  hipLaunchKernelGGL(k_sensor_splat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

extern "C" int drrt_sensor_splat_f32(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                                     const float plane_p[3], const float plane_n[3], const float t1[3],
                                     const float t2[3], int res, float span, float* image, unsigned flags,
                                     void* stream) {
  return splat_fwd(n, x, v, e, e_scalar, plane_p, plane_n, t1, t2, nullptr, res, span, image, flags, stream);
}
extern "C" int drrt_sensor_splat_dframe_f32(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                                            const float* frame12, int res, float span, float* image, unsigned flags,
                                            void* stream) {
  if (!frame12) return sensor_fail(DRRT_ERR_ARG, "null frame pointer");
  return splat_fwd(n, x, v, e, e_scalar, kZero3, kZero3, kZero3, kZero3, frame12, res, span, image, flags, stream);
}

static int splat_bwd(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                     const float plane_p[3], const float plane_n[3], const float t1[3], const float t2[3],
                     const float* frame_dev, int res, float span, const float* grad_image, float* grad_x, float* grad_v,
                     void* stream) {
  SensorArgs a{};
  int rc = fill_args(a, n, x, v, e, e_scalar, plane_p, plane_n, t1, t2, res, span); if (rc) return rc;
  a.frame_dev = frame_dev;
  if (!grad_image || !grad_x || !grad_v) return sensor_fail(DRRT_ERR_ARG, "null gradient pointer");
  if (n == 0) return DRRT_OK;
  a.grad_image = grad_image; a.grad_x = grad_x; a.grad_v = grad_v;
  hipLaunchKernelGGL(k_sensor_splat_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

extern "C" int drrt_sensor_splat_bwd_f32(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                                         const float plane_p[3], const float plane_n[3], const float t1[3],
                                         const float t2[3], int res, float span, const float* grad_image,
                                         float* grad_x, float* grad_v, void* stream) {
  return splat_bwd(n, x, v, e, e_scalar, plane_p, plane_n, t1, t2, nullptr, res, span, grad_image, grad_x, grad_v, stream);
}
extern "C" int drrt_sensor_splat_dframe_bwd_f32(size_t n, const float* x, const float* v, const float* e, float e_scalar,
                                                const float* frame12, int res, float span, const float* grad_image,
                                                float* grad_x, float* grad_v, void* stream) {
  if (!frame12) return sensor_fail(DRRT_ERR_ARG, "null frame pointer");
  return splat_bwd(n, x, v, e, e_scalar, kZero3, kZero3, kZero3, kZero3, frame12, res, span, grad_image, grad_x, grad_v, stream);
}

// ---- far-field sensor (core/sensor.py:31-53 generate_inf_sensor, called at core/image_opt.py:116) ----------------
// Same splat kernels with SensorArgs::far set: `ang_cut` = sin(0.5 * deg2rad(angle_span)) is computed by the caller
// (the reference evaluates it in the rays' dtype, sensor.py:38); the image spans [0, 2*ang_cut)^2.
static int fill_far(SensorArgs& a, size_t n, const float* v, const float* e, float e_scalar, const float t1[3],
                    const float t2[3], const float* frame_dev, int res, float ang_cut) {
  const float zero[3] = {0.f, 0.f, 0.f};
  int rc = fill_args(a, n, v, v, e, e_scalar, zero, zero, t1, t2, res, 2.0f * ang_cut); if (rc) return rc;
  a.frame_dev = frame_dev;              // (the far field reads t1, t2 only; p and n are not used)
  a.half_span = ang_cut; a.inv_hs = 1.0f / (2.0f * ang_cut / (float)res);   // Grid(zeros, 2*ang_cut/res), :44
  a.far = 1;
  return DRRT_OK;
}

static int far_fwd(size_t n, const float* v, const float* e, float e_scalar, const float t1[3], const float t2[3],
                   const float* frame_dev, int res, float ang_cut, float* image, unsigned flags, void* stream) {
  SensorArgs a{};
  int rc = fill_far(a, n, v, e, e_scalar, t1, t2, frame_dev, res, ang_cut); if (rc) return rc;
  if (!image) return sensor_fail(DRRT_ERR_ARG, "null image pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & DRRT_FLAG_NO_ZERO)) {
    hipError_t e_ = hipMemsetAsync(image, 0, (size_t)res * res * sizeof(float), s);
    if (e_ != hipSuccess) return sensor_fail(DRRT_ERR_HIP, hipGetErrorString(e_));
  }
  if (n == 0) return DRRT_OK;
  a.image = image;
  hipLaunchKernelGGL(k_sensor_splat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

extern "C" int drrt_sensor_far_splat_f32(size_t n, const float* v, const float* e, float e_scalar, const float t1[3],
                                         const float t2[3], int res, float ang_cut, float* image, unsigned flags,
                                         void* stream) {
  return far_fwd(n, v, e, e_scalar, t1, t2, nullptr, res, ang_cut, image, flags, stream);
}
extern "C" int drrt_sensor_far_splat_dframe_f32(size_t n, const float* v, const float* e, float e_scalar,
                                                const float* frame12, int res, float ang_cut, float* image,
                                                unsigned flags, void* stream) {
  if (!frame12) return sensor_fail(DRRT_ERR_ARG, "null frame pointer");
  return far_fwd(n, v, e, e_scalar, kZero3, kZero3, frame12, res, ang_cut, image, flags, stream);
}

static int far_bwd(size_t n, const float* v, const float* e, float e_scalar, const float t1[3], const float t2[3],
                   const float* frame_dev, int res, float ang_cut, const float* grad_image, float* grad_x, float* grad_v,
                   void* stream) {
  SensorArgs a{};
  int rc = fill_far(a, n, v, e, e_scalar, t1, t2, frame_dev, res, ang_cut); if (rc) return rc;
  if (!grad_image || !grad_x || !grad_v) return sensor_fail(DRRT_ERR_ARG, "null gradient pointer");
  if (n == 0) return DRRT_OK;
  a.grad_image = grad_image; a.grad_x = grad_x; a.grad_v = grad_v;
  hipLaunchKernelGGL(k_sensor_splat_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

extern "C" int drrt_sensor_far_splat_bwd_f32(size_t n, const float* v, const float* e, float e_scalar, const float t1[3],
                                             const float t2[3], int res, float ang_cut, const float* grad_image,
                                             float* grad_x, float* grad_v, void* stream) {
  return far_bwd(n, v, e, e_scalar, t1, t2, nullptr, res, ang_cut, grad_image, grad_x, grad_v, stream);
}
extern "C" int drrt_sensor_far_splat_dframe_bwd_f32(size_t n, const float* v, const float* e, float e_scalar,
                                                    const float* frame12, int res, float ang_cut,
                                                    const float* grad_image, float* grad_x, float* grad_v, void* stream) {
  if (!frame12) return sensor_fail(DRRT_ERR_ARG, "null frame pointer");
  return far_bwd(n, v, e, e_scalar, kZero3, kZero3, frame12, res, ang_cut, grad_image, grad_x, grad_v, stream);
}

// ---- texture lookups at the sensor (core/sensor.py:102-138), see k_sensor_tex_get ------------------------------------
// mode 0: get_sdf_vals_near (rays -> plane -> sensor frame, cell size span / res);
// mode 1: get_sdf_vals_far  (coordinates v . T + ang_cut from the direction as it is, cell size 2 ang_cut / res; pass
//         span = 2 * ang_cut).
static int fill_tex(SensorArgs& a, size_t n, const float* x, const float* v, const float p[3], const float nrm[3],
                    const float t1[3], const float t2[3], const float* frame_dev, int res, float span, int mode,
                    const float* tex) {
  int rc = fill_args(a, n, x, v, nullptr, 1.f, p, nrm, t1, t2, res, span); if (rc) return rc;
  a.frame_dev = frame_dev;
  if (!tex) return sensor_fail(DRRT_ERR_ARG, "null texture pointer");
  if (mode != 0 && mode != 1) return sensor_fail(DRRT_ERR_ARG, "mode must be 0 (near) or 1 (far)");
  a.far = mode == 1 ? 2 : 0;
  a.grad_image = tex;
  return DRRT_OK;
}

static int tex_fwd(size_t n, const float* x, const float* v, const float plane_p[3], const float plane_n[3],
                   const float t1[3], const float t2[3], const float* frame_dev, const float* tex, int res, float span,
                   int mode, float* f_out, void* stream) {
  SensorArgs a{};
  int rc = fill_tex(a, n, x, v, plane_p, plane_n, t1, t2, frame_dev, res, span, mode, tex); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!f_out) return sensor_fail(DRRT_ERR_ARG, "null output pointer");
  a.f_out = f_out;
  hipLaunchKernelGGL(k_sensor_tex_get, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

extern "C" int drrt_sensor_tex_get_f32(size_t n, const float* x, const float* v, const float plane_p[3],
                                       const float plane_n[3], const float t1[3], const float t2[3], const float* tex,
                                       int res, float span, int mode, float* f_out, void* stream) {
  return tex_fwd(n, x, v, plane_p, plane_n, t1, t2, nullptr, tex, res, span, mode, f_out, stream);
}
extern "C" int drrt_sensor_tex_get_dframe_f32(size_t n, const float* x, const float* v, const float* frame12,
                                              const float* tex, int res, float span, int mode, float* f_out, void* stream) {
  if (!frame12) return sensor_fail(DRRT_ERR_ARG, "null frame pointer");
  return tex_fwd(n, x, v, kZero3, kZero3, kZero3, kZero3, frame12, tex, res, span, mode, f_out, stream);
}

static int tex_bwd(size_t n, const float* x, const float* v, const float plane_p[3], const float plane_n[3],
                   const float t1[3], const float t2[3], const float* frame_dev, const float* tex, int res, float span,
                   int mode, const float* grad_f, float* grad_x, float* grad_v, void* stream) {
  SensorArgs a{};
  int rc = fill_tex(a, n, x, v, plane_p, plane_n, t1, t2, frame_dev, res, span, mode, tex); if (rc) return rc;
  if (n == 0) return DRRT_OK;
  if (!grad_f || !grad_x || !grad_v) return sensor_fail(DRRT_ERR_ARG, "null gradient pointer");
  a.grad_f = grad_f; a.grad_x = grad_x; a.grad_v = grad_v;
  hipLaunchKernelGGL(k_sensor_tex_get_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

extern "C" int drrt_sensor_tex_get_bwd_f32(size_t n, const float* x, const float* v, const float plane_p[3],
                                           const float plane_n[3], const float t1[3], const float t2[3],
                                           const float* tex, int res, float span, int mode, const float* grad_f,
                                           float* grad_x, float* grad_v, void* stream) {
  return tex_bwd(n, x, v, plane_p, plane_n, t1, t2, nullptr, tex, res, span, mode, grad_f, grad_x, grad_v, stream);
}
extern "C" int drrt_sensor_tex_get_dframe_bwd_f32(size_t n, const float* x, const float* v, const float* frame12,
                                                  const float* tex, int res, float span, int mode, const float* grad_f,
                                                  float* grad_x, float* grad_v, void* stream) {
  if (!frame12) return sensor_fail(DRRT_ERR_ARG, "null frame pointer");
  return tex_bwd(n, x, v, kZero3, kZero3, kZero3, kZero3, frame12, tex, res, span, mode, grad_f, grad_x, grad_v, stream);
}

// =============================================================================================
// multires up-sampling of a volume (SURVEY.md 8.8 "next" row 3)
//
// Reference: core/optimizer.py:7-10 upres_scene -> core/grid.py:318-330 upres_volume: trilinear
// resampling of the (R,R,R) volume at linspace(0,1,S)^3 through Grid.GetLinear (:227-273), carried
// out in float64 and cast back.  The reference materialises an (S^3, 3) float64 point list plus ~20
// (S^3, 8)-sized temporaries (several GB at S = 256); this is one pass, one thread per output voxel.
// =============================================================================================
namespace drrt {

__global__ void __launch_bounds__(256) k_upres(const float* __restrict__ src, int r0, int r1, int r2,
                                               float* __restrict__ dst, int s0, int s1, int s2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)s0 * s1 * s2;
  if (i >= total) return;
  const int k = (int)(i % s2), j = (int)((i / s2) % s1), m = (int)(i / ((size_t)s2 * s1));
  const int idx[3] = {m, j, k}, sn[3] = {s0, s1, s2};
  const int rr = r0;                                   // the reference clips every axis with res[0] (grid.py:242)
  const double h = 1.0 / (double)(rr > 1 ? rr - 1 : 1);   // grid.py:319-320
  int i0[3], i1[3]; double w[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // torch.linspace(0, 1, s): step*i for the first half, 1 - step*(s-1-i) for the second
    const double step = sn[a] > 1 ? 1.0 / (double)(sn[a] - 1) : 0.0;
    const double x = (idx[a] < sn[a] / 2) ? step * idx[a] : 1.0 - step * (double)(sn[a] - 1 - idx[a]);
    const double nx = x / h;                           // grid.py:232
    const double fl = floor(nx);
    double ww = nx - fl; ww = ww < 0.0 ? 0.0 : (ww > 1.0 ? 1.0 : ww);   // :235
    w[a] = ww;
    const int b = (int)fl;
    i0[a] = min(max(b, 0), rr - 1); i1[a] = min(max(b + 1, 0), rr - 1);  // :243
  }
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int a0 = (c & 4) ? i1[0] : i0[0], a1 = (c & 2) ? i1[1] : i0[1], a2 = (c & 1) ? i1[2] : i0[2];
    const double ww = ((c & 4) ? w[0] : 1.0 - w[0]) * ((c & 2) ? w[1] : 1.0 - w[1]) * ((c & 1) ? w[2] : 1.0 - w[2]);
    acc += ww * (double)src[((size_t)a0 * r1 + a1) * r2 + a2];
  }
  dst[i] = (float)acc;
}

}  // namespace drrt

extern "C" int drrt_upres_volume_f32(const float* src, const int src_shape[3], float* dst, const int dst_shape[3],
                                     void* stream) {
  if (!src || !dst || !src_shape || !dst_shape) return sensor_fail(DRRT_ERR_ARG, "null pointer");
  for (int a = 0; a < 3; ++a)
    if (src_shape[a] < 1 || dst_shape[a] < 1) return sensor_fail(DRRT_ERR_ARG, "bad shape");
  if (src_shape[0] != src_shape[1] || src_shape[0] != src_shape[2])
    return sensor_fail(DRRT_ERR_ARG, "upres_volume expects a cubic source volume (the reference clips all axes with res[0])");
  const size_t total = (size_t)dst_shape[0] * dst_shape[1] * dst_shape[2];
  hipLaunchKernelGGL(k_upres, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     src_shape[0], src_shape[1], src_shape[2], dst, dst_shape[0], dst_shape[1], dst_shape[2]);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

// =============================================================================================
// one optimiser iteration's tail (SURVEY.md 8.8 "next" row 3): boundary-gradient mask + Adam + clamp
//
// Reference: core/optimizer.py:57-69 -- `n.grad[mask] = 0` (mask = the outermost voxel layer, :54-55),
// `opto.step()` with torch.optim.Adam, `n.clamp_(min=1)`.  In torch that is a boolean-mask index_put (a nonzero()
// with a host sync), ~10 element-wise launches and three extra passes over the volume and its two moments; here it
// is ONE pass: 16 B read + 12 B written per voxel.  The update is torch's Adam (torch/optim/adam.py
// _single_tensor_adam, amsgrad = maximize = False), bias corrections computed by the caller in double:
//   g     = grad (0 on the boundary layer; written back there like the reference's in-place mask) + weight_decay * p
//   m    += (g - m) * (1 - beta1)                      (exp_avg.lerp_)
//   v     = beta2 * v + (1 - beta2) * g * g
//   p    -= step_size * m / (sqrt(v) / sqrt(bias_correction2) + eps),   step_size = lr / bias_correction1
//   p     = p < clamp_min ? clamp_min : p              (NaN stays NaN, like clamp_)
// =============================================================================================
namespace drrt {

struct AdamArgs {
  float* p; float* g; float* m; float* v;
  size_t n; int s0, s1, s2;             // torch shape (z, y, x): x fastest
  float step_size, bc2_sqrt, beta2, omb1, omb2, eps, weight_decay, clamp_min;   // omb = 1 - beta, rounded from double like torch's scalars
  int mask_boundary, clamp;
};

__global__ void __launch_bounds__(256) k_adam_masked(AdamArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  float g = a.g[i];
  if (a.mask_boundary) {
    const int x = (int)(i % (size_t)a.s2), y = (int)((i / (size_t)a.s2) % (size_t)a.s1), z = (int)(i / ((size_t)a.s2 * a.s1));
    if ((x == 0) | (x == a.s2 - 1) | (y == 0) | (y == a.s1 - 1) | (z == 0) | (z == a.s0 - 1)) { g = 0.f; a.g[i] = 0.f; }
  }
  float p = a.p[i], m = a.m[i], v = a.v[i];
  if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
  m = fmaf(g - m, a.omb1, m);
  v = fmaf(a.omb2, g * g, a.beta2 * v);
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = fmaf(-a.step_size, m / denom, p);
  if (a.clamp) p = (p < a.clamp_min) ? a.clamp_min : p;
  a.p[i] = p; a.m[i] = m; a.v[i] = v;
}

}  // namespace drrt

extern "C" int drrt_adam_step_f32(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const int shape[3],
                                  double step, double lr, double beta1, double beta2, double eps, double weight_decay,
                                  double clamp_min, unsigned flags, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !shape) return sensor_fail(DRRT_ERR_ARG, "null pointer");
  for (int k = 0; k < 3; ++k) if (shape[k] < 1) return sensor_fail(DRRT_ERR_ARG, "bad shape");
  if (!(step >= 1.0)) return sensor_fail(DRRT_ERR_ARG, "step must be >= 1 (the value AFTER the increment, like torch's)");
  drrt::AdamArgs a{};
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq;
  a.s0 = shape[0]; a.s1 = shape[1]; a.s2 = shape[2];
  a.n = (size_t)shape[0] * shape[1] * shape[2];
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);       // torch/optim/adam.py
  a.step_size = (float)(lr / bc1); a.bc2_sqrt = (float)sqrt(bc2);
  a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps; a.weight_decay = (float)weight_decay;
  a.clamp_min = (float)clamp_min;
  a.mask_boundary = (flags & DRRT_ADAM_MASK_BOUNDARY) ? 1 : 0;
  a.clamp = (flags & DRRT_ADAM_CLAMP_MIN) ? 1 : 0;
  hipLaunchKernelGGL(drrt::k_adam_masked, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

// =============================================================================================
// ray -> plane intersection as its own operator (the statement right after the march)
//
// Reference: core/sensor.py:195-202 trace_rays_to_plane: t = n.(p - x) / n.v ; x_out = x + t v ; v unchanged --
// written with torch.matmul on (N,1,3) x (N,3,1) operands, i.e. a batched matmul of N one-by-three products, which on
// the GPU costs ~25 ms forward and ~55 ms backward for 1M rays (tools/bench_iteration.py): 15x the march itself.
// Here: one thread per ray, forward and analytic backward (gradients w.r.t. the rays; the planes are constants in
// every experiment of the reference -- the Python wrapper falls back to the torch expressions if they require grad):
//   a = n.(p - x), b = n.v, t = a / b
//   d x_out / d x = I - v n^T / b            d x_out / d v = t I - (t / b) v n^T
//   => gx = g - (g.v / b) n                  gv = t g - (t / b)(g.v) n          (g = dL/dx_out)
// plane_stride = 3: one plane per ray; 0: one plane for all rays.
// =============================================================================================
namespace drrt {

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

__global__ void __launch_bounds__(256) k_rays_to_plane(size_t n, const float* __restrict__ x, const float* __restrict__ v,
                                                       const float* __restrict__ p, const float* __restrict__ nr,
                                                       int plane_stride, float* __restrict__ xo) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t k = i * (size_t)plane_stride;
  const float x0 = x[3 * i], x1 = x[3 * i + 1], x2 = x[3 * i + 2], v0 = v[3 * i], v1 = v[3 * i + 1], v2 = v[3 * i + 2];
  const float n0 = nr[k], n1 = nr[k + 1], n2 = nr[k + 2];
  const float a = dot3(n0, n1, n2, p[k] - x0, p[k + 1] - x1, p[k + 2] - x2);        // :199
  const float t = a / dot3(n0, n1, n2, v0, v1, v2);                                   // :200
  xo[3 * i] = fmaf(t, v0, x0); xo[3 * i + 1] = fmaf(t, v1, x1); xo[3 * i + 2] = fmaf(t, v2, x2);   // :202
}

__global__ void __launch_bounds__(256) k_rays_to_plane_bwd(size_t n, const float* __restrict__ x, const float* __restrict__ v,
                                                           const float* __restrict__ p, const float* __restrict__ nr,
                                                           int plane_stride, const float* __restrict__ g,
                                                           float* __restrict__ gx, float* __restrict__ gv) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t k = i * (size_t)plane_stride;
  const float x0 = x[3 * i], x1 = x[3 * i + 1], x2 = x[3 * i + 2], v0 = v[3 * i], v1 = v[3 * i + 1], v2 = v[3 * i + 2];
  const float n0 = nr[k], n1 = nr[k + 1], n2 = nr[k + 2];
  const float g0 = g[3 * i], g1 = g[3 * i + 1], g2 = g[3 * i + 2];
  const float a = dot3(n0, n1, n2, p[k] - x0, p[k + 1] - x1, p[k + 2] - x2);
  const float inv_b = 1.f / dot3(n0, n1, n2, v0, v1, v2);
  const float t = a * inv_b;
  const float c = dot3(g0, g1, g2, v0, v1, v2) * inv_b;       // g.v / b
  gx[3 * i] = fmaf(-c, n0, g0); gx[3 * i + 1] = fmaf(-c, n1, g1); gx[3 * i + 2] = fmaf(-c, n2, g2);
  const float tc = t * c;
  gv[3 * i] = fmaf(t, g0, -tc * n0); gv[3 * i + 1] = fmaf(t, g1, -tc * n1); gv[3 * i + 2] = fmaf(t, g2, -tc * n2);
}

}  // namespace drrt

extern "C" int drrt_rays_to_plane_f32(size_t n, const float* x, const float* v, const float* plane_p, const float* plane_n,
                                      int plane_stride, float* x_out, void* stream) {
  if (plane_stride != 0 && plane_stride != 3) return sensor_fail(DRRT_ERR_ARG, "plane_stride must be 0 or 3");
  if (n == 0) return DRRT_OK;
  if (!x || !v || !plane_p || !plane_n || !x_out) return sensor_fail(DRRT_ERR_ARG, "null pointer");
  hipLaunchKernelGGL(drrt::k_rays_to_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, x, v,
                     plane_p, plane_n, plane_stride, x_out);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

extern "C" int drrt_rays_to_plane_bwd_f32(size_t n, const float* x, const float* v, const float* plane_p,
                                          const float* plane_n, int plane_stride, const float* grad_x_out,
                                          float* grad_x, float* grad_v, void* stream) {
  if (plane_stride != 0 && plane_stride != 3) return sensor_fail(DRRT_ERR_ARG, "plane_stride must be 0 or 3");
  if (n == 0) return DRRT_OK;
  if (!x || !v || !plane_p || !plane_n || !grad_x_out || !grad_x || !grad_v) return sensor_fail(DRRT_ERR_ARG, "null pointer");
  hipLaunchKernelGGL(drrt::k_rays_to_plane_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, x,
                     v, plane_p, plane_n, plane_stride, grad_x_out, grad_x, grad_v);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}

