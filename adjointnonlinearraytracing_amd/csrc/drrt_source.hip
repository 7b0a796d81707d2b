// drrt_source.hip -- ray generation on the device ("next" row 2): the step
// that precedes the march in every 3-D script of the reference.
//
// Reference semantics (all torch on the host, /root/reference/core/source.py):
//   :54-69    plane_source3_rand     jittered (or independent) points on the y = 0 plane, spp samples
//                                    per pixel, candidate order (s, i, j)
//   :72-104   point_source3_rand     rays from the point (0, -w/2, 0) through jittered pixel centres of the plane
//                                    y = w/2 (kind 1; :360-365 rand_ptrays_in_sphere concatenates the views)
//   :275-293  rotate_pts_to_source   optional disc mask r < width/2 (order-preserving compaction),
//                                    x = p R^T + width/2 - width v/2, v = R e_y, t = R e_z,
//                                    sensor plane (point, normal, tangent) per ray
//   :303-312  rotate_ray3            R about z (or about x when `vert`), entries rounded to fp32
//   :352-357  rand_rays_in_sphere    views at linspace(0, angle_span, nviews+1)[:-1], concatenated
//   :398-412  rand_rays_cube         four views about z + two about x, concatenated
//   :555-563  random_rotate_ic       a second rotation M about the cube centre of x, v and the planes
// The reference builds the rays with ~15 host tensors per view, a boolean-mask gather, two (N,3)x(3,3)
// matmuls per quantity and a torch.cat over views, then uploads (N,3)+(N,3)+(N,3,3) floats every
// iteration.  Here all views of a call are three launches: per-block keep counts, one scan of the
// counts, and the write pass that recomputes the candidate (cheaper than storing it) and emits x, v and
// the planes at its compacted position.  Only the jitter uniforms are read: 8 B per candidate in, 60 B
// per kept ray out -- an HBM-write-bound kernel.
//
// Arithmetic follows the reference's fp32 expression order (built with -ffp-contract=off); the two
// matmuls are accumulated left to right, which can differ from the host BLAS by an ulp -- the parity
// tolerance is stated in tests/test_source.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/drrt_hip.h"

namespace drrt {

int sensor_fail(int code, const char* msg);   // drrt_kernels.hip

constexpr int GEN_BLOCK = 256;
constexpr int GEN_ITEMS = 4;                         // consecutive 256-candidate slabs per block
constexpr int GEN_CHUNK = GEN_BLOCK * GEN_ITEMS;

struct GenArgs {
  const float* u;          // (V, 2*spp, P0, P1) uniforms in [0,1)
  const float* view_rot;   // (V, 9) row-major
  float ic_rot[9];         // random_rotate_ic matrix (identity when !has_ic)
  int has_ic;
  int n_views, spp, p0, p1;
  int circle, independent;
  int kind;                // 0 = plane source (source.py:54-69), 1 = point source (:72-104), 2 = cone source (:186-203)
  float cone_cos;          // kind 2: cos(cone_angle / 2), evaluated by the caller in fp32 as hatbox_sample does (:533-534)
  float width, half_width, plane_scale, half_span;
  unsigned cand_per_view, blocks_per_view;
  float* x; float* v; float* planes;
  int* block_counts;       // (V*blocks_per_view + 1) kept rays per block, then their exclusive scan
  int* view_counts;        // (V + 1) exclusive prefix of kept rays per view
};

// Candidate c of one view -> point on the source plane (px, 0, pz) and the disc-mask decision.
__device__ __forceinline__ bool gen_point(const GenArgs& a, int view, unsigned c, float& px, float& pz) {
  const unsigned pp = (unsigned)a.p0 * (unsigned)a.p1;
  const unsigned s = c / pp, rem = c - s * pp;
  const unsigned i = rem / (unsigned)a.p1, j = rem - i * (unsigned)a.p1;
  const float* uv = a.u + (size_t)view * 2u * a.spp * pp;
  if (a.kind == 2) {                          // cone source: every candidate is a ray; px, pz carry its two uniforms
    px = uv[c];                               // hatbox_sample's first torch.rand(N): z   (source.py:535)
    pz = uv[(size_t)a.cand_per_view + c];     // its second: theta                        (:536)
    return true;
  }
  const float o0 = uv[(size_t)s * pp + rem] * a.width;                       // source.py:56
  const float o1 = uv[(size_t)(a.spp + s) * pp + rem] * a.width;
  if (a.kind == 1) {                                                         // point source, :73-83
    // jitter is (u - 0.5) in ABSOLUTE units around the pixel centres, as written in the reference
    const float r0 = a.width * (((float)i + 0.5f) / (float)a.p0 - 0.5f);
    const float r1 = a.width * (((float)j + 0.5f) / (float)a.p1 - 0.5f);
    px = r0 + (uv[(size_t)s * pp + rem] - 0.5f);
    pz = r1 + (uv[(size_t)(a.spp + s) * pp + rem] - 0.5f);
  } else if (a.independent) {                                                // :61-63
    px = o0 - a.half_width;
    pz = o1 - a.half_width;
  } else {                                                                   // :57-58, 65-68
    const float r0 = a.width * ((float)i / (float)a.p0 - 0.5f);
    const float r1 = a.width * ((float)j / (float)a.p1 - 0.5f);
    px = r0 + o0 / (float)a.p0;
    pz = r1 + o1 / (float)a.p1;
  }
  if (!a.circle) return true;
  return sqrtf(px * px + pz * pz) < a.half_width;                            // :278-280 / :81-83
}

__device__ __forceinline__ void mat3(const float* R, const float p[3], float out[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = (p[0] * R[3 * k] + p[1] * R[3 * k + 1]) + p[2] * R[3 * k + 2];
}

// Block-wide exclusive scan of a 0/1 flag over 256 threads (4 waves): returns this thread's rank
// among the kept and the block total.
__device__ __forceinline__ unsigned block_rank(bool keep, unsigned& total, unsigned* wave_tot) {
  const unsigned long long m = __ballot(keep);
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned below = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[w] = __popcll(m);
  __syncthreads();
  unsigned base = 0;
#pragma unroll
  for (unsigned k = 0; k < GEN_BLOCK / 64; ++k) base += (k < w) ? wave_tot[k] : 0u;
  total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  __syncthreads();
  return base + below;
}

__global__ void __launch_bounds__(GEN_BLOCK) k_gen_count(GenArgs a) {
  __shared__ unsigned wave_tot[GEN_BLOCK / 64];
  const int view = blockIdx.y;
  unsigned kept = 0;
#pragma unroll
  for (int it = 0; it < GEN_ITEMS; ++it) {
    const unsigned c = blockIdx.x * GEN_CHUNK + it * GEN_BLOCK + threadIdx.x;
    float px, pz;
    const bool keep = c < a.cand_per_view && gen_point(a, view, c, px, pz);
    unsigned tot;
    block_rank(keep, tot, wave_tot);
    kept += tot;
  }
  if (threadIdx.x == 0) a.block_counts[(size_t)view * a.blocks_per_view + blockIdx.x] = (int)kept;
}

// Exclusive scan of the per-block counts (a few thousand entries): one block, running carry.
__global__ void __launch_bounds__(1024) k_gen_scan(int* counts, unsigned n_blocks, unsigned blocks_per_view,
                                                   int* view_counts, int n_views) {
  __shared__ int part[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (unsigned base = 0; base < n_blocks; base += 1024) {
    const unsigned i = base + threadIdx.x;
    const int mine = i < n_blocks ? counts[i] : 0;
    part[threadIdx.x] = mine;
    __syncthreads();
    for (unsigned off = 1; off < 1024; off <<= 1) {           // Hillis-Steele inclusive scan
      const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    const int excl = carry + part[threadIdx.x] - mine;
    if (i < n_blocks) {
      counts[i] = excl;
      if (i % blocks_per_view == 0) view_counts[i / blocks_per_view] = excl;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) { counts[n_blocks] = carry; view_counts[n_views] = carry; }
}

__global__ void __launch_bounds__(GEN_BLOCK) k_gen_write(GenArgs a) {
  __shared__ unsigned wave_tot[GEN_BLOCK / 64];
  const int view = blockIdx.y;
  const float* R = a.view_rot + 9 * view;
  // per-view constants: v = R e_y, t = R e_z (exact), plane point (source.py:281-292)
  float vdir[3], tdir[3], pl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    vdir[k] = R[3 * k + 1];
    tdir[k] = R[3 * k + 2];
    pl[k] = (a.kind == 1) ? (a.plane_scale * vdir[k]) / 2.f + a.half_width       // :102
                          : a.plane_scale * vdir[k] + a.half_width;            // :290
  }
  float vv[3] = {vdir[0], vdir[1], vdir[2]}, tt[3] = {tdir[0], tdir[1], tdir[2]}, pp[3] = {pl[0], pl[1], pl[2]};
  if (a.has_ic) {                                                            // source.py:557-561
    const float q[3] = {pl[0] - a.half_span, pl[1] - a.half_span, pl[2] - a.half_span};
    mat3(a.ic_rot, q, pp);
#pragma unroll
    for (int k = 0; k < 3; ++k) pp[k] += a.half_span;
    mat3(a.ic_rot, vdir, vv);
    mat3(a.ic_rot, tdir, tt);
  }
  unsigned out_base = (unsigned)a.block_counts[(size_t)view * a.blocks_per_view + blockIdx.x];
#pragma unroll 1
  for (int it = 0; it < GEN_ITEMS; ++it) {
    const unsigned c = blockIdx.x * GEN_CHUNK + it * GEN_BLOCK + threadIdx.x;
    float px = 0.f, pz = 0.f;
    const bool keep = c < a.cand_per_view && gen_point(a, view, c, px, pz);
    unsigned tot;
    const unsigned rank = block_rank(keep, tot, wave_tot);
    if (keep) {
      float x[3], vray[3] = {vv[0], vv[1], vv[2]};
      if (a.kind == 2) {
        // hatbox_sample (source.py:531-545) around e_y with basis e_z: t1 = e_z x e_y = -e_x, t2 = t1 x e_y = -e_z,
        // so the direction is (-cos(theta) s, z, -sin(theta) s) with z uniform in [cos(cone/2), 1), s = sqrt(1 - z^2)
        const float zc = px * (1.f - a.cone_cos) + a.cone_cos;
        const float th = 6.283185307179586f * pz;
        const float sc = sqrtf(1.f - zc * zc);
        const float d[3] = {-(cosf(th) * sc), zc, -(sinf(th) * sc)};
        float vr[3];
        mat3(R, d, vr);                                                      // :192
        const float o[3] = {0.f, -a.half_width, 0.f};
        mat3(R, o, x);                                                       // :191
#pragma unroll
        for (int k = 0; k < 3; ++k) { x[k] += a.half_width; vray[k] = vr[k]; }
        if (a.has_ic) mat3(a.ic_rot, vr, vray);
      } else if (a.kind == 1) {
        // direction (px, width, pz) normalised (:85-89), rotated (:96); origin R (0, -w/2, 0) + w/2 (:94-95)
        const float nrm = sqrtf((px * px + a.width * a.width) + pz * pz);
        const float d[3] = {px / nrm, a.width / nrm, pz / nrm};
        float vr[3];
        mat3(R, d, vr);
        const float o[3] = {0.f, -a.half_width, 0.f};
        mat3(R, o, x);
#pragma unroll
        for (int k = 0; k < 3; ++k) { x[k] += a.half_width; vray[k] = vr[k]; }
        if (a.has_ic) mat3(a.ic_rot, vr, vray);
      } else {
        const float p[3] = {px, 0.f, pz};
        mat3(R, p, x);                                                       // :284
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          x[k] += a.half_width;
          x[k] -= (a.width * vdir[k]) / 2.f;                                 // :287
        }
      }
      if (a.has_ic) {
        const float q[3] = {x[0] - a.half_span, x[1] - a.half_span, x[2] - a.half_span};
        mat3(a.ic_rot, q, x);
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] += a.half_span;
      }
      const size_t o = (size_t)out_base + rank;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a.x[3 * o + k] = x[k];
        a.v[3 * o + k] = vray[k];
        a.planes[9 * o + k] = pp[k];
        a.planes[9 * o + 3 + k] = vv[k];
        a.planes[9 * o + 6 + k] = tt[k];
      }
    }
    out_base += tot;
  }
}

}  // namespace drrt

extern "C" size_t drrt_gen_workspace_bytes(int n_views, int spp, int p0, int p1) {
  if (n_views < 1 || spp < 1 || p0 < 1 || p1 < 1) return 0;
  const size_t cand = (size_t)spp * p0 * p1;
  const size_t bpv = (cand + drrt::GEN_CHUNK - 1) / drrt::GEN_CHUNK;
  return ((size_t)n_views * bpv + 1) * sizeof(int);
}

static int gen_rays_impl(int kind, double cone_cos, const float* u, const float* view_rot, int n_views, int spp, int p0, int p1,
                         double width, double sensor_dist, int circle, int independent,
                         const float* ic_rot_host, double span, float* x, float* v, float* planes,
                         int* view_counts, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int drrt_gen_rays_f32(int kind, const float* u, const float* view_rot, int n_views, int spp, int p0, int p1,
                                 double width, double sensor_dist, int circle, int independent,
                                 const float* ic_rot_host, double span, float* x, float* v, float* planes,
                                 int* view_counts, void* workspace, size_t workspace_bytes, void* stream) {
  if (kind != 0 && kind != 1) return drrt::sensor_fail(DRRT_ERR_ARG, "source kind must be 0 (plane) or 1 (point)");
  return gen_rays_impl(kind, 1.0, u, view_rot, n_views, spp, p0, p1, width, sensor_dist, circle, independent, ic_rot_host,
                       span, x, v, planes, view_counts, workspace, workspace_bytes, stream);
}

extern "C" int drrt_gen_cone_rays_f32(const float* u, const float* view_rot, int n_views, int spp, int p0, int p1,
                                      double width, double sensor_dist, double cone_cos, const float* ic_rot_host,
                                      double span, float* x, float* v, float* planes, int* view_counts, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (!(cone_cos >= -1.0 && cone_cos <= 1.0)) return drrt::sensor_fail(DRRT_ERR_ARG, "cone_cos must lie in [-1, 1]");
  return gen_rays_impl(2, cone_cos, u, view_rot, n_views, spp, p0, p1, width, sensor_dist, 0, 0, ic_rot_host, span, x, v,
                       planes, view_counts, workspace, workspace_bytes, stream);
}

static int gen_rays_impl(int kind, double cone_cos, const float* u, const float* view_rot, int n_views, int spp, int p0, int p1,
                         double width, double sensor_dist, int circle, int independent,
                         const float* ic_rot_host, double span, float* x, float* v, float* planes,
                         int* view_counts, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace drrt;
  if (!u || !view_rot || !x || !v || !planes || !view_counts || !workspace)
    return sensor_fail(DRRT_ERR_ARG, "null pointer");
  if (n_views < 1 || n_views > 65535 || spp < 1 || p0 < 1 || p1 < 1) return sensor_fail(DRRT_ERR_ARG, "bad view / pixel counts");
  const unsigned long long cand = (unsigned long long)spp * p0 * p1;
  if (cand * (unsigned long long)n_views >= (1ull << 31)) return sensor_fail(DRRT_ERR_ARG, "too many candidate rays (>= 2^31)");
  if (!(width > 0.0)) return sensor_fail(DRRT_ERR_ARG, "width must be positive");
  if (workspace_bytes < drrt_gen_workspace_bytes(n_views, spp, p0, p1)) return sensor_fail(DRRT_ERR_ARG, "workspace too small");
  GenArgs a;
  a.u = u; a.view_rot = view_rot;
  a.has_ic = ic_rot_host != nullptr;
  for (int k = 0; k < 9; ++k) a.ic_rot[k] = a.has_ic ? ic_rot_host[k] : ((k % 4 == 0) ? 1.f : 0.f);
  a.n_views = n_views; a.spp = spp; a.p0 = p0; a.p1 = p1;
  a.circle = circle != 0; a.independent = independent != 0;
  // python scalars of the reference are doubles, rounded to fp32 when they meet an fp32 tensor
  a.width = (float)width; a.half_width = (float)(width / 2.0);
  a.kind = kind;
  a.cone_cos = (float)cone_cos;
  a.plane_scale = kind == 1 ? (float)(sensor_dist * width)                    // source.py:102
                            : (float)(sensor_dist + width / 2.0);             // source.py:290
  a.half_span = (float)(span / 2.0);
  a.cand_per_view = (unsigned)cand;
  a.blocks_per_view = (unsigned)((cand + GEN_CHUNK - 1) / GEN_CHUNK);
  a.x = x; a.v = v; a.planes = planes;
  a.block_counts = (int*)workspace; a.view_counts = view_counts;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(a.blocks_per_view, (unsigned)n_views);
  hipLaunchKernelGGL(k_gen_count, grid, dim3(GEN_BLOCK), 0, s, a);
  hipLaunchKernelGGL(k_gen_scan, dim3(1), dim3(1024), 0, s, a.block_counts, a.blocks_per_view * (unsigned)n_views,
                     a.blocks_per_view, view_counts, n_views);
  hipLaunchKernelGGL(k_gen_write, grid, dim3(GEN_BLOCK), 0, s, a);
  hipError_t le = hipGetLastError();
  return le == hipSuccess ? DRRT_OK : sensor_fail(DRRT_ERR_HIP, hipGetErrorString(le));
}
