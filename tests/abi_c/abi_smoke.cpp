// abi_smoke.cpp -- the drop-in boundary exercised WITHOUT python or torch (TEST INFRASTRUCTURE):
// a plain C++ program that loads libdrrt_hip.so through its C ABI (include/drrt_hip.h), feeds it buffers from
// hipMalloc on the default stream, and checks trace + backtrace against the CPU oracle's C library in
// `factored` arithmetic (exit rays bit-exact, gradient to summation-order tolerance).
//   usage: abi_smoke <libdrrt_hip.so> <libdrrt_oracle.so>
// Built by tests/test_abi_c.py:  hipcc -O2 -std=c++17 abi_smoke.cpp -ldl -o _build/abi_smoke
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/drrt_hip.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

typedef size_t (*ws_fn)(size_t, unsigned);
typedef const char* (*str_fn)(void);
typedef int (*trace_fn)(const float*, long long, const int*, size_t, const float*, const float*, float, float, float*,
                        float*, drrt_stats*, void*, size_t, unsigned, void*);
typedef int (*back_fn)(const float*, long long, const int*, size_t, const float*, const float*, const float*,
                       const float*, float, float, float*, drrt_stats*, void*, size_t, unsigned, void*);
typedef void (*arith_fn)(int);
typedef int (*otrace_fn)(const float*, const int*, long long, size_t, const float*, const float*, float, float, float*,
                         float*, int*, long long*, int*);
typedef int (*oback_fn)(const float*, const int*, long long, size_t, const float*, const float*, const float*,
                        const float*, float, float, float, float*, long long*);

template <typename T> static T sym(void* h, const char* name) {
  void* p = dlsym(h, name);
  if (!p) { printf("missing symbol %s\n", name); exit(3); }
  return (T)p;
}

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: %s libdrrt_hip.so libdrrt_oracle.so\n", argv[0]); return 1; }
  void* L = dlopen(argv[1], RTLD_NOW);
  void* O = dlopen(argv[2], RTLD_NOW);
  if (!L || !O) { printf("dlopen failed: %s\n", dlerror()); return 1; }
  auto ws_bytes = sym<ws_fn>(L, "drrt_workspace_bytes");
  auto version = sym<str_fn>(L, "drrt_version");
  auto last_error = sym<str_fn>(L, "drrt_last_error");
  auto trace = sym<trace_fn>(L, "drrt_trace_f32");
  auto back = sym<back_fn>(L, "drrt_backtrace_f32");
  auto set_arith = sym<arith_fn>(O, "oracle_set_arith");
  auto otrace = sym<otrace_fn>(O, "oracle_trace_f32");
  auto oback = sym<oback_fn>(O, "oracle_backtrace_f32");
  printf("library: %s\n", version());

  // Luneburg ball on a 33^3 grid, 4096 slightly tilted rays from the y = 0 face
  const int R = 33; const int res[3] = {R, R, R};
  const long long nvox = (long long)R * R * R;
  const float span = 1.0f, h = span / (R - 1), ds = h / 2;
  std::vector<float> rif(nvox);
  for (int z = 0; z < R; ++z) for (int y = 0; y < R; ++y) for (int x = 0; x < R; ++x) {
    const double dx = x * (double)h - 0.5, dy = y * (double)h - 0.5, dz = z * (double)h - 0.5;
    const double r = fmin(sqrt(dx * dx + dy * dy + dz * dz) / 0.5, 1.0);
    rif[((size_t)z * R + y) * R + x] = (float)sqrt(2.0 - r * r);
  }
  const size_t n = 4096;
  std::vector<float> pos(3 * n), vel(3 * n), seed(3 * n);
  unsigned s = 12345u;
  auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffffff) / 16777216.0f; };
  for (size_t i = 0; i < n; ++i) {
    pos[3 * i] = 0.05f + 0.9f * rnd(); pos[3 * i + 1] = -0.3f * ds; pos[3 * i + 2] = 0.05f + 0.9f * rnd();
    float vx = 0.1f * (rnd() - 0.5f), vz = 0.1f * (rnd() - 0.5f), nr = sqrtf(vx * vx + 1.f + vz * vz);
    vel[3 * i] = vx / nr; vel[3 * i + 1] = 1.f / nr; vel[3 * i + 2] = vz / nr;
    seed[3 * i] = rnd() - 0.5f; seed[3 * i + 1] = rnd() - 0.5f; seed[3 * i + 2] = rnd() - 0.5f;
  }

  float *d_rif, *d_pos, *d_vel, *d_xt, *d_vt, *d_dx, *d_grad; drrt_stats* d_st; void* d_ws;
  const unsigned flags = DRRT_FLAG_SORT_RAYS;
  const size_t wsb = ws_bytes(n, flags);
  HIPCHECK(hipMalloc(&d_rif, nvox * 4)); HIPCHECK(hipMalloc(&d_pos, n * 12)); HIPCHECK(hipMalloc(&d_vel, n * 12));
  HIPCHECK(hipMalloc(&d_xt, n * 12)); HIPCHECK(hipMalloc(&d_vt, n * 12)); HIPCHECK(hipMalloc(&d_dx, n * 12));
  HIPCHECK(hipMalloc(&d_grad, nvox * 4)); HIPCHECK(hipMalloc(&d_st, sizeof(drrt_stats))); HIPCHECK(hipMalloc(&d_ws, wsb));
  HIPCHECK(hipMemcpy(d_rif, rif.data(), nvox * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_pos, pos.data(), n * 12, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_vel, vel.data(), n * 12, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_dx, seed.data(), n * 12, hipMemcpyHostToDevice));

  int rc = trace(d_rif, nvox, res, n, d_pos, d_vel, h, ds, d_xt, d_vt, d_st, d_ws, wsb, flags, nullptr);
  if (rc) { printf("drrt_trace_f32 failed: %s\n", last_error()); return 4; }
  std::vector<float> xt(3 * n), vt(3 * n); drrt_stats st;
  HIPCHECK(hipMemcpy(xt.data(), d_xt, n * 12, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(vt.data(), d_vt, n * 12, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(&st, d_st, sizeof(st), hipMemcpyDeviceToHost));

  set_arith(1);                                   // factored: the op sequence the kernels implement
  std::vector<float> oxt(3 * n), ovt(3 * n); std::vector<int> osteps(n); long long nf = 0; int iters = 0;
  otrace(rif.data(), res, nvox, n, pos.data(), vel.data(), h, ds, oxt.data(), ovt.data(), osteps.data(), &nf, &iters);
  long long osum = 0; for (size_t i = 0; i < n; ++i) osum += osteps[i];
  const bool fwd_ok = !memcmp(xt.data(), oxt.data(), n * 12) && !memcmp(vt.data(), ovt.data(), n * 12) &&
                      (long long)st.ray_steps == osum && (long long)st.n_failed == nf && (int)st.iters == iters;
  printf("forward: exit rays bit-exact %d, ray_steps %llu (oracle %lld), iters %u (oracle %d)\n", (int)fwd_ok,
         st.ray_steps, osum, st.iters, iters);

  rc = back(d_rif, nvox, res, n, d_xt, d_vt, d_dx, d_dx, h, ds, d_grad, d_st, d_ws, wsb, flags, nullptr);
  if (rc) { printf("drrt_backtrace_f32 failed: %s\n", last_error()); return 4; }
  std::vector<float> grad(nvox), ograd(nvox, 0.f); long long ost = 0;
  HIPCHECK(hipMemcpy(grad.data(), d_grad, nvox * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(&st, d_st, sizeof(st), hipMemcpyDeviceToHost));
  oback(rif.data(), res, nvox, n, oxt.data(), ovt.data(), seed.data(), seed.data(), h, ds, 1.0f, ograd.data(), &ost);
  double num = 0, den = 0;
  for (long long k = 0; k < nvox; ++k) { const double d = (double)grad[k] - ograd[k]; num += d * d; den += (double)ograd[k] * ograd[k]; }
  const double rel = sqrt(num / (den > 0 ? den : 1));
  const bool adj_ok = rel <= 2e-5 && (long long)st.ray_steps == ost;
  printf("adjoint: rel-L2 %.3e, ray_steps %llu (oracle %lld)\n", rel, st.ray_steps, ost);

  // error contract: a resolution that does not match the data is refused with the reference's message
  const int bad[3] = {R, R, R + 1};
  rc = trace(d_rif, nvox, bad, n, d_pos, d_vel, h, ds, d_xt, d_vt, d_st, d_ws, wsb, flags, nullptr);
  const bool err_ok = rc == DRRT_ERR_RES_MISMATCH && strstr(last_error(), "Resolution doesn't match data");
  printf("error contract: rc %d \"%s\"\n", rc, last_error());
  HIPCHECK(hipDeviceSynchronize());
  printf(fwd_ok && adj_ok && err_ok ? "ABI_SMOKE_OK\n" : "ABI_SMOKE_FAILED\n");
  return fwd_ok && adj_ok && err_ok ? 0 : 5;
}
