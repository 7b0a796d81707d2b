/*
 * drrt_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (parity oracle, never shipped as product).
 *
 * Plain-C, array-at-a-time CPU restatement of the reference hot path of
 * ArjunTeh/AdjointNonlinearRayTracing.  This file is included twice by
 * drrt_oracle.c, once with REAL=float (suffix _f32) and once with REAL=double
 * (suffix _f64).  Every function cites the reference file:line it follows; the
 * arithmetic keeps the reference's expression order (no factoring), because this
 * is the statement of WHAT the reference computes, not a fast implementation.
 *
 * Parity status: the reference's native path depends on enoki (un-vendored, empty
 * submodule, version unpinned) and the repository holds no tests or golden vectors
 * => "parity unpinned" against the native code.  The pieces that CAN be pinned are
 * pinned by tests/golden (generated from the reference's importable torch helpers):
 * eval_grad vs core/grid.py Grid.GetLinear, cylinder eval_grad vs core/cable.py
 * Cable.GetLinear, and the adjoint vs torch.autograd through a torch restatement.
 *
 * Conventions:
 *   - ray arrays are (N,3) row-major (as torch hands them over);
 *   - flat voxel index = (z*H + y)*W + x with (W,H,D) = res[0..2]  (Q2);
 *   - masked gather returns 0 for masked-out lanes (enoki semantics).
 */

#ifndef REAL
#error "include from drrt_oracle.c"
#endif

/* ---- helpers ------------------------------------------------------------------- */

static inline int FN(clampi)(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef struct {
  int idx[8];      /* 000,100,010,110,001,101,011,111  (bit0 = x, bit1 = y, bit2 = z) */
  REAL w0[3], w1[3];
} FN(cell_t);

/* index + weight computation shared by eval_grad / eval_hess / splat
 * reference: src/volume.cpp:128-141 (== :55-68 == :200-214)                         */
static inline void FN(locate)(const int res[3], REAL h, const REAL p[3], FN(cell_t)* c) {
  const int width = res[0], height = res[1];
  const REAL rh = (REAL)1 / h;                         /* rcp(h_)  (:128)           */
  int pos0[3], pos1[3];
  for (int a = 0; a < 3; ++a) {
    REAL pm = p[a] * rh;
    REAL fl = FLOOR(pm);
    int ip = (int)fl;                                  /* floor2int (:129)          */
    c->w0[a] = pm - (REAL)ip;                          /* w0 = pm - pos (:130)      */
    c->w1[a] = (REAL)1 - c->w0[a];
    pos0[a] = FN(clampi)(ip, 0, res[a] - 1);           /* :131                      */
    pos1[a] = FN(clampi)(ip + 1, 0, res[a] - 1);       /* :132                      */
  }
  for (int k = 0; k < 8; ++k) {
    int x = (k & 1) ? pos1[0] : pos0[0];
    int y = (k & 2) ? pos1[1] : pos0[1];
    int z = (k & 4) ? pos1[2] : pos0[2];
    c->idx[k] = (z * height + y) * width + x;          /* :134-141                  */
  }
}

/* trajectory signature hook (tests only, see drrt_oracle.c): the floor cell of p, computed like locate() */
static inline void FN(sig_cell)(size_t i, REAL h, const REAL p[3]) {
  if (!g_sig) return;
  const REAL rh = (REAL)1 / h;
  sig_visit(i, (int)FLOOR(p[0] * rh), (int)FLOOR(p[1] * rh), (int)FLOOR(p[2] * rh));
}

/* ---- volume::eval_grad  (src/volume.cpp:101-181) --------------------------------- */
static inline void FN(vol_eval_grad)(const REAL* data, const int res[3], REAL h,
                                     const REAL p[3], int mask, REAL* n, REAL g[3]) {
  if (res[0] == 1 && res[1] == 1 && res[2] == 1) {     /* :117-121 (mask not applied) */
    *n = data[0]; g[0] = g[1] = g[2] = 0; return;
  }
  FN(cell_t) c; FN(locate)(res, h, p, &c);
  REAL v[8];
  for (int k = 0; k < 8; ++k) v[k] = mask ? data[c.idx[k]] : (REAL)0;   /* :143-150 */
  const REAL *w0 = c.w0, *w1 = c.w1;
  const REAL v000=v[0], v100=v[1], v010=v[2], v110=v[3], v001=v[4], v101=v[5], v011=v[6], v111=v[7];
  REAL w000 = w1[0]*w1[1]*w1[2], w100 = w0[0]*w1[1]*w1[2];
  REAL w010 = w1[0]*w0[1]*w1[2], w110 = w0[0]*w0[1]*w1[2];
  REAL w001 = w1[0]*w1[1]*w0[2], w101 = w0[0]*w1[1]*w0[2];
  REAL w011 = w1[0]*w0[1]*w0[2], w111 = w0[0]*w0[1]*w0[2];                 /* :152-159 */
  *n = w000*v000 + w100*v100 + w010*v010 + w110*v110 +
       w001*v001 + w101*v101 + w011*v011 + w111*v111;                       /* :162-163 */
  REAL nx = (v100*w1[1]*w1[2] + v101*w1[1]*w0[2] + v110*w0[1]*w1[2] + v111*w0[1]*w0[2])
          - (v000*w1[1]*w1[2] + v001*w1[1]*w0[2] + v010*w0[1]*w1[2] + v011*w0[1]*w0[2]);
  REAL ny = (v010*w1[0]*w1[2] + v011*w1[0]*w0[2] + v110*w0[0]*w1[2] + v111*w0[0]*w0[2])
          - (v000*w1[0]*w1[2] + v001*w1[0]*w0[2] + v100*w0[0]*w1[2] + v101*w0[0]*w0[2]);
  REAL nz = (v001*w1[0]*w1[1] + v011*w1[0]*w0[1] + v101*w0[0]*w1[1] + v111*w0[0]*w0[1])
          - (v000*w1[0]*w1[1] + v010*w1[0]*w0[1] + v100*w0[0]*w1[1] + v110*w0[0]*w0[1]);
  const REAL rh = (REAL)1 / h;
  g[0] = nx * rh; g[1] = ny * rh; g[2] = nz * rh;                           /* :178 */
}

/* ---- volume::eval_hess  (src/volume.cpp:40-99): only mixed partials, /h/h (Q10) --- */
static inline void FN(vol_eval_hess)(const REAL* data, const int res[3], REAL h,
                                     const REAL p[3], int mask, REAL H[3] /* xy,xz,yz */) {
  FN(cell_t) c; FN(locate)(res, h, p, &c);
  REAL v[8];
  for (int k = 0; k < 8; ++k) v[k] = mask ? data[c.idx[k]] : (REAL)0;   /* :70-77 */
  const REAL v000=v[0], v100=v[1], v010=v[2], v110=v[3], v001=v[4], v101=v[5], v011=v[6], v111=v[7];
  /* enoki lerp(a,b,t) = fmadd(b, t, fnmadd(a, t, a))                                */
  REAL a, b, t;
  a = v110 - v010 - v100 + v000; b = v111 - v011 - v101 + v001; t = c.w0[2];
  REAL dxdy = FMA(b, t, FMA(-a, t, a));                                   /* :79-81 */
  a = v101 - v001 - v100 + v000; b = v111 - v011 - v110 + v010; t = c.w0[1];
  REAL dxdz = FMA(b, t, FMA(-a, t, a));                                   /* :82-84 */
  a = v011 - v001 - v010 + v000; b = v111 - v101 - v110 + v100; t = c.w0[0];
  REAL dydz = FMA(b, t, FMA(-a, t, a));                                   /* :85-87 */
  H[0] = dxdy / h / h; H[1] = dxdz / h / h; H[2] = dydz / h / h;           /* :98 */
}

/* ---- volume::splat  (src/volume.cpp:182-244).  Serial scatter => deterministic.
 *      NOTE Q3: the gradient part carries NO 1/h (as written in the reference);
 *      grad_scale lets the caller opt in to the corrected form (1/h).               */
static inline void FN(vol_splat)(REAL* data, const int res[3], REAL h, const REAL p[3],
                                 REAL val, const REAL grad[3], int active, REAL grad_scale) {
  if (!active) return;
  FN(cell_t) c; FN(locate)(res, h, p, &c);
  const REAL *w0 = c.w0, *w1 = c.w1;
  data[c.idx[0]] += val*w1[0]*w1[1]*w1[2];
  data[c.idx[1]] += val*w0[0]*w1[1]*w1[2];
  data[c.idx[2]] += val*w1[0]*w0[1]*w1[2];
  data[c.idx[3]] += val*w0[0]*w0[1]*w1[2];
  data[c.idx[4]] += val*w1[0]*w1[1]*w0[2];
  data[c.idx[5]] += val*w0[0]*w1[1]*w0[2];
  data[c.idx[6]] += val*w1[0]*w0[1]*w0[2];
  data[c.idx[7]] += val*w0[0]*w0[1]*w0[2];                                   /* :217-224 */
  const REAL gx = grad[0]*grad_scale, gy = grad[1]*grad_scale, gz = grad[2]*grad_scale;
  REAL v000 = -gx*w1[1]*w1[2] - gy*w1[0]*w1[2] - gz*w1[0]*w1[1];
  REAL v100 =  gx*w1[1]*w1[2] - gy*w0[0]*w1[2] - gz*w0[0]*w1[1];
  REAL v010 = -gx*w0[1]*w1[2] + gy*w1[0]*w1[2] - gz*w1[0]*w0[1];
  REAL v110 =  gx*w0[1]*w1[2] + gy*w0[0]*w1[2] - gz*w0[0]*w0[1];
  REAL v001 = -gx*w1[1]*w0[2] - gy*w1[0]*w0[2] + gz*w1[0]*w1[1];
  REAL v101 =  gx*w1[1]*w0[2] - gy*w0[0]*w0[2] + gz*w0[0]*w1[1];
  REAL v011 = -gx*w0[1]*w0[2] + gy*w1[0]*w0[2] + gz*w1[0]*w0[1];
  REAL v111 =  gx*w0[1]*w0[2] + gy*w0[0]*w0[2] + gz*w0[0]*w0[1];           /* :227-234 */
  data[c.idx[0]] += v000; data[c.idx[1]] += v100; data[c.idx[2]] += v010; data[c.idx[3]] += v110;
  data[c.idx[4]] += v001; data[c.idx[5]] += v101; data[c.idx[6]] += v011; data[c.idx[7]] += v111;
}

/* ---- volume::inbounds / escaped  (src/volume.cpp:246-271, Q8) ---------------------- */
static inline int FN(vol_inbounds)(const int res[3], REAL h, const REAL p[3]) {
  int below = (p[0] >= 0) & (p[1] >= 0) & (p[2] >= 0);
  int above = (p[0] < ((REAL)(res[0]-1) * h)) & (p[1] < ((REAL)(res[1]-1) * h)) &
              (p[2] < ((REAL)(res[2]-1) * h));
  return below & above;
}
static inline int FN(vol_escaped)(const int res[3], REAL h, const REAL p[3], const REAL v[3]) {
  int e = 0;
  for (int a = 0; a < 3; ++a)
    e |= ((p[a] < 0) & (v[a] < 0)) | ((p[a] >= ((REAL)(res[a]-1) * h)) & (v[a] > 0));
  return e;
}

static int FN(check_res)(const int res[3], long long nvox) {
  if ((long long)res[0] * res[1] * res[2] != nvox) return -1;   /* volume.cpp:34-37 */
  if (!(res[0] == 1 && res[1] == 1 && res[2] == 1) && (res[0] < 2 || res[1] < 2)) return -2; /* :123 */
  return 0;
}

static inline int FN(max3i)(const int r[3]) { int m = r[0]; if (r[1]>m) m=r[1]; if (r[2]>m) m=r[2]; return m; }

/* ================================================================================== */
/* Tracer::trace (src/tracer.cpp:35-100), trace_plane (:102-172), trace_sdf (:244-310)  */
/* mode: 0 = trace, 1 = trace_plane, 2 = trace_sdf.  Array-at-a-time over all rays,     */
/* global break on all(escaped) exactly like the reference loop.                        */
/* steps_out (nullable): number of loop iterations executed before ray i was flagged    */
/* escaped (its contribution to the ray-steps metric); n_failed: #rays still active.    */
/* ================================================================================== */
static int FN(trace_generic)(int mode, const REAL* rif, const REAL* sdf, const int res[3],
                             long long nvox, size_t N, const REAL* pos, const REAL* vel,
                             const REAL* pln_o, const REAL* pln_d, REAL h, REAL ds,
                             REAL* xt, REAL* vt, unsigned char* failmask,
                             int* steps_out, long long* n_failed, int* iters_out) {
  int rc = FN(check_res)(res, nvox); if (rc) return rc;
  /* :51 / :120 / :262 -- float expression truncated to int (Q5) */
  int max_steps = (mode == 2) ? (int)((REAL)2 * h * (REAL)FN(max3i)(res) / ds)
                              : (int)((REAL)4 * h * (REAL)FN(max3i)(res) / ds);
  REAL* x = (REAL*)malloc(sizeof(REAL) * 3 * N);
  REAL* v = (REAL*)malloc(sizeof(REAL) * 3 * N);
  unsigned char* inside  = (unsigned char*)malloc(N);
  unsigned char* escaped = (unsigned char*)malloc(N);
  unsigned char* active  = (unsigned char*)malloc(N);
  memcpy(x, pos, sizeof(REAL) * 3 * N);  memcpy(v, vel, sizeof(REAL) * 3 * N);
  memcpy(xt, pos, sizeof(REAL) * 3 * N); memcpy(vt, vel, sizeof(REAL) * 3 * N);   /* :53-57 */
  for (size_t i = 0; i < N; ++i) {
    inside[i] = (unsigned char)FN(vol_inbounds)(res, h, x + 3*i);                /* :61 */
    escaped[i] = 0; active[i] = 1;                                              /* :62-63 */
    if (steps_out) steps_out[i] = 0;
    if (mode == 2) {                                                            /* :276-277 */
      REAL d, dg[3]; FN(vol_eval_grad)(sdf, res, h, x + 3*i, active[i], &d, dg);
      active[i] = d < 0;
    }
  }
  int it;
  for (it = 0; it < max_steps; ++it) {
    int all_escaped = 1;
    for (size_t i = 0; i < N; ++i) {
      REAL *xi = x + 3*i, *vi = v + 3*i;
      REAL n, g[3];
      FN(vol_eval_grad)(rif, res, h, xi, inside[i], &n, g);                     /* :68 */
      REAL dsn = ds * n;
      for (int a = 0; a < 3; ++a) vi[a] = FMA(dsn, g[a], vi[a]);                /* :70 */
      for (int a = 0; a < 3; ++a) xi[a] = FMA(ds, vi[a], xi[a]);                /* :71 */
      int cur_inside;
      if (mode == 2) {                                                          /* :287-288 */
        REAL d, dg[3]; FN(vol_eval_grad)(sdf, res, h, xi, inside[i], &d, dg);
        cur_inside = d < 0;
      } else {
        cur_inside = FN(vol_inbounds)(res, h, xi);                              /* :73 */
        if (mode == 1) {                                                        /* :144-145 */
          const REAL *o = pln_o + 3*i, *d = pln_d + 3*i;
          REAL dot = (xi[0]-o[0])*d[0] + (xi[1]-o[1])*d[1] + (xi[2]-o[2])*d[2];
          cur_inside = cur_inside & !(dot > 0);
        }
      }
      int was_escaped = escaped[i];
      int cross = inside[i] & !cur_inside;                                      /* :74 */
      escaped[i] |= (unsigned char)cross;                                       /* :75 */
      escaped[i] |= (unsigned char)FN(vol_escaped)(res, h, xi, vi);             /* :76 */
      active[i] &= !escaped[i];                                                 /* :77 */
      if (cross) { for (int a = 0; a < 3; ++a) { xt[3*i+a] = xi[a]; vt[3*i+a] = vi[a]; } } /* :79-80 */
      if (steps_out && !was_escaped) steps_out[i] = it + 1;
      all_escaped &= escaped[i];
      inside[i] = (unsigned char)cur_inside;                                    /* :86 */
    }
    if (all_escaped) { ++it; break; }                                           /* :82-84 */
  }
  long long nf = 0;
  for (size_t i = 0; i < N; ++i) nf += active[i];
  if (mode != 2) {
    if (nf > 0)                                                                 /* :89-96 */
      for (size_t i = 0; i < N; ++i)
        if (!escaped[i]) for (int a = 0; a < 3; ++a) xt[3*i+a] = x[3*i+a];
  }
  if (failmask) for (size_t i = 0; i < N; ++i) failmask[i] = !escaped[i];       /* :171 */
  if (n_failed) *n_failed = nf;
  if (iters_out) *iters_out = it;
  free(x); free(v); free(inside); free(escaped); free(active);
  return 0;
}

/* Tracer::trace_target (src/tracer.cpp:174-242): closest approach to a per-ray target.
 * The closest-approach update is NOT gated by `escaped`, so every ray keeps updating
 * until the global all(escaped) break.                                                */
static int FN(trace_target_impl)(const REAL* rif, const int res[3], long long nvox, size_t N,
                                 const REAL* pos, const REAL* vel, const REAL* target,
                                 REAL h, REAL ds, REAL* xt, REAL* vt, REAL* dist2,
                                 long long* n_failed, int* iters_out) {
  int rc = FN(check_res)(res, nvox); if (rc) return rc;
  int max_steps = (int)((REAL)4 * h * (REAL)FN(max3i)(res) / ds);               /* :192 */
  REAL* x = (REAL*)malloc(sizeof(REAL) * 3 * N);
  REAL* v = (REAL*)malloc(sizeof(REAL) * 3 * N);
  unsigned char* inside  = (unsigned char*)malloc(N);
  unsigned char* escaped = (unsigned char*)malloc(N);
  unsigned char* active  = (unsigned char*)malloc(N);
  memcpy(x, pos, sizeof(REAL) * 3 * N);  memcpy(v, vel, sizeof(REAL) * 3 * N);
  memcpy(xt, pos, sizeof(REAL) * 3 * N); memcpy(vt, vel, sizeof(REAL) * 3 * N);
  for (size_t i = 0; i < N; ++i) {
    REAL d0 = x[3*i]-target[3*i], d1 = x[3*i+1]-target[3*i+1], d2 = x[3*i+2]-target[3*i+2];
    dist2[i] = d0*d0 + d1*d1 + d2*d2;                                           /* :200 */
    inside[i] = (unsigned char)FN(vol_inbounds)(res, h, x + 3*i);
    escaped[i] = 0; active[i] = 1;
  }
  int it;
  for (it = 0; it < max_steps; ++it) {
    int all_escaped = 1;
    for (size_t i = 0; i < N; ++i) {
      REAL *xi = x + 3*i, *vi = v + 3*i;
      REAL n, g[3];
      FN(vol_eval_grad)(rif, res, h, xi, inside[i], &n, g);                     /* :211 */
      REAL dsn = ds * n;
      for (int a = 0; a < 3; ++a) vi[a] = FMA(dsn, g[a], vi[a]);
      for (int a = 0; a < 3; ++a) xi[a] = FMA(ds, vi[a], xi[a]);
      REAL d0 = xi[0]-target[3*i], d1 = xi[1]-target[3*i+1], d2 = xi[2]-target[3*i+2];
      REAL cur = d0*d0 + d1*d1 + d2*d2;                                         /* :216 */
      int closer = cur < dist2[i];
      int cur_inside = FN(vol_inbounds)(res, h, xi);
      int cross = inside[i] & !cur_inside;
      escaped[i] |= (unsigned char)cross;
      escaped[i] |= (unsigned char)FN(vol_escaped)(res, h, xi, vi);
      active[i] &= !escaped[i];
      if (closer) {                                                             /* :225-227 */
        for (int a = 0; a < 3; ++a) { xt[3*i+a] = xi[a]; vt[3*i+a] = vi[a]; }
        dist2[i] = cur;
      }
      all_escaped &= escaped[i];
      inside[i] = (unsigned char)cur_inside;
    }
    if (all_escaped) { ++it; break; }
  }
  long long nf = 0; for (size_t i = 0; i < N; ++i) nf += active[i];
  if (n_failed) *n_failed = nf;
  if (iters_out) *iters_out = it;
  free(x); free(v); free(inside); free(escaped); free(active);
  return 0;
}

/* ================================================================================== */
/* Tracer::backtrace (src/tracer.cpp:384-440) and backtrace_sdf (:443-509)              */
/* grad must hold nvox entries; it is zeroed here (reference: fresh zero array :401).   */
/* grad_scale = 1 reproduces the reference as written (Q3); 1/h is the corrected form.  */
/* ================================================================================== */
static int FN(backtrace_generic)(int use_sdf, const REAL* rif, const REAL* sdf, const int res[3],
                                 long long nvox, size_t N, const REAL* xt, const REAL* vt,
                                 const REAL* dx, const REAL* dv, REAL h, REAL ds,
                                 REAL grad_scale, REAL* grad, long long* steps_total) {
  int rc = FN(check_res)(res, nvox); if (rc) return rc;
  memset(grad, 0, sizeof(REAL) * (size_t)nvox);
  int max_steps = (int)((REAL)2 * h * (REAL)FN(max3i)(res) / ds);               /* :417 */
  REAL* x  = (REAL*)malloc(sizeof(REAL) * 3 * N);
  REAL* v  = (REAL*)malloc(sizeof(REAL) * 3 * N);
  REAL* la = (REAL*)malloc(sizeof(REAL) * 3 * N);
  REAL* mu = (REAL*)malloc(sizeof(REAL) * 3 * N);
  unsigned char* active  = (unsigned char*)malloc(N);
  unsigned char* outside = (unsigned char*)malloc(N);
  memcpy(x, xt, sizeof(REAL) * 3 * N); memcpy(v, vt, sizeof(REAL) * 3 * N);
  long long steps = 0;
  for (size_t i = 0; i < N; ++i) {
    for (int a = 0; a < 3; ++a) {
      la[3*i+a] = dx[3*i+a];                                                    /* :409 */
      mu[3*i+a] = dv[3*i+a] + ds * dx[3*i+a];                                   /* :410 */
    }
    REAL nv[3] = { -v[3*i], -v[3*i+1], -v[3*i+2] };
    active[i] = !FN(vol_escaped)(res, h, x + 3*i, nv);                          /* :413-414 */
    outside[i] = 0;
    if (use_sdf) {                                                              /* :476-477 */
      REAL d, dg[3]; FN(vol_eval_grad)(sdf, res, h, x + 3*i, active[i], &d, dg);
      outside[i] = d >= 0;
    }
  }
  for (int it = 0; it < max_steps; ++it) {
    int any_active = 0;
    /* The reference breaks BEFORE splatting when none(active) (:426-428).  The work of
       one iteration is ray-separable and a ray that has gone inactive contributes
       nothing, so one fused pass over the rays followed by the break test is exact.   */
    for (size_t i = 0; i < N; ++i) {
      REAL *xi = x + 3*i, *vi = v + 3*i, *li = la + 3*i, *mi = mu + 3*i;
      for (int a = 0; a < 3; ++a) xi[a] = FMA(-ds, vi[a], xi[a]);               /* :420 */
      REAL n, g[3], Hm[3];
      FN(vol_eval_grad)(rif, res, h, xi, active[i], &n, g);                     /* :421 */
      if (active[i]) FN(vol_eval_hess)(rif, res, h, xi, 1, Hm);                 /* :422 */
      else Hm[0] = Hm[1] = Hm[2] = 0;
      REAL mdsn = -ds * n;
      for (int a = 0; a < 3; ++a) vi[a] = FMA(mdsn, g[a], vi[a]);               /* :423 */
      REAL nv[3] = { -vi[0], -vi[1], -vi[2] };
      active[i] &= !FN(vol_escaped)(res, h, xi, nv);                            /* :425 */
      if (use_sdf) {                                                            /* :488-497 */
        REAL d, dg[3]; FN(vol_eval_grad)(sdf, res, h, xi, 1, &d, dg);
        /* reference gathers the sdf with mask=active(before update); a masked-out lane
           reads 0 => dist>=0, but such a lane is already inactive, so no effect.       */
        int now_out = d >= 0;
        int cross = (!outside[i]) & now_out;
        active[i] &= !cross;
        outside[i] = (unsigned char)now_out;
      }
      if (!active[i]) continue;
      any_active = 1; ++steps;
      FN(sig_cell)(i, h, xi);
      REAL dn = mi[0]*g[0] + mi[1]*g[1] + mi[2]*g[2];                           /* :430 */
      REAL dnx[3] = { n*mi[0]*ds, n*mi[1]*ds, n*mi[2]*ds };                     /* :431-432 */
      FN(vol_splat)(grad, res, h, xi, dn*ds, dnx, 1, grad_scale);               /* :432 */
      /* Hess*mu with zero diagonal: H = [[0,xy,xz],[xy,0,yz],[xz,yz,0]]               */
      REAL Hmu[3] = { Hm[0]*mi[1] + Hm[1]*mi[2], Hm[0]*mi[0] + Hm[2]*mi[2], Hm[1]*mi[0] + Hm[2]*mi[1] };
      for (int a = 0; a < 3; ++a) li[a] = li[a] + ds * (dn * g[a] + n * Hmu[a]); /* :434 */
      for (int a = 0; a < 3; ++a) mi[a] = mi[a] + ds * li[a];                   /* :435 */
    }
    if (!any_active) break;                                                     /* :426-428 */
  }
  if (steps_total) *steps_total = steps;
  free(x); free(v); free(la); free(mu); free(active); free(outside);
  return 0;
}

/* ================================================================================== */
/* cylinder_volume  (src/cylinder_volume.cpp, Q13): radial profile around the y axis     */
/* through x = z = radius.                                                              */
/* ================================================================================== */
#define CYL_EPS ((REAL)1e-6)                                                    /* :15 */

typedef struct { int i0, i1; REAL w0, w1, r, h, xs[3], rhat[3]; } FN(cyl_t);

static inline void FN(cyl_locate)(size_t rres, REAL radius, const REAL p[3], FN(cyl_t)* c) {
  c->xs[0] = p[0] - radius; c->xs[1] = 0; c->xs[2] = p[2] - radius;             /* :37-38 */
  c->r = SQRT(c->xs[0]*c->xs[0] + c->xs[1]*c->xs[1] + c->xs[2]*c->xs[2]);       /* :41 */
  c->h = radius / (REAL)(rres - 1);                                             /* :42 */
  REAL rm = c->r / c->h;                                                        /* :44 */
  int ir = (int)FLOOR(rm);
  c->i0 = FN(clampi)(ir, 0, (int)rres - 1);                                     /* :45 */
  c->i1 = FN(clampi)(c->i0 + 1, 0, (int)rres - 1);                              /* :46 */
  c->w0 = rm - (REAL)c->i0; c->w1 = (REAL)1 - c->w0;                            /* :48 (clamped idx0!) */
  /* normalize(xs) = xs * rsqrt(|xs|^2); at r = 0 this is 0*inf = NaN in enoki, then
     overwritten by the r < eps select (:56, :91, :143).                               */
  if (c->r < CYL_EPS) { c->rhat[0] = c->rhat[1] = c->rhat[2] = 0; }
  else { c->rhat[0] = c->xs[0] / c->r; c->rhat[1] = 0; c->rhat[2] = c->xs[2] / c->r; }
}

static inline void FN(cyl_eval_grad)(const REAL* data, size_t rres, REAL radius,
                                     const REAL p[3], REAL* f, REAL g[3]) {
  FN(cyl_t) c; FN(cyl_locate)(rres, radius, p, &c);
  REAL val0 = data[c.i0], val1 = data[c.i1];                                    /* :50-51 unmasked */
  *f = val0*c.w1 + val1*c.w0;                                                   /* :53 */
  REAL rx = (val1 - val0) / c.h;                                                /* :54 */
  for (int a = 0; a < 3; ++a) g[a] = rx * c.rhat[a];                            /* :55-56 */
}

/* returns the 4 nonzero entries H00,H02,H20,H22 (row 1 / col 1 are zero, :97-105)      */
static inline void FN(cyl_eval_hess)(const REAL* data, size_t rres, REAL radius,
                                     const REAL p[3], REAL H[4]) {
  FN(cyl_t) c; FN(cyl_locate)(rres, radius, p, &c);
  if (c.r < CYL_EPS) { H[0] = H[1] = H[2] = H[3] = 0; return; }                 /* :108 */
  REAL val0 = data[c.i0], val1 = data[c.i1];
  REAL rx = (val1 - val0) / c.h;                                                /* :88 */
  REAL s = rx / c.r;                                                            /* :107 */
  H[0] = ((REAL)1 - c.rhat[0]*c.rhat[0]) * s;
  H[1] = -(c.rhat[0]*c.rhat[2]) * s;
  H[2] = -(c.rhat[2]*c.rhat[0]) * s;
  H[3] = ((REAL)1 - c.rhat[2]*c.rhat[2]) * s;
}

static inline void FN(cyl_splat)(REAL* data, size_t rres, REAL radius, const REAL p[3],
                                 REAL val, const REAL grad[3], int active) {
  if (!active) return;
  FN(cyl_t) c; FN(cyl_locate)(rres, radius, p, &c);
  data[c.i0] += val*c.w1;                                                       /* :139 */
  data[c.i1] += val*c.w0;                                                       /* :140 */
  REAL gv = grad[0]*c.rhat[0] + grad[1]*c.rhat[1] + grad[2]*c.rhat[2];          /* :142-143 */
  data[c.i0] += -gv / c.h;                                                      /* :146 */
  data[c.i1] +=  gv / c.h;                                                      /* :147 */
}

static inline int FN(cyl_inbounds)(REAL radius, REAL length, const REAL p[3]) {  /* :150-156 */
  REAL px = p[0] - radius, pz = p[2] - radius;
  REAL r = px*px + pz*pz;
  int inlength = (p[1] < length) & (p[1] >= 0);
  return (r < (radius*radius)) & inlength;
}
static inline int FN(cyl_escaped)(REAL radius, REAL length, const REAL p[3], const REAL v[3]) { /* :158-170 */
  REAL px = p[0] - radius, pz = p[2] - radius;
  int esc_length = ((p[1] < 0) & (v[1] < 0)) | ((p[1] > length) & (v[1] > 0));
  int out_radius = (px*px + pz*pz) >= (radius*radius);
  int esc_radius = (px*v[0] + pz*v[2]) > 0;
  return (out_radius & esc_radius) | esc_length;
}

/* Tracer::trace_cable (src/tracer.cpp:312-382): state update masked by `active` (Q7).  */
static int FN(trace_cable_impl)(const REAL* rif, size_t rres, REAL radius, REAL length, size_t N,
                                const REAL* pos, const REAL* vel, const REAL* target, REAL ds,
                                REAL* xt, REAL* vt, REAL* dist2, long long* n_failed,
                                long long* steps_total) {
  if (rres < 2) return -2;
  int max_steps = (int)((REAL)4 * length / ds);                                 /* :332 */
  REAL* x = (REAL*)malloc(sizeof(REAL) * 3 * N);
  REAL* v = (REAL*)malloc(sizeof(REAL) * 3 * N);
  unsigned char* inside  = (unsigned char*)malloc(N);
  unsigned char* escaped = (unsigned char*)malloc(N);
  unsigned char* active  = (unsigned char*)malloc(N);
  memcpy(x, pos, sizeof(REAL) * 3 * N);  memcpy(v, vel, sizeof(REAL) * 3 * N);
  memcpy(xt, pos, sizeof(REAL) * 3 * N); memcpy(vt, vel, sizeof(REAL) * 3 * N);
  long long steps = 0;
  for (size_t i = 0; i < N; ++i) {
    REAL d0 = x[3*i]-target[3*i], d1 = x[3*i+1]-target[3*i+1], d2 = x[3*i+2]-target[3*i+2];
    dist2[i] = d0*d0 + d1*d1 + d2*d2;                                           /* :340 */
    inside[i] = (unsigned char)FN(cyl_inbounds)(radius, length, x + 3*i);       /* :344 */
    escaped[i] = 0; active[i] = 1;
  }
  for (int it = 0; it < max_steps; ++it) {
    int all_escaped = 1;
    for (size_t i = 0; i < N; ++i) {
      REAL *xi = x + 3*i, *vi = v + 3*i;
      REAL n, g[3];
      FN(cyl_eval_grad)(rif, rres, radius, xi, &n, g);                          /* :351 */
      if (active[i]) {                                                          /* :353-354 */
        REAL dsn = ds * n;
        for (int a = 0; a < 3; ++a) vi[a] = FMA(dsn, g[a], vi[a]);
        for (int a = 0; a < 3; ++a) xi[a] = FMA(ds, vi[a], xi[a]);
        ++steps;
      }
      REAL d0 = xi[0]-target[3*i], d1 = xi[1]-target[3*i+1], d2 = xi[2]-target[3*i+2];
      REAL cur = d0*d0 + d1*d1 + d2*d2;                                         /* :356 */
      int closer = cur < dist2[i];
      int cur_inside = FN(cyl_inbounds)(radius, length, xi);                    /* :359 */
      int cross = inside[i] & !cur_inside;
      escaped[i] |= (unsigned char)cross;
      escaped[i] |= (unsigned char)FN(cyl_escaped)(radius, length, xi, vi);     /* :362 */
      active[i] &= !escaped[i];
      if (closer) {                                                             /* :365-367 */
        for (int a = 0; a < 3; ++a) { xt[3*i+a] = xi[a]; vt[3*i+a] = vi[a]; }
        dist2[i] = cur;
      }
      all_escaped &= escaped[i];
      inside[i] = (unsigned char)cur_inside;
    }
    if (all_escaped) break;
  }
  long long nf = 0; for (size_t i = 0; i < N; ++i) nf += active[i];
  if (n_failed) *n_failed = nf;
  if (steps_total) *steps_total = steps;
  free(x); free(v); free(inside); free(escaped); free(active);
  return 0;
}

/* Tracer::backtrace_cable (src/tracer.cpp:511-567)                                     */
static int FN(backtrace_cable_impl)(const REAL* rif, size_t rres, REAL radius, REAL length, size_t N,
                                    const REAL* xt, const REAL* vt, const REAL* dx, const REAL* dv,
                                    REAL ds, REAL* grad, long long* steps_total) {
  if (rres < 2) return -2;
  memset(grad, 0, sizeof(REAL) * rres);
  int max_steps = (int)((REAL)4 * length / ds);                                 /* :544 */
  long long steps = 0;
  /* ray-separable (a finished ray contributes nothing, the global break is a no-op):
     march ray by ray; the scatter order is then ray-major, deterministic.             */
  for (size_t i = 0; i < N; ++i) {
    REAL x[3], v[3], la[3], mu[3];
    for (int a = 0; a < 3; ++a) {
      x[a] = xt[3*i+a]; v[a] = vt[3*i+a];
      la[a] = dx[3*i+a]; mu[a] = dv[3*i+a] + ds * dx[3*i+a];                    /* :536-537 */
    }
    REAL nv[3] = { -v[0], -v[1], -v[2] };
    int active = !FN(cyl_escaped)(radius, length, x, nv);                       /* :540-541 */
    for (int it = 0; it < max_steps && active; ++it) {
      for (int a = 0; a < 3; ++a) x[a] = FMA(-ds, v[a], x[a]);                  /* :547 */
      REAL n, g[3], H[4];
      FN(cyl_eval_grad)(rif, rres, radius, x, &n, g);                           /* :548 */
      FN(cyl_eval_hess)(rif, rres, radius, x, H);                               /* :549 */
      REAL mdsn = -ds * n;
      for (int a = 0; a < 3; ++a) v[a] = FMA(mdsn, g[a], v[a]);                 /* :550 */
      nv[0] = -v[0]; nv[1] = -v[1]; nv[2] = -v[2];
      active &= !FN(cyl_escaped)(radius, length, x, nv);                        /* :552 */
      if (!active) break;
      ++steps;
      REAL dn = mu[0]*g[0] + mu[1]*g[1] + mu[2]*g[2];                           /* :557 */
      REAL dnx[3] = { n*mu[0]*ds, n*mu[1]*ds, n*mu[2]*ds };                     /* :558-559 */
      FN(cyl_splat)(grad, rres, radius, x, dn*ds, dnx, 1);
      REAL Hmu[3] = { H[0]*mu[0] + H[1]*mu[2], 0, H[2]*mu[0] + H[3]*mu[2] };
      for (int a = 0; a < 3; ++a) la[a] = la[a] + ds * (dn * g[a] + n * Hmu[a]); /* :561 */
      for (int a = 0; a < 3; ++a) mu[a] = mu[a] + ds * la[a];                   /* :562 */
    }
  }
  if (steps_total) *steps_total = steps;
  return 0;
}

/* ================================================================================== */
/* "factored" arithmetic mode                                                           */
/*                                                                                      */
/* Same algorithm, re-associated: n, grad n and the mixed partials come from one lerp   */
/* tree (differences first) and volume::splat's 16 scatter_adds are fused into 8 corner */
/* sums.  This is the explicit IEEE operation sequence that the HIP kernels implement   */
/* (adjointnonlinearraytracing_amd/csrc/drrt_device.h, "ARITHMETIC CONTRACT"), restated  */
/* here independently in plain C so that GPU trajectories can be compared bit for bit.   */
/* It is validated against the literal mode above in float64 by tests/test_oracle.py     */
/* (agreement ~1e-12), so the chain is: reference text == literal mode ~ factored mode   */
/* == GPU.  The loops keep the reference's array-at-a-time structure and global breaks.  */
/* ================================================================================== */

typedef struct {
  const REAL* data; int W, H, D, sy, sz; REAL inv_h, inv_h2, bx, by, bz;
} FN(fvol_t);

typedef struct { int base, ox, oy, oz; REAL wx, wy, wz; } FN(fcell_t);
typedef struct { REAL n, gx, gy, gz, hxy, hxz, hyz; } FN(fsample_t);

static inline void FN(fvol_make)(FN(fvol_t)* V, const REAL* data, const int res[3], REAL h) {
  V->data = data; V->W = res[0]; V->H = res[1]; V->D = res[2]; V->sy = res[0]; V->sz = res[0]*res[1];
  V->inv_h = (REAL)1 / h; V->inv_h2 = V->inv_h * V->inv_h;
  V->bx = (REAL)(res[0]-1) * h; V->by = (REAL)(res[1]-1) * h; V->bz = (REAL)(res[2]-1) * h;
}

static inline int FN(f2i_sat)(REAL f) {
  if (!(f == f)) return 0;
  if (f >= (REAL)2147483648.0) return 2147483647;
  if (f <= (REAL)-2147483648.0) return (-2147483647 - 1);
  return (int)f;
}

static inline FN(fcell_t) FN(flocate)(const FN(fvol_t)* V, REAL px, REAL py, REAL pz) {
  FN(fcell_t) c;
  REAL fx = px * V->inv_h, fy = py * V->inv_h, fz = pz * V->inv_h;
  REAL flx = FLOOR(fx), fly = FLOOR(fy), flz = FLOOR(fz);
  c.wx = fx - flx; c.wy = fy - fly; c.wz = fz - flz;
  int ix = FN(f2i_sat)(flx), iy = FN(f2i_sat)(fly), iz = FN(f2i_sat)(flz);
  int x0 = FN(clampi)(ix, 0, V->W-1), x1 = FN(clampi)(ix+1, 0, V->W-1);
  int y0 = FN(clampi)(iy, 0, V->H-1), y1 = FN(clampi)(iy+1, 0, V->H-1);
  int z0 = FN(clampi)(iz, 0, V->D-1), z1 = FN(clampi)(iz+1, 0, V->D-1);
  c.base = z0*V->sz + y0*V->sy + x0;
  c.ox = x1 - x0; c.oy = (y1 - y0)*V->sy; c.oz = (z1 - z0)*V->sz;
  return c;
}

static inline FN(fsample_t) FN(finterp)(const REAL* d, const FN(fcell_t)* c) {
  const REAL* p = d + c->base;
  REAL v000 = p[0], v100 = p[c->ox], v010 = p[c->oy], v110 = p[c->oy + c->ox];
  REAL v001 = p[c->oz], v101 = p[c->oz + c->ox], v011 = p[c->oz + c->oy], v111 = p[c->oz + c->oy + c->ox];
  REAL wx = c->wx, wy = c->wy, wz = c->wz;
  FN(fsample_t) s;
  REAL d00 = v100 - v000, d10 = v110 - v010, d01 = v101 - v001, d11 = v111 - v011;
  REAL c00 = FMA(wx, d00, v000), c10 = FMA(wx, d10, v010);
  REAL c01 = FMA(wx, d01, v001), c11 = FMA(wx, d11, v011);
  REAL e0 = c10 - c00, e1 = c11 - c01;
  REAL l0 = FMA(wy, e0, c00), l1 = FMA(wy, e1, c01);
  REAL dz = l1 - l0;
  s.gz = dz;
  s.n = FMA(wz, dz, l0);
  REAL eyz = e1 - e0;
  s.gy = FMA(wz, eyz, e0);
  REAL dxy0 = d10 - d00, dxy1 = d11 - d01;
  REAL gx0 = FMA(wy, dxy0, d00), gx1 = FMA(wy, dxy1, d01);
  REAL gxz = gx1 - gx0;
  s.gx = FMA(wz, gxz, gx0);
  s.hxy = FMA(wz, dxy1 - dxy0, dxy0);
  s.hxz = gxz;
  s.hyz = eyz;
  return s;
}

static inline int FN(finbounds)(const FN(fvol_t)* V, REAL px, REAL py, REAL pz) {
  return (px >= 0) & (py >= 0) & (pz >= 0) & (px < V->bx) & (py < V->by) & (pz < V->bz);
}
static inline int FN(fescaped)(const FN(fvol_t)* V, REAL px, REAL py, REAL pz, REAL vx, REAL vy, REAL vz) {
  int ex = ((px < 0) & (vx < 0)) | ((px >= V->bx) & (vx > 0));
  int ey = ((py < 0) & (vy < 0)) | ((py >= V->by) & (vy > 0));
  int ez = ((pz < 0) & (vz < 0)) | ((pz >= V->bz) & (vz > 0));
  return ex | ey | ez;
}
static inline REAL FN(fdot3)(REAL ax, REAL ay, REAL az, REAL bx, REAL by, REAL bz) {
  return FMA(az, bz, FMA(ay, by, ax * bx));
}

typedef struct {
  REAL x, y, z, vx, vy, vz, xtx, xty, xtz, vtx, vty, vtz, a0, a1, a2, a3, a4, a5;
  int inside, esc, act; int steps;
} FN(fstate_t);

/* one forward iteration; mode 0 trace, 1 plane, 2 sdf, 3 target */
static inline void FN(ffwd_step)(int mode, const FN(fvol_t)* V, const REAL* sdf, REAL ds, FN(fstate_t)* s) {
  REAL n = 0, gx = 0, gy = 0, gz = 0;
  if (s->inside) {
    FN(fcell_t) c = FN(flocate)(V, s->x, s->y, s->z);
    FN(fsample_t) q = FN(finterp)(V->data, &c);
    n = q.n; gx = q.gx * V->inv_h; gy = q.gy * V->inv_h; gz = q.gz * V->inv_h;
  }
  REAL dsn = ds * n;
  s->vx = FMA(dsn, gx, s->vx); s->vy = FMA(dsn, gy, s->vy); s->vz = FMA(dsn, gz, s->vz);
  s->x = FMA(ds, s->vx, s->x); s->y = FMA(ds, s->vy, s->y); s->z = FMA(ds, s->vz, s->z);
  int cur_inside;
  if (mode == 2) {
    REAL d = 0;
    if (s->inside) { FN(fcell_t) c = FN(flocate)(V, s->x, s->y, s->z); d = FN(finterp)(sdf, &c).n; }
    cur_inside = d < 0;
  } else {
    cur_inside = FN(finbounds)(V, s->x, s->y, s->z);
    if (mode == 1) {
      REAL d = FN(fdot3)(s->x - s->a0, s->y - s->a1, s->z - s->a2, s->a3, s->a4, s->a5);
      cur_inside = cur_inside & !(d > 0);
    }
  }
  int cross = s->inside & !cur_inside;
  s->esc = s->esc | cross | FN(fescaped)(V, s->x, s->y, s->z, s->vx, s->vy, s->vz);
  if (mode == 3) {
    REAL ex = s->x - s->a0, ey = s->y - s->a1, ez = s->z - s->a2;
    REAL cur = FN(fdot3)(ex, ey, ez, ex, ey, ez);
    if (cur < s->a3) { s->xtx = s->x; s->xty = s->y; s->xtz = s->z; s->vtx = s->vx; s->vty = s->vy; s->vtz = s->vz; s->a3 = cur; }
  } else if (cross) {
    s->xtx = s->x; s->xty = s->y; s->xtz = s->z; s->vtx = s->vx; s->vty = s->vy; s->vtz = s->vz;
  }
  s->inside = cur_inside;
}

/* array-at-a-time forward march with the reference's global all(escaped) break
 * (src/tracer.cpp:66-87, :135-159, :209-234, :280-302), factored arithmetic.           */
static int FN(trace_fact)(int mode, const REAL* rif, const REAL* sdf, const int res[3], long long nvox,
                          size_t N, const REAL* pos, const REAL* vel, const REAL* aux_a, const REAL* aux_b,
                          REAL h, REAL ds, REAL* xt, REAL* vt, REAL* dist2, unsigned char* failmask,
                          int* steps_out, long long* n_failed, int* iters_out) {
  int rc = FN(check_res)(res, nvox); if (rc) return rc;
  int max_steps = (mode == 2) ? (int)((REAL)2 * h * (REAL)FN(max3i)(res) / ds)
                              : (int)((REAL)4 * h * (REAL)FN(max3i)(res) / ds);
  FN(fvol_t) V; FN(fvol_make)(&V, rif, res, h);
  FN(fstate_t)* S = (FN(fstate_t)*)malloc(sizeof(FN(fstate_t)) * (N ? N : 1));
  for (size_t i = 0; i < N; ++i) {
    FN(fstate_t)* s = S + i;
    s->x = pos[3*i]; s->y = pos[3*i+1]; s->z = pos[3*i+2];
    s->vx = vel[3*i]; s->vy = vel[3*i+1]; s->vz = vel[3*i+2];
    s->xtx = s->x; s->xty = s->y; s->xtz = s->z; s->vtx = s->vx; s->vty = s->vy; s->vtz = s->vz;
    s->a0 = s->a1 = s->a2 = s->a3 = s->a4 = s->a5 = 0;
    if (mode == 1) { s->a0 = aux_a[3*i]; s->a1 = aux_a[3*i+1]; s->a2 = aux_a[3*i+2];
                     s->a3 = aux_b[3*i]; s->a4 = aux_b[3*i+1]; s->a5 = aux_b[3*i+2]; }
    if (mode == 3) { s->a0 = aux_a[3*i]; s->a1 = aux_a[3*i+1]; s->a2 = aux_a[3*i+2];
                     REAL ex = s->x - s->a0, ey = s->y - s->a1, ez = s->z - s->a2;
                     s->a3 = FN(fdot3)(ex, ey, ez, ex, ey, ez); }
    s->inside = FN(finbounds)(&V, s->x, s->y, s->z);
    s->esc = 0; s->act = 1; s->steps = 0;
    if (mode == 2) { FN(fcell_t) c = FN(flocate)(&V, s->x, s->y, s->z); s->act = FN(finterp)(sdf, &c).n < 0; }
  }
  int it;
  for (it = 0; it < max_steps; ++it) {
    int all_escaped = 1;
    for (size_t i = 0; i < N; ++i) {
      FN(fstate_t)* s = S + i;
      int was = s->esc;
      FN(ffwd_step)(mode, &V, sdf, ds, s);          /* escaped rays keep updating (Q7) */
      if (!was) s->steps = it + 1;
      all_escaped &= s->esc;
    }
    if (all_escaped) { ++it; break; }
  }
  long long nf = 0;
  for (size_t i = 0; i < N; ++i) {
    FN(fstate_t)* s = S + i;
    s->act = s->act & !s->esc;
    nf += s->act;
    if ((mode == 0 || mode == 1) && !s->esc) { s->xtx = s->x; s->xty = s->y; s->xtz = s->z; }
    xt[3*i] = s->xtx; xt[3*i+1] = s->xty; xt[3*i+2] = s->xtz;
    vt[3*i] = s->vtx; vt[3*i+1] = s->vty; vt[3*i+2] = s->vtz;
    if (dist2) dist2[i] = s->a3;
    if (failmask) failmask[i] = !s->esc;
    if (steps_out) steps_out[i] = s->steps;
  }
  if (mode == 3) { nf = 0; for (size_t i = 0; i < N; ++i) nf += !S[i].esc; }
  if (n_failed) *n_failed = nf;
  if (iters_out) *iters_out = it;
  free(S);
  return 0;
}

typedef struct { REAL x, y, z, vx, vy, vz, lx, ly, lz, mx, my, mz; int active, outside; } FN(fadj_t);

static int FN(backtrace_fact)(int use_sdf, const REAL* rif, const REAL* sdf, const int res[3], long long nvox,
                              size_t N, const REAL* xt, const REAL* vt, const REAL* dx, const REAL* dv,
                              REAL h, REAL ds, REAL grad_scale, REAL* grad, long long* steps_total) {
  int rc = FN(check_res)(res, nvox); if (rc) return rc;
  memset(grad, 0, sizeof(REAL) * (size_t)nvox);
  int max_steps = (int)((REAL)2 * h * (REAL)FN(max3i)(res) / ds);
  FN(fvol_t) V; FN(fvol_make)(&V, rif, res, h);
  FN(fadj_t)* S = (FN(fadj_t)*)malloc(sizeof(FN(fadj_t)) * (N ? N : 1));
  long long steps = 0;
  for (size_t i = 0; i < N; ++i) {
    FN(fadj_t)* s = S + i;
    s->x = xt[3*i]; s->y = xt[3*i+1]; s->z = xt[3*i+2]; s->vx = vt[3*i]; s->vy = vt[3*i+1]; s->vz = vt[3*i+2];
    s->lx = dx[3*i]; s->ly = dx[3*i+1]; s->lz = dx[3*i+2];
    s->mx = FMA(ds, dx[3*i], dv[3*i]); s->my = FMA(ds, dx[3*i+1], dv[3*i+1]); s->mz = FMA(ds, dx[3*i+2], dv[3*i+2]);
    s->active = !FN(fescaped)(&V, s->x, s->y, s->z, -s->vx, -s->vy, -s->vz);
    s->outside = 0;
    if (use_sdf && s->active) { FN(fcell_t) c = FN(flocate)(&V, s->x, s->y, s->z); s->outside = FN(finterp)(sdf, &c).n >= 0; }
  }
  for (int it = 0; it < max_steps; ++it) {
    int any_active = 0;
    for (size_t i = 0; i < N; ++i) {
      FN(fadj_t)* s = S + i;
      if (!s->active) continue;
      s->x = FMA(-ds, s->vx, s->x); s->y = FMA(-ds, s->vy, s->y); s->z = FMA(-ds, s->vz, s->z);
      FN(fcell_t) c = FN(flocate)(&V, s->x, s->y, s->z);
      FN(fsample_t) q = FN(finterp)(rif, &c);
      REAL n = q.n, gx = q.gx * V.inv_h, gy = q.gy * V.inv_h, gz = q.gz * V.inv_h;
      REAL mdsn = -ds * n;
      s->vx = FMA(mdsn, gx, s->vx); s->vy = FMA(mdsn, gy, s->vy); s->vz = FMA(mdsn, gz, s->vz);
      int active = !FN(fescaped)(&V, s->x, s->y, s->z, -s->vx, -s->vy, -s->vz);
      if (use_sdf) {
        int now_out = FN(finterp)(sdf, &c).n >= 0;
        active = active & !((!s->outside) & now_out);
        s->outside = now_out;
      }
      s->active = active;
      if (!active) continue;
      any_active = 1; ++steps;
      { const REAL pp[3] = { s->x, s->y, s->z }; FN(sig_cell)(i, h, pp); }
      REAL dn = FN(fdot3)(s->mx, s->my, s->mz, gx, gy, gz);
      REAL nds = (n * ds) * grad_scale;
      REAL val = dn * ds, ggx = nds * s->mx, ggy = nds * s->my, ggz = nds * s->mz;
      {   /* fused splat: 8 corner sums */
        REAL x1 = c.wx, x0 = (REAL)1 - c.wx, y1 = c.wy, y0 = (REAL)1 - c.wy, z1 = c.wz, z0 = (REAL)1 - c.wz;
        REAL a0 = FMA(val, x0, -ggx), a1 = FMA(val, x1, ggx);
        REAL yz00 = y0*z0, yz10 = y1*z0, yz01 = y0*z1, yz11 = y1*z1;
        REAL gyz0 = ggy*z0, gyz1 = ggy*z1, gzy0 = ggz*y0, gzy1 = ggz*y1;
        REAL b00 = -gyz0 - gzy0, b10 = gyz0 - gzy1, b01 = gzy0 - gyz1, b11 = gyz1 + gzy1;
        REAL* g = grad + c.base;
        g[0]                   += FMA(yz00, a0, x0*b00);  g[c.ox]               += FMA(yz00, a1, x1*b00);
        g[c.oy]                += FMA(yz10, a0, x0*b10);  g[c.oy + c.ox]        += FMA(yz10, a1, x1*b10);
        g[c.oz]                += FMA(yz01, a0, x0*b01);  g[c.oz + c.ox]        += FMA(yz01, a1, x1*b01);
        g[c.oz + c.oy]         += FMA(yz11, a0, x0*b11);  g[c.oz + c.oy + c.ox] += FMA(yz11, a1, x1*b11);
      }
      REAL hxy = q.hxy * V.inv_h2, hxz = q.hxz * V.inv_h2, hyz = q.hyz * V.inv_h2;
      REAL hmx = FMA(hxz, s->mz, hxy * s->my);
      REAL hmy = FMA(hyz, s->mz, hxy * s->mx);
      REAL hmz = FMA(hyz, s->my, hxz * s->mx);
      s->lx = FMA(ds, FMA(dn, gx, n*hmx), s->lx);
      s->ly = FMA(ds, FMA(dn, gy, n*hmy), s->ly);
      s->lz = FMA(ds, FMA(dn, gz, n*hmz), s->lz);
      s->mx = FMA(ds, s->lx, s->mx); s->my = FMA(ds, s->ly, s->my); s->mz = FMA(ds, s->lz, s->mz);
    }
    if (!any_active) break;
  }
  if (steps_total) *steps_total = steps;
  free(S);
  return 0;
}

/* ---- cable, factored ---------------------------------------------------------------- */
typedef struct { const REAL* data; int rres; REAL radius, length, h, inv_h, r2; } FN(fcyl_t);
typedef struct { int i0, i1; REAL w0, r, rhx, rhz; int tiny; } FN(fcylcell_t);

static inline FN(fcyl_t) FN(fcyl_make)(const REAL* data, int rres, REAL radius, REAL length) {
  FN(fcyl_t) C; C.data = data; C.rres = rres; C.radius = radius; C.length = length;
  C.h = radius / (REAL)(rres - 1); C.inv_h = (REAL)1 / C.h; C.r2 = radius * radius;
  return C;
}
static inline FN(fcylcell_t) FN(fcyl_locate)(const FN(fcyl_t)* C, REAL px, REAL pz) {
  FN(fcylcell_t) c;
  REAL xs = px - C->radius, zs = pz - C->radius;
  c.r = SQRT(FMA(xs, xs, zs*zs));
  REAL rm = c.r * C->inv_h;
  int ir = FN(f2i_sat)(FLOOR(rm));
  c.i0 = FN(clampi)(ir, 0, C->rres - 1);
  c.i1 = FN(clampi)(c.i0 + 1, 0, C->rres - 1);
  c.w0 = rm - (REAL)c.i0;
  c.tiny = c.r < (REAL)1e-6f;
  REAL inv_r = c.tiny ? (REAL)0 : (REAL)1 / c.r;
  c.rhx = xs * inv_r; c.rhz = zs * inv_r;
  return c;
}
static inline int FN(fcyl_inbounds)(const FN(fcyl_t)* C, REAL px, REAL py, REAL pz) {
  REAL xs = px - C->radius, zs = pz - C->radius;
  return (FMA(xs, xs, zs*zs) < C->r2) & (py < C->length) & (py >= 0);
}
static inline int FN(fcyl_escaped)(const FN(fcyl_t)* C, REAL px, REAL py, REAL pz, REAL vx, REAL vy, REAL vz) {
  REAL xs = px - C->radius, zs = pz - C->radius;
  int esc_len = ((py < 0) & (vy < 0)) | ((py > C->length) & (vy > 0));
  int out_r = FMA(xs, xs, zs*zs) >= C->r2;
  int esc_r = FMA(xs, vx, zs*vz) > 0;
  return (out_r & esc_r) | esc_len;
}

static int FN(trace_cable_fact)(const REAL* rif, size_t rres, REAL radius, REAL length, size_t N,
                                const REAL* pos, const REAL* vel, const REAL* target, REAL ds,
                                REAL* xt, REAL* vt, REAL* dist2, long long* n_failed, long long* steps_total) {
  if (rres < 2) return -2;
  int max_steps = (int)((REAL)4 * length / ds);
  FN(fcyl_t) C = FN(fcyl_make)(rif, (int)rres, radius, length);
  long long steps = 0, nf = 0;
  for (size_t i = 0; i < N; ++i) {   /* ray-separable: masked state update freezes escaped rays */
    REAL x = pos[3*i], y = pos[3*i+1], z = pos[3*i+2], vx = vel[3*i], vy = vel[3*i+1], vz = vel[3*i+2];
    REAL xtx = x, xty = y, xtz = z, vtx = vx, vty = vy, vtz = vz;
    REAL t0 = target[3*i], t1 = target[3*i+1], t2 = target[3*i+2];
    REAL ex = x - t0, ey = y - t1, ez = z - t2;
    REAL best = FN(fdot3)(ex, ey, ez, ex, ey, ez);
    int inside = FN(fcyl_inbounds)(&C, x, y, z), esc = 0;
    for (int it = 0; it < max_steps; ++it) {
      FN(fcylcell_t) c = FN(fcyl_locate)(&C, x, z);
      REAL v0 = rif[c.i0], v1 = rif[c.i1];
      REAL f = FMA(v1, c.w0, v0 * ((REAL)1 - c.w0));
      REAL rx = (v1 - v0) * C.inv_h;
      REAL dsn = ds * f;
      vx = FMA(dsn, rx * c.rhx, vx); vz = FMA(dsn, rx * c.rhz, vz);
      x = FMA(ds, vx, x); y = FMA(ds, vy, y); z = FMA(ds, vz, z);
      ex = x - t0; ey = y - t1; ez = z - t2;
      REAL cur = FN(fdot3)(ex, ey, ez, ex, ey, ez);
      int cur_inside = FN(fcyl_inbounds)(&C, x, y, z);
      int cross = inside & !cur_inside;
      esc = esc | cross | FN(fcyl_escaped)(&C, x, y, z, vx, vy, vz);
      if (cur < best) { xtx = x; xty = y; xtz = z; vtx = vx; vty = vy; vtz = vz; best = cur; }
      ++steps;
      if (esc) break;
      inside = cur_inside;
    }
    nf += !esc;
    xt[3*i] = xtx; xt[3*i+1] = xty; xt[3*i+2] = xtz; vt[3*i] = vtx; vt[3*i+1] = vty; vt[3*i+2] = vtz;
    dist2[i] = best;
  }
  if (n_failed) *n_failed = nf;
  if (steps_total) *steps_total = steps;
  return 0;
}

static int FN(backtrace_cable_fact)(const REAL* rif, size_t rres, REAL radius, REAL length, size_t N,
                                    const REAL* xt, const REAL* vt, const REAL* dx, const REAL* dv,
                                    REAL ds, REAL* grad, long long* steps_total) {
  if (rres < 2) return -2;
  memset(grad, 0, sizeof(REAL) * rres);
  int max_steps = (int)((REAL)4 * length / ds);
  FN(fcyl_t) C = FN(fcyl_make)(rif, (int)rres, radius, length);
  long long steps = 0;
  for (size_t i = 0; i < N; ++i) {
    REAL x = xt[3*i], y = xt[3*i+1], z = xt[3*i+2], vx = vt[3*i], vy = vt[3*i+1], vz = vt[3*i+2];
    REAL lx = dx[3*i], ly = dx[3*i+1], lz = dx[3*i+2];
    REAL mx = FMA(ds, dx[3*i], dv[3*i]), my = FMA(ds, dx[3*i+1], dv[3*i+1]), mz = FMA(ds, dx[3*i+2], dv[3*i+2]);
    int active = !FN(fcyl_escaped)(&C, x, y, z, -vx, -vy, -vz);
    for (int it = 0; it < max_steps && active; ++it) {
      x = FMA(-ds, vx, x); y = FMA(-ds, vy, y); z = FMA(-ds, vz, z);
      FN(fcylcell_t) c = FN(fcyl_locate)(&C, x, z);
      REAL v0 = rif[c.i0], v1 = rif[c.i1];
      REAL w0 = c.w0, w1 = (REAL)1 - c.w0;
      REAL n = FMA(v1, w0, v0 * w1);
      REAL rx = (v1 - v0) * C.inv_h;
      REAL gx = rx * c.rhx, gz = rx * c.rhz;
      REAL mdsn = -ds * n;
      vx = FMA(mdsn, gx, vx); vz = FMA(mdsn, gz, vz);
      active = !FN(fcyl_escaped)(&C, x, y, z, -vx, -vy, -vz);
      if (!active) break;
      ++steps;
      REAL dn = FMA(mz, gz, mx * gx);
      REAL val = dn * ds;
      REAL gv = (n * ds) * FMA(mz, c.rhz, mx * c.rhx);
      REAL gvh = gv * C.inv_h;
      grad[c.i0] += FMA(val, w1, -gvh);
      grad[c.i1] += FMA(val, w0, gvh);
      REAL sH = c.tiny ? (REAL)0 : rx / c.r;
      REAL h00 = ((REAL)1 - c.rhx*c.rhx) * sH, h02 = -(c.rhx*c.rhz) * sH, h22 = ((REAL)1 - c.rhz*c.rhz) * sH;
      REAL hmx = FMA(h02, mz, h00 * mx), hmz = FMA(h22, mz, h02 * mx);
      lx = FMA(ds, FMA(dn, gx, n*hmx), lx);
      lz = FMA(ds, FMA(dn, gz, n*hmz), lz);
      mx = FMA(ds, lx, mx); my = FMA(ds, ly, my); mz = FMA(ds, lz, mz);
    }
  }
  if (steps_total) *steps_total = steps;
  return 0;
}
