/*
 * oracle/sdfr_oracle_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the sdfest hot path, written from the maths of the
 * reference (file:line cited per function), used ONLY as the parity checker in
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
 * under sdfest_amd/ may include, link or call this.
 *
 * This body is included twice by sdfr_oracle.c: once with REAL=float (suffix
 * _f32; the arithmetic type the reference CUDA kernels compute in) and once
 * with REAL=double (suffix _f64; the type of the reference's numpy twin, which
 * is what the golden vectors under tests/golden/ were captured from).
 *
 * Pinning: tests/test_oracle_golden.py checks every function below against
 * golden vectors produced by importing the reference itself
 * (tools/make_goldens.py; reference files simple_renderer.py, losses.py,
 * sdf_vae.py).  The CUDA-only "cuda_compat" d/dsdf weight permutation
 * (sdf_renderer_cuda.cu:373-388) has no importable reference and is pinned by
 * reading only -- it is NOT the default anywhere.
 *
 * Conventions:
 *   sdf      R*R*R, C-contiguous, index order sdf[x][y][z]
 *   quat     (x, y, z, w) scalar-last
 *   camera   OpenGL: looks down -z, y up; image row 0 is the top row
 *   depth    B*H*W row-major; 0 = no hit
 */

#ifndef REAL
#error "include via sdfr_oracle.c"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef struct { REAL x, y, z; } FN(v3);

static inline FN(v3) FN(v3_make)(REAL x, REAL y, REAL z) { FN(v3) r = {x, y, z}; return r; }
static inline FN(v3) FN(v3_add)(FN(v3) a, FN(v3) b) { return FN(v3_make)(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline FN(v3) FN(v3_sub)(FN(v3) a, FN(v3) b) { return FN(v3_make)(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline FN(v3) FN(v3_scale)(REAL s, FN(v3) a) { return FN(v3_make)(s * a.x, s * a.y, s * a.z); }
static inline REAL FN(v3_dot)(FN(v3) a, FN(v3) b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline FN(v3) FN(v3_cross)(FN(v3) a, FN(v3) b) {
  return FN(v3_make)(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* 3x3 rotation matrix of a scalar-last quaternion, in the "1 - 2(..)" form the
 * reference evaluates (sdf_renderer_cuda.cu:112-121).  m[r][c]. */
static inline void FN(quat_matrix)(const REAL* q, REAL m[3][3]) {
  const REAL x = q[0], y = q[1], z = q[2], w = q[3];
  m[0][0] = 1 - 2 * (y * y + z * z); m[0][1] = 2 * (x * y - w * z);     m[0][2] = 2 * (x * z + w * y);
  m[1][0] = 2 * (x * y + w * z);     m[1][1] = 1 - 2 * (x * x + z * z); m[1][2] = 2 * (y * z - w * x);
  m[2][0] = 2 * (x * z - w * y);     m[2][1] = 2 * (y * z + w * x);     m[2][2] = 1 - 2 * (x * x + y * y);
}
static inline FN(v3) FN(mat_apply)(const REAL m[3][3], FN(v3) v) {
  return FN(v3_make)(m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z,
                     m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                     m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z);
}
static inline FN(v3) FN(mat_apply_t)(const REAL m[3][3], FN(v3) v) {
  return FN(v3_make)(m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z,
                     m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
                     m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z);
}

/* Pixel -> unit ray direction.  sdf_renderer_cuda.cu:137-154 (the reference
 * forms the two ratios in double and rounds; so does this). */
static inline FN(v3) FN(pixel_ray)(int row, int col, double cx, double cy, double fx, double fy) {
  REAL dx = (REAL)(((double)col + 0.5 - cx) / fx);
  REAL dy = (REAL)(-((double)row + 0.5 - cy) / fy);
  REAL inv_len = (REAL)1 / SQRT(dx * dx + dy * dy + 1);
  return FN(v3_make)(dx * inv_len, dy * inv_len, -inv_len);
}

/* Ray (origin 0) against the oriented cube of half-width `scale` centred at p.
 * Slab method; sdf_renderer_cuda.cu:156-194 == simple_renderer.py:71-118.
 * `margin` (optional) is lowered to the smallest |lhs - rhs| of any comparison
 * that decided the outcome, so a test can tell fragile pixels from robust ones. */
static inline int FN(slab_clip)(FN(v3) d, FN(v3) p, const REAL rot[3][3], REAL scale,
                                REAL* t_near, REAL* t_far, REAL* margin) {
  REAL lo = (REAL)-1e-10, hi = (REAL)1e10;
  for (int axis = 0; axis < 3; ++axis) {
    FN(v3) a = FN(v3_make)(rot[0][axis], rot[1][axis], rot[2][axis]); /* R * e_axis */
    REAL e = FN(v3_dot)(a, p);
    REAL f = FN(v3_dot)(a, d);
    if (FABS(f) > (REAL)1e-20) {
      REAL ta = (e + scale) / f, tb = (e - scale) / f;
      if (ta > tb) { REAL s = ta; ta = tb; tb = s; }
      if (ta > lo) lo = ta;
      if (tb < hi) hi = tb;
      if (margin) {
        REAL m1 = FABS(lo - hi), m2 = FABS(hi);
        if (m1 < *margin) *margin = m1;
        if (m2 < *margin) *margin = m2;
      }
      if (lo > hi || hi < 0) return 0;
    } else if (-e > scale || -e < -scale) {
      return 0;
    }
  }
  *t_near = lo > 0 ? lo : 0;
  *t_far = hi;
  return 1;
}

typedef struct {
  int base[3];   /* clamped cell index */
  REAL off[3];   /* cell-local coordinate, NOT clamped (extrapolates) */
  REAL v[8];     /* corner values, index = 4*ix + 2*iy + iz */
} FN(cell);

/* Locate the cell of a point given in normalised object coordinates [-1,1]^3
 * and fetch its corners.  sdf_renderer_cuda.cu:196-239 == simple_renderer.py:158-219. */
static inline void FN(locate)(const REAL* sdf, int R, FN(v3) pn, FN(cell)* c) {
  const REAL g = (REAL)(2.0 / (R - 1));
  const REAL ginv = (REAL)((R - 1) / 2.0);
  const REAL pv[3] = {pn.x, pn.y, pn.z};
  for (int a = 0; a < 3; ++a) {
    int i = (int)FLOOR((pv[a] + (REAL)1) * (R - 1) * (REAL)0.5);
    if (i > R - 2) i = R - 2;
    if (i < 0) i = 0;
    c->base[a] = i;
    REAL pos0 = i * g - (REAL)1;
    c->off[a] = ginv * (pv[a] - pos0);
  }
  const size_t RR = (size_t)R * R;
  const REAL* p = sdf + c->base[0] * RR + (size_t)c->base[1] * R + c->base[2];
  c->v[0] = p[0];      c->v[1] = p[1];
  c->v[2] = p[R];      c->v[3] = p[R + 1];
  c->v[4] = p[RR];     c->v[5] = p[RR + 1];
  c->v[6] = p[RR + R]; c->v[7] = p[RR + R + 1];
}

/* Trilinear value; lerp order x, then y, then z (sdf_renderer_cuda.cu:231-238). */
static inline REAL FN(interp)(const FN(cell)* c) {
  const REAL ox = c->off[0], oy = c->off[1], oz = c->off[2];
  REAL c00 = c->v[0] * (1 - ox) + c->v[4] * ox;
  REAL c01 = c->v[1] * (1 - ox) + c->v[5] * ox;
  REAL c10 = c->v[2] * (1 - ox) + c->v[6] * ox;
  REAL c11 = c->v[3] * (1 - ox) + c->v[7] * ox;
  REAL c0 = c00 * (1 - oy) + c10 * oy;
  REAL c1 = c01 * (1 - oy) + c11 * oy;
  return c0 * (1 - oz) + c1 * oz;
}

/* Gradient of the trilinear value w.r.t. the cell-local coordinate. */
static inline FN(v3) FN(interp_grad)(const FN(cell)* c) {
  const REAL ox = c->off[0], oy = c->off[1], oz = c->off[2];
  const REAL* v = c->v;
  REAL c00 = v[0] * (1 - ox) + v[4] * ox, c01 = v[1] * (1 - ox) + v[5] * ox;
  REAL c10 = v[2] * (1 - ox) + v[6] * ox, c11 = v[3] * (1 - ox) + v[7] * ox;
  REAL gx = ((v[4] - v[0]) * (1 - oy) + (v[6] - v[2]) * oy) * (1 - oz) +
            ((v[5] - v[1]) * (1 - oy) + (v[7] - v[3]) * oy) * oz;
  REAL gy = (c10 - c00) * (1 - oz) + (c11 - c01) * oz;
  REAL gz = (c01 * (1 - oy) + c11 * oy) - (c00 * (1 - oy) + c10 * oy);
  return FN(v3_make)(gx, gy, gz);
}

/* Weight of corner k (= 4*ix+2*iy+iz) in the trilinear value.
 * mode 0: the mathematically correct weights (simple_renderer.py:399-408).
 * mode 1: the permuted weights the CUDA kernel really adds
 *         (sdf_renderer_cuda.cu:373-388). */
static inline void FN(corner_weights)(const REAL off[3], int mode, REAL w[8]) {
  const REAL x1 = off[0], y1 = off[1], z1 = off[2];
  const REAL x0 = 1 - x1, y0 = 1 - y1, z0 = 1 - z1;
  if (mode == 0) {
    w[0] = x0 * y0 * z0; w[1] = x0 * y0 * z1; w[2] = x0 * y1 * z0; w[3] = x0 * y1 * z1;
    w[4] = x1 * y0 * z0; w[5] = x1 * y0 * z1; w[6] = x1 * y1 * z0; w[7] = x1 * y1 * z1;
  } else {
    w[0] = x0 * y0 * z1; w[1] = x0 * y1 * z0; w[2] = x0 * y1 * z1; w[3] = x1 * y0 * z0;
    w[4] = x1 * y0 * z1; w[5] = x1 * y0 * z1; w[6] = x1 * y1 * z0; w[7] = x1 * y1 * z1;
  }
}

/* ------------------------------------------------------------------------ */
/* Depth render, forward.  sdf_renderer_cuda.cu:241-298 ==
 * simple_renderer.py:253-314 + :120-156.
 * steps  (optional, B*H*W int)  number of SDF evaluations per ray
 * margin (optional, B*H*W REAL) smallest |lhs-rhs| over every branch decision
 * max_steps <= 0: unbounded like the reference. */
void FN(sdfo_render_forward)(const REAL* sdf, int R, const REAL* pos, const REAL* quat,
                             const REAL* inv_scale, int B, int W, int H, double cx, double cy,
                             double fx, double fy, double threshold, REAL* depth, int* steps,
                             REAL* margin, int max_steps) {
  const REAL thr = (REAL)threshold;
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
  for (int b = 0; b < B; ++b) {
    for (int row = 0; row < H; ++row) {
      const REAL isc = inv_scale[b];
      const REAL scale = (REAL)1 / isc;
      REAL rot[3][3];
      FN(quat_matrix)(quat + 4 * b, rot);
      const FN(v3) p = FN(v3_make)(pos[3 * b], pos[3 * b + 1], pos[3 * b + 2]);
      const FN(v3) org = FN(mat_apply_t)(rot, FN(v3_make)(-p.x, -p.y, -p.z));
      for (int col = 0; col < W; ++col) {
        const size_t pix = ((size_t)b * H + row) * W + col;
        REAL out = 0, mg = (REAL)1e30;
        int n = 0;
        FN(v3) d = FN(pixel_ray)(row, col, cx, cy, fx, fy);
        REAL t, t_far;
        if (FN(slab_clip)(d, p, rot, scale, &t, &t_far, (margin && !g_margin_hit_tests_only) ? &mg : 0)) {
          FN(v3) dobj = FN(mat_apply_t)(rot, d);
          while (t < t_far) {
            if (max_steps > 0 && n >= max_steps) break;
            FN(v3) pt = FN(v3_add)(org, FN(v3_scale)(t, dobj));
            FN(cell) c;
            FN(locate)(sdf, R, FN(v3_scale)(isc, pt), &c);
            REAL dist = FN(interp)(&c) * scale;
            ++n;
            REAL m = FABS(dist - thr * t);
            if (m < mg) mg = m;
            if (dist < thr * t) { out = -t * d.z; break; }
            t += dist;
            m = FABS(t - t_far);
            if (m < mg && !g_margin_hit_tests_only) mg = m;
          }
        }
        depth[pix] = out;
        if (steps) steps[pix] = n;
        if (margin) margin[pix] = mg;
      }
    }
  }
}

/* Per-pixel first-order derivatives of depth.  Fills dz[8] in the order
 * (px, py, pz, qx, qy, qz, qw, inv_scale), the cell, and f = scale*|d.z|.
 * sdf_renderer_cuda.cu:334-457 == simple_renderer.py:317-458, re-derived:
 *   c      = s * Rhom(q)^T (x - p) + const,      s = inv_scale / g
 *   dc/dp_j = -s * R[j][:]
 *   dc/dq_k = s * ( d/dq_k[Rhom^T v] - 2 q_k o ),  v = x - p, o = R^T v
 *   dc/ds^-1 = o / g
 *   dz      = scale * |d.z| * (grad_tri . dc);   dz[7] -= tri * scale^2 * |d.z|
 * with Rhom^T v = (w^2-|u|^2) v + 2 u (u.v) - 2 w (u x v), u = (qx,qy,qz). */
static inline void FN(pixel_derivs)(const REAL* sdf, int R, REAL z, FN(v3) d, FN(v3) p,
                                    const REAL* q, const REAL rot[3][3], REAL isc, FN(cell)* c,
                                    REAL dz[8], REAL* fout) {
  const REAL scale = (REAL)1 / isc;
  const REAL g = (REAL)(2.0 / (R - 1));
  const REAL ginv = (REAL)1 / g;
  const REAL t = -z / d.z;
  FN(v3) xw = FN(v3_scale)(t, d);
  FN(v3) org = FN(mat_apply_t)(rot, FN(v3_make)(-p.x, -p.y, -p.z));
  FN(v3) dobj = FN(mat_apply_t)(rot, d);
  FN(v3) o = FN(v3_add)(org, FN(v3_scale)(t, dobj));
  FN(locate)(sdf, R, FN(v3_scale)(isc, o), c);
  const REAL tri = FN(interp)(c);
  const FN(v3) G = FN(interp_grad)(c);
  const REAL adz = FABS(d.z);
  const REAL s = isc * ginv;
  const FN(v3) v = FN(v3_sub)(xw, p);
  const FN(v3) u = FN(v3_make)(q[0], q[1], q[2]);
  const REAL w = q[3];
  /* position: dc/dp_j = -s * row j of R */
  const FN(v3) RG = FN(mat_apply)(rot, G);
  dz[0] = scale * (-s * RG.x) * adz;
  dz[1] = scale * (-s * RG.y) * adz;
  dz[2] = scale * (-s * RG.z) * adz;
  /* quaternion */
  const REAL udv = FN(v3_dot)(u, v);
  const FN(v3) uxv = FN(v3_cross)(u, v);
  const FN(v3) e[3] = {FN(v3_make)(1, 0, 0), FN(v3_make)(0, 1, 0), FN(v3_make)(0, 0, 1)};
  const REAL uk[3] = {u.x, u.y, u.z}, vk[3] = {v.x, v.y, v.z};
  for (int k = 0; k < 3; ++k) {
    /* d/du_k = -2 u_k v + 2 e_k (u.v) + 2 u v_k - 2 w (e_k x v) */
    FN(v3) a = FN(v3_scale)(-2 * uk[k], v);
    a = FN(v3_add)(a, FN(v3_scale)(2 * udv, e[k]));
    a = FN(v3_add)(a, FN(v3_scale)(2 * vk[k], u));
    a = FN(v3_sub)(a, FN(v3_scale)(2 * w, FN(v3_cross)(e[k], v)));
    a = FN(v3_sub)(a, FN(v3_scale)(2 * uk[k], o));
    dz[3 + k] = scale * (s * FN(v3_dot)(G, a)) * adz;
  }
  {
    /* d/dw = 2 w v - 2 (u x v) */
    FN(v3) a = FN(v3_sub)(FN(v3_scale)(2 * w, v), FN(v3_scale)(2, uxv));
    a = FN(v3_sub)(a, FN(v3_scale)(2 * w, o));
    dz[6] = scale * (s * FN(v3_dot)(G, a)) * adz;
  }
  dz[7] = scale * (ginv * FN(v3_dot)(G, o)) * adz - (tri * scale * scale) * adz;
  *fout = scale * adz;
}

/* Depth render, backward.  sdf_renderer_cuda.cu:300-468 (+ host launcher
 * :512-556 for the zero-fill) == simple_renderer.py:317-458 reduced as in
 * sdf_renderer.py:242-261.  Sums are carried in double in both builds so that
 * the oracle is the summation-order-free reference value (the _f32 build: every
 * per-pixel term in float, summed exactly -- the floor a float kernel can reach).
 * g_sdf R^3 is the sum over all B views; g_pos B*3, g_quat B*4, g_inv_scale B. */
void FN(sdfo_render_backward)(const REAL* grad_depth, const REAL* depth, const REAL* sdf, int R,
                              const REAL* pos, const REAL* quat, const REAL* inv_scale, int B,
                              int W, int H, double cx, double cy, double fx, double fy,
                              int sdf_grad_mode, REAL* g_sdf, REAL* g_pos, REAL* g_quat,
                              REAL* g_inv_scale) {
  const size_t nvox = (size_t)R * R * R;
  const size_t RR = (size_t)R * R;
  /* (view, row) pairs in parallel; every thread adds into its OWN double volume and its own per-view pose sums,
   * both reduced over the threads afterwards in thread order (doubles: order-dependent at the 1e-16 level only) */
  const int T = omp_get_max_threads();
  sdfo_slab* slabs = sdfo_slabs(T);
  double* pose_acc = (double*)calloc((size_t)T * B * 8, sizeof(double));
  for (int t = 0; t < T; ++t) { slabs[t].lo = nvox; slabs[t].hi = 0; }
  const double tm0 = sdfo_now();
#pragma omp parallel
  {
    const int t = omp_get_thread_num();
    sdfo_slab* mine = slabs + t;
    double* acc_sdf = NULL;
    double* accs = pose_acc + (size_t)t * B * 8;
    size_t lo = nvox, hi = 0;
#pragma omp for collapse(2) schedule(dynamic, 8)
    for (int b = 0; b < B; ++b) {
      for (int row = 0; row < H; ++row) {
        REAL rot[3][3];
        FN(quat_matrix)(quat + 4 * b, rot);
        const FN(v3) p = FN(v3_make)(pos[3 * b], pos[3 * b + 1], pos[3 * b + 2]);
        double* acc = accs + (size_t)b * 8;
        for (int col = 0; col < W; ++col) {
          const size_t pix = ((size_t)b * H + row) * W + col;
          const REAL z = depth[pix];
          if (z == 0) continue;
          if (!acc_sdf) acc_sdf = sdfo_slab_of(mine, nvox);
          const REAL go = grad_depth[pix];
          FN(v3) d = FN(pixel_ray)(row, col, cx, cy, fx, fy);
          FN(cell) c;
          REAL dz[8], f, wgt[8];
          FN(pixel_derivs)(sdf, R, z, d, p, quat + 4 * b, rot, inv_scale[b], &c, dz, &f);
          for (int k = 0; k < 8; ++k) acc[k] += (double)(dz[k] * go);
          FN(corner_weights)(c.off, sdf_grad_mode, wgt);
          const size_t first = c.base[0] * RR + (size_t)c.base[1] * R + c.base[2];
          double* a = acc_sdf + first;
          const size_t offs[8] = {0, 1, (size_t)R, (size_t)R + 1, RR, RR + 1, RR + R, RR + R + 1};
          for (int k = 0; k < 8; ++k) a[offs[k]] += (double)(go * wgt[k] * f);
          if (first < lo) lo = first;
          if (first + RR + R + 2 > hi) hi = first + RR + R + 2;
        }
      }
    }
    mine->lo = lo;
    mine->hi = hi;
  }
  const double tm1 = sdfo_now();
  size_t LO = nvox, HI = 0;
  for (int t = 0; t < T; ++t) {
    if (slabs[t].lo < LO) LO = slabs[t].lo;
    if (slabs[t].hi > HI) HI = slabs[t].hi;
  }
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < nvox; ++i) {
    double sum = 0;
    if (i >= LO && i < HI)
      for (int t = 0; t < T; ++t)
        if (i >= slabs[t].lo && i < slabs[t].hi) { sum += slabs[t].v[i]; slabs[t].v[i] = 0; }
    g_sdf[i] = (REAL)sum;
  }
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < 8; ++k) acc[k] += pose_acc[((size_t)t * B + b) * 8 + k];
    g_pos[3 * b] = (REAL)acc[0]; g_pos[3 * b + 1] = (REAL)acc[1]; g_pos[3 * b + 2] = (REAL)acc[2];
    g_quat[4 * b] = (REAL)acc[3]; g_quat[4 * b + 1] = (REAL)acc[4];
    g_quat[4 * b + 2] = (REAL)acc[5]; g_quat[4 * b + 3] = (REAL)acc[6];
    g_inv_scale[b] = (REAL)acc[7];
  }
  free(pose_acc);
  if (g_sdfo_timing)
    fprintf(stderr, "sdfo backward: %d threads, %d views: pixels %.2f ms, reduce %.2f ms\n", T, B, (tm1 - tm0) * 1e3,
            (sdfo_now() - tm1) * 1e3);
}

/* Per-pixel derivative images, dimg[B][H][W][8] in the order of pixel_derivs;
 * zero where depth == 0.  Lets the golden test compare against the numpy
 * twin's derivative images one by one (simple_renderer.py:444-458). */
void FN(sdfo_render_derivative_images)(const REAL* depth, const REAL* sdf, int R, const REAL* pos,
                                       const REAL* quat, const REAL* inv_scale, int B, int W,
                                       int H, double cx, double cy, double fx, double fy,
                                       REAL* dimg) {
  for (int b = 0; b < B; ++b) {
    REAL rot[3][3];
    FN(quat_matrix)(quat + 4 * b, rot);
    const FN(v3) p = FN(v3_make)(pos[3 * b], pos[3 * b + 1], pos[3 * b + 2]);
    for (int row = 0; row < H; ++row)
      for (int col = 0; col < W; ++col) {
        const size_t pix = ((size_t)b * H + row) * W + col;
        REAL* out = dimg + 8 * pix;
        for (int k = 0; k < 8; ++k) out[k] = 0;
        if (depth[pix] == 0) continue;
        FN(v3) d = FN(pixel_ray)(row, col, cx, cy, fx, fy);
        FN(cell) c;
        REAL f;
        FN(pixel_derivs)(sdf, R, depth[pix], d, p, quat + 4 * b, rot, inv_scale[b], &c, out, &f);
      }
  }
}

/* ------------------------------------------------------------------------ */
/* Trilinear SDF sampler of the point-cloud loss.  estimation/losses.py:32-135.
 * Normalises q, rotates by conj(q^), divides by scale, floors to a cell,
 * points whose UNclamped cell index leaves [0, R-2] on any axis give 0.
 * `pc` (optional) receives per point: cell base (3 ints as REAL), off(3), mask. */
typedef struct {
  REAL qn[4];      /* normalised quaternion */
  REAL norm;       /* |q| */
  REAL rot[3][3];  /* R(q^) (object->world); the sampler applies its transpose */
} FN(pc_frame);

static inline void FN(pc_setup)(const REAL* quat, FN(pc_frame)* fr) {
  REAL n = SQRT(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
  fr->norm = n;
  for (int i = 0; i < 4; ++i) fr->qn[i] = quat[i] / n;
  FN(quat_matrix)(fr->qn, fr->rot);
}

/* returns 1 if the point is inside the volume (mask == False in the reference) */
static inline int FN(pc_locate)(const REAL* sdf, int R, FN(v3) pn, FN(cell)* c) {
  const REAL g = (REAL)(2.0 / (R - 1));
  const REAL pv[3] = {pn.x, pn.y, pn.z};
  int inside = 1;
  for (int a = 0; a < 3; ++a) {
    REAL cf = FLOOR((pv[a] + (REAL)1.0) * (R - 1) * (REAL)0.5);
    if (cf < 0 || cf > R - 2) inside = 0;
    if (cf < 0) cf = 0;
    if (cf > R - 2) cf = (REAL)(R - 2);
    c->base[a] = (int)cf;
    REAL cellpos = cf * g - (REAL)1.0;
    c->off[a] = (pv[a] - cellpos) / g;
  }
  const size_t RR = (size_t)R * R;
  const REAL* p = sdf + c->base[0] * RR + (size_t)c->base[1] * R + c->base[2];
  c->v[0] = p[0];      c->v[1] = p[1];
  c->v[2] = p[R];      c->v[3] = p[R + 1];
  c->v[4] = p[RR];     c->v[5] = p[RR + 1];
  c->v[6] = p[RR + R]; c->v[7] = p[RR + R + 1];
  return inside;
}

void FN(sdfo_pc_loss_forward)(const REAL* points, int M, const REAL* pos, const REAL* quat,
                              const REAL* scale_p, const REAL* sdf, int R, REAL* out) {
  FN(pc_frame) fr;
  FN(pc_setup)(quat, &fr);
  const REAL scale = scale_p[0];
  const FN(v3) p = FN(v3_make)(pos[0], pos[1], pos[2]);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; ++i) {
    FN(v3) P = FN(v3_make)(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    FN(v3) o = FN(mat_apply_t)(fr.rot, FN(v3_sub)(P, p));
    FN(v3) pn = FN(v3_make)(o.x / scale, o.y / scale, o.z / scale);
    FN(cell) c;
    int inside = FN(pc_locate)(sdf, R, pn, &c);
    out[i] = inside ? FN(interp)(&c) * scale : 0;
  }
}

/* VJP of the sampler for upstream grad_out[M]; equals torch autograd through
 * losses.py:32-135 (floor/clip/take have zero/identity gradient; the masked
 * assignment cuts the gradient of outside points; the q normalisation is
 * differentiated through).  g_sdf is R^3 (overwritten). */
void FN(sdfo_pc_loss_backward)(const REAL* grad_out, const REAL* points, int M, const REAL* pos,
                               const REAL* quat, const REAL* scale_p, const REAL* sdf, int R,
                               REAL* g_sdf, REAL* g_pos, REAL* g_quat, REAL* g_scale) {
  const size_t nvox = (size_t)R * R * R;
  const size_t RR = (size_t)R * R;
  FN(pc_frame) fr;
  FN(pc_setup)(quat, &fr);
  const REAL scale = scale_p[0];
  const REAL g = (REAL)(2.0 / (R - 1));
  const FN(v3) p = FN(v3_make)(pos[0], pos[1], pos[2]);
  const FN(v3) u = FN(v3_make)(fr.qn[0], fr.qn[1], fr.qn[2]);
  const REAL w = fr.qn[3];
  double* acc_sdf = (double*)calloc(nvox, sizeof(double));
  double ap[3] = {0, 0, 0}, aq[4] = {0, 0, 0, 0}, as = 0;
  for (int i = 0; i < M; ++i) {
    FN(v3) P = FN(v3_make)(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    FN(v3) v = FN(v3_sub)(P, p);
    FN(v3) o = FN(mat_apply_t)(fr.rot, v);
    FN(v3) pn = FN(v3_make)(o.x / scale, o.y / scale, o.z / scale);
    FN(cell) c;
    if (!FN(pc_locate)(sdf, R, pn, &c)) continue;
    const REAL go = grad_out[i];
    const REAL tri = FN(interp)(&c);
    const FN(v3) G = FN(interp_grad)(&c); /* d tri / d off */
    REAL wgt[8];
    FN(corner_weights)(c.off, 0, wgt);
    double* a = acc_sdf + c.base[0] * RR + (size_t)c.base[1] * R + c.base[2];
    const REAL gs = go * scale;
    a[0] += (double)(gs * wgt[0]);      a[1] += (double)(gs * wgt[1]);
    a[R] += (double)(gs * wgt[2]);      a[R + 1] += (double)(gs * wgt[3]);
    a[RR] += (double)(gs * wgt[4]);     a[RR + 1] += (double)(gs * wgt[5]);
    a[RR + R] += (double)(gs * wgt[6]); a[RR + R + 1] += (double)(gs * wgt[7]);
    /* value = tri(off) * scale, off = (o/scale - cellpos)/g
     * d value / d o = G / g           (scale cancels)
     * d value / d scale = tri - (G . o) / (g * scale) */
    const FN(v3) dvo = FN(v3_make)(G.x / g, G.y / g, G.z / g);
    as += (double)(go * (tri - FN(v3_dot)(dvo, o) / scale));
    /* o = R^T (P - p): d/dp = -R dvo */
    FN(v3) rg = FN(mat_apply)(fr.rot, dvo);
    ap[0] -= (double)(go * rg.x); ap[1] -= (double)(go * rg.y); ap[2] -= (double)(go * rg.z);
    /* d o / d q^ for the "1-2(..)" matrix form: differentiate R^T v entrywise.
     * R^T v = v - 2|u|^2 v + 2 u (u.v) - 2 w (u x v)   (unit-norm-free form used
     * by losses.py:65-77, i.e. NOT the homogeneous form) */
    const REAL udv = FN(v3_dot)(u, v);
    const FN(v3) e[3] = {FN(v3_make)(1, 0, 0), FN(v3_make)(0, 1, 0), FN(v3_make)(0, 0, 1)};
    const REAL uk[3] = {u.x, u.y, u.z}, vk[3] = {v.x, v.y, v.z};
    REAL gq[4];
    for (int k = 0; k < 3; ++k) {
      FN(v3) d = FN(v3_scale)(-4 * uk[k], v);
      d = FN(v3_add)(d, FN(v3_scale)(2 * udv, e[k]));
      d = FN(v3_add)(d, FN(v3_scale)(2 * vk[k], u));
      d = FN(v3_sub)(d, FN(v3_scale)(2 * w, FN(v3_cross)(e[k], v)));
      gq[k] = FN(v3_dot)(dvo, d);
    }
    gq[3] = FN(v3_dot)(dvo, FN(v3_scale)(-2, FN(v3_cross)(u, v)));
    for (int k = 0; k < 4; ++k) aq[k] += (double)(go * gq[k]);
  }
  /* through q^ = q/|q|:  g_q = (g_q^ - q^ (q^ . g_q^)) / |q| */
  double dotq = 0;
  for (int k = 0; k < 4; ++k) dotq += aq[k] * (double)fr.qn[k];
  for (int k = 0; k < 4; ++k) g_quat[k] = (REAL)((aq[k] - (double)fr.qn[k] * dotq) / (double)fr.norm);
  for (int k = 0; k < 3; ++k) g_pos[k] = (REAL)ap[k];
  g_scale[0] = (REAL)as;
  for (size_t i = 0; i < nvox; ++i) g_sdf[i] = (REAL)acc_sdf[i];
  free(acc_sdf);
}

/* ------------------------------------------------------------------------ */
/* VAE decoder forward.  vae/sdf_vae.py:217-259 (SDFDecoder.forward).
 * params: flat buffer in state_dict order: fc{i}.weight [out,in], fc{i}.bias,
 * ..., conv{i}.weight [cout,cin,k,k,k], conv{i}.bias.
 * Trilinear resize = ATen upsample_trilinear3d, align_corners=False:
 *   src = max((dst+0.5)*in/out - 0.5, 0); i0=floor(src); i1=i0+(i0<in-1);
 *   l1=src-i0; l0=1-l1   (separable per axis). */
static void FN(resize3)(const REAL* in, int C, int n_in, int n_out, REAL* out) {
  int* i0 = (int*)malloc(sizeof(int) * n_out);
  int* i1 = (int*)malloc(sizeof(int) * n_out);
  REAL* l1 = (REAL*)malloc(sizeof(REAL) * n_out);
  const REAL ratio = (REAL)n_in / (REAL)n_out;
  for (int d = 0; d < n_out; ++d) {
    REAL src = ratio * ((REAL)d + (REAL)0.5) - (REAL)0.5;
    if (src < 0) src = 0;
    int a = (int)src;
    if (a > n_in - 1) a = n_in - 1;
    i0[d] = a;
    i1[d] = a + (a < n_in - 1 ? 1 : 0);
    l1[d] = src - a;
  }
  const size_t si = (size_t)n_in * n_in * n_in, so = (size_t)n_out * n_out * n_out;
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < C; ++c)
    for (int x = 0; x < n_out; ++x) {
      const REAL* ic = in + c * si;
      REAL* oc = out + c * so;
      const REAL wx1 = l1[x], wx0 = 1 - wx1;
      for (int y = 0; y < n_out; ++y) {
        const REAL wy1 = l1[y], wy0 = 1 - wy1;
        for (int z = 0; z < n_out; ++z) {
          const REAL wz1 = l1[z], wz0 = 1 - wz1;
#define AT(ix, iy, iz) ic[((size_t)(ix) * n_in + (iy)) * n_in + (iz)]
          REAL v = wx0 * (wy0 * (wz0 * AT(i0[x], i0[y], i0[z]) + wz1 * AT(i0[x], i0[y], i1[z])) +
                          wy1 * (wz0 * AT(i0[x], i1[y], i0[z]) + wz1 * AT(i0[x], i1[y], i1[z]))) +
                   wx1 * (wy0 * (wz0 * AT(i1[x], i0[y], i0[z]) + wz1 * AT(i1[x], i0[y], i1[z])) +
                          wy1 * (wz0 * AT(i1[x], i1[y], i0[z]) + wz1 * AT(i1[x], i1[y], i1[z])));
#undef AT
          oc[((size_t)x * n_out + y) * n_out + z] = v;
        }
      }
    }
  free(i0); free(i1); free(l1);
}

static void FN(conv3_valid)(const REAL* in, int cin, int n, const REAL* wgt, const REAL* bias,
                            int cout, int k, int relu, REAL* out) {
  const int m = n - k + 1;
  const size_t si = (size_t)n * n * n, so = (size_t)m * m * m;
#pragma omp parallel for collapse(2) schedule(static)
  for (int co = 0; co < cout; ++co)
    for (int x = 0; x < m; ++x)
      for (int y = 0; y < m; ++y)
        for (int z = 0; z < m; ++z) {
          REAL acc = bias[co];
          for (int ci = 0; ci < cin; ++ci)
            for (int a = 0; a < k; ++a)
              for (int b = 0; b < k; ++b)
                for (int c = 0; c < k; ++c)
                  acc += wgt[((((size_t)co * cin + ci) * k + a) * k + b) * k + c] *
                         in[ci * si + ((size_t)(x + a) * n + (y + b)) * n + (z + c)];
          if (relu && acc < 0) acc = 0;
          out[co * so + ((size_t)x * m + y) * m + z] = acc;
        }
}

/* Returns 0 on success.  out: N * volume^3.  Layer description arrays are the
 * yaml's decoder.fc_layers[].out and decoder.conv_layers[].{in_size,
 * in_channels,out_channels,kernel_size,relu}. */
int FN(sdfo_decoder_forward)(const REAL* params, int latent, int n_fc, const int* fc_out,
                             int n_conv, const int* conv_in_size, const int* conv_cin,
                             const int* conv_cout, const int* conv_k, const int* conv_relu,
                             int volume, double tsdf, int enforce_tsdf, const REAL* z, int N,
                             REAL* out) {
  size_t maxbuf = 0;
  {
    size_t s = (size_t)latent;
    for (int i = 0; i < n_fc; ++i) if ((size_t)fc_out[i] > s) s = fc_out[i];
    maxbuf = s;
    for (int i = 0; i < n_conv; ++i) {
      size_t a = (size_t)conv_cin[i] * conv_in_size[i] * conv_in_size[i] * conv_in_size[i];
      int m = conv_in_size[i] - conv_k[i] + 1;
      size_t b = (size_t)conv_cout[i] * m * m * m;
      if (a > maxbuf) maxbuf = a;
      if (b > maxbuf) maxbuf = b;
    }
    size_t v = (size_t)volume * volume * volume;
    if (v > maxbuf) maxbuf = v;
  }
  REAL* bufa = (REAL*)malloc(sizeof(REAL) * maxbuf);
  REAL* bufb = (REAL*)malloc(sizeof(REAL) * maxbuf);
  const size_t vox = (size_t)volume * volume * volume;
  for (int nidx = 0; nidx < N; ++nidx) {
    const REAL* pp = params;
    int width = latent;
    for (int i = 0; i < latent; ++i) bufa[i] = z[(size_t)nidx * latent + i];
    for (int l = 0; l < n_fc; ++l) {
      const REAL* wgt = pp;
      const REAL* bias = pp + (size_t)fc_out[l] * width;
      for (int o = 0; o < fc_out[l]; ++o) {
        REAL acc = bias[o];
        for (int i = 0; i < width; ++i) acc += wgt[(size_t)o * width + i] * bufa[i];
        bufb[o] = acc > 0 ? acc : 0;
      }
      pp = bias + fc_out[l];
      width = fc_out[l];
      REAL* t = bufa; bufa = bufb; bufb = t;
    }
    int cur_c = conv_cin[0], cur_n = conv_in_size[0];
    if ((size_t)cur_c * cur_n * cur_n * cur_n != (size_t)width) { free(bufa); free(bufb); return -1; }
    for (int l = 0; l < n_conv; ++l) {
      if (cur_c != conv_cin[l]) { free(bufa); free(bufb); return -2; }
      if (cur_n != conv_in_size[l]) {
        FN(resize3)(bufa, cur_c, cur_n, conv_in_size[l], bufb);
        cur_n = conv_in_size[l];
        REAL* t = bufa; bufa = bufb; bufb = t;
      }
      const int k = conv_k[l];
      const REAL* wgt = pp;
      const REAL* bias = pp + (size_t)conv_cout[l] * conv_cin[l] * k * k * k;
      FN(conv3_valid)(bufa, cur_c, cur_n, wgt, bias, conv_cout[l], k, conv_relu[l], bufb);
      pp = bias + conv_cout[l];
      cur_c = conv_cout[l];
      cur_n = cur_n - k + 1;
      REAL* t = bufa; bufa = bufb; bufb = t;
    }
    if (cur_c != 1) { free(bufa); free(bufb); return -3; }
    if (cur_n != volume) {
      FN(resize3)(bufa, 1, cur_n, volume, bufb);
      REAL* t = bufa; bufa = bufb; bufb = t;
    }
    REAL* o = out + (size_t)nidx * vox;
    for (size_t i = 0; i < vox; ++i) {
      REAL v = bufa[i];
      if (enforce_tsdf && tsdf > 0) {
        if (v < (REAL)-tsdf) v = (REAL)-tsdf;
        if (v > (REAL)tsdf) v = (REAL)tsdf;
      }
      o[i] = v;
    }
  }
  free(bufa); free(bufb);
  return 0;
}

/* Depth image -> point cloud, OpenGL convention.
 * initialization/pointset_utils.py:57-77 (pixel-centre-0 intrinsics: the
 * caller passes cx0 = cx - pixel_center etc.).  Emits points for depth != 0 in
 * row-major order (torch.nonzero order); returns the count. */
int FN(sdfo_depth_to_pointcloud)(const REAL* depth, int W, int H, double fx, double fy, double cx0,
                                 double cy0, REAL* points) {
  int n = 0;
  for (int row = 0; row < H; ++row)
    for (int col = 0; col < W; ++col) {
      REAL zv = depth[(size_t)row * W + col];
      if (zv == 0) continue;
      points[3 * n] = ((REAL)col - (REAL)cx0) * zv / (REAL)fx;
      points[3 * n + 1] = -((REAL)row - (REAL)cy0) * zv / (REAL)fy;
      points[3 * n + 2] = -zv;
      ++n;
    }
  return n;
}

#undef CAT_
#undef CAT
#undef FN
