/*
 * oracle/sdfr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Instantiates the CPU restatement in sdfr_oracle_impl.h for float (_f32) and
 * double (_f64).  Built by oracle/Makefile into oracle/libsdfr_oracle.so and
 * loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
 *
 * Parity status: PINNED against golden vectors captured by importing the
 * reference's numpy renderer, torch pc_loss and torch SDFDecoder in the build
 * container (tools/make_goldens.py -> tests/golden/ npz files).  The reference's
 * native CUDA path cannot be compiled here (no nvcc; CUDA-only sources), so
 * there is no oracle/_ref build; see DESIGN.md "Oracle".
 */
#include <math.h>
#include <stdlib.h>
#include <stddef.h>

/* Which decisions `margin` of sdfo_render_forward covers: 0 (default) every branch of the pixel's ray -- slab tests,
 * hit tests, the march's loop bound; 1 the HIT TESTS alone (|dist - threshold * t| of every sample): the decisions
 * that change a pixel's depth.  (A ray that grazes the cube's silhouette or takes one sample more before it leaves
 * the cube flips the other decisions without changing anything it returns.) */
static int g_margin_hit_tests_only = 0;
void sdfo_set_margin_mode(int hit_tests_only) { g_margin_hit_tests_only = hit_tests_only != 0; }

/* Per-thread d/dSDF slabs of the renderer's backward ("per-thread grad_sdf slabs + tree reduce"; until round 6
 * every contribution was an `omp atomic` on one shared volume, and the port stopped scaling at 64 threads).  A slab is a
 * thread's own R^3 doubles, kept between calls and left all-zero by the reduction; each thread notes the range of
 * voxels it touched, so a single view (C1 / C2) reduces over thousands of words, not over threads x R^3. */
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_max_threads(void) { return 1; }
static int omp_get_thread_num(void) { return 0; }
#endif
#include <stdio.h>
#include <time.h>
static int g_sdfo_timing = 0;   /* sdfo_set_timing(1): the backward prints its phases on stderr (bench experiments) */
void sdfo_set_timing(int on) { g_sdfo_timing = on; }
static double sdfo_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
typedef struct { double* v; size_t cap, lo, hi; } sdfo_slab;
static sdfo_slab* g_slabs = NULL;
static int g_nslabs = 0;
static sdfo_slab* sdfo_slabs(int threads) {   /* (called outside parallel regions) */
  if (threads > g_nslabs) {
    g_slabs = (sdfo_slab*)realloc(g_slabs, (size_t)threads * sizeof(sdfo_slab));
    for (int t = g_nslabs; t < threads; ++t) { g_slabs[t].v = NULL; g_slabs[t].cap = 0; }
    g_nslabs = threads;
  }
  return g_slabs;
}
static double* sdfo_slab_of(sdfo_slab* s, size_t nvox) {   /* (by the owning thread: first touch places the pages) */
  if (s->cap < nvox) {
    free(s->v);
    s->v = (double*)calloc(nvox, sizeof(double));
    s->cap = nvox;
  }
  return s->v;
}

#define REAL float
#define SUFFIX _f32
#define SQRT sqrtf
#define FABS fabsf
#define FLOOR floorf
#include "sdfr_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef SQRT
#undef FABS
#undef FLOOR

#define REAL double
#define SUFFIX _f64
#define SQRT sqrt
#define FABS fabs
#define FLOOR floor
#include "sdfr_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef SQRT
#undef FABS
#undef FLOOR

int sdfo_version(void) { return 1; }

#ifdef _OPENMP
void sdfo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int sdfo_max_threads(void) { return omp_get_max_threads(); }
#else
void sdfo_set_threads(int n) { (void)n; }
int sdfo_max_threads(void) { return 1; }
#endif
