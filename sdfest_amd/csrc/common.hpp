// common.hpp -- shared host/device helpers of libsdfr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "sdfr.h"
#include "tuning.hpp"

namespace sdfr {

// ---- host: error reporting ------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define SDFR_HIP_TRY(expr)                                  \
  do {                                                      \
    hipError_t e__ = (expr);                                \
    if (e__ != hipSuccess) return ::sdfr::hip_fail(e__, #expr); \
  } while (0)

// ---- device: small vector algebra -----------------------------------------------------------
struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Per-view record written once by the set-up kernel and then read through the
// scalar cache by every workgroup of that view.
struct alignas(128) ViewSetup {
  float rot[9];   // R(q), row-major: object -> camera
  float e[3];     // R^T p  (object-frame position of the camera origin is -e)
  float og[3];    // ray origin in grid coordinates: (-e * inv_scale + 1) * h
  float p[3];     // object position
  float q[4];     // quaternion (x, y, z, w)
  float scale;    // 1 / inv_scale
  float isc;      // inv_scale
  int rect[4];    // conservative screen bounds of the OBB: x0, y0, x1, y1 (pixels, half-open)
  float ep[3];    // e + scale  (slab planes, object frame)
  float em[3];    // e - scale
  float dgk;      // inv_scale * (R-1)/2: object-frame direction -> grid-space direction
  float tp[3];    // e + hi, e + lo of the MAY-HIT box (object frame): the part of the cube outside of
  float tm[3];    //   which no sample can pass the hit test (render.hip, plane minima); = ep, em without it
  int bwd_big;    // batch backward: 1 = this view's tiles are 32 x 32 pixels, 0 = 64 x 8 (backward_big_tiles)
  int spans;      // 1 = the view's band spans are valid (render.hip, band spans); 0 = cull with `rect` only
  float pad[21];
};
static_assert(sizeof(ViewSetup) == 256, "ViewSetup must stay 256 bytes");

// A workgroup owns a tile of SX x SY sub-tiles of 32 x 8 pixels; a sub-tile is four wave-sized patches
// (8 x 8 pixels in the forward, 16 x 4 in the backward).  Two geometries are compiled: the 64 x 8 macro
// tile (2 x 1 sub-tiles; forward: 2 waves walking 4 patches each, backward: 4 waves, 2 patches each) for
// batches, and the single 32 x 8 sub-tile (4 waves, one patch each) when a call has too few macro tiles
// to fill 256 CUs (single-view calls of the drop-in autograd path).
constexpr int kSubW = 32;
constexpr int kSubH = 8;
struct TileGeom {
  int sx, sy;  // sub-tiles per tile
  __host__ __device__ constexpr int w() const { return sx * kSubW; }
  __host__ __device__ constexpr int h() const { return sy * kSubH; }
  __host__ __device__ constexpr int nx(int W) const { return (W + w() - 1) / w(); }
  __host__ __device__ constexpr int ny(int H) const { return (H + h() - 1) / h(); }
};
constexpr TileGeom kMacroTile{SDFR_MACRO_SX, SDFR_MACRO_SY};
constexpr TileGeom kSmallTile{1, 1};
// Both image kernels take the macro tile (64 x 8 pixels) once a call has >= 16384 of them (B >= ~55
// at 640x480) and the 32 x 8 sub-tile below that: the backward gains more LDS pre-summation per
// global atomic (B=256: 421 vs 265 us with small tiles), the forward fewer workgroups to dispatch;
// single views need the finer grid to fill 256 CUs (B=1: 14 vs 46 us).
constexpr long long kBackwardMacroMinTiles = SDFR_BWD_MACRO_MIN;
// the forward's batch tile (timing experiments: -DSDFR_FWD_SX / -DSDFR_FWD_SY; the backward keeps kMacroTile)
constexpr TileGeom kFwdMacroTile{SDFR_FWD_SX, SDFR_FWD_SY};
// Wide images take 128 x 8-pixel tiles walked by 4 waves -- the same work per wave as 64 x 8 by 2, half as many
// workgroups.  What a mostly-culled grid costs is the dispatcher's rate (~0.42 ns per workgroup whatever it does,
// DESIGN.md section 8): forward us per 256 views of 640x480, 64 x 8 by 2 waves / 128 x 8 by 4: the benchmark 164.6 / 165.3,
// objects a third of that size 81.8 / 71.3, B = 32 ... 56 5-10 % faster; but 160x120 images (two 128-pixel tiles per
// row) 106.8 / 165.5, B = 512 300.7 / 311.0, and in the step harness the benchmark itself 160.6 / 162.7 -- hence the
// two bounds (the benchmark's 153 600 tiles stay on 64 x 8).
constexpr TileGeom kFwdWideTile{4, 1};
constexpr int kFwdWideMinWidth = 256;
constexpr long long kFwdWideMaxTiles = 100000;   // of the 64 x 8 tiling (B <= 166 at 640x480)
inline TileGeom forward_geom(int B, int W, int H) {
  const long long tiles = (long long)B * kMacroTile.nx(W) * kMacroTile.ny(H);
  if (tiles < SDFR_FWD_MACRO_MIN) return kSmallTile;
  if (SDFR_FWD_WIDE && W >= kFwdWideMinWidth && tiles <= kFwdWideMaxTiles) return kFwdWideTile;
  return kFwdMacroTile;
}
inline TileGeom backward_geom(int B, int W, int H) {
  return ((long long)B * kMacroTile.nx(W) * kMacroTile.ny(H) >= kBackwardMacroMinTiles) ? kMacroTile
                                                                                         : kSmallTile;
}

// points per workgroup of the sampler's kernels = points per partial record of its backward (sampler.hip; the
// deferred gradient chain in loop.hip reads those records)
constexpr int kSamplerPts = 256;
constexpr int kDeferredMaxViews = 64;  // sdfr_views_to_pose_grad_deferred: views per call
// floats of a forward tile's loss record: a (sum, count) pair per wave, four waves (render.hip, forward_tile LOSS)
constexpr int kLossRec = 8;
constexpr int kSamplerGridTarget = SDFR_PC_GRID_TARGET;
constexpr int kSamplerMinGroups = SDFR_PC_MIN_GROUPS;

// The batch backward picks its tile shape PER VIEW.  What a tile costs is its flush -- one global float atomic per
// voxel its hit pixels touched -- and what it saves is the pre-summation of the pixels that share those voxels, so
// the tile should be as large and as square as the LDS table (512 z-runs) allows.  Measured on 256 views of
// 640x480 (backward us) against the pixels one voxel spans on the screen, r = f * voxel size / distance:
//   r ~ 4.6: 64x8 238, 32x32 178 | 3.2 (C3): 156 / 126 | 2.3: 134 / 137 | 1.6: 117 / 150 | 1.1: 104 / 166
// (32x24: 185 / 131 / 121 / 126 / 143, 64x16: 182 / 134 / 135 / 138 / 152, 32x16: 222 / 140 / 123 / 115 / 114):
// 32 x 32 tiles from r = 2.0 (thresholds 1.7 / 2.0 / 2.2 / 2.4 / 2.8 measured), 64 x 8 below -- where a 32 x 32 tile spans more cells than the table holds and the
// overflow goes to global atomics pixel by pixel.
constexpr TileGeom kBwdBigTile{1, 4};
// One launch serves both tilings without an integer division (a culled workgroup's life is its instruction
// chain): the grid is the 64 x 8 tiling's, (nx, rows) workgroups per view, and workgroup (bx, by) of a view with
// 32 x 32 tiles takes tile (2 bx + (by & 1), by >> 1) -- or leaves.  Its record of partial sums is by * nx + bx.
__host__ __device__ constexpr int backward_batch_rows(int H) {   // rows of partial RECORDS per view
  return kMacroTile.ny(H) > 2 * kBwdBigTile.ny(H) ? kMacroTile.ny(H) : 2 * kBwdBigTile.ny(H);
}
// Rows of WORKGROUPS per view.  A view with 32 x 32 tiles uses 2 * ny(32 x 32) rows of the launch and a view with
// 64 x 8 tiles ny(64 x 8) = twice as many.  SDFR_BWD_HALF_GRID (a caller's hint, include/sdfr.h): the grid has only
// the former, and a workgroup of a 64 x 8 view takes the tiles (bx, 2 by) and (bx, 2 by + 1) one after the other --
// half the workgroups to dispatch when the views are close (all of the benchmark's: backward 125 -> 112 us), but
// a 64 x 8 view's hit tiles then run in pairs (objects of ~1 pixel per voxel: 103 -> 174 us).
__host__ __device__ constexpr int backward_half_rows(int H) {
  return 2 * kBwdBigTile.ny(H) > (kMacroTile.ny(H) + 1) / 2 ? 2 * kBwdBigTile.ny(H) : (kMacroTile.ny(H) + 1) / 2;
}
__host__ __device__ constexpr int backward_tile_stride(int W, int H) {   // records per view
  return kMacroTile.nx(W) * backward_batch_rows(H);
}
__host__ __device__ constexpr int backward_big_record(int tx, int ty, int W) {
  return (2 * ty + (tx & 1)) * kMacroTile.nx(W) + (tx >> 1);
}

// packed cell records are used for a grid shared by >= kPackedMinViews views, up to kPackedMaxR
constexpr int kPackedMinViews = SDFR_PACKED_MIN_VIEWS;
constexpr int kPackedMaxR = 128;

// Every renderer workspace starts the same way: [B view records][sync region][the call's own scratch].  The sync
// region -- a 128-byte header (word 0: the prologue's epoch, word 1: how many view set-ups fell back to the full
// cube because the plane minima did not arrive in time, render.hip) and 6 x kPackedMaxR tagged plane-minimum
// entries of 16 bytes -- has ONE size and ONE place in the forward, backward, step and loss layouts, so no call
// that shares a workspace with another ever writes over it.
// Behind the sync region: the BAND SPANS, one word per view and band of 8 image rows = the columns [x0, x1) in which
// the view's may-hit box can be seen in those rows (x0 | x1 << 16), from the outline of the projected box rather than
// its bounding rectangle (a third of the rectangle's 8 x 8 patches lies outside the outline).  Written by the
// forward's one-wave-per-view set-ups; views set up by a single thread (small calls, the stand-alone backward) carry
// ViewSetup::spans = 0 and are culled with the rectangle alone.
__host__ __device__ constexpr int span_bands(int H) { return (H + 7) >> 3; }
__host__ __device__ constexpr int span_stride_words(int H) { return (span_bands(H) + 31) & ~31; }   // 128-byte rows
constexpr int kSyncHeaderWords = 32;
constexpr size_t kSyncBytes = (size_t)kSyncHeaderWords * 4 + (size_t)6 * kPackedMaxR * 16;
static_assert(kSyncBytes % 128 == 0, "the scratch behind the sync region stays 128-byte aligned");
inline size_t spans_offset_bytes(int B) { return (size_t)(B > 0 ? B : 0) * 256 + kSyncBytes; }
inline size_t scratch_offset_bytes(int B, int H) {
  return spans_offset_bytes(B) + (size_t)(B > 0 ? B : 0) * (H > 0 ? span_stride_words(H) : 0) * 4;
}


// Device-side fill / copy as ordinary kernels.  The entry points are captured into hipGraphs
// (FusedRenderAndCompare); with hipMemsetAsync / hipMemcpyAsync nodes in the captured sequence,
// replays of the loop after the first run produced results that depended on the process's memory
// layout (the eager sequence never did).  Kernel nodes only: the replayed graph is then exactly the
// launch sequence.  n_floats is in 4-byte words.
static __global__ void zero_words_kernel(float* __restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.0f;
}
static __global__ void copy_words_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
inline void zero_words_async(float* p, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
}
inline void copy_words_async(float* dst, const float* src, size_t n, hipStream_t st) {
  if (n) hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n);
}

}  // namespace sdfr
