// common.hpp -- shared host/device helpers of libsdfr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "sdfr.h"

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
// scalar cache by every workgroup of that view (128 bytes = one s_load_dwordx16 x2).
struct alignas(128) ViewSetup {
  float rot[9];   // R(q), row-major: object -> camera
  float e[3];     // R^T p  (object-frame position of the camera origin is -e)
  float og[3];    // ray origin in grid coordinates: (-e * inv_scale + 1) * h
  float p[3];     // object position
  float q[4];     // quaternion (x, y, z, w)
  float scale;    // 1 / inv_scale
  float isc;      // inv_scale
  int rect[4];    // conservative screen bounds of the OBB: x0, y0, x1, y1 (pixels, half-open)
  float pad[4];
};
static_assert(sizeof(ViewSetup) == 128, "ViewSetup must stay 128 bytes");

// A workgroup (4 waves) owns a 64 x 32 pixel macro-tile and walks its eight 32 x 8 sub-tiles;
// inside a sub-tile each wave is an 8 x 8 pixel patch.
constexpr int kTileW = 64;
constexpr int kTileH = 32;
constexpr int kSubW = 32;
constexpr int kSubH = 8;
constexpr int kSubs = (kTileW / kSubW) * (kTileH / kSubH);  // 8

// packed cell records are used for a grid shared by >= kPackedMinViews views, up to kPackedMaxR
#ifndef SDFR_PACKED_MIN_VIEWS
#define SDFR_PACKED_MIN_VIEWS 4
#endif
constexpr int kPackedMinViews = SDFR_PACKED_MIN_VIEWS;
constexpr int kPackedMaxR = 128;

inline int tiles_x(int W) { return (W + kTileW - 1) / kTileW; }
inline int tiles_y(int H) { return (H + kTileH - 1) / kTileH; }

}  // namespace sdfr
