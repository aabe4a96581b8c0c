// render.hip -- sphere-tracing depth render of an SDF grid, forward + analytic backward,
// hand-written for MI355X (gfx950, wave64).
//
// What it computes is fixed by the reference (one image per call there, a batch of B here):
//   forward   sdfest/differentiable_renderer/csrc/sdf_renderer_cuda.cu:241-298
//   backward  sdfest/differentiable_renderer/csrc/sdf_renderer_cuda.cu:300-468
// How it computes it is not a translation:
//   * a set-up kernel turns each pose into a 256-byte record (rotation, object-frame camera
//     position, grid-space ray origin, conservative screen rectangle of the bounding cube);
//     the image kernels read it through the scalar cache instead of re-deriving it per pixel;
//   * a workgroup owns a 64x8-pixel tile (2 sub-tiles of 32x8, a wave = an 8x8 patch; 32x8 for
//     small batches): tiles outside the rectangle only stream float4 zeros (forward) or exit
//     without touching memory (backward) -- in realistic scenes that is most of the frame.
//     (A persistent grid walking the tile list with a stride was measured: 40 us less dispatch,
//     but 100+ us more tail -- the hardware dispatcher balances the very uneven tiles better.)
//   * the slab test and the march run in the object frame / in grid coordinates, so one step
//     is 3 FMAs + floor/clamp instead of the reference's scale-normalise-index-denormalise chain;
//   * the 64^3 grid (1 MiB) cannot live in the 160 KiB LDS and a per-corner gather costs ~28
//     L1 tag look-ups per load instruction (measured: the L1, not HBM or the VALU, bounded v1),
//     so a pre-pass re-packs the grid into 16-byte face records (4 corners side by side): one
//     march step is two 16-byte loads;
//   * the 8 pose-gradient sums stay in registers across a macro-tile, then wave-shuffle -> LDS ->
//     one 32-byte partial per macro-tile -> a fixed-order reduction kernel (bitwise
//     reproducible); the reference issues 8 same-address float atomics per hit pixel
//     (sdf_renderer_cuda.cu:459-466);
//   * d/dsdf contributions of a tile are pre-summed in an LDS hash of short z-runs, in 64-bit
//     fixed point (device.hpp, RunHash), and flushed as runs of global float atomics (one per
//     touched voxel per tile);
//   * the depth term of the render-and-compare loss can ride inside both kernels (LOSS): no loss
//     kernel, no gradient image.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.hpp"
#include "device.hpp"
#include "sampler_device.hpp"


namespace sdfr {
namespace {

constexpr int kBlock = 256;
constexpr int kFwdWaves = SDFR_FWD_WAVES;  // waves per workgroup of the batch forward (macro tiles)
constexpr int kInlineSetupMaxViews = SDFR_INLINE_MAX_VIEWS;    // a step's forward without a prologue launch (render_forward_kernel, INLINE)


// ---------------------------------------------------------------------------------------------
// set-up: one thread per view
// ---------------------------------------------------------------------------------------------
// plane_min (nullable): 3 x R minima of the grid over its x-, y- and z-planes (plane_min_block).
// With it the view's MAY-HIT box is formed: a sample can pass the hit test dist < threshold * t
// (cu:289) only if its trilinear value is below vmax = threshold * t_far / scale, a trilinear value
// is a convex combination of its cell's corners, and the cells of slab i of an axis have their
// corners in planes i and i+1 -- so outside the slabs with min(plane i, plane i+1) < vmax every ray
// is a certain miss.  Rays are then culled against that box (tiles against its screen rectangle)
// and marched only up to its far side; rays that do cross it march from the FULL cube's near plane
// exactly as before (the trajectory of a hit is the reference's).  Measured on the benchmark scene:
// the box keeps 57 % of the in-cube rays, 81 % of the march steps and 60 % of the rectangle area
// (tools/analysis/aabb_pruning.py).
// pose part of a view's record: everything that does not depend on the grid
__device__ __forceinline__ void setup_pose(int b, const float* __restrict__ pos, const float* __restrict__ quat,
                                           const float* __restrict__ inv_scale, int R, float fx, float fy,
                                           ViewSetup& s) {
  const float x = quat[4 * b], y = quat[4 * b + 1], z = quat[4 * b + 2], w = quat[4 * b + 3];
  const V3 p = mk(pos[3 * b], pos[3 * b + 1], pos[3 * b + 2]);
  const float isc = inv_scale[b];
  const float scale = 1.0f / isc;
  // rotation matrix in the form the reference evaluates (sdf_renderer_cuda.cu:112-121)
  s.rot[0] = 1 - 2 * (y * y + z * z); s.rot[1] = 2 * (x * y - w * z);     s.rot[2] = 2 * (x * z + w * y);
  s.rot[3] = 2 * (x * y + w * z);     s.rot[4] = 1 - 2 * (x * x + z * z); s.rot[5] = 2 * (y * z - w * x);
  s.rot[6] = 2 * (x * z - w * y);     s.rot[7] = 2 * (y * z + w * x);     s.rot[8] = 1 - 2 * (x * x + y * y);
  // e = R^T p
  s.e[0] = s.rot[0] * p.x + s.rot[3] * p.y + s.rot[6] * p.z;
  s.e[1] = s.rot[1] * p.x + s.rot[4] * p.y + s.rot[7] * p.z;
  s.e[2] = s.rot[2] * p.x + s.rot[5] * p.y + s.rot[8] * p.z;
  const float h = 0.5f * (float)(R - 1);
  for (int k = 0; k < 3; ++k) s.og[k] = (-s.e[k] * isc + 1.0f) * h;
  s.p[0] = p.x; s.p[1] = p.y; s.p[2] = p.z;
  s.q[0] = x; s.q[1] = y; s.q[2] = z; s.q[3] = w;
  s.scale = scale;
  s.isc = isc;
  s.dgk = isc * h;
  {  // pixels one voxel spans on the screen at the object's distance (common.hpp, kBwdBigTile)
    const float dist = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
    const float r = sqrtf(fabsf(fx * fy)) * (scale / h) / fmaxf(dist, 1e-20f);
    s.bwd_big = (r >= SDFR_BWD_BIG_MIN_RATIO) ? 1 : 0;   // (NaN -> 0)
  }
  s.spans = 0;
  for (int k = 0; k < 21; ++k) s.pad[k] = 0.0f;
}

// grid part: the may-hit box from the plane minima (nullable), its screen rectangle, the slab planes.
// WAVE: the 64 lanes of a wave share the work of one view (all lanes end up with the same record).
// spans (WAVE only, nullable): this view's row of the band-span array (common.hpp): lane k writes band k.
template <bool WAVE>
__device__ __forceinline__ void setup_box(ViewSetup& s, int R, int W, int H, float cx, float cy, float fx,
                                          float fy, const float* __restrict__ plane_min, float threshold,
                                          unsigned* __restrict__ spans = nullptr) {
  const V3 p = mk(s.p[0], s.p[1], s.p[2]);
  const float scale = s.scale, isc = s.isc;
  const float h = 0.5f * (float)(R - 1);
  const int lane = threadIdx.x & 63;
  // may-hit box in object coordinates, lo[a] .. hi[a] (the whole cube without plane minima)
  float lo[3] = {-scale, -scale, -scale}, hi[3] = {scale, scale, scale};
  bool empty = false;
  if (plane_min && scale > 0.0f && scale < 1e30f) {
    const float plen = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
    const float vhit = fmaxf(threshold, 0.0f) * (plen + 1.7321f * scale) * isc * 1.0001f;  // >= thr * t / scale
    if (vhit == vhit && vhit < 1e30f) {
      const float cell = scale / h;  // one grid cell in object units
      for (int a = 0; a < 3; ++a) {
        const float* pm = plane_min + a * R;
        int first = R, last = -1;
        if (WAVE) {
          // the whole wave scans the R-1 slabs of the axis: 64 per round, first / last set bit of the ballot
          // (a serial scan by one thread is 3(R-1) dependent loads: 17 us per launch at R = 64, measured)
          for (int i0 = 0; i0 + 1 < R; i0 += 64) {
            const int i = i0 + lane;
            const bool f = (i + 1 < R) && (fminf(pm[i], pm[i + 1]) < vhit);
            const unsigned long long m = __ballot(f);
            if (m) {
              if (first == R) first = i0 + __builtin_ctzll(m);
              last = i0 + 63 - __builtin_clzll(m);
            }
          }
        } else {
          for (int i = 0; i + 1 < R; ++i) {
            if (fminf(pm[i], pm[i + 1]) < vhit) {
              if (first == R) first = i;
              last = i;
            }
          }
        }
        if (last < 0) { empty = true; break; }
        // slab i spans grid coordinates [i, i+1]; 1/16 cell of slack for the rounding of the slab
        // arithmetic and of the march's own floor()
        lo[a] = fmaxf(-scale, ((float)first - 0.0625f) * cell - scale);
        hi[a] = fminf(scale, ((float)(last + 1) + 0.0625f) * cell - scale);
      }
    }
  }
  // Conservative screen rectangle of the box.  A ray can only pass the slab test if its pixel
  // centre lies inside the projection of the box, which (box entirely in front of the camera)
  // lies inside the bounding rectangle of the 8 projected corners.
  float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
  bool in_front = true;
  auto corner = [&](int c) {
    const float sx = (c & 1) ? hi[0] : lo[0], sy = (c & 2) ? hi[1] : lo[1],
                sz = (c & 4) ? hi[2] : lo[2];
    const float X = p.x + s.rot[0] * sx + s.rot[1] * sy + s.rot[2] * sz;
    const float Y = p.y + s.rot[3] * sx + s.rot[4] * sy + s.rot[5] * sz;
    const float Z = p.z + s.rot[6] * sx + s.rot[7] * sy + s.rot[8] * sz;
    if (!(Z < -1e-6f)) in_front = false;
    const float iz = 1.0f / fmaxf(-Z, 1e-30f);
    const float u = cx + fx * X * iz;
    const float v = cy - fy * Y * iz;
    umin = fminf(umin, u); umax = fmaxf(umax, u);
    vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
  };
  if (WAVE) {
    // corner lane & 7 per lane, then the 8-lane groups fold (min / max are exact: the same values in any order)
    corner(lane & 7);
#pragma unroll
    for (int off = 1; off <= 4; off <<= 1) {
      umin = fminf(umin, __shfl_xor(umin, off, 64)); umax = fmaxf(umax, __shfl_xor(umax, off, 64));
      vmin = fminf(vmin, __shfl_xor(vmin, off, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    }
    in_front = __ballot(!in_front) == 0ull;
  } else {
    for (int c = 0; c < 8; ++c) corner(c);
  }
  int x0 = 0, y0 = 0, x1 = W, y1 = H;
  if (in_front) {
    const float mx = 2.0f + 1e-5f * fabsf(fx), my = 2.0f + 1e-5f * fabsf(fy);
    const float fx0 = fminf(fmaxf(floorf(umin - 0.5f - mx), 0.0f), (float)W);
    const float fx1 = fminf(fmaxf(ceilf(umax - 0.5f + mx) + 1.0f, 0.0f), (float)W);
    const float fy0 = fminf(fmaxf(floorf(vmin - 0.5f - my), 0.0f), (float)H);
    const float fy1 = fminf(fmaxf(ceilf(vmax - 0.5f + my) + 1.0f, 0.0f), (float)H);
    x0 = (int)fx0; x1 = (int)fx1; y0 = (int)fy0; y1 = (int)fy1;
  }
  if (empty) { x0 = y0 = x1 = y1 = 0; }  // no cell of the grid can be hit from this pose
  s.rect[0] = x0; s.rect[1] = y0; s.rect[2] = x1; s.rect[3] = y1;
  if (WAVE && spans) {
    // Band spans: the box projects onto the convex hull of its 8 projected corners, whose outline consists of
    // projected box EDGES.  The hull's column extent over a band of rows is therefore the extent of (the 12 edges
    // cut by the band's two boundary lines) + (the corners inside the band); same 2-pixel margins as the rectangle.
    // Only when every corner projects to moderate coordinates (the cuts are then good to ~0.03 pixels in fp32).
    float cu[8], cv[8];
    float big = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float sx = (c & 1) ? hi[0] : lo[0], sy = (c & 2) ? hi[1] : lo[1], sz = (c & 4) ? hi[2] : lo[2];
      const float X = p.x + s.rot[0] * sx + s.rot[1] * sy + s.rot[2] * sz;
      const float Y = p.y + s.rot[3] * sx + s.rot[4] * sy + s.rot[5] * sz;
      const float Z = p.z + s.rot[6] * sx + s.rot[7] * sy + s.rot[8] * sz;
      const float iz = 1.0f / fmaxf(-Z, 1e-30f);
      cu[c] = cx + fx * X * iz;
      cv[c] = cy - fy * Y * iz;
      big = fmaxf(big, fmaxf(fabsf(cu[c]), fabsf(cv[c])));
    }
    bool sane = in_front && !empty && big < 4.0e4f && W <= 65535;
#pragma unroll
    for (int c = 0; c < 8; ++c) sane = sane && (cu[c] == cu[c]) && (cv[c] == cv[c]);   // (fmaxf drops a NaN)
    s.spans = sane ? 1 : 0;
    if (sane) {
      const float mx = 2.0f + 1e-5f * fabsf(fx), my = 2.0f + 1e-5f * fabsf(fy);
      const int nb = span_bands(H);
      // per edge, once: du/dv (an edge along the rows: +-inf or NaN, harmless below)
      float slope[12];
#pragma unroll
      for (int a = 0, e = 0; a < 8; ++a)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
          if (!(a & (1 << ax))) {
            const int c2 = a | (1 << ax);
            slope[e++] = (cu[c2] - cu[a]) * __builtin_amdgcn_rcpf(cv[c2] - cv[a]);
          }
      for (int k0 = 0; k0 < nb; k0 += 64) {
        const int k = k0 + lane;
        const float ya = (float)(8 * k) + 0.5f - my, yb = (float)(8 * k) + 7.5f + my;
        float xmin = 3.0e38f, xmax = -3.0e38f;
        float da[8], db[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          da[a] = ya - cv[a];
          db[a] = yb - cv[a];
          const float u = (da[a] <= 0.0f && db[a] >= 0.0f) ? cu[a] : __int_as_float(0x7fc00000);   // corner in the band
          xmin = fminf(xmin, u); xmax = fmaxf(xmax, u);
        }
#pragma unroll
        for (int a = 0, e = 0; a < 8; ++a)
#pragma unroll
          for (int ax = 0; ax < 3; ++ax)
            if (!(a & (1 << ax))) {
              const int c2 = a | (1 << ax);   // box edge a -- c2 cut by the band's two boundary rows
              // (a cut exists where the corners lie on different sides; x = u_a + (row - v_a) du/dv; an edge lying
              // in the row gives NaN, which fminf / fmaxf drop -- its corners are in the band anyway)
              const float xa = (da[a] * da[c2] <= 0.0f) ? fmaf(da[a], slope[e], cu[a]) : __int_as_float(0x7fc00000);
              const float xb = (db[a] * db[c2] <= 0.0f) ? fmaf(db[a], slope[e], cu[a]) : __int_as_float(0x7fc00000);
              xmin = fminf(xmin, fminf(xa, xb)); xmax = fmaxf(xmax, fmaxf(xa, xb));
              ++e;
            }
        int sx0 = 0, sx1 = 0;
        if (xmin <= xmax && 8 * k < y1 && 8 * k + 8 > y0) {
          sx0 = max(x0, (int)fminf(fmaxf(floorf(xmin - 0.5f - mx), 0.0f), (float)W));
          sx1 = min(x1, (int)fminf(fmaxf(ceilf(xmax - 0.5f + mx) + 1.0f, 0.0f), (float)W));
          if (sx1 <= sx0) sx0 = sx1 = 0;
        }
        if (k < nb) spans[k] = (unsigned)sx0 | ((unsigned)sx1 << 16);
      }
    }
  }
  for (int k = 0; k < 3; ++k) {
    s.ep[k] = s.e[k] + scale;
    s.em[k] = s.e[k] - scale;
    s.tp[k] = s.e[k] + hi[k];
    s.tm[k] = s.e[k] + lo[k];
  }
}

template <bool WAVE = false>
__device__ __forceinline__ void compute_view_setup(int b, const float* __restrict__ pos,
                                                   const float* __restrict__ quat,
                                                   const float* __restrict__ inv_scale, int R, int W, int H,
                                                   float cx, float cy, float fx, float fy,
                                                   ViewSetup* __restrict__ out,
                                                   const float* __restrict__ plane_min = nullptr,
                                                   float threshold = 0.0f, unsigned* __restrict__ spans = nullptr) {
  ViewSetup s;
  setup_pose(b, pos, quat, inv_scale, R, fx, fy, s);
  setup_box<WAVE>(s, R, W, H, cx, cy, fx, fy, plane_min, threshold,
                  spans ? spans + (size_t)b * span_stride_words(H) : nullptr);
  if (!WAVE || (threadIdx.x & 63) == 0) out[b] = s;
}

// How many of a forward's views are CLOSE (ViewSetup::bwd_big: the views the batch backward gives 32 x 32 tiles) --
// the fact behind the SDFR_BWD_HALF_GRID hint, counted on the device so that no caller has to look at poses that
// live in device memory.  Called by one lane per view -- of the view's first workgroup in the forward kernel, after
// its tile: the prologue launch is a few us long and a returning atomic per view at its end showed (9.9 -> 13.4 us);
// in the 150 us image kernel it is free.  Sync header word 2: (views counted << 16 | close
// views) of the running launch, ONE atomic per view -- the view that finds B - 1 before it is the last, knows the
// total from the value the atomic returned, and resets the word: no fence anywhere (a __threadfence() is an L2
// write-back on this part: two per view made the prologue 16 instead of 10 us) -- word 4: launches counted, word 5:
// close views of the last complete launch.  `word` (nullable, a kernel ARGUMENT, never an address kept in memory):
// the caller's host-visible 64-bit word, which receives (launches counted << 32 | close views) in ONE store when the
// last view has been counted; a host that reads it any time later gets some complete launch's count, never a torn one.
__device__ __forceinline__ void count_close_view(unsigned* __restrict__ sync, int B, bool close,
                                                 unsigned long long* __restrict__ word) {
  const unsigned old = atomicAdd(&sync[2], 0x10000u | (close ? 1u : 0u));   // B <= 65535 (check_common)
  if ((old >> 16) + 1u != (unsigned)B) return;
  const unsigned c = (old & 0xffffu) + (close ? 1u : 0u);
  atomicExch(&sync[2], 0u);
  const unsigned seq = sync[4] + 1u;
  sync[4] = seq;
  sync[5] = c;
  if (word) __hip_atomic_store(word, ((unsigned long long)seq << 32) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void view_setup_kernel(const float* __restrict__ pos, const float* __restrict__ quat,
                                  const float* __restrict__ inv_scale, int B, int R, int W, int H,
                                  float cx, float cy, float fx, float fy,
                                  ViewSetup* __restrict__ out, const float* __restrict__ plane_min,
                                  float threshold, unsigned* __restrict__ spans) {
  // one wave per view (the wave shares the scan of the plane minima, lane 0 writes the record)
  const int b = blockIdx.x;
  if (b < B)
    compute_view_setup<true>(b, pos, quat, inv_scale, R, W, H, cx, cy, fx, fy, out, plane_min, threshold, spans);
}

// The plane minima reach the set-up waves of the SAME launch (forward_prologue_kernel) as tagged entries: 16 bytes
// per plane, two 64-bit words {bits(min), tag} and {~bits(min), tag}, each written with one agent-scope atomic
// store.  A reader that sees the current tag in both words and complementary payloads holds the value (single-copy
// atomicity of the 64-bit words: no fence, no counter to reset); the tag is the workspace's epoch + 1, which the
// forward kernel advances after the launch, so entries of earlier calls -- or whatever else the caller kept in the
// workspace -- never read as ready (two matching tags and complementary payloads: 2^-96 for random bytes).
struct PlaneEntry {
  unsigned long long a, b;
};
__device__ __forceinline__ void publish_plane_min(float* __restrict__ plane_min, PlaneEntry* __restrict__ ent,
                                                  unsigned tag, int j, float m) {
  if (ent) {
    const unsigned bits = __float_as_uint(m);
    __hip_atomic_store(&ent[j].a, ((unsigned long long)tag << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ent[j].b, ((unsigned long long)tag << 32) | (unsigned)~bits, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  } else {
    plane_min[j] = m;
  }
}

// Minimum of the grid over one plane: block j = axis * R + i reduces the R^2 values with index i
// along `axis`.  (NaN values are ignored: a sample that sees one never passes the hit test.)
// z-planes (axis 2) are strided by R floats: a block takes FOUR of them with 16-byte loads (R % 4 == 0; block
// j = 2R + i does planes i .. i+3 when i % 4 == 0 and nothing otherwise), a quarter of the lines per plane.
__device__ __forceinline__ void plane_min_block(const float* __restrict__ sdf, int R, int j,
                                                float* __restrict__ plane_min, PlaneEntry* __restrict__ ent = nullptr,
                                                unsigned tag = 0u) {
  __shared__ float red[4][4];
  const int axis = j / R, i = j % R, RR = R * R;
  const int tid = threadIdx.x;
  if (axis == 2 && (R & 3) == 0 && ((uintptr_t)sdf & 15) == 0) {
    if (i & 3) return;
    float m0 = 3.0e38f, m1 = 3.0e38f, m2 = 3.0e38f, m3 = 3.0e38f;
    for (int k = tid; k < RR; k += 256) {
      const float4 v = *reinterpret_cast<const float4*>(sdf + (size_t)k * R + i);
      m0 = fminf(m0, v.x); m1 = fminf(m1, v.y); m2 = fminf(m2, v.z); m3 = fminf(m3, v.w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      m0 = fminf(m0, __shfl_xor(m0, off, 64)); m1 = fminf(m1, __shfl_xor(m1, off, 64));
      m2 = fminf(m2, __shfl_xor(m2, off, 64)); m3 = fminf(m3, __shfl_xor(m3, off, 64));
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = m0; red[tid >> 6][1] = m1; red[tid >> 6][2] = m2; red[tid >> 6][3] = m3; }
    __syncthreads();
    if (tid < 4)
      publish_plane_min(plane_min, ent, tag, j + tid,
                        fminf(fminf(red[0][tid], red[1][tid]), fminf(red[2][tid], red[3][tid])));
    return;
  }
  float m = 3.0e38f;
  for (int k = tid; k < RR; k += 256) {
    const int u = k / R, v = k % R;
    const int idx = axis == 0 ? i * RR + k : (axis == 1 ? u * RR + i * R + v : k * R + i);
    m = fminf(m, sdf[idx]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
  if ((tid & 63) == 0) red[tid >> 6][0] = m;
  __syncthreads();
  if (tid == 0) publish_plane_min(plane_min, ent, tag, j, fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0])));
}

// The backward's prologue in one launch: zero the gradient volume(s) and set the views up (every
// launch costs ~5 us of a ~30 us single-view call whatever it does).  grid: ceil(max(n_words, B) / 256)
__global__ __launch_bounds__(256) void backward_prologue_kernel(
    float* __restrict__ g_sdf, size_t n_words, const float* __restrict__ pos,
    const float* __restrict__ quat, const float* __restrict__ inv_scale, int B, int R, int W, int H,
    float cx, float cy, float fx, float fy, ViewSetup* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_words) g_sdf[i] = 0.0f;
  if (i < (size_t)B) compute_view_setup((int)i, pos, quat, inv_scale, R, W, H, cx, cy, fx, fy, out);
}

// ---------------------------------------------------------------------------------------------
// face records: rec[lin] = the 4 corners of the cell face x = const whose corner 00 is voxel lin:
// (v(x,y,z), v(x,y,z+1), v(x,y+1,z), v(x,y+1,z+1)) as one float4.  A cell is the two records
// the record of (x,y,z) and the record of (x+1,y,z): two 16-byte loads per march step instead of
// four 8-byte z-pair loads.  4 MiB at R = 64 (a full 32-byte cell record per voxel was measured
// too: 8 MiB, 2.5 % slower -- it overflows the 4 MiB per-XCD L2).  Storage order: device.hpp,
// record_index.
// ---------------------------------------------------------------------------------------------
// (A 2x2x2-blocked record order was measured: +10 integer ops per step, no gain -- the march is
// bound by dependent-load latency, not by lines per access.  Records stay in grid order.)
// The launch's last 3R blocks compute the plane minima for the may-hit boxes (compute_view_setup);
// the views are set up by the next launch, which needs them.
// Plane minima for the one-launch prologue (R % 4 == 0, 16-byte aligned grid): 3R blocks, every thread a few
// 16-byte loads that are all in flight together (the set-up waves of the launch wait for the slowest block).
//   block i      in [0, R):    x-plane i, 16 KiB contiguous
//   block R + i  in [R, 2R):   y-plane i, R rows of R floats
//   block 2R + k in [2R, 3R):  z-planes 4 zq .. 4 zq + 3 (zq = k % (R/4)) over a QUARTER of the (x, y) rows
//                              (part = k / (R/4)): the loads of a z-plane are strided by a grid row, so the work
//                              is spread over four blocks and the readers take the minimum of the four entries
// Entries: x -> i, y -> R + i, z -> 2R + part * R + z   (6R entries of 16 bytes).
__device__ __forceinline__ float4 min4(float4 a, float4 b) {
  return make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w));
}
__device__ __forceinline__ void plane_min_fast_block(const float* __restrict__ sdf, int R, int blk,
                                                     PlaneEntry* __restrict__ ent, unsigned tag) {
  __shared__ float4 red4[4];
  const int tid = threadIdx.x, RR = R * R, Rq = R >> 2;
  const float4* v = reinterpret_cast<const float4*>(sdf);
  float4 m = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f);
  const int axis = blk / R, i = blk - axis * R;
  if (axis == 0) {
    const float4* base = v + (size_t)i * (RR >> 2);
#pragma unroll 4
    for (int q = tid; q < (RR >> 2); q += 256) m = min4(m, base[q]);
  } else if (axis == 1) {
    const float4* base = v + (size_t)i * Rq;
#pragma unroll 4
    for (int q = tid; q < R * Rq; q += 256) m = min4(m, base[(size_t)(q / Rq) * (RR >> 2) + (q % Rq)]);
  } else {
    const int zq = i % Rq, part = i / Rq, rows = RR >> 2;
    const float4* base = v + (size_t)part * rows * Rq + zq;
#pragma unroll 4
    for (int r = tid; r < rows; r += 256) m = min4(m, base[(size_t)r * Rq]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    m.x = fminf(m.x, __shfl_xor(m.x, off, 64)); m.y = fminf(m.y, __shfl_xor(m.y, off, 64));
    m.z = fminf(m.z, __shfl_xor(m.z, off, 64)); m.w = fminf(m.w, __shfl_xor(m.w, off, 64));
  }
  if ((tid & 63) == 0) red4[tid >> 6] = m;
  __syncthreads();
  if (tid < 4) {
    const float4 a = min4(min4(red4[0], red4[1]), min4(red4[2], red4[3]));
    if (axis < 2) {
      if (tid == 0) publish_plane_min(nullptr, ent, tag, blk, fminf(fminf(a.x, a.y), fminf(a.z, a.w)));
    } else {
      const int zq = i % Rq, part = i / Rq;
      const float val = tid == 0 ? a.x : (tid == 1 ? a.y : (tid == 2 ? a.z : a.w));
      publish_plane_min(nullptr, ent, tag, 2 * R + part * R + 4 * zq + tid, val);
    }
  }
}

__device__ __forceinline__ void pack_cell(const float* __restrict__ sdf, int R, float4* __restrict__ cells, int lin) {
  const int RR = R * R;
  if (lin >= RR * R) return;
  const int z = lin % R, y = (lin / R) % R;
  if (y >= R - 1 || z >= R - 1) return;
  const float* p = sdf + lin;
  cells[record_index(lin / RR, y, z, (R + 1) >> 1)] = make_float4(p[0], p[1], p[R], p[R + 1]);
}

__global__ __launch_bounds__(256) void pack_cells_kernel(const float* __restrict__ sdf, int R,
                                                         float4* __restrict__ cells, int n_pack_blocks,
                                                         float* __restrict__ plane_min) {
  // the plane-minimum blocks come FIRST: they take longest (16 strided loads per thread, a reduction)
  const int n_plane = (int)gridDim.x - n_pack_blocks;
  if ((int)blockIdx.x < n_plane) {
    plane_min_block(sdf, R, (int)blockIdx.x, plane_min);
    return;
  }
  pack_cell(sdf, R, cells, ((int)blockIdx.x - n_plane) * (int)blockDim.x + (int)threadIdx.x);
}

// The forward's prologue in ONE launch (it was two: the view set-up needs the plane minima of the grid, and every
// launch of the step costs ~5 us whatever it does).  Blocks, in dispatch order:
//   [0, n_plane)            plane minima, published as tagged entries (PlaneEntry)
//   [n_plane, +n_setup)     4 views each, one wave per view: the block polls the 3R entries (agent-scope atomic
//                           loads) until all carry this launch's tag, then sets its views up from them
//   the rest                face records of the grid (pack_cell) and the zero fill of `g_zero` (the gradient
//                           volume of the step's backward, sdfr_render_step_forward)
// The set-up blocks can only wait for blocks dispatched before them, which never wait themselves; and the wait is
// bounded: after kPrologueMaxPolls rounds a block sets its views up WITHOUT plane minima (the whole cube as the
// may-hit box: slower, same depth), so no schedule can hang the launch.
// A view that was set up that way is counted in the sync header's word 1 (sdfr_render_prologue_fallbacks): the
// count stays 0 unless the launch was serialised (a profiler replaying block by block, a CU mask of one CU).
constexpr int kPrologueMaxPolls = 1 << 16;
// The bound of the wait below, in polling rounds, is a property of the WORKSPACE: sync header words 6 / 7 =
// {SDFR_SYNC_POLLS_MAGIC, rounds} override the default for the forwards on that workspace (0 rounds: every view set-up
// takes the fall-back path -- how the tests reach it); any other word 6, e.g. of a workspace nobody initialised, means
// the default.  Nothing process-wide.
__global__ __launch_bounds__(256) void forward_prologue_kernel(
    const float* __restrict__ sdf, int R, float4* __restrict__ cells, int n_plane, int n_setup,
    unsigned* __restrict__ sync, const float* __restrict__ pos, const float* __restrict__ quat,
    const float* __restrict__ inv_scale, int B, int W, int H, float cx, float cy, float fx, float fy,
    ViewSetup* __restrict__ out, float threshold, float* __restrict__ g_zero, size_t n_zero, int default_polls,
    unsigned* __restrict__ spans) {
  const unsigned tag = sync[0] + 1u;  // the epoch the last forward on this workspace left, + 1
  PlaneEntry* ent = reinterpret_cast<PlaneEntry*>(sync + kSyncHeaderWords);
  int blk = (int)blockIdx.x;
  if (blk < n_plane) {
    plane_min_fast_block(sdf, R, blk, ent, tag);
    return;
  }
  blk -= n_plane;
  if (blk < n_setup) {
    __shared__ float pm_s[6 * kPackedMaxR];
    const int tid = (int)threadIdx.x, n_ent = 6 * R;
    // the pose part of this wave's view first: its loads and arithmetic run while the plane blocks work
    const int b = blk * 4 + (tid >> 6);
    ViewSetup s;
    if (b < B) setup_pose(b, pos, quat, inv_scale, R, fx, fy, s);
    bool ready = false;
    const int max_polls = sync[6] == SDFR_SYNC_POLLS_MAGIC ? (int)sync[7] : default_polls;   // (workgroup-uniform)
    for (int poll = 0; poll < max_polls; ++poll) {
      bool ok = true;
      for (int j = tid; j < n_ent; j += 256) {
        const unsigned long long a = __hip_atomic_load(&ent[j].a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long c = __hip_atomic_load(&ent[j].b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool good = (unsigned)(a >> 32) == tag && (unsigned)(c >> 32) == tag && (unsigned)a == ~(unsigned)c;
        if (good) pm_s[j] = __uint_as_float((unsigned)a);
        ok = ok && good;
      }
      if (__syncthreads_and(ok)) { ready = true; break; }   // (also orders the pm_s writes before the reads below)
      __builtin_amdgcn_s_sleep(1);
    }
    if (ready) {  // z-planes: the minimum of the four partial entries (thread z reads and writes only column z)
      for (int z = tid; z < R; z += 256)
        pm_s[2 * R + z] = fminf(fminf(pm_s[2 * R + z], pm_s[3 * R + z]), fminf(pm_s[4 * R + z], pm_s[5 * R + z]));
      __syncthreads();
    }
    if (b < B) {
      setup_box<true>(s, R, W, H, cx, cy, fx, fy, ready ? pm_s : nullptr, threshold,
                      spans + (size_t)b * span_stride_words(H));
      if ((tid & 63) == 0) {
        out[b] = s;
        if (!ready) atomicAdd(&sync[1], 1u);
      }
    }
    return;
  }
  blk -= n_setup;
  const int lin = blk * 256 + (int)threadIdx.x;
  const size_t vox = (size_t)R * R * R;
  for (size_t i = (size_t)lin; i < n_zero; i += vox) g_zero[i] = 0.0f;   // (lin < vox or the block has no thread)
  pack_cell(sdf, R, cells, lin);
}

struct Rect {
  int x0, y0, x1, y1;
};
__device__ __forceinline__ bool overlaps(const Rect& r, int px, int py, int w, int h) {
  return (px < r.x1) && (px + w > r.x0) && (py < r.y1) && (py + h > r.y0);
}

// ---------------------------------------------------------------------------------------------
// forward.  grid = (macro-tiles x, macro-tiles y, views)
// ---------------------------------------------------------------------------------------------
// One ray per lane.  (K rays per lane -- K independent load chains to hide the gather latency --
// was built and measured in rounds 1 and 3: slower at every K, see DESIGN.md.  Where the march stands at the end of
// round 3 (slopes): its two 16-byte gathers per step come first -- one more costs +34 us per 256 views,
// 16 more VALU instructions +9 -- with the VALU at 70-85 % close behind; the loop below keeps both small: packed fp32
// in march_sample, and per-step bookkeeping of two compares and one add.)
//
// LOSS: the masked depth-L1 of simple_setup.py:129-135 is folded in: a hit pixel
// also reads the observed depth and the tile leaves (sum |est - obs|, count) over the overlap
// mask (obs > 0) & (est > 0) in `loss_part`; est = 0 off the object, so only hit pixels can be in
// the mask and culled tiles never touch the observed image.
// TIGHT: rays are also tested against the view's may-hit box (compute_view_setup) and march only up
// to its far side.
// One lane's ray after the slab tests: unit ray (cu:137-154) rotated into the object frame with d.z = -1 folded
// in, then the slab test in the object frame (the cube is axis-aligned there, its centre at +e from the ray origin,
// f_i = dobj_i): same accept/reject as cu:156-194.  A ray parallel to a slab (f = 0) needs no special case: 1/f =
// +-inf puts both plane distances at the same infinity when the origin is outside the slab (-> t_near > t_far or
// t_far < 0) and at opposite infinities when it is inside (-> the axis does not constrain the interval).  Pixels
// outside the screen rectangle fail this test by construction of the rectangle.
// TIGHT: the may-hit box lies inside the cube and its planes come from the same products (e + hi <= e + scale,
// rounding is monotonic), so a ray that crosses the box crosses the cube: of the cube only the near distance is
// needed -- the march starts there (cu:262-268) -- and the box decides hit-or-miss and the far end.
// (plain minimum / maximum instructions: fminf / fmaxf make the compiler quiet signalling NaNs with an extra
// v_max per operand -- a sixth of this block; a NaN operand is dropped here as it is there)
struct Ray {
  float dg[3];    // direction in grid coordinates per unit of t
  float t, tf;    // first sample (the FULL cube's near plane, cu:262-268) and the far end of the march
  float inv_len;  // -d.z of the unit ray: depth = t * inv_len
  bool go;        // the ray crosses the (may-hit) box in front of the camera
};
template <bool TIGHT>
__device__ __forceinline__ Ray ray_setup(const ViewSetup& s, int row, int col, bool inside, float cx, float cy,
                                         float rfx, float rfy) {
  Ray r;
  const float dx = ((float)col + 0.5f - cx) * rfx;
  const float dy = -((float)row + 0.5f - cy) * rfy;
  const float inv_len = __builtin_amdgcn_rsqf(fmaf(dx, dx, fmaf(dy, dy, 1.0f)));
  const float ux = fmaf(s.rot[0], dx, fmaf(s.rot[3], dy, -s.rot[6]));
  const float uy = fmaf(s.rot[1], dx, fmaf(s.rot[4], dy, -s.rot[7]));
  const float uz = fmaf(s.rot[2], dx, fmaf(s.rot[5], dy, -s.rot[8]));
  const float dv[3] = {ux * inv_len, uy * inv_len, uz * inv_len};
  float t_near = -1e-10f, tf = 1e10f, t_near2 = -1e-10f, tf2 = 1e10f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float inv = __builtin_amdgcn_rcpf(dv[a]);
    const float ta = s.ep[a] * inv, tb = s.em[a] * inv;
    t_near = vmax(t_near, vmin(ta, tb));
    if (TIGHT) {
      const float tc = s.tp[a] * inv, td = s.tm[a] * inv;
      t_near2 = vmax(t_near2, vmin(tc, td));
      tf2 = vmin(tf2, vmax(tc, td));
    } else {
      tf = vmin(tf, vmax(ta, tb));
    }
  }
  if (TIGHT) tf = tf2;   // no sample behind the may-hit box can be a hit
  const bool miss = !inside || (TIGHT ? ((t_near2 > tf2) || (tf2 < 0.0f)) : ((t_near > tf) || (tf < 0.0f)));
  r.t = vmax(t_near, 0.0f);
  r.tf = tf;
  r.inv_len = inv_len;
  r.go = !miss && (r.t < tf);
  const float kgrid = s.dgk;
  r.dg[0] = dv[0] * kgrid; r.dg[1] = dv[1] * kgrid; r.dg[2] = dv[2] * kgrid;
  return r;
}

template <int RT, bool PACKED, int SX, int SY, bool LOSS, bool TIGHT, int NW>
__device__ __forceinline__ void forward_tile(
    int tile_x, int tile_y, int ntx, int nty, int b, const float* __restrict__ src, int R,
    long long src_view_stride, const ViewSetup& s, int W, int H, float cx, float cy,
    float rfx, float rfy, float threshold, int vec_ok, float* __restrict__ depth,
    const float* __restrict__ target, float* __restrict__ loss_part, const unsigned* __restrict__ spans) {
  constexpr int kSubs = SX * SY, kTileW = SX * kSubW, kTileH = SY * kSubH;
  static_assert(kTileH == 8, "a forward tile is one band of the span array");
  constexpr int kThreads = NW * 64;  // NW waves walk the tile's 4 * kSubs 8x8 patches
  using PF = Patch<kPatchWFwd>;
  const int px0 = tile_x * kTileW, py0 = tile_y * kTileH;
  Rect rc{s.rect[0], s.rect[1], s.rect[2], s.rect[3]};
  const bool in_rect = overlaps(rc, px0, py0, kTileW, kTileH);
  // the band's column span of the projected may-hit box (one scalar load): tighter than the rectangle's columns
  if (s.spans) {
    const unsigned sp = spans[(size_t)b * span_stride_words(H) + tile_y];
    rc.x0 = (int)(sp & 0xffffu);
    rc.x1 = (int)(sp >> 16);
  }
  float* img = depth + (size_t)b * H * W;
  const int tid = threadIdx.x;

  if (!overlaps(rc, px0, py0, kTileW, kTileH)) {
    // nothing of the cube projects here: stream zeros.  (LOSS: the view's reduce sums the records of every tile of
    // the rectangle, also of those the band span leaves out)
    if (LOSS && in_rect && tid < kLossRec) loss_part[(((size_t)b * nty + tile_y) * ntx + tile_x) * kLossRec + tid] = 0.0f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (vec_ok) {
#pragma unroll
      for (int i = tid; i < kTileW * kTileH / 4; i += kThreads) {
        const int row = py0 + i / (kTileW / 4), col = px0 + (i % (kTileW / 4)) * 4;
        // whole 64-byte runs, written once and not re-read here: keep them out of L2 (measured
        // -2 %; on the scattered per-pixel result stores the same hint cost +28 % WRITE_SIZE)
        if (row < H && col < W) __builtin_nontemporal_store(zero4, reinterpret_cast<f32x4*>(img + (size_t)row * W + col));
      }
    } else {
      for (int i = tid; i < kTileW * kTileH; i += kThreads) {
        const int row = py0 + i / kTileW, col = px0 + i % kTileW;
        if (row < H && col < W) img[(size_t)row * W + col] = 0.0f;
      }
    }
    return;
  }

  const int wave = tid >> 6, lane = tid & 63;
  const float scale = s.scale;
  const float* vol = src + (size_t)b * src_view_stride;
  const int Rr = RT > 0 ? RT : R;
  // the record array / the grid as a buffer resource: 32-bit byte offsets, hardware range check
  const unsigned src_bytes = PACKED ? (unsigned)Rr * record_slab(Rr) * 16u : (unsigned)Rr * Rr * Rr * 4u;
  const __amdgpu_buffer_rsrc_t vsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vol), 0, (int)src_bytes, 0x00020000);

  float l_sum = 0.0f, l_cnt = 0.0f;  // LOSS: this lane's share of the tile's (sum, count)
  const float* obs = LOSS ? target + (size_t)b * H * W : nullptr;
  const float og[3] = {s.og[0], s.og[1], s.og[2]};
  for (int it = wave; it < 4 * kSubs; it += NW) {
    const int sub = it >> 2, pw = it & 3;  // sub-tile and 8x8 patch inside it (NW = 4: patch = wave)
    const int sx = px0 + (sub % SX) * kSubW, sy = py0 + (sub / SX) * kSubH;
    const int col = sx + PF::ox(pw) + PF::x(lane), row = sy + PF::oy(pw) + PF::y(lane);
    const bool inside = (col < W) && (row < H);
    float result = 0.0f;
    // wave-uniform: does this wave's 8x8 patch touch the rectangle at all?
    if (overlaps(rc, sx + PF::ox(pw), sy + PF::oy(pw), PF::W, PF::H)) {
      const Ray r = ray_setup<TIGHT>(s, row, col, inside, cx, cy, rfx, rfy);
      if (r.go) {
        const f32x2 dgxy = {r.dg[0], r.dg[1]}, ogxy = {og[0], og[1]};
        float t = r.t;
        // sphere tracing, cu:196-260: sample, hit if dist < threshold * t (t = the sample's own
        // t), else advance by dist and stop at the far plane or at the step cap.  A lane that
        // leaves keeps its t: the depth of a hit is formed once, after the loop.
        bool hit;
        int n = 0;  // samples taken; the same for every lane still in the loop
        for (;;) {
          ++n;
          const f32x2 t2 = {t, t};
          const float value = march_sample<RT, PACKED>(vsrc, R, __builtin_elementwise_fma(t2, dgxy, ogxy),
                                                       fmaf(t, r.dg[2], og[2]));
          const float dist = value * scale;
          hit = dist < threshold * t;
          if (hit) break;
          const float tn = t + dist;
          if (!(tn < r.tf) || n >= SDFR_MAX_MARCH_STEPS) break;
          t = tn;
        }
        result = hit ? t * r.inv_len : 0.0f;  // inv_len = -d.z of the unit ray
        if (LOSS && result > 0.0f) {
          const float o = obs[row * W + col];
          if (o > 0.0f) {
            l_sum += fabsf(result - o);
            l_cnt += 1.0f;
          }
        }
      }
    }
    if (inside) img[row * W + col] = result;
  }
  if (LOSS) {
    // The tile's record: one (sum, count) pair per WAVE (4 slots; a tile walked by fewer waves zero-fills the rest),
    // each a fixed-order butterfly over the wave's lanes.  No LDS and no barrier: a wave that is done leaves, and
    // whoever reads the record adds the four pairs as (w0 + w1) + (w2 + w3) -- the order in which the tile itself
    // used to add them behind a barrier, so the sums are the same bit for bit.
    l_sum = wave_sum(l_sum);
    l_cnt = wave_sum(l_cnt);
    float* rec = loss_part + (((size_t)b * nty + tile_y) * ntx + tile_x) * kLossRec;
    if (lane == 0) *reinterpret_cast<float2*>(rec + 2 * wave) = make_float2(l_sum, l_cnt);
    if (NW < 4 && wave == 0 && lane >= 2 * NW && lane < kLossRec) rec[lane] = 0.0f;
  }
}

// One workgroup per tile.  (More than half of the tiles of a batch only store zeros -- or, in the
// backward, do nothing -- and an all-culled launch of 153 600 workgroups takes 40 us, so fewer,
// fatter workgroups looked attractive.  Measured, B=256, forward/backward us: one tile per
// workgroup 257/211, 1x2 tiles 264/311, 1x4 270/317, 2x4 295/363; a persistent grid striding over
// the whole tile list 377/472: the dispatcher hides the very uneven tile costs only when it has
// many independent workgroups.)
// NW waves per workgroup.  The dispatcher's cost is per workgroup, not per wave (all-culled launches of
// 1-, 2- and 4-wave workgroups: 66 / 66 / 67 us), and a new workgroup needs a free slot on NW SIMDs at
// once: 64 x 8 tiles walked by 2 waves (4 patches each) 167 us, by 4 waves 172, by 1 wave 183; 8- and
// 16-wave workgroups on 128 x 8 ... 128 x 16 tiles 186 ... 237.
// INLINE (a step over a few views of the plain grid, e.g. the single view of the reference's autograd pair,
// sdf_renderer.py:311-357): the launch has no prologue.  Every workgroup derives its view's record itself -- the
// same arithmetic in every lane, ~0.7 us of latency instead of a launch of ~4 us in front of a 14 us kernel -- the
// view's first workgroup leaves the record for the step's backward, and the threads of the grid zero-fill the
// gradient volume between them.
struct InlineSetup {
  const float* pos;
  const float* quat;
  const float* inv_scale;
  float fx, fy;
  ViewSetup* out;
  float* g_zero;
  size_t n_zero;
};
template <int RT, bool PACKED, int SX, int SY, bool LOSS, int NW, bool INLINE = false>
__global__ __launch_bounds__(NW * 64) void render_forward_kernel(
    const float* __restrict__ src, int R, long long src_view_stride,
    const ViewSetup* __restrict__ setup, int W, int H, int ntx, int nty, float cx, float cy,
    float rfx, float rfy, float threshold, int vec_ok, float* __restrict__ depth,
    const float* __restrict__ target, float* __restrict__ loss_part, unsigned* __restrict__ epoch,
    const unsigned* __restrict__ spans, InlineSetup in, unsigned* __restrict__ count_sync,
    unsigned long long* __restrict__ close_word) {
  if (INLINE) {
    const int b = blockIdx.z;
    ViewSetup s;
    setup_pose(b, in.pos, in.quat, in.inv_scale, R, in.fx, in.fy, s);
    setup_box<false>(s, R, W, H, cx, cy, in.fx, in.fy, nullptr, threshold);
    if ((blockIdx.x | blockIdx.y) == 0 && threadIdx.x == 0) in.out[b] = s;
    const size_t nthreads = (size_t)gridDim.x * gridDim.y * gridDim.z * (NW * 64);
    const size_t me = (((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NW * 64) + threadIdx.x;
    for (size_t i = me; i < in.n_zero; i += nthreads) in.g_zero[i] = 0.0f;
    forward_tile<RT, PACKED, SX, SY, LOSS, PACKED, NW>(blockIdx.x, blockIdx.y, ntx, nty, b, src, R, src_view_stride,
                                                       s, W, H, cx, cy, rfx, rfy, threshold, vec_ok, depth, target,
                                                       loss_part, spans);
    return;
  }
  forward_tile<RT, PACKED, SX, SY, LOSS, PACKED, NW>(blockIdx.x, blockIdx.y, ntx, nty, blockIdx.z, src, R,
                                         src_view_stride, setup[blockIdx.z], W, H, cx, cy, rfx, rfy, threshold,
                                         vec_ok, depth, target, loss_part, spans);
  if (count_sync && (blockIdx.x | blockIdx.y) == 0 && threadIdx.x == 0)
    count_close_view(count_sync, (int)gridDim.z, setup[blockIdx.z].bwd_big != 0, close_word);
  // The workspace's epoch advances once per forward call, after its prologue launch (forward_prologue_kernel), and
  // the entries that launch published are wiped (equal words never read as an entry: the payloads of a valid one
  // are complementary).  Either alone keeps an entry of this call from reading as ready in a later one; the wipe
  // also covers a caller whose other work overwrote the epoch word with the value it had before.
  if (epoch && (blockIdx.x | blockIdx.y | blockIdx.z) == 0) {
    if (threadIdx.x == 0) epoch[0] += 1u;
    unsigned long long* ent = reinterpret_cast<unsigned long long*>(epoch + kSyncHeaderWords);
    for (int j = threadIdx.x; j < 2 * 6 * kPackedMaxR; j += NW * 64) ent[j] = 0ull;
  }
}

// ---------------------------------------------------------------------------------------------
// backward.  grid = (macro-tiles x, macro-tiles y, views)
// ---------------------------------------------------------------------------------------------
// d/dSDF of a tile is pre-summed in LDS before it reaches the volume's global float atomics (one per touched voxel
// per tile).  Two tables share the workgroup's LDS:
//   * the DENSE BOX (round 3): the tile's hit pixels first agree on the box of grid cells they touch (one pass of ray
//     / cell arithmetic, min / max over the wave by DPP, one LDS atomic per wave and bound); if the box has at most
//     kDenseCap voxels the table is simply that box, one int32 per voxel, addressed by arithmetic: 8 ds_add_u32 per
//     hit pixel that return nothing -- no key, no compare-and-swap, no LDS round trip for the wave to wait for.  The
//     run hash's look-ups were 29 % + 25 % of a hit tile's time (stamps).  On the benchmark 88 %
//     of the hit tiles (84 % of the hit pixels) fit (tools/analysis/tile_boxes_r03.py);
//   * the z-run hash (device.hpp) for the tiles whose box is larger (silhouettes seen at a grazing angle, far objects
//     under 64 x 8 tiles).
// Both hold fixed-point sums, so a tile's contribution does not depend on the order in which its lanes arrive.
constexpr int kDenseCap = SDFR_DENSE_CAP;

template <typename Hash>
struct BackwardLds {
  union {
    Hash hash;
    int dense[kDenseCap > 0 ? kDenseCap : 4];
  };
  float wave_part[4][8];
  int tile_max_bits;
  int box[6];   // cells of the tile's hit pixels: min x, y, z, max x, y, z (index of the cell's corner 000)
  float loss_pair[4][2];   // REG: the waves' (sum |est - obs|, count)
};

// fractional bits of a tile's fixed-point sums in 32-bit words: a voxel receives at most one contribution per pixel,
// each below 2^(bits - 1) in magnitude (2^bits for the few extrapolating ones that pass the weight limit)
__host__ __device__ constexpr int tile_fixed_bits(int pixels, int table_bits) {
  int b = 31;
  for (int p = 1; p < pixels; p <<= 1) --b;
  return b < table_bits ? b : table_bits;
}
template <typename Hash> struct HashIs32 { static constexpr bool value = false; };
template <int SLOTS> struct HashIs32<PairRunHash<SLOTS>> { static constexpr bool value = true; };

// grid-space hit point of a pixel of depth z: the arithmetic of the reference's backward (cu:334-345) in the object
// frame.  ONE function for the bounds pass and the main pass of a tile: the same instruction sequence on the same
// inputs chooses the same cell in both.
struct HitPoint {
  V3 d;        // unit ray, camera frame
  V3 o;        // hit point, object frame
  float t;
  float gx, gy, gz;
};
__device__ __forceinline__ HitPoint hit_point(const ViewSetup& s, int row, int col, float z, float cx, float cy,
                                              float rfx, float rfy, float isc, float h) {
  HitPoint p;
  p.d = pixel_ray(row, col, cx, cy, rfx, rfy);
  const V3 dobj = rot_t(s, p.d);
  p.t = z * __builtin_amdgcn_rcpf(-p.d.z);  // -z / d.z   (cu:339)
  p.o = mk(fmaf(p.t, dobj.x, -s.e[0]), fmaf(p.t, dobj.y, -s.e[1]), fmaf(p.t, dobj.z, -s.e[2]));
  p.gx = fmaf(p.o.x * isc, h, h);
  p.gy = fmaf(p.o.y * isc, h, h);
  p.gz = fmaf(p.o.z * isc, h, h);
  return p;
}

// (sum, count) of one forward tile from its record of four per-wave pairs (forward_tile, LOSS): the one association
// every reader uses
__device__ __forceinline__ float2 loss_tile_record(const float* __restrict__ part, size_t tile) {
  const float4* r = reinterpret_cast<const float4*>(part + tile * kLossRec);
  const float4 a = r[0], c = r[1];   // (s0, n0, s1, n1), (s2, n2, s3, n3)
  return make_float2((a.x + a.z) + (c.x + c.z), (a.y + a.w) + (c.y + c.w));
}

// A loss-fused STEP whose forward left the reduction of its (sum, count) tile records to the backward (deferred:
// sdfr_render_step_forward_l1 with loss = loss_stats = NULL): no launch between the two image kernels.  A backward
// tile that has a hit pixel sums the view's counts itself -- integers held in floats below 2^24: the sum is exact in
// any order, so k = weight / count is the very number the reduce launch would have handed it -- and workgroup (0, 0)
// of every view (a corner of the image: nearly always a culled tile) runs the fixed-order reduction of
// loss_reduce_kernel on the side: loss[b], stats[b] for whoever reads them after this launch.
struct LossTiles {
  const float* part;   // the forward's records, [view][nty][ntx][kLossRec]; nullptr: the forward reduced them itself
  int ntx, nty, wl, hl;   // the FORWARD's tiling (not the backward's): tiles of 2^wl x 2^hl pixels
  float* loss;         // [B]
  float* stats;        // [B][2]
};
// every thread of the workgroup calls it; `scratch`: 4 floats of LDS nobody else uses until the next barrier
__device__ __forceinline__ float view_overlap_count(const LossTiles& lt, const ViewSetup& s, int b, float* scratch) {
  const int x0 = s.rect[0], y0 = s.rect[1], x1 = s.rect[2], y1 = s.rect[3];
  float cnt = 0.0f;
  if (x1 > x0 && y1 > y0) {
    // (scalar divisions: the rectangle and the tiling are workgroup-uniform; the threads walk the records as an
    // 8 x 32 block -- no per-thread division in a path every tile with a hit pixel takes)
    const int tx0 = x0 >> lt.wl, tx1 = (x1 - 1) >> lt.wl, ty0 = y0 >> lt.hl, ty1 = (y1 - 1) >> lt.hl;   // (x0, y0 >= 0)
    const size_t base = (size_t)b * lt.ntx * lt.nty;
    const int c = (int)threadIdx.x & 31, r = (int)threadIdx.x >> 5;
    for (int ty = ty0 + r; ty <= ty1; ty += kBlock / 32)
      for (int tx = tx0 + c; tx <= tx1; tx += 32) cnt += loss_tile_record(lt.part, base + (size_t)ty * lt.ntx + tx).y;
  }
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = cnt;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}
// one wave: the body of loss_reduce_kernel (same lane-strided order, same butterfly: the same bits)
__device__ __forceinline__ void reduce_view_loss(const LossTiles& lt, const ViewSetup& s, int b, int lane) {
  const int x0 = s.rect[0], y0 = s.rect[1], x1 = s.rect[2], y1 = s.rect[3];
  float sum = 0.0f, cnt = 0.0f;
  if (x1 > x0 && y1 > y0) {
    const int tx0 = x0 >> lt.wl, tx1 = (x1 - 1) >> lt.wl, ty0 = y0 >> lt.hl, ty1 = (y1 - 1) >> lt.hl;
    const int nx = tx1 - tx0 + 1, n = nx * (ty1 - ty0 + 1);
    const size_t base = (size_t)b * lt.ntx * lt.nty;
    for (int i = lane; i < n; i += 64) {
      const float2 p = loss_tile_record(lt.part, base + (size_t)(ty0 + i / nx) * lt.ntx + tx0 + i % nx);
      sum += p.x;
      cnt += p.y;
    }
  }
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  if (lane == 0) {
    lt.loss[b] = sum / cnt;
    lt.stats[2 * b] = sum;
    lt.stats[2 * b + 1] = cnt;
  }
}

// What a loss-fused tile needs to form its upstream gradient -- looked at only once the tile is known to hold a hit
// pixel: a culled workgroup's life is its chain of dependent scalar loads, and three out of four workgroups are culled.
struct LossArgs {
  const float* loss_grad;    // [B] or nullptr
  const float* loss_stats;   // [B][2] from the forward (nullptr with lt.part)
  float loss_weight;
  LossTiles lt;
};

// One tile of the backward.  Every return is workgroup-uniform.
// LOSS: `grad_depth` is the OBSERVED depth image and the upstream gradient is formed on the fly,
// go = +-k on the overlap mask (obs > 0) & (est > 0), k = weight * dL/dloss_b / count_b
// (the gradient of simple_setup.py:129-135's masked mean of |est - obs|; 0 where est == obs).
// DET: deterministic d/dSDF (SDFR_SDF_GRAD_DETERMINISTIC): `g_sdf` is the 64-bit fixed-point volume, every
// contribution is rounded ONCE, per pixel, to the fixed quantum 2^-kDetQuantumBits -- a function of the pixel alone,
// not of its tile -- and everything after that is integer addition (64-bit LDS run table, 64-bit global atomics).
constexpr int kDetQuantumBits = SDFR_FIXED_QUANTUM_BITS;
// REG (render_fused_l1_pc_kernel: forward and backward of a tile in one launch): the thread's pixel comes in registers
// -- its depth as the march has just left it, the SIGN of (estimate - observation) on the overlap as its upstream
// gradient (the view's k = weight / count is not known before the launch ends: whoever consumes the sums multiplies by
// it) -- and the tile also leaves its (sum |est - obs|, count) pair and adds its count to the view's.  `setup` points at
// the workgroup's own record.  SDFG = false: pose sums only (a loop that does not optimise the shape).
struct RegPixel {
  float z = 0.0f, sign = 0.0f, l_sum = 0.0f, l_cnt = 0.0f;
  int row = 0, col = 0;
  float* tile_loss = nullptr;   // this tile's (sum, count)
  float* view_cnt = nullptr;    // the view's overlap count (atomic: integers below 2^24, exact in any order)
  bool use_table = false;       // d/dSDF pre-summed in the tile's LDS table (a view of many pixels) or sent straight
};
template <int RT, int SX, int SY, typename Hash, bool LOSS, bool DET = false, bool REG = false, bool SDFG = true>
__device__ __forceinline__ void backward_tile(
    BackwardLds<Hash>& lds, int tile_x, int tile_y, size_t record, int b, const LossArgs& la,
    const float* __restrict__ grad_depth, const float* __restrict__ depth,
    const float* __restrict__ sdf, int R, long long sdf_view_stride,
    const ViewSetup* __restrict__ setup, int W, int H, float cx, float cy, float rfx, float rfy,
    int sdf_grad_mode, float* __restrict__ g_sdf, long long g_sdf_view_stride,
    float* __restrict__ partials, const RegPixel& rp = RegPixel{}) {
  static_assert(!REG || (SX * SY == 1 && LOSS && !DET), "the register form: one pixel per thread, loss-fused");
  // d/dSDF pre-summed in the tile's LDS table -- or straight to the volume's float atomics.  REG (the loop's one-launch
  // step) decides per VIEW (rp.use_table, workgroup-uniform): a small object is a few dozen tiles with hit pixels, and
  // what they take is the depth of a tile's chain of phases (bounds pass, clear, adds, flush: three barriers and a table
  // walk), not the number of atomics -- the C5 mug's 4 k pixels: 0.1042 -> 0.0970 ms per iteration without the table;
  // an object that fills the image sends so many that they queue up on the volume -- 25 k pixels 0.119 -> 0.145 ms,
  // 83 k: 0.135 -> 0.306 -- and keeps the table (tools/microbench/fused_render_close.py)
  const bool TABLE = SDFG && (REG ? rp.use_table : !(SDFR_SMALL_DIRECT && std::is_same<Hash, SmallHash>::value && !DET));
  Hash& hash = lds.hash;
  float (*wave_part)[8] = lds.wave_part;

  constexpr int kSubs = SX * SY, kTileW = SX * kSubW, kTileH = SY * kSubH;
  using PB = Patch<kPatchWBwd>;
  const int Rr = RT > 0 ? RT : R;
  const int px0 = tile_x * kTileW, py0 = tile_y * kTileH;
  const ViewSetup& s = REG ? *setup : setup[b];
  const Rect rc{s.rect[0], s.rect[1], s.rect[2], s.rect[3]};
  if (!REG && !overlaps(rc, px0, py0, kTileW, kTileH)) return;  // depth is 0 there by construction
  float* part = partials + record * 8;  // this tile's pose sums

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* zimg = REG ? nullptr : depth + (size_t)b * H * W;
  const float* gimg = REG ? nullptr : grad_depth + (size_t)b * H * W;
  // the thread's pixel of sub-tile `sub`
  auto pixel = [&](int sub, int& row, int& col) {
    if (REG) { row = rp.row; col = rp.col; return; }
    col = px0 + (sub % SX) * kSubW + PB::ox(wave) + PB::x(lane);
    row = py0 + (sub / SX) * kSubH + PB::oy(wave) + PB::y(lane);
  };

  // all depth reads of the macro-tile up front (independent loads), then the upstream
  // gradient of the hit pixels only
  float zs[kSubs], gos[kSubs];
  bool any_hit = false;
#pragma unroll
  for (int sub = 0; sub < kSubs; ++sub) {
    int row, col;
    pixel(sub, row, col);
    zs[sub] = REG ? rp.z : ((col < W && row < H) ? zimg[(size_t)row * W + col] : 0.0f);  // (nt loads: +4 %, measured)
  }
  if (tid == 0) {   // (ordered before the atomics below by the barrier of __syncthreads_or)
    lds.tile_max_bits = 0;
    lds.box[0] = lds.box[1] = lds.box[2] = 0x7fffffff;
    lds.box[3] = lds.box[4] = lds.box[5] = -1;
  }
  float gmax = 0.0f;
#pragma unroll
  for (int sub = 0; sub < kSubs; ++sub) {
    int row, col;
    pixel(sub, row, col);
    const bool hit = zs[sub] != 0.0f;
    gos[sub] = REG ? rp.sign : (hit ? gimg[(size_t)row * W + col] : 0.0f);
    if (LOSS && !REG) {   // the sign of (estimate - observation) on the overlap; times k below
      const float e = zs[sub], o = gos[sub];
      gos[sub] = (e > 0.0f && o > 0.0f) ? ((e > o) ? 1.0f : ((e < o) ? -1.0f : 0.0f)) : 0.0f;
    }
    gmax = fmaxf(gmax, fabsf(gos[sub]));
    any_hit = any_hit || hit;
  }
  if (!__syncthreads_or(any_hit)) {
    if (tid < 8) part[tid] = 0.0f;
    if (REG && tid == 0) *reinterpret_cast<float2*>(rp.tile_loss) = make_float2(0.0f, 0.0f);
    return;
  }
  if (LOSS && !REG) {
    // same expression as depth_l1_grad_kernel (loop.hip): k = weight / count, 0 if the overlap is empty
    const float w = la.loss_grad ? la.loss_weight * la.loss_grad[b] : la.loss_weight;
    const float cnt = la.lt.part ? view_overlap_count(la.lt, s, b, &lds.wave_part[0][0])   // (workgroup-uniform)
                                 : la.loss_stats[2 * b + 1];
    const float k = cnt > 0.0f ? w / cnt : 0.0f;
#pragma unroll
    for (int sub = 0; sub < kSubs; ++sub) gos[sub] = gos[sub] > 0.0f ? k : (gos[sub] < 0.0f ? -k : 0.0f);
    gmax = gmax > 0.0f ? fabsf(k) : 0.0f;
  }
  const float h = 0.5f * (float)(Rr - 1);
  const float scale = s.scale, isc = s.isc;
  const float top = (float)(Rr - 2);
  // bounds pass: the box of cells under this tile's hit pixels (clamped like the cell choice itself, cu:196-207, so
  // the bounds are in [0, R - 2] whatever the depth image holds)
  if (TABLE && __ballot(any_hit) != 0ull) {
    int lo0 = 0x7fffffff, lo1 = 0x7fffffff, lo2 = 0x7fffffff, hi0 = -1, hi1 = -1, hi2 = -1;
#pragma unroll
    for (int sub = 0; sub < kSubs; ++sub) {
      if (zs[sub] == 0.0f) continue;
      int row, col;
      pixel(sub, row, col);
      const HitPoint hp = hit_point(s, row, col, zs[sub], cx, cy, rfx, rfy, isc, h);
      const int ix = (int)fminf(fmaxf(floorf(hp.gx), 0.0f), top);
      const int iy = (int)fminf(fmaxf(floorf(hp.gy), 0.0f), top);
      const int iz = (int)fminf(fmaxf(floorf(hp.gz), 0.0f), top);
      lo0 = min(lo0, ix); lo1 = min(lo1, iy); lo2 = min(lo2, iz);
      hi0 = max(hi0, ix); hi1 = max(hi1, iy); hi2 = max(hi2, iz);
    }
    lo0 = wave_min_to63(lo0); lo1 = wave_min_to63(lo1); lo2 = wave_min_to63(lo2);
    hi0 = wave_max_to63(hi0); hi1 = wave_max_to63(hi1); hi2 = wave_max_to63(hi2);
    const int gbits = wave_max_to63(__float_as_int(gmax));   // non-negative floats order as ints (NaN: above all)
    if (lane == 63) {
      atomicMin(&lds.box[0], lo0); atomicMin(&lds.box[1], lo1); atomicMin(&lds.box[2], lo2);
      atomicMax(&lds.box[3], hi0); atomicMax(&lds.box[4], hi1); atomicMax(&lds.box[5], hi2);
      atomicMax(&lds.tile_max_bits, gbits);
    }
  }
  if (TABLE) __syncthreads();
  // the box in voxels: corners reach one past the last cell.  Workgroup-uniform -> scalar registers.
  const int bx0 = __builtin_amdgcn_readfirstlane(lds.box[0]), by0 = __builtin_amdgcn_readfirstlane(lds.box[1]),
            bz0 = __builtin_amdgcn_readfirstlane(lds.box[2]);
  const int ny = __builtin_amdgcn_readfirstlane(lds.box[4]) - by0 + 2,
            nz = __builtin_amdgcn_readfirstlane(lds.box[5]) - bz0 + 2;
  const int nvox = (__builtin_amdgcn_readfirstlane(lds.box[3]) - bx0 + 2) * ny * nz;
  const int tile_max_bits = __builtin_amdgcn_readfirstlane(lds.tile_max_bits);
  const bool dense = TABLE && !DET && nvox <= kDenseCap;
  constexpr int kDenseBits = tile_fixed_bits(kSubs * kBlock, 22);
  constexpr int kHashBits = HashIs32<Hash>::value ? tile_fixed_bits(kSubs * kBlock, Hash::kBits) : Hash::kBits;
  if (dense) {
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    i32x4v* d4 = reinterpret_cast<i32x4v*>(lds.dense);
    for (int i = tid; i < (nvox + 3) >> 2; i += kBlock) d4[i] = i32x4v{0, 0, 0, 0};
  } else if (TABLE) {
    hash.clear(tid, kBlock);
  }
  if (TABLE) __syncthreads();

  const float* vol = sdf + (size_t)b * sdf_view_stride;
  float* gvol = g_sdf + (size_t)b * g_sdf_view_stride;
  // fixed-point scale: 2^(bits - e) with 2^e >= 2 * max|go| * scale  (power of two)
  const float bound = 2.0f * __int_as_float(tile_max_bits) * fabsf(scale);
  int e2;
  (void)frexpf(bound, &e2);  // bound = m * 2^e2, m in [0.5, 1)  ->  2^e2 > bound
  static_assert(!DET || !HashIs32<Hash>::value, "the deterministic mode needs 64-bit table sums");
  const bool fixed_ok = TABLE && (DET || ((bound > 0.0f) && (bound < 1e30f) && (e2 > -80)));
  const int bits = dense ? kDenseBits : kHashBits;
  const float to_fixed = DET ? (float)(1ll << kDetQuantumBits) : (fixed_ok ? ldexpf(1.0f, bits - e2) : 0.0f);
  const float from_fixed = (fixed_ok && !DET) ? ldexpf(1.0f, e2 - bits) : 0.0f;
  // scaled weights at or above the limit (extrapolating cells) bypass the table
  const float weight_limit = DET ? 1.0e15f   /* < 2^50: fixed_from_float is exact below 2^51 */
                             : dense ? (float)(1 << kDenseBits)
                                     : (HashIs32<Hash>::value ? (float)(1 << (kHashBits < 30 ? kHashBits : 30)) : Hash::kWeightLimit);
  // dense addressing: word = (ix * ny + iy) * nz + iz - c0, formed in float (exact: < 2^24)
  const float f_nz = (float)nz, f_nynz = (float)(ny * nz);
  const int c0 = (bx0 * ny + by0) * nz + bz0;
  const int rel_max = nvox - (ny * nz + nz + 1) - 1;   // last word a cell's corner 000 may take
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};

#pragma unroll
  for (int sub = 0; sub < kSubs; ++sub) {
    float z = zs[sub];
    if (z == 0.0f) continue;
    asm volatile("" : "+v"(z));   // recompute below what the bounds pass computed, do not keep it in registers
    int row, col;
    pixel(sub, row, col);
    const float go = gos[sub];
    const HitPoint hp = hit_point(s, row, col, z, cx, cy, rfx, rfy, isc, h);
    const V3 d = hp.d, o = hp.o;
    const float t = hp.t;
    Cell c;
    gather_cell<RT>(vol, R, hp.gx, hp.gy, hp.gz, c);
    const float tri = trilerp(c);
    // gradient of the trilinear value w.r.t. the cell coordinate
    const float ax = 1.0f - c.ox, ay = 1.0f - c.oy, az = 1.0f - c.oz;
    const float c00 = fmaf(c.v[4], c.ox, c.v[0] * ax), c01 = fmaf(c.v[5], c.ox, c.v[1] * ax);
    const float c10 = fmaf(c.v[6], c.ox, c.v[2] * ax), c11 = fmaf(c.v[7], c.ox, c.v[3] * ax);
    V3 G;
    G.x = ((c.v[4] - c.v[0]) * ay + (c.v[6] - c.v[2]) * c.oy) * az +
          ((c.v[5] - c.v[1]) * ay + (c.v[7] - c.v[3]) * c.oy) * c.oz;
    G.y = (c10 - c00) * az + (c11 - c01) * c.oz;
    G.z = fmaf(c11, c.oy, c01 * ay) - fmaf(c10, c.oy, c00 * ay);

    const float adz = fabsf(d.z);
    const float f = scale * adz;   // cu:372
    const float sg = isc * h;      // s = inv_scale / grid_size (cu:391)
    const float kf = f * sg * go;  // common factor of the pose terms, times upstream grad
    // position: dc/dp_j = -s * R[j][:]
    const V3 RG = rot_f(s, G);
    acc[0] -= kf * RG.x; acc[1] -= kf * RG.y; acc[2] -= kf * RG.z;
    // quaternion: dc/dq_k = s * (d/dq_k[Rhom^T v] - 2 q_k o), v = t d - p
    const V3 v = mk(fmaf(t, d.x, -s.p[0]), fmaf(t, d.y, -s.p[1]), fmaf(t, d.z, -s.p[2]));
    const V3 u = mk(s.q[0], s.q[1], s.q[2]);
    const float w = s.q[3];
    const float udv = dot(u, v);
    const V3 uxv = cross(u, v);
    const float Gv = dot(G, v), Go = dot(G, o), Gu = dot(G, u);
    const V3 vxG = cross(v, G);  // G . (e_k x v) = (v x G)_k
    // d/du_k: -2 u_k v + 2 e_k (u.v) + 2 u v_k - 2 w (e_k x v) - 2 u_k o
    acc[3] += kf * 2.0f * (-u.x * Gv + udv * G.x + v.x * Gu - w * vxG.x - u.x * Go);
    acc[4] += kf * 2.0f * (-u.y * Gv + udv * G.y + v.y * Gu - w * vxG.y - u.y * Go);
    acc[5] += kf * 2.0f * (-u.z * Gv + udv * G.z + v.z * Gu - w * vxG.z - u.z * Go);
    // d/dw: 2 w v - 2 (u x v) - 2 w o
    acc[6] += kf * 2.0f * (w * Gv - dot(G, uxv) - w * Go);
    // inverse scale: dc/ds^-1 = o / g, plus the product rule on scale (cu:439, :457)
    acc[7] += go * (f * h * Go - tri * scale * scale * adz);

    const float gf = go * f;
    const float x1w = c.ox * gf, x0w = ax * gf;
    float w0, w1, w2, w3, w4, w5, w6, w7;
    if (sdf_grad_mode == SDFR_SDF_GRAD_EXACT) {
      w0 = x0w * ay * az;   w1 = x0w * ay * c.oz;   w2 = x0w * c.oy * az;   w3 = x0w * c.oy * c.oz;
      w4 = x1w * ay * az;   w5 = x1w * ay * c.oz;   w6 = x1w * c.oy * az;   w7 = x1w * c.oy * c.oz;
    } else {  // the weights the CUDA kernel really adds (cu:373-388)
      w0 = x0w * ay * c.oz; w1 = x0w * c.oy * az;   w2 = x0w * c.oy * c.oz; w3 = x1w * ay * az;
      w4 = x1w * ay * c.oz; w5 = x1w * ay * c.oz;   w6 = x1w * c.oy * az;   w7 = x1w * c.oy * c.oz;
    }
    // Outside [0,1] cell coordinates (extrapolation) a weight can exceed the bound the fixed-
    // point scale assumes; such a pixel (never the case for a hit inside the volume) and the
    // degenerate-scale case go straight to global float atomics.  So does a pixel whose upstream
    // gradient is NaN (all 8 weights are then NaN, fmaxf(NaN, NaN) = NaN and the comparison below is
    // false): NaN reaches g_sdf as it does through the reference's atomicAdd.
    const float wmax = fmaxf(fmaxf(fmaxf(fabsf(w0), fabsf(w1)), fmaxf(fabsf(w2), fabsf(w3))),
                             fmaxf(fmaxf(fabsf(w4), fabsf(w5)), fmaxf(fabsf(w6), fabsf(w7))));
    if (!SDFG || (REG && go == 0.0f)) {   // (REG: the sign is 0 off the overlap -- nothing to add)
    } else if (fixed_ok && wmax * to_fixed < weight_limit) {
      const float wk[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
      if (dense) {
        // word of corner 000 (the clamp cannot act -- the bounds pass saw this very cell -- and keeps every
        // access inside the box whatever happens)
        int rel = (int)fmaf(c.bx, f_nynz, fmaf(c.by, f_nz, c.bz)) - c0;
        rel = min(max(rel, 0), rel_max);
        int* p0 = lds.dense + rel;
        int* p1 = p0 + ny * nz;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // round to nearest through the mantissa: |w * to_fixed| < 2^22
          const int q = __float_as_int(fmaf(wk[j], to_fixed, 12582912.0f)) - 0x4b400000;
          int* p = ((j & 4) ? p1 : p0) + ((j & 2) ? nz : 0) + (j & 1);
          atomicAdd(p, q);   // result unused: ds_add_u32, nothing to wait for
        }
      } else {
        hash.template add_cell<DET>(gvol, c.lin, Rr, wk, to_fixed);
      }
    } else if (DET) {
      // beyond the table's range (or not finite: the conversion saturates, NaN counts as 0 -- a 64-bit integer
      // volume cannot carry them, include/sdfr.h)
      unsigned long long* g0 = reinterpret_cast<unsigned long long*>(gvol) + c.lin;
      const float wk[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
#pragma unroll
      for (int j = 0; j < 8; ++j)
        atomicAdd(g0 + ((j & 4) ? Rr * Rr : 0) + ((j & 2) ? Rr : 0) + (j & 1),
                  (unsigned long long)__float2ll_rn(wk[j] * to_fixed));
    } else {
      float* g0 = gvol + c.lin;
      atomicAdd(g0, w0);                atomicAdd(g0 + 1, w1);
      atomicAdd(g0 + Rr, w2);           atomicAdd(g0 + Rr + 1, w3);
      atomicAdd(g0 + Rr * Rr, w4);      atomicAdd(g0 + Rr * Rr + 1, w5);
      atomicAdd(g0 + Rr * Rr + Rr, w6); atomicAdd(g0 + Rr * Rr + Rr + 1, w7);
    }
  }

  // pose sums: registers -> wave shuffle -> LDS -> macro-tile partial (a wave without a hit pixel
  // has nothing to reduce)
  if (__ballot(any_hit) != 0ull) {
    const float sk = wave_sum8(acc, lane);  // (8 separate butterflies: +6 us per launch)
    if ((lane & 7) == 0) wave_part[wave][lane >> 3] = sk;
  } else if (lane < 8) {
    wave_part[wave][lane] = 0.0f;
  }
  if (REG) {   // the tile's share of the view's depth loss: four waves' pairs, added as the records' readers add them
    const float ls = wave_sum(rp.l_sum), lc = wave_sum(rp.l_cnt);
    if (lane == 0) { lds.loss_pair[wave][0] = ls; lds.loss_pair[wave][1] = lc; }
  }
  __syncthreads();
  if (tid < 8) part[tid] = (wave_part[0][tid] + wave_part[1][tid]) + (wave_part[2][tid] + wave_part[3][tid]);
  if (REG && tid == 8) {
    const float ls = (lds.loss_pair[0][0] + lds.loss_pair[1][0]) + (lds.loss_pair[2][0] + lds.loss_pair[3][0]);
    const float lc = (lds.loss_pair[0][1] + lds.loss_pair[1][1]) + (lds.loss_pair[2][1] + lds.loss_pair[3][1]);
    *reinterpret_cast<float2*>(rp.tile_loss) = make_float2(ls, lc);
    if (lc > 0.0f) atomicAdd(rp.view_cnt, lc);
  }

  if (!TABLE) return;
  if (dense) {
    // consecutive lanes take consecutive words = consecutive z of a box row: contiguous global float atomics
    const unsigned m_nz = 0xffffffffu / (unsigned)nz + 1u, m_ny = 0xffffffffu / (unsigned)ny + 1u;   // exact for < 2^16
    for (int i = tid; i < nvox; i += kBlock) {
      const int q = lds.dense[i];
      if (q == 0) continue;
      const int rowi = (int)__umulhi((unsigned)i, m_nz), zz = i - rowi * nz;
      const int xx = (int)__umulhi((unsigned)rowi, m_ny), yy = rowi - xx * ny;
      const int lin = ((bx0 + xx) * Rr + by0 + yy) * Rr + bz0 + zz;
      atomicAdd(gvol + lin, (float)q * from_fixed);
    }
  } else {
    hash.template flush<DET>(gvol, Rr * Rr * Rr, from_fixed, tid, kBlock);
  }
}

// Batches pre-sum in the z-pair run table (device.hpp, PairRunHash), small calls in 2-voxel runs x 1024 slots.
// (Until the pair table, batches of low-resolution images took a 4 x 1024 run table of their own: 320x240
// 117.6 -> 109.8 us, 160x120 107.1 -> 73.1 us per 256 views with the one pair table for all image sizes.)
using BatchTable = PairRunHash<SDFR_BWD_SLOTS>;

// One tile of view b.  BATCH: workgroup (bx, by) of the view's own tiling -- 32 x 32 pixels or 64 x 8, chosen per
// view by the set-up (ViewSetup::bwd_big; common.hpp, kBwdBigTile) -- otherwise the 32 x 8 tile (bx, by) of a
// small call.  Returns are workgroup-uniform.
template <int RT, bool BATCH, bool LOSS, bool DET = false, bool PAIR = false>
__device__ __forceinline__ void backward_dispatch(
    unsigned char* raw, int bx, int by, int ntx, int nty, int stride, int b,
    const float* __restrict__ grad_depth, const float* __restrict__ depth, const float* __restrict__ sdf, int R,
    long long sdf_view_stride, const ViewSetup* __restrict__ setup, int W, int H, float cx, float cy, float rfx,
    float rfy, int sdf_grad_mode, float* __restrict__ g_sdf, long long g_sdf_view_stride,
    float* __restrict__ partials, const LossArgs& la = LossArgs{}) {
  if (BATCH) {
    using Table = typename std::conditional<DET, BatchHash, BatchTable>::type;   // DET: 64-bit sums
    auto& lds = *reinterpret_cast<BackwardLds<Table>*>(raw);
    const size_t record = (size_t)b * stride + by * ntx + bx;   // ntx = the 64 x 8 tiling's
    if (setup[b].bwd_big) {
      const int tx = 2 * bx + (by & 1), ty = by >> 1;
      if (tx >= kBwdBigTile.nx(W) || ty >= kBwdBigTile.ny(H)) return;
      backward_tile<RT, kBwdBigTile.sx, kBwdBigTile.sy, Table, LOSS, DET>(
          lds, tx, ty, record, b, la, grad_depth, depth, sdf, R, sdf_view_stride, setup, W, H, cx, cy, rfx,
          rfy, sdf_grad_mode, g_sdf, g_sdf_view_stride, partials);
    } else if (PAIR) {
#pragma unroll 1
      // two tiles, one after the other.  Every exit of a tile is workgroup-uniform, and no barrier is needed between
      // them: whatever the second tile does to the table comes after ITS first barrier, which the waves reach only
      // when they have left the first tile (whose last table reads are in its flush); before that barrier the second
      // tile writes only `box` / `tile_max_bits`, which the first one has not read since its own second barrier.
      for (int k = 0; k < 2; ++k) {
        const int ty = 2 * by + k;
        if (ty >= nty) break;
        backward_tile<RT, SDFR_MACRO_SX, SDFR_MACRO_SY, Table, LOSS, DET>(
            lds, bx, ty, (size_t)b * stride + ty * ntx + bx, b, la, grad_depth, depth, sdf, R, sdf_view_stride,
            setup, W, H, cx, cy, rfx, rfy, sdf_grad_mode, g_sdf, g_sdf_view_stride, partials);
      }
    } else {
      if (by >= nty) return;
      backward_tile<RT, SDFR_MACRO_SX, SDFR_MACRO_SY, Table, LOSS, DET>(
          lds, bx, by, record, b, la, grad_depth, depth, sdf, R, sdf_view_stride, setup, W, H, cx, cy, rfx,
          rfy, sdf_grad_mode, g_sdf, g_sdf_view_stride, partials);
    }
  } else {
    auto& lds = *reinterpret_cast<BackwardLds<SmallHash>*>(raw);
    backward_tile<RT, 1, 1, SmallHash, LOSS, DET>(
        lds, bx, by, ((size_t)b * nty + by) * ntx + bx, b, la, grad_depth, depth, sdf, R, sdf_view_stride,
        setup, W, H, cx, cy, rfx, rfy, sdf_grad_mode, g_sdf, g_sdf_view_stride, partials);
  }
}

// grid: (tiles x, rows, views): BATCH: the 64 x 8 tiling's columns x backward_batch_rows(H); else the 32 x 8 tiling
// 8 waves per SIMD for the backward: the register allocator is held to 64 VGPRs (it takes 67 on its own: 7 waves
// per SIMD; the price is 8 bytes of scratch): stand-alone backward 134.6 -> 129.3 us, a step's 117.8 -> 116.6 us at the
// benchmark, mug-sized objects unchanged.  0: the compiler's own choice (timing experiments).
// (The single-view tiling's workgroups hold 21 KB of LDS each: seven of them, 7 waves per SIMD, is what a CU takes
// whatever the registers -- there the target is 7, which the compiler's own 67 VGPRs meet without a spill; asking for
// 8 only earned "failed to meet occupancy target" from every such instantiation of a clean build.)
#if SDFR_BWD_WAVES_PER_EU
#define SDFR_BWD_OCC __attribute__((amdgpu_waves_per_eu(BATCH ? SDFR_BWD_WAVES_PER_EU : (SDFR_BWD_WAVES_PER_EU > 7 ? 7 : SDFR_BWD_WAVES_PER_EU), \
                                                        BATCH ? SDFR_BWD_WAVES_PER_EU : (SDFR_BWD_WAVES_PER_EU > 7 ? 7 : SDFR_BWD_WAVES_PER_EU))))
#else
#define SDFR_BWD_OCC
#endif
template <int RT, bool BATCH, bool LOSS, bool DET = false, bool PAIR = false>
__global__ __launch_bounds__(kBlock) SDFR_BWD_OCC void render_backward_kernel(
    const float* __restrict__ grad_depth, const float* __restrict__ depth,
    const float* __restrict__ sdf, int R, long long sdf_view_stride,
    const ViewSetup* __restrict__ setup, int W, int H, int ntx, int nty, int stride, float cx, float cy,
    float rfx, float rfy, int sdf_grad_mode, float* __restrict__ g_sdf,
    long long g_sdf_view_stride, float* __restrict__ partials, const float* __restrict__ loss_grad,
    const float* __restrict__ loss_stats, float loss_weight, LossTiles lt) {
  constexpr size_t kBatchLds = sizeof(BackwardLds<BatchTable>) > sizeof(BackwardLds<BatchHash>)
                                   ? sizeof(BackwardLds<BatchTable>) : sizeof(BackwardLds<BatchHash>);
  __shared__ __attribute__((aligned(16))) unsigned char raw[BATCH ? (DET ? kBatchLds : sizeof(BackwardLds<BatchTable>))
                                                                  : sizeof(BackwardLds<SmallHash>)];
  backward_dispatch<RT, BATCH, LOSS, DET, PAIR>(raw, blockIdx.x, blockIdx.y, ntx, nty, stride, blockIdx.z,
                                     grad_depth, depth, sdf, R, sdf_view_stride, setup, W, H, cx, cy, rfx, rfy,
                                     sdf_grad_mode, g_sdf, g_sdf_view_stride, partials,
                                     LossArgs{loss_grad, loss_stats, loss_weight, lt});
  // (after the tile, not in front of it: nothing of this is on a culled workgroup's way to its exit)
  if (LOSS && lt.part && (blockIdx.x | blockIdx.y) == 0 && threadIdx.x < 64)
    reduce_view_loss(lt, setup[blockIdx.z], (int)blockIdx.z, (int)threadIdx.x);
}

// The renderer's backward (depth-L1 form) and the sampler's backward (point-cloud L1 form) of one loop iteration in
// ONE launch.  They are independent -- both read the SDF and the poses and add into the same d/dSDF volume --
// and in the captured loop each is a launch of its own that cannot fill the chip (1 200 small tiles, 13 blocks
// of points): side by side they take max(12, 14) us instead of 12 + 14.  The first `pc_rows` rows of the grid
// are the sampler's blocks (they take longest, so they start first), the rest the image tiles.
template <int RT, bool BATCH, bool DET = false>
__global__ __launch_bounds__(kBlock) void render_backward_pc_kernel(
    const float* __restrict__ target, const float* __restrict__ depth,
    const float* __restrict__ sdf, int R, long long sdf_view_stride,
    const ViewSetup* __restrict__ setup, int W, int H, int ntx, int nty, int stride, float cx, float cy,
    float rfx, float rfy, int sdf_grad_mode, float* __restrict__ g_sdf,
    long long g_sdf_view_stride, float* __restrict__ partials, const float* __restrict__ loss_grad,
    const float* __restrict__ loss_stats, float loss_weight, int pc_rows, PcBackwardArgs pa, LossTiles lt) {
  constexpr size_t kBatchLds = (DET && sizeof(BackwardLds<BatchHash>) > sizeof(BackwardLds<BatchTable>))
                                   ? sizeof(BackwardLds<BatchHash>) : sizeof(BackwardLds<BatchTable>);
  constexpr size_t kTileLds = BATCH ? kBatchLds : sizeof(BackwardLds<SmallHash>);
  constexpr size_t kLdsBytes = kTileLds > sizeof(PcBackwardLds) ? kTileLds : sizeof(PcBackwardLds);
  __shared__ __attribute__((aligned(16))) unsigned char raw[kLdsBytes];
  const int b = blockIdx.z;
  if ((int)blockIdx.y < pc_rows) {
    const int bx = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    if (bx < (pa.groups > 0 ? pa.groups : pa.nblk))
      pc_backward_block<RT, true, DET, (SDFR_SMALL_DIRECT && !BATCH && !DET)>(*reinterpret_cast<PcBackwardLds*>(raw), pa, bx, b);
    return;
  }
  backward_dispatch<RT, BATCH, true, DET>(raw, blockIdx.x, (int)blockIdx.y - pc_rows, ntx, nty, stride, b, target, depth,
                                     sdf, R, sdf_view_stride, setup, W, H, cx, cy, rfx, rfy, sdf_grad_mode, g_sdf,
                                     g_sdf_view_stride, partials, LossArgs{loss_grad, loss_stats, loss_weight, lt});
  if (lt.part && blockIdx.x == 0 && (int)blockIdx.y == pc_rows && threadIdx.x < 64)
    reduce_view_loss(lt, setup[b], b, (int)threadIdx.x);
}

// ONE launch for the render pair of a loop iteration over a few views (and the sampler's blocks beside it, as in
// render_backward_pc_kernel): a tile marches its rays (forward_tile's arithmetic, INLINE set-up, 32 x 8 pixels, one
// 8 x 8 patch per wave) and, while the depths are still in registers, runs the backward of its hit pixels
// (backward_tile, REG).  What a tile cannot know is the view's overlap count -- the depth loss is a MEAN over pixels
// every tile contributes to -- so the upstream gradient is the bare sign of (est - obs):
//   * d/dSDF of the depth term goes UNSCALED into the view's own volume g_depth[b] (the sampler's blocks add the
//     point-cloud term, whose scale is known, into pa.g_sdf): the consumer forms  g_sdf + sum_b k_b g_depth[b],
//     k_b = weight / count_b  (sdfr_decoder_backward_latent_deferred_scaled);
//   * the tile's pose sums are unscaled too, its (sum |est - obs|, count) pair lies in `tile_loss`, and the view's count
//     is summed atomically in `view_cnt` (integers below 2^24 held in floats: exact whatever the order) -- the loop's
//     tail multiplies and resets (sdfr_loop_tail_fused).
// A view whose observed point set is small sends d/dSDF straight to the volumes' float atomics -- tiles and sampler blocks
// alike; a larger one pre-sums in the LDS tables as the two launches do (`direct` below; backward_tile REG / TABLE,
// pc_backward_block DIRECT).
// Nothing in this launch zero-fills: the volumes are cleared by their consumer (the decoder VJP's last launch).
// The depth image is the forward's, bit for bit; the gradients differ from the two launches' in rounding only
// (k is applied to sums instead of to terms).     grid (tiles x, pc_rows + tiles y, views)
// PCD: the sampler's blocks add straight to the (shared) point-cloud volume -- one or two views; more views collide
// there, and their blocks pre-sum in the LDS run table as in the two-launch form.
template <int RT, bool SDFG, bool PCD>
__global__ __launch_bounds__(kBlock) void render_fused_l1_pc_kernel(
    const float* __restrict__ sdf, int R, long long sdf_view_stride, int W, int H, int ntx, int nty, float cx, float cy,
    float rfx, float rfy, float threshold, int vec_ok, float* __restrict__ depth, const float* __restrict__ target,
    InlineSetup in, int sdf_grad_mode, float* __restrict__ g_depth, float* __restrict__ partials,
    float* __restrict__ tile_loss, float* __restrict__ view_cnt, int pc_rows, PcBackwardArgs pa) {
  constexpr size_t kTileLds = sizeof(BackwardLds<SmallHash>);
  constexpr size_t kLdsBytes = kTileLds > sizeof(PcBackwardLds) ? kTileLds : sizeof(PcBackwardLds);
  __shared__ __attribute__((aligned(16))) unsigned char raw[kLdsBytes];
  const int b = blockIdx.z;
  // straight atomics or the LDS tables: by the size of the view's observed point set -- its mask's pixels, which is
  // what the estimate comes to cover (workgroup-uniform, one scalar load; backward_tile, TABLE)
  const int n_obs = pa.offsets ? pa.offsets[b + 1] - pa.offsets[b] : pa.n_single;
  const bool direct = n_obs <= SDFR_FUSED_DIRECT_MAX_POINTS;
  if ((int)blockIdx.y < pc_rows) {
    const int bx = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    if (bx < (pa.groups > 0 ? pa.groups : pa.nblk)) {
      if (PCD && direct) pc_backward_block<RT, true, false, true, SDFG>(*reinterpret_cast<PcBackwardLds*>(raw), pa, bx, b);
      else pc_backward_block<RT, true, false, false, SDFG>(*reinterpret_cast<PcBackwardLds*>(raw), pa, bx, b);
    }
    return;
  }
  const int tile_x = blockIdx.x, tile_y = (int)blockIdx.y - pc_rows;
  ViewSetup s;
  setup_pose(b, in.pos, in.quat, in.inv_scale, R, in.fx, in.fy, s);
  setup_box<false>(s, R, W, H, cx, cy, in.fx, in.fy, nullptr, threshold);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if ((tile_x | tile_y) == 0 && tid == 0) in.out[b] = s;   // (the tail's reductions walk the view's rectangle)
  using PF = Patch<kPatchWFwd>;
  const int px0 = tile_x * kSubW, py0 = tile_y * kSubH;
  const Rect rc{s.rect[0], s.rect[1], s.rect[2], s.rect[3]};
  float* img = depth + (size_t)b * H * W;
  const size_t tile = ((size_t)b * nty + tile_y) * ntx + tile_x;
  if (!overlaps(rc, px0, py0, kSubW, kSubH)) {   // nothing of the cube projects here: zeros, and no record anyone reads
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (vec_ok) {
      if (tid < kSubW * kSubH / 4) {
        const int row = py0 + tid / (kSubW / 4), col = px0 + (tid % (kSubW / 4)) * 4;
        if (row < H && col < W) __builtin_nontemporal_store(zero4, reinterpret_cast<f32x4*>(img + (size_t)row * W + col));
      }
    } else {
      const int row = py0 + tid / kSubW, col = px0 + tid % kSubW;
      if (row < H && col < W) img[(size_t)row * W + col] = 0.0f;
    }
    return;
  }
  const int Rr = RT > 0 ? RT : R;
  const float* vol = sdf + (size_t)b * sdf_view_stride;
  const __amdgpu_buffer_rsrc_t vsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vol), 0, (int)((unsigned)Rr * Rr * Rr * 4u), 0x00020000);
  RegPixel rp;
  rp.col = px0 + PF::ox(wave) + PF::x(lane);
  rp.row = py0 + PF::oy(wave) + PF::y(lane);
  rp.tile_loss = tile_loss + tile * kLossRec;
  rp.view_cnt = view_cnt + b;
  rp.use_table = !direct;
  const bool inside = (rp.col < W) && (rp.row < H);
  // the march: forward_tile's (plain grid, full cube), one pixel per lane
  if (overlaps(rc, px0 + PF::ox(wave), py0 + PF::oy(wave), PF::W, PF::H)) {
    const Ray r = ray_setup<false>(s, rp.row, rp.col, inside, cx, cy, rfx, rfy);
    if (r.go) {
      const f32x2 dgxy = {r.dg[0], r.dg[1]}, ogxy = {s.og[0], s.og[1]};
      float t = r.t;
      bool hit;
      int n = 0;
      for (;;) {
        ++n;
        const f32x2 t2 = {t, t};
        const float value = march_sample<RT, false>(vsrc, R, __builtin_elementwise_fma(t2, dgxy, ogxy),
                                                    fmaf(t, r.dg[2], s.og[2]));
        const float dist = value * s.scale;
        hit = dist < threshold * t;
        if (hit) break;
        const float tn = t + dist;
        if (!(tn < r.tf) || n >= SDFR_MAX_MARCH_STEPS) break;
        t = tn;
      }
      rp.z = hit ? t * r.inv_len : 0.0f;
      if (rp.z > 0.0f) {
        const float o = target[(size_t)b * H * W + rp.row * W + rp.col];
        if (o > 0.0f) {
          rp.l_sum = fabsf(rp.z - o);
          rp.l_cnt = 1.0f;
          rp.sign = (rp.z > o) ? 1.0f : ((rp.z < o) ? -1.0f : 0.0f);
        }
      }
    }
  }
  if (inside) img[rp.row * W + rp.col] = rp.z;
  // (every view its own unscaled volume: its k is its own)
  backward_tile<RT, 1, 1, SmallHash, true, false, true, SDFG>(
      *reinterpret_cast<BackwardLds<SmallHash>*>(raw), tile_x, tile_y, tile, b, LossArgs{}, nullptr, nullptr, sdf, R,
      sdf_view_stride, &s, W, H, cx, cy, rfx, rfy, sdf_grad_mode, g_depth, SDFG ? (long long)Rr * Rr * Rr : 0, partials,
      rp);
}

// deterministic mode: the 64-bit fixed-point volume -> float (one rounding per voxel)
__global__ __launch_bounds__(256) void fixed_to_float_kernel(const long long* __restrict__ fixed, size_t n,
                                                             float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)fixed[i] * (1.0f / (float)(1ll << kDetQuantumBits));
}

// Fixed-order sum of a view's tile partials: one wave per view.  stride > 0: a batch backward (the view's own
// tiling, `stride` records per view); else the tiling (ntx, nty, tile_w, tile_h) for every view.
__global__ __launch_bounds__(64) void pose_reduce_kernel(const float* __restrict__ partials,
                                                         const ViewSetup* __restrict__ setup, int W, int H,
                                                         int ntx, int nty, int tile_w, int tile_h, int stride,
                                                         float* __restrict__ g_pos,
                                                         float* __restrict__ g_quat,
                                                         float* __restrict__ g_inv_scale) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const ViewSetup& s = setup[b];
  size_t first = (size_t)b * ntx * nty;
  const bool big = stride > 0 && s.bwd_big;
  if (stride > 0) first = (size_t)b * stride;
  if (big) { tile_w = kBwdBigTile.w(); tile_h = kBwdBigTile.h(); }
  // live tile range of this view (same predicate as the image kernels)
  const int x0 = s.rect[0], y0 = s.rect[1], x1 = s.rect[2], y1 = s.rect[3];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (x1 > x0 && y1 > y0) {
    const int tx0 = x0 / tile_w, tx1 = (x1 - 1) / tile_w, ty0 = y0 / tile_h, ty1 = (y1 - 1) / tile_h;
    const int nx = tx1 - tx0 + 1, n = nx * (ty1 - ty0 + 1);
    const float* base = partials + first * 8;
    for (int i = lane; i < n; i += 64) {
      const int ty = ty0 + i / nx, tx = tx0 + i % nx;
      const size_t rec = big ? (size_t)backward_big_record(tx, ty, W) : (size_t)ty * ntx + tx;
      const float4* p = reinterpret_cast<const float4*>(base + rec * 8);
      const float4 a = p[0], c = p[1];
      acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
      acc[4] += c.x; acc[5] += c.y; acc[6] += c.z; acc[7] += c.w;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
    g_pos[3 * b] = acc[0]; g_pos[3 * b + 1] = acc[1]; g_pos[3 * b + 2] = acc[2];
    g_quat[4 * b] = acc[3]; g_quat[4 * b + 1] = acc[4]; g_quat[4 * b + 2] = acc[5];
    g_quat[4 * b + 3] = acc[6];
    g_inv_scale[b] = acc[7];
  }
}

// Fixed-order sum of a view's per-tile (sum, count) records: one wave per view.
// loss[b] = sum / count (NaN for an empty overlap, like torch.mean of an empty selection).
__global__ __launch_bounds__(64) void loss_reduce_kernel(const float* __restrict__ loss_part,
                                                         const ViewSetup* __restrict__ setup,
                                                         int ntx, int nty, int tile_w, int tile_h,
                                                         float* __restrict__ loss,
                                                         float* __restrict__ stats) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const ViewSetup& s = setup[b];
  const int x0 = s.rect[0], y0 = s.rect[1], x1 = s.rect[2], y1 = s.rect[3];
  float sum = 0.0f, cnt = 0.0f;
  if (x1 > x0 && y1 > y0) {
    const int tx0 = x0 / tile_w, tx1 = (x1 - 1) / tile_w, ty0 = y0 / tile_h, ty1 = (y1 - 1) / tile_h;
    const int nx = tx1 - tx0 + 1, n = nx * (ty1 - ty0 + 1);
    const size_t base = (size_t)b * ntx * nty;
    for (int i = lane; i < n; i += 64) {
      const float2 p = loss_tile_record(loss_part, base + (size_t)(ty0 + i / nx) * ntx + tx0 + i % nx);
      sum += p.x;
      cnt += p.y;
    }
  }
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  if (lane == 0) {
    loss[b] = sum / cnt;
    stats[2 * b] = sum;
    stats[2 * b + 1] = cnt;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int check_common(int R, int B, int W, int H, float fx, float fy) {
  if (R < 2 || R > 1023) return fail(SDFR_E_INVALID, "R=%d out of range [2,1023]", R);
  if (B < 0 || W < 0 || H < 0) return fail(SDFR_E_INVALID, "negative size B=%d W=%d H=%d", B, W, H);
  if (B > 65535) return fail(SDFR_E_INVALID, "B=%d exceeds 65535 views per call", B);
  if (kSmallTile.ny(H) > 65535) return fail(SDFR_E_INVALID, "H=%d too large", H);
  if ((long long)W * H > 0x7fffffffLL) return fail(SDFR_E_INVALID, "image of %d x %d pixels too large", W, H);
  if ((long long)W * H * (long long)(B > 0 ? B : 1) > (1LL << 40))
    return fail(SDFR_E_INVALID, "image batch too large");
  if (!(fx != 0.0f) || !(fy != 0.0f)) return fail(SDFR_E_INVALID, "focal length must be non-zero");
  return 0;
}

size_t setup_bytes(int B) { return (size_t)(B > 0 ? B : 0) * sizeof(ViewSetup); }

// The packed-record march pays a re-pack of the grid per call (R^3 x 32 bytes written); worth it
// when the grid is shared by several views and small enough to stay cache resident.
bool use_packed(int R, int B, long long sdf_view_stride) {
  return sdf_view_stride == 0 && B >= kPackedMinViews && R <= kPackedMaxR;
}
size_t packed_bytes(int R) { return (size_t)R * record_slab(R) * 4 * sizeof(float); }
// Workspace layouts (common.hpp, kSyncBytes): [view records][sync region] and then
//   forward        [face records]            (+ [loss records] for the depth-L1 form)
//   backward       [tile partials]
//   step           [face records][tile partials]
// (between the sync region and the call's scratch: the band spans, common.hpp)
size_t sync_offset(int B) { return setup_bytes(B); }
size_t scratch_offset(int B, int H) { return scratch_offset_bytes(B, H); }
size_t cells_bytes(int R) { return (R >= 2 && R <= kPackedMaxR) ? packed_bytes(R) : 0; }
size_t step_partials_offset(int R, int B, int H) { return scratch_offset(B, H) + cells_bytes(R); }
// deterministic d/dSDF: a 64-bit fixed-point volume behind the tile partials (grids up to kDetMaxR)
constexpr int kDetMaxR = 128;
size_t fixed_bytes(int R) { return (R >= 2 && R <= kDetMaxR) ? (size_t)R * R * R * sizeof(long long) : 0; }
size_t partials_bytes(int B, int W, int H) {
  if (B <= 0 || W <= 0 || H <= 0) return 0;
  // one 32-byte record per tile of the finer geometry (or per workgroup of a batch launch)
  const size_t small = (size_t)kSmallTile.nx(W) * kSmallTile.ny(H), batch = (size_t)backward_tile_stride(W, H);
  return (size_t)B * (small > batch ? small : batch) * 8 * sizeof(float);
}

}  // namespace
}  // namespace sdfr

using namespace sdfr;

extern "C" size_t sdfr_render_forward_workspace_bytes(int R, int B, int W, int H) {
  (void)W; (void)H;
  return scratch_offset(B, H) + cells_bytes(R);
}

extern "C" size_t sdfr_render_sync_offset(int B) { return sync_offset(B); }

extern "C" size_t sdfr_render_fixed_volume_offset(int R, int B, int W, int H, int step_layout) {
  if (R < 2 || B <= 0) return 0;
  return (step_layout ? step_partials_offset(R, B, H) : scratch_offset(B, H)) + partials_bytes(B, W, H);
}

extern "C" size_t sdfr_render_partials_offset(int R, int B, int W, int H, int step_layout) {
  (void)W;
  if (R < 2 || B <= 0) return 0;
  return step_layout ? step_partials_offset(R, B, H) : scratch_offset(B, H);
}

extern "C" int sdfr_fixed_to_float(const long long* fixed, size_t n, float* out, int device, void* stream) {
  if (n == 0) return 0;
  if (!fixed || !out) return fail(SDFR_E_NULL, "sdfr_fixed_to_float: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fixed,
                     n, out);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t sdfr_render_forward_l1_workspace_bytes(int R, int B, int W, int H) {
  size_t n = (sdfr_render_forward_workspace_bytes(R, B, W, H) + 127) & ~(size_t)127;
  // one record (four per-wave (sum, count) pairs) per tile of the finer geometry
  if (B > 0 && W > 0 && H > 0) n += (size_t)B * kSmallTile.nx(W) * kSmallTile.ny(H) * kLossRec * sizeof(float);
  return n;
}

// (behind the int64 volume: the (sum, count) tile records of the loss-fused step, sdfr_render_step_forward_l1)
size_t step_loss_offset(int R, int B, int W, int H) {
  return step_partials_offset(R, B, H) + partials_bytes(B, W, H) + fixed_bytes(R);
}
extern "C" size_t sdfr_render_step_workspace_bytes(int R, int B, int W, int H) {
  if (R < 2 || B <= 0) return 256;
  size_t n = step_loss_offset(R, B, W, H);
  if (W > 0 && H > 0) n += (size_t)B * kSmallTile.nx(W) * kSmallTile.ny(H) * kLossRec * sizeof(float);
  return n;
}

extern "C" size_t sdfr_render_backward_workspace_bytes(int R, int B, int W, int H) {
  if (B <= 0 || W <= 0 || H <= 0) return scratch_offset(B, H);
  return scratch_offset(B, H) + partials_bytes(B, W, H) + fixed_bytes(R);
}

namespace {
// where a forward call keeps its pieces in the caller's workspace
struct ForwardLayout {
  ViewSetup* setup;
  float* cells;      // face records (packed_bytes)
  unsigned* sync;    // sync region (kSyncBytes): epoch, fallback count, plane-minimum entries
  unsigned* spans;   // band spans (common.hpp): B rows of span_stride_words(H) words
  float* loss_part;  // LOSS: (sum, count) per tile
};
ForwardLayout plain_forward_layout(void* workspace, int R, int B, int W, int H) {
  char* w = (char*)workspace;
  ForwardLayout l;
  l.setup = (ViewSetup*)w;
  l.sync = (unsigned*)(w + sync_offset(B));
  l.spans = (unsigned*)(w + spans_offset_bytes(B));
  l.cells = (float*)(w + scratch_offset(B, H));  // 128-byte aligned
  l.loss_part = (float*)(w + ((sdfr_render_forward_workspace_bytes(R, B, W, H) + 127) & ~(size_t)127));
  return l;
}

// target == nullptr: the plain forward.  Otherwise the forward with the depth-L1 folded in.
// g_zero (nullable): n_zero words zero-filled by the prologue (the step's gradient volume).
int forward_impl(const char* fn, const float* sdf, int R, long long sdf_view_stride, const float* pos,
                 const float* quat, const float* inv_scale, int B, int W, int H, float cx, float cy,
                 float fx, float fy, float threshold, float* depth, const float* target, float* loss,
                 float* loss_stats, void* workspace, size_t workspace_bytes, size_t need, const ForwardLayout& lay,
                 float* g_zero, size_t n_zero, int device, void* stream, unsigned long long* close_word = nullptr) {
  const bool with_loss = target != nullptr;
  if (int rc = check_common(R, B, W, H, fx, fy)) return rc;
  if (sdf_view_stride != 0 && sdf_view_stride < (long long)R * R * R)
    return fail(SDFR_E_INVALID, "sdf_view_stride must be 0 or >= R^3");
  if (B == 0 || W == 0 || H == 0) {
    if (g_zero && n_zero) {
      SDFR_HIP_TRY(hipSetDevice(device));
      zero_words_async(g_zero, n_zero, (hipStream_t)stream);
    }
    return 0;
  }
  if (!sdf || !pos || !quat || !inv_scale || !depth || !workspace)
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (workspace_bytes < need)
    return fail(SDFR_E_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, workspace_bytes, need);
  if ((uintptr_t)workspace % alignof(ViewSetup))
    return fail(SDFR_E_INVALID, "workspace must be %zu-byte aligned", alignof(ViewSetup));
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  ViewSetup* setup = lay.setup;
  float* cells = lay.cells;
  float* loss_part = lay.loss_part;
  const bool packed = use_packed(R, B, sdf_view_stride);
  const TileGeom geom = forward_geom(B, W, H);
  // (small tiles: the set-up each workgroup repeats is amortised over one 32 x 8 tile, so only where launches are
  // what a call costs)
  const bool inline_setup = g_zero && !packed && geom.sx * geom.sy == 1 && B <= kInlineSetupMaxViews;
  unsigned* epoch = nullptr;
  // (the one-launch prologue reads the grid with 16-byte loads: other grids take the two-launch form)
  if (packed && (R & 3) == 0 && ((uintptr_t)sdf & 15) == 0) {
    const int n_pack = (R * R * R + 255) / 256, n_setup = (B + 3) / 4;
    hipLaunchKernelGGL(forward_prologue_kernel, dim3(3 * R + n_setup + n_pack), dim3(256), 0, st, sdf, R,
                       (float4*)cells, 3 * R, n_setup, lay.sync, pos, quat, inv_scale, B, W, H, cx, cy, fx, fy,
                       setup, threshold, g_zero, n_zero, kPrologueMaxPolls,
                       lay.spans);
    epoch = lay.sync;
  } else if (g_zero && !packed && inline_setup) {
    // a step over a few views of the plain grid: no prologue launch at all (render_forward_kernel, INLINE)
  } else if (g_zero && !packed) {
    // a step over a few views (or with one grid per view): zero fill + set-up in ONE launch -- the stand-alone
    // backward's prologue, run early.  Without plane minima the record does not depend on the threshold, and the
    // one-thread and one-wave forms of the set-up give the same record (min / max of the same 8 corners).
    hipLaunchKernelGGL(backward_prologue_kernel, dim3((unsigned)((std::max(n_zero, (size_t)B) + 255) / 256)),
                       dim3(256), 0, st, g_zero, n_zero, pos, quat, inv_scale, B, R, W, H, cx, cy, fx, fy, setup);
  } else {
    if (g_zero) zero_words_async(g_zero, n_zero, st);
    float* plane_min = nullptr;
    if (packed) {
      const int n_pack = (R * R * R + 255) / 256;
      plane_min = (float*)(lay.sync + kSyncHeaderWords);
      hipLaunchKernelGGL(pack_cells_kernel, dim3(n_pack + 3 * R), dim3(256), 0, st, sdf, R,
                         (float4*)cells, n_pack, plane_min);
    }
    hipLaunchKernelGGL(view_setup_kernel, dim3(B), dim3(64), 0, st, pos, quat, inv_scale, B, R,
                       W, H, cx, cy, fx, fy, setup, plane_min, threshold, lay.spans);
  }
  const bool macro = geom.sx * geom.sy > 1;
  const int ntx = geom.nx(W), nty = geom.ny(H);
  const dim3 grid_tile((unsigned)ntx, (unsigned)nty, (unsigned)B);
  const float rfx = (float)(1.0 / (double)fx), rfy = (float)(1.0 / (double)fy);
  // a step's forward over a batch counts its close views (count_close_view): what the backward's half-grid hint needs
  unsigned* count_sync = (g_zero && !inline_setup && macro) ? lay.sync : nullptr;
  const int vec_ok = (W % 4 == 0) && ((uintptr_t)depth % 16 == 0);
#define SDFR_LAUNCH_FWD_L(RT, PK, SRC, STRIDE, SX, SY, LOSS)                                          \
  hipLaunchKernelGGL((render_forward_kernel<RT, PK, SX, SY, LOSS, (SX * SY > 1 && SX < 4 ? kFwdWaves : 4)>), grid_tile, \
                     dim3((SX * SY > 1 && SX < 4 ? kFwdWaves : 4) * 64), 0, st, \
                     SRC, R, STRIDE, setup, W, H, ntx, nty, cx, cy, rfx, rfy, threshold, vec_ok,      \
                     depth, target, loss_part, epoch, lay.spans, InlineSetup{}, count_sync, close_word)
#define SDFR_LAUNCH_FWD_G(RT, PK, SRC, STRIDE, SX, SY)                                               \
  do {                                                                                               \
    if (with_loss) SDFR_LAUNCH_FWD_L(RT, PK, SRC, STRIDE, SX, SY, true);                             \
    else SDFR_LAUNCH_FWD_L(RT, PK, SRC, STRIDE, SX, SY, false);                                      \
  } while (0)
#define SDFR_LAUNCH_FWD(RT, PK, SRC, STRIDE)                                                         \
  do {                                                                                               \
    if (geom.sx == kFwdWideTile.sx && geom.sy == kFwdWideTile.sy) SDFR_LAUNCH_FWD_G(RT, PK, SRC, STRIDE, 4, 1);  \
    else if (macro) SDFR_LAUNCH_FWD_G(RT, PK, SRC, STRIDE, SDFR_FWD_SX, SDFR_FWD_SY);                \
    else SDFR_LAUNCH_FWD_G(RT, PK, SRC, STRIDE, 1, 1);                                               \
  } while (0)
  if (inline_setup) {
    const InlineSetup in{pos, quat, inv_scale, fx, fy, setup, g_zero, n_zero};
#define SDFR_LAUNCH_INLINE(RT, LOSS)                                                                              \
  hipLaunchKernelGGL((render_forward_kernel<RT, false, 1, 1, LOSS, 4, true>), grid_tile, dim3(256), 0, st, sdf, R, \
                     sdf_view_stride, setup, W, H, ntx, nty, cx, cy, rfx, rfy, threshold, vec_ok, depth, target,  \
                     loss_part, epoch, lay.spans, in, (unsigned*)nullptr, (unsigned long long*)nullptr)
    if (R == 64) { if (with_loss) SDFR_LAUNCH_INLINE(64, true); else SDFR_LAUNCH_INLINE(64, false); }
    else { if (with_loss) SDFR_LAUNCH_INLINE(0, true); else SDFR_LAUNCH_INLINE(0, false); }
#undef SDFR_LAUNCH_INLINE
  } else if (packed) {
    if (R == 64) SDFR_LAUNCH_FWD(64, true, cells, 0LL); else SDFR_LAUNCH_FWD(0, true, cells, 0LL);
  } else {
    if (R == 64) SDFR_LAUNCH_FWD(64, false, sdf, sdf_view_stride); else SDFR_LAUNCH_FWD(0, false, sdf, sdf_view_stride);
  }
#undef SDFR_LAUNCH_FWD
#undef SDFR_LAUNCH_FWD_G
#undef SDFR_LAUNCH_FWD_L
  if (with_loss && loss)   // (loss == NULL: a step whose backward reduces the records, LossTiles)
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(B), dim3(64), 0, st, loss_part, setup, ntx, nty, geom.w(),
                       geom.h(), loss, loss_stats);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int sdfr_render_forward(const float* sdf, int R, long long sdf_view_stride,
                                   const float* pos, const float* quat, const float* inv_scale,
                                   int B, int W, int H, float cx, float cy, float fx, float fy,
                                   float threshold, float* depth, void* workspace,
                                   size_t workspace_bytes, int device, void* stream) {
  return forward_impl("sdfr_render_forward", sdf, R, sdf_view_stride, pos, quat, inv_scale, B, W, H, cx,
                      cy, fx, fy, threshold, depth, nullptr, nullptr, nullptr, workspace,
                      workspace_bytes, sdfr_render_forward_workspace_bytes(R, B, W, H),
                      plain_forward_layout(workspace, R, B, W, H), nullptr, 0, device, stream);
}

extern "C" int sdfr_render_forward_l1(const float* sdf, int R, long long sdf_view_stride,
                                      const float* pos, const float* quat, const float* inv_scale,
                                      int B, int W, int H, float cx, float cy, float fx, float fy,
                                      float threshold, const float* target, float* depth, float* loss,
                                      float* loss_stats, void* workspace, size_t workspace_bytes,
                                      int device, void* stream) {
  if (B > 0 && W > 0 && H > 0 && (!target || !loss || !loss_stats))
    return fail(SDFR_E_NULL, "sdfr_render_forward_l1: NULL pointer argument");
  return forward_impl("sdfr_render_forward_l1", sdf, R, sdf_view_stride, pos, quat, inv_scale, B, W, H,
                      cx, cy, fx, fy, threshold, depth, target, loss, loss_stats, workspace,
                      workspace_bytes, sdfr_render_forward_l1_workspace_bytes(R, B, W, H),
                      plain_forward_layout(workspace, R, B, W, H), nullptr, 0, device, stream);
}


namespace {
// loss_stats == nullptr: the plain backward (grad_depth = upstream gradient image).  Otherwise
// grad_depth is the observed depth image and the gradient of the folded-in depth-L1 is formed in
// the kernel.
int backward_impl(const char* fn, const float* grad_depth, const float* depth, const float* sdf, int R,
                  long long sdf_view_stride, const float* pos, const float* quat,
                  const float* inv_scale, int B, int W, int H, float cx, float cy, float fx, float fy,
                  int sdf_grad_mode, float* g_sdf, long long g_sdf_view_stride, float* g_pos,
                  float* g_quat, float* g_inv_scale, const float* loss_grad, const float* loss_stats,
                  float loss_weight, void* workspace, size_t workspace_bytes, int device,
                  void* stream, const PcBackwardArgs* pc = nullptr, bool prepared = false,
                  float* loss_out = nullptr, float* stats_out = nullptr) {
  // loss_out / stats_out (a prepared, loss-fused step only): the step's forward left its (sum, count) tile records
  // unreduced (sdfr_render_step_forward_l1 with loss = loss_stats = NULL); this launch reduces them (LossTiles)
  // prepared: a step's backward (sdfr_render_step_backward) -- the views were set up and g_sdf zero-filled by
  // the step's forward, in the step layout of the workspace; pos / quat / inv_scale are not read
  const bool loss_deferred = loss_out != nullptr;
  const bool with_loss = loss_stats != nullptr || loss_deferred;
  if (loss_deferred && (!prepared || !stats_out || loss_stats))
    return fail(SDFR_E_INVALID, "%s: loss / loss_stats outputs go with a step whose forward deferred them", fn);
  if (int rc = check_common(R, B, W, H, fx, fy)) return rc;
  const long long vox = (long long)R * R * R;
  if (sdf_view_stride != 0 && sdf_view_stride < vox)
    return fail(SDFR_E_INVALID, "sdf_view_stride must be 0 or >= R^3");
  if (g_sdf_view_stride != 0 && g_sdf_view_stride != vox)
    return fail(SDFR_E_INVALID, "g_sdf_view_stride must be 0 or R^3");
  const bool det = (sdf_grad_mode & SDFR_SDF_GRAD_DETERMINISTIC) != 0;
  const bool half_hint = (sdf_grad_mode & SDFR_BWD_HALF_GRID) != 0;
  const bool small_tiles = (sdf_grad_mode & SDFR_BWD_SMALL_TILES) != 0;
  sdf_grad_mode &= ~(SDFR_SDF_GRAD_DETERMINISTIC | SDFR_BWD_HALF_GRID | SDFR_BWD_SMALL_TILES);
  if (sdf_grad_mode != SDFR_SDF_GRAD_EXACT && sdf_grad_mode != SDFR_SDF_GRAD_CUDA_COMPAT)
    return fail(SDFR_E_INVALID, "unknown sdf_grad_mode %d", sdf_grad_mode);
  if (det && (g_sdf_view_stride != 0 || R > kDetMaxR))
    return fail(SDFR_E_INVALID, "%s: SDFR_SDF_GRAD_DETERMINISTIC needs one shared gradient volume "
                "(g_sdf_view_stride = 0) and R <= %d", fn, kDetMaxR);
  if (!g_sdf) return fail(SDFR_E_NULL, "%s: g_sdf is NULL", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const size_t g_words = (size_t)vox * (g_sdf_view_stride ? (size_t)(B > 0 ? B : 1) : 1);
  if (B == 0) {
    zero_words_async(g_sdf, g_words, st);
    return 0;
  }
  // deferred: the tile partials stay in the workspace for sdfr_views_to_pose_grad_deferred (one launch less)
  const bool deferred = !g_pos && !g_quat && !g_inv_scale;
  if (deferred && (W == 0 || H == 0))
    return fail(SDFR_E_INVALID, "%s: deferred pose gradients need a non-empty image", fn);
  if ((!deferred && (!g_pos || !g_quat || !g_inv_scale)) || (!prepared && (!pos || !quat || !inv_scale)))
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (W == 0 || H == 0) {
    zero_words_async(g_sdf, g_words, st);
    zero_words_async(g_pos, (size_t)B * 3, st);
    zero_words_async(g_quat, (size_t)B * 4, st);
    zero_words_async(g_inv_scale, (size_t)B, st);
    return 0;
  }
  if (!grad_depth || !depth || !sdf || !workspace)
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  const size_t need = prepared ? sdfr_render_step_workspace_bytes(R, B, W, H)
                               : sdfr_render_backward_workspace_bytes(R, B, W, H);
  if (workspace_bytes < need)
    return fail(SDFR_E_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, workspace_bytes, need);
  if ((uintptr_t)workspace % alignof(ViewSetup))
    return fail(SDFR_E_INVALID, "workspace must be %zu-byte aligned", alignof(ViewSetup));
  ViewSetup* setup = (ViewSetup*)workspace;
  float* partials = (float*)((char*)workspace + (prepared ? step_partials_offset(R, B, H) : scratch_offset(B, H)));
  // (a stand-alone backward does not know the forward's threshold, so its rectangles are those of the full cube;
  // depth is 0 outside the forward's may-hit rectangle anyway.  A step's backward culls with the forward's own,
  // tighter rectangles and launches no prologue.)
  if (!prepared)
    hipLaunchKernelGGL(backward_prologue_kernel, dim3((unsigned)((std::max(g_words, (size_t)B) + 255) / 256)),
                       dim3(256), 0, st, g_sdf, g_words, pos, quat, inv_scale, B, R, W, H, cx, cy, fx, fy, setup);
  // deterministic mode: the image kernel adds integers into the workspace's 64-bit volume, converted at the end
  long long* fixed = nullptr;
  float* g_out = g_sdf;
  if (det) {
    fixed = (long long*)((char*)partials + partials_bytes(B, W, H));
    zero_words_async((float*)fixed, (size_t)vox * 2, st);
    g_sdf = (float*)fixed;
  }
  PcBackwardArgs pc_det;
  if (det && pc) {   // the sampler's blocks add into the same 64-bit volume
    pc_det = *pc;
    pc_det.g_sdf = (float*)fixed;
    pc = &pc_det;
  }
  // batch: `stride` workgroups per view, each view in its own tiling (common.hpp, kBwdBigTile); else 32 x 8 tiles
  // (SDFR_BWD_SMALL_TILES: always those -- a view's pose sums then do not depend on how many views share the launch)
  const TileGeom geom = small_tiles ? kSmallTile : backward_geom(B, W, H);
  const bool batch = geom.sx * geom.sy > 1;
  const int ntx = geom.nx(W), nty = geom.ny(H);
  const int stride = batch ? backward_tile_stride(W, H) : 0;
  const bool half = half_hint && batch && !det && !pc;   // (a hint: ignored where it does not apply)
  const int rows = batch ? (half ? backward_half_rows(H) : backward_batch_rows(H)) : nty;
  const int pc_groups = pc ? (pc->groups > 0 ? pc->groups : pc->nblk) : 0;   // workgroups of the sampler per view
  const int pc_rows = pc ? (pc_groups + ntx - 1) / ntx : 0;
  const dim3 grid_tile((unsigned)ntx, (unsigned)(rows + pc_rows), (unsigned)B);
  const float rfx = (float)(1.0 / (double)fx), rfy = (float)(1.0 / (double)fy);
  LossTiles lt_arg{};
  if (loss_deferred) {   // the records of the step's forward, in ITS tiling (forward_impl)
    const TileGeom fg = forward_geom(B, W, H);
    auto log2i = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };   // tiles are 32 sx x 8 sy, sx, sy powers of two
    if ((1 << log2i(fg.w())) != fg.w() || (1 << log2i(fg.h())) != fg.h())
      return fail(SDFR_E_INVALID, "%s: forward tiles of %d x %d pixels are not powers of two", fn, fg.w(), fg.h());
    lt_arg = LossTiles{(const float*)((char*)workspace + step_loss_offset(R, B, W, H)), fg.nx(W), fg.ny(H),
                       log2i(fg.w()), log2i(fg.h()), loss_out, stats_out};
  }
#define SDFR_BWD_ARGS                                                                                \
  grad_depth, depth, sdf, R, sdf_view_stride, setup, W, H, ntx, nty, stride, cx, cy, rfx, rfy,       \
      sdf_grad_mode, g_sdf, g_sdf_view_stride, partials, loss_grad, loss_stats, loss_weight
  // (culling a step's backward tiles with the forward's band spans was built and measured: the two dependent scalar
  // loads in front of every tile of the rectangle cost more than the culled tiles' depth loads, step +3.5 us)
#define SDFR_LAUNCH_BWD(RT, BATCH)                                                                   \
  do {                                                                                               \
    if (pc && det)                                                                                   \
      hipLaunchKernelGGL((render_backward_pc_kernel<RT, BATCH, true>), grid_tile, dim3(kBlock), 0, st, \
                         SDFR_BWD_ARGS, pc_rows, *pc, lt_arg);                                       \
    else if (pc)                                                                                     \
      hipLaunchKernelGGL((render_backward_pc_kernel<RT, BATCH>), grid_tile, dim3(kBlock), 0, st,     \
                         SDFR_BWD_ARGS, pc_rows, *pc, lt_arg);                                       \
    else if (with_loss && det)                                                                       \
      hipLaunchKernelGGL((render_backward_kernel<RT, BATCH, true, true>), grid_tile, dim3(kBlock), 0, st, \
                         SDFR_BWD_ARGS, lt_arg);                                              \
    else if (with_loss && half)                                                                      \
      hipLaunchKernelGGL((render_backward_kernel<RT, BATCH, true, false, BATCH>), grid_tile, dim3(kBlock), 0, st, \
                         SDFR_BWD_ARGS, lt_arg);                                              \
    else if (with_loss)                                                                              \
      hipLaunchKernelGGL((render_backward_kernel<RT, BATCH, true>), grid_tile, dim3(kBlock), 0, st,  \
                         SDFR_BWD_ARGS, lt_arg);                                              \
    else if (det)                                                                                    \
      hipLaunchKernelGGL((render_backward_kernel<RT, BATCH, false, true>), grid_tile, dim3(kBlock), 0, st, \
                         SDFR_BWD_ARGS, lt_arg);                                              \
    else if (half)                                                                                   \
      hipLaunchKernelGGL((render_backward_kernel<RT, BATCH, false, false, BATCH>), grid_tile, dim3(kBlock), 0, st, \
                         SDFR_BWD_ARGS, lt_arg);                                              \
    else                                                                                             \
      hipLaunchKernelGGL((render_backward_kernel<RT, BATCH, false>), grid_tile, dim3(kBlock), 0, st, \
                         SDFR_BWD_ARGS, lt_arg);                                              \
  } while (0)
  if (R == 64) { if (batch) SDFR_LAUNCH_BWD(64, true); else SDFR_LAUNCH_BWD(64, false); }
  else { if (batch) SDFR_LAUNCH_BWD(0, true); else SDFR_LAUNCH_BWD(0, false); }
#undef SDFR_LAUNCH_BWD
#undef SDFR_BWD_ARGS
  if (!deferred)
    hipLaunchKernelGGL(pose_reduce_kernel, dim3(B), dim3(64), 0, st, partials, setup, W, H, ntx, nty,
                       geom.w(), geom.h(), stride, g_pos, g_quat, g_inv_scale);
  if (det)
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((vox + 255) / 256)), dim3(256), 0, st, fixed,
                       (size_t)vox, g_out);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int sdfr_render_backward(const float* grad_depth, const float* depth, const float* sdf,
                                    int R, long long sdf_view_stride, const float* pos,
                                    const float* quat, const float* inv_scale, int B, int W, int H,
                                    float cx, float cy, float fx, float fy, int sdf_grad_mode,
                                    float* g_sdf, long long g_sdf_view_stride, float* g_pos,
                                    float* g_quat, float* g_inv_scale, void* workspace,
                                    size_t workspace_bytes, int device, void* stream) {
  return backward_impl("sdfr_render_backward", grad_depth, depth, sdf, R, sdf_view_stride, pos, quat,
                       inv_scale, B, W, H, cx, cy, fx, fy, sdf_grad_mode, g_sdf, g_sdf_view_stride, g_pos,
                       g_quat, g_inv_scale, nullptr, nullptr, 0.0f, workspace, workspace_bytes, device,
                       stream);
}

extern "C" int sdfr_render_step_forward(const float* sdf, int R, long long sdf_view_stride, const float* pos,
                                        const float* quat, const float* inv_scale, int B, int W, int H, float cx,
                                        float cy, float fx, float fy, float threshold, float* depth, float* g_sdf,
                                        long long g_sdf_view_stride, void* workspace, size_t workspace_bytes,
                                        int device, void* stream) {
  return sdfr_render_step_forward_counted(sdf, R, sdf_view_stride, pos, quat, inv_scale, B, W, H, cx, cy, fx, fy,
                                          threshold, depth, g_sdf, g_sdf_view_stride, workspace, workspace_bytes,
                                          nullptr, device, stream);
}

extern "C" int sdfr_render_step_forward_counted(const float* sdf, int R, long long sdf_view_stride, const float* pos,
                                                const float* quat, const float* inv_scale, int B, int W, int H,
                                                float cx, float cy, float fx, float fy, float threshold, float* depth,
                                                float* g_sdf, long long g_sdf_view_stride, void* workspace,
                                                size_t workspace_bytes, unsigned long long* close_views_word,
                                                int device, void* stream) {
  const char* fn = "sdfr_render_step_forward";
  if ((uintptr_t)close_views_word % 8) return fail(SDFR_E_INVALID, "%s: close_views_word must be 8-byte aligned", fn);
  const long long vox = (long long)R * R * R;
  if (R >= 2 && R <= 1023 && g_sdf_view_stride != 0 && g_sdf_view_stride != vox)
    return fail(SDFR_E_INVALID, "g_sdf_view_stride must be 0 or R^3");
  if (B > 0 && !g_sdf) return fail(SDFR_E_NULL, "%s: g_sdf is NULL", fn);
  ForwardLayout lay{};
  if (workspace && R >= 2 && R <= 1023 && B > 0) {
    char* w = (char*)workspace;
    lay.setup = (ViewSetup*)w;
    lay.sync = (unsigned*)(w + sync_offset(B));
    lay.spans = (unsigned*)(w + spans_offset_bytes(B));
    lay.cells = (float*)(w + scratch_offset(B, H));
    lay.loss_part = nullptr;
  }
  const size_t g_words = (R >= 2 && R <= 1023) ? (size_t)vox * (g_sdf_view_stride ? (size_t)(B > 0 ? B : 1) : 1) : 0;
  return forward_impl(fn, sdf, R, sdf_view_stride, pos, quat, inv_scale, B, W, H, cx, cy, fx, fy, threshold, depth,
                      nullptr, nullptr, nullptr, workspace, workspace_bytes,
                      sdfr_render_step_workspace_bytes(R, B, W, H), lay, g_sdf, g_words, device, stream,
                      close_views_word);
}

extern "C" int sdfr_render_step_forward_l1(const float* sdf, int R, long long sdf_view_stride, const float* pos,
                                           const float* quat, const float* inv_scale, int B, int W, int H, float cx,
                                           float cy, float fx, float fy, float threshold, const float* target,
                                           float* depth, float* loss, float* loss_stats, float* g_sdf,
                                           long long g_sdf_view_stride, void* workspace, size_t workspace_bytes,
                                           unsigned long long* close_views_word, int device, void* stream) {
  const char* fn = "sdfr_render_step_forward_l1";
  if ((uintptr_t)close_views_word % 8) return fail(SDFR_E_INVALID, "%s: close_views_word must be 8-byte aligned", fn);
  const long long vox = (long long)R * R * R;
  if (R >= 2 && R <= 1023 && g_sdf_view_stride != 0 && g_sdf_view_stride != vox)
    return fail(SDFR_E_INVALID, "g_sdf_view_stride must be 0 or R^3");
  if (B > 0 && !g_sdf) return fail(SDFR_E_NULL, "%s: g_sdf is NULL", fn);
  if (B > 0 && W > 0 && H > 0 && (!target || (!loss) != (!loss_stats)))
    return fail(SDFR_E_NULL, "%s: NULL pointer argument (loss and loss_stats: both, or neither = deferred)", fn);
  ForwardLayout lay{};
  if (workspace && R >= 2 && R <= 1023 && B > 0) {
    char* w = (char*)workspace;
    lay.setup = (ViewSetup*)w;
    lay.sync = (unsigned*)(w + sync_offset(B));
    lay.spans = (unsigned*)(w + spans_offset_bytes(B));
    lay.cells = (float*)(w + scratch_offset(B, H));
    lay.loss_part = (float*)(w + step_loss_offset(R, B, W, H));
  }
  const size_t g_words = (R >= 2 && R <= 1023) ? (size_t)vox * (g_sdf_view_stride ? (size_t)(B > 0 ? B : 1) : 1) : 0;
  return forward_impl(fn, sdf, R, sdf_view_stride, pos, quat, inv_scale, B, W, H, cx, cy, fx, fy, threshold, depth,
                      target, loss, loss_stats, workspace, workspace_bytes,
                      sdfr_render_step_workspace_bytes(R, B, W, H), lay, g_sdf, g_words, device, stream,
                      close_views_word);
}

extern "C" int sdfr_render_step_backward(const float* grad_depth, const float* depth, const float* sdf, int R,
                                         long long sdf_view_stride, int B, int W, int H, float cx, float cy,
                                         float fx, float fy, int sdf_grad_mode, float* g_sdf,
                                         long long g_sdf_view_stride, float* g_pos, float* g_quat,
                                         float* g_inv_scale, void* workspace, size_t workspace_bytes, int device,
                                         void* stream) {
  if (B == 0 || W == 0 || H == 0) {
    // nothing was rendered: g_sdf is zero already (the step's forward filled it); the pose gradients are zero
    if (int rc = check_common(R, B, W, H, fx, fy)) return rc;
    if (B > 0 && g_pos && g_quat && g_inv_scale) {
      SDFR_HIP_TRY(hipSetDevice(device));
      zero_words_async(g_pos, (size_t)B * 3, (hipStream_t)stream);
      zero_words_async(g_quat, (size_t)B * 4, (hipStream_t)stream);
      zero_words_async(g_inv_scale, (size_t)B, (hipStream_t)stream);
    }
    return 0;
  }
  return backward_impl("sdfr_render_step_backward", grad_depth, depth, sdf, R, sdf_view_stride, nullptr, nullptr,
                       nullptr, B, W, H, cx, cy, fx, fy, sdf_grad_mode, g_sdf, g_sdf_view_stride, g_pos, g_quat,
                       g_inv_scale, nullptr, nullptr, 0.0f, workspace, workspace_bytes, device, stream, nullptr,
                       true);
}

extern "C" int sdfr_render_backward_l1(const float* loss_grad, float loss_weight,
                                       const float* loss_stats, const float* target,
                                       const float* depth, const float* sdf, int R,
                                       long long sdf_view_stride, const float* pos, const float* quat,
                                       const float* inv_scale, int B, int W, int H, float cx, float cy,
                                       float fx, float fy, int sdf_grad_mode, float* g_sdf,
                                       long long g_sdf_view_stride, float* g_pos, float* g_quat,
                                       float* g_inv_scale, void* workspace, size_t workspace_bytes,
                                       int device, void* stream) {
  if (B > 0 && !loss_stats) return fail(SDFR_E_NULL, "sdfr_render_backward_l1: loss_stats is NULL");
  return backward_impl("sdfr_render_backward_l1", target, depth, sdf, R, sdf_view_stride, pos, quat,
                       inv_scale, B, W, H, cx, cy, fx, fy, sdf_grad_mode, g_sdf, g_sdf_view_stride, g_pos,
                       g_quat, g_inv_scale, loss_grad, loss_stats, loss_weight, workspace,
                       workspace_bytes, device, stream);
}

extern "C" int sdfr_render_step_backward_l1(const float* loss_grad, float loss_weight, const float* loss_stats,
                                            const float* target, const float* depth, const float* sdf, int R,
                                            long long sdf_view_stride, int B, int W, int H, float cx, float cy,
                                            float fx, float fy, int sdf_grad_mode, float* g_sdf,
                                            long long g_sdf_view_stride, float* g_pos, float* g_quat,
                                            float* g_inv_scale, void* workspace, size_t workspace_bytes, float* loss,
                                            float* loss_stats_out, int device, void* stream) {
  const char* fn = "sdfr_render_step_backward_l1";
  if (B > 0 && !loss_stats && !loss) return fail(SDFR_E_NULL, "%s: loss_stats is NULL (and no deferred outputs)", fn);
  if (B == 0 || W == 0 || H == 0) {   // nothing was rendered (as sdfr_render_step_backward)
    if (int rc = check_common(R, B, W, H, fx, fy)) return rc;
    if (B > 0 && g_pos && g_quat && g_inv_scale) {
      SDFR_HIP_TRY(hipSetDevice(device));
      zero_words_async(g_pos, (size_t)B * 3, (hipStream_t)stream);
      zero_words_async(g_quat, (size_t)B * 4, (hipStream_t)stream);
      zero_words_async(g_inv_scale, (size_t)B, (hipStream_t)stream);
    }
    return 0;
  }
  return backward_impl(fn, target, depth, sdf, R, sdf_view_stride, nullptr, nullptr, nullptr, B, W, H, cx, cy, fx, fy,
                       sdf_grad_mode, g_sdf, g_sdf_view_stride, g_pos, g_quat, g_inv_scale, loss_grad, loss_stats,
                       loss_weight, workspace, workspace_bytes, device, stream, nullptr, true, loss, loss_stats_out);
}

namespace {
int backward_l1_pc_impl(const char* fn, bool prepared,
    const float* loss_grad, float loss_weight, const float* loss_stats, const float* target, const float* depth,
    const float* sdf, int R, long long sdf_view_stride, const float* pos, const float* quat, const float* inv_scale,
    int B, int W, int H, float cx, float cy, float fx, float fy, int sdf_grad_mode, float* g_sdf,
    long long g_sdf_view_stride, void* workspace, size_t workspace_bytes, float pc_weight, const float* points,
    const int* offsets, int max_view_points, const float* scale, void* pc_workspace, size_t pc_workspace_bytes,
    int device, void* stream, float* loss_out = nullptr, float* stats_out = nullptr) {
  if (B <= 0 || W <= 0 || H <= 0 || max_view_points <= 0)
    return fail(SDFR_E_INVALID, "%s: needs B, W, H, max_view_points > 0 (B=%d W=%d H=%d points=%d)", fn, B, W, H,
                max_view_points);
  if (R > 1023) return fail(SDFR_E_INVALID, "%s: R=%d out of range", fn, R);
  if ((!loss_stats && !loss_out) || !points || !scale || !pc_workspace)
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (!offsets && B > 1) return fail(SDFR_E_NULL, "offsets may be NULL only for a single view");
  if (pc_workspace_bytes < sdfr_pc_loss_backward_workspace_bytes(B, max_view_points))
    return fail(SDFR_E_WORKSPACE, "%s: sampler workspace %zu < %zu bytes", fn, pc_workspace_bytes,
                sdfr_pc_loss_backward_workspace_bytes(B, max_view_points));
  if ((uintptr_t)pc_workspace % 16) return fail(SDFR_E_INVALID, "sampler workspace must be 16-byte aligned");
  const int nblk = (max_view_points + kSamplerPts - 1) / kSamplerPts;
  float* pc_part = (float*)pc_workspace;
  // workgroups of the sampler per view (sampler_device.hpp, GROUPS): one per block while the launch is small, but
  // never a grid that grows with the point buffers' capacity times the views -- ~8192 workgroups keep the chip busy
  // (2 048 resident at 256 threads), and at least 64 per view: a mug's ~60 blocks still run side by side
  const int groups = std::min(nblk, std::max(kSamplerMinGroups, (kSamplerGridTarget + B - 1) / B));
  const PcBackwardArgs pa{nullptr, points, offsets, max_view_points, pos, quat, scale, sdf, R, sdf_view_stride,
                          g_sdf, g_sdf_view_stride, pc_part, nblk, pc_weight, pc_part + (size_t)B * nblk * 8, groups};
  if (!pos || !quat || !inv_scale) return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  return backward_impl(fn, target, depth, sdf, R, sdf_view_stride, pos, quat, inv_scale, B, W, H, cx, cy, fx, fy,
                       sdf_grad_mode, g_sdf, g_sdf_view_stride, nullptr, nullptr, nullptr, loss_grad, loss_stats,
                       loss_weight, workspace, workspace_bytes, device, stream, &pa, prepared, loss_out, stats_out);
}
}  // namespace

extern "C" int sdfr_render_backward_l1_pc(
    const float* loss_grad, float loss_weight, const float* loss_stats, const float* target, const float* depth,
    const float* sdf, int R, long long sdf_view_stride, const float* pos, const float* quat, const float* inv_scale,
    int B, int W, int H, float cx, float cy, float fx, float fy, int sdf_grad_mode, float* g_sdf,
    long long g_sdf_view_stride, void* workspace, size_t workspace_bytes, float pc_weight, const float* points,
    const int* offsets, int max_view_points, const float* scale, void* pc_workspace, size_t pc_workspace_bytes,
    int device, void* stream) {
  return backward_l1_pc_impl("sdfr_render_backward_l1_pc", false, loss_grad, loss_weight, loss_stats, target, depth, sdf,
                             R, sdf_view_stride, pos, quat, inv_scale, B, W, H, cx, cy, fx, fy, sdf_grad_mode, g_sdf,
                             g_sdf_view_stride, workspace, workspace_bytes, pc_weight, points, offsets, max_view_points,
                             scale, pc_workspace, pc_workspace_bytes, device, stream);
}

extern "C" int sdfr_render_step_backward_l1_pc(
    const float* loss_grad, float loss_weight, const float* loss_stats, const float* target, const float* depth,
    const float* sdf, int R, long long sdf_view_stride, const float* pos, const float* quat, const float* inv_scale,
    int B, int W, int H, float cx, float cy, float fx, float fy, int sdf_grad_mode, float* g_sdf,
    long long g_sdf_view_stride, void* workspace, size_t workspace_bytes, float pc_weight, const float* points,
    const int* offsets, int max_view_points, const float* scale, void* pc_workspace, size_t pc_workspace_bytes,
    float* loss, float* loss_stats_out, int device, void* stream) {
  return backward_l1_pc_impl("sdfr_render_step_backward_l1_pc", true, loss_grad, loss_weight, loss_stats, target, depth,
                             sdf, R, sdf_view_stride, pos, quat, inv_scale, B, W, H, cx, cy, fx, fy, sdf_grad_mode, g_sdf,
                             g_sdf_view_stride, workspace, workspace_bytes, pc_weight, points, offsets, max_view_points,
                             scale, pc_workspace, pc_workspace_bytes, device, stream, loss, loss_stats_out);
}

extern "C" size_t sdfr_render_fused_view_count_offset(int B, int H) { return scratch_offset(B, H); }
extern "C" size_t sdfr_render_fused_tile_loss_offset(int R, int B, int W, int H) {
  return (R < 2 || B <= 0) ? 0 : step_loss_offset(R, B, W, H);
}

extern "C" int sdfr_render_step_fused_l1_pc(
    const float* sdf, int R, long long sdf_view_stride, const float* pos, const float* quat, const float* inv_scale,
    const float* scale, int B, int W, int H, float cx, float cy, float fx, float fy, float threshold,
    const float* target, float* depth, int sdf_grad_mode, float* g_sdf, float* g_depth, void* workspace,
    size_t workspace_bytes, float pc_weight, const float* points, const int* offsets, int max_view_points,
    void* pc_workspace, size_t pc_workspace_bytes, int device, void* stream) {
  const char* fn = "sdfr_render_step_fused_l1_pc";
  if (int rc = check_common(R, B, W, H, fx, fy)) return rc;
  if (B <= 0 || B > SDFR_FUSED_MAX_VIEWS || W <= 0 || H <= 0 || max_view_points <= 0)
    return fail(SDFR_E_INVALID, "%s: 1 .. %d views of a non-empty image with observed points (B=%d W=%d H=%d points=%d)",
                fn, SDFR_FUSED_MAX_VIEWS, B, W, H, max_view_points);
  if ((g_sdf != nullptr) != (g_depth != nullptr))
    return fail(SDFR_E_NULL, "%s: g_sdf and g_depth go together (both NULL: nobody wants d/dSDF)", fn);
  if (cells_bytes(R) == 0)   // (the workspace region the counts live in)
    return fail(SDFR_E_INVALID, "%s: grids up to R = %d", fn, kPackedMaxR);
  if (sdf_grad_mode != SDFR_SDF_GRAD_EXACT && sdf_grad_mode != SDFR_SDF_GRAD_CUDA_COMPAT)
    return fail(SDFR_E_INVALID, "%s: sdf_grad_mode %d (the weights only: no flags in the one-launch form)", fn,
                sdf_grad_mode);
  if (sdf_view_stride != 0 && sdf_view_stride < (long long)R * R * R)
    return fail(SDFR_E_INVALID, "sdf_view_stride must be 0 or >= R^3");
  if (!sdf || !pos || !quat || !inv_scale || !scale || !target || !depth || !workspace || !points || !pc_workspace)
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (!offsets && B > 1) return fail(SDFR_E_NULL, "offsets may be NULL only for a single view");
  if (workspace_bytes < sdfr_render_step_workspace_bytes(R, B, W, H))
    return fail(SDFR_E_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, workspace_bytes,
                sdfr_render_step_workspace_bytes(R, B, W, H));
  if ((uintptr_t)workspace % alignof(ViewSetup))
    return fail(SDFR_E_INVALID, "workspace must be %zu-byte aligned", alignof(ViewSetup));
  if (pc_workspace_bytes < sdfr_pc_loss_backward_workspace_bytes(B, max_view_points))
    return fail(SDFR_E_WORKSPACE, "%s: sampler workspace %zu < %zu bytes", fn, pc_workspace_bytes,
                sdfr_pc_loss_backward_workspace_bytes(B, max_view_points));
  if ((uintptr_t)pc_workspace % 16) return fail(SDFR_E_INVALID, "sampler workspace must be 16-byte aligned");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)workspace;
  ViewSetup* setup = (ViewSetup*)w;
  float* partials = (float*)(w + step_partials_offset(R, B, H));
  float* tile_loss = (float*)(w + step_loss_offset(R, B, W, H));
  float* view_cnt = (float*)(w + scratch_offset(B, H));                   // (the face records' place: the plain grid is marched)
  const int nblk = (max_view_points + kSamplerPts - 1) / kSamplerPts;
  float* pc_part = (float*)pc_workspace;
  const int groups = std::min(nblk, std::max(kSamplerMinGroups, (kSamplerGridTarget + B - 1) / B));
  // (pose only: SDFG = false, neither the tiles nor the sampler's blocks touch a volume)
  const PcBackwardArgs pa{nullptr, points, offsets, max_view_points, pos, quat, scale, sdf, R, sdf_view_stride,
                          g_sdf, 0, pc_part, nblk, pc_weight, pc_part + (size_t)B * nblk * 8, groups};
  const int ntx = kSmallTile.nx(W), nty = kSmallTile.ny(H);
  const int pc_rows = (groups + ntx - 1) / ntx;
  const dim3 grid((unsigned)ntx, (unsigned)(nty + pc_rows), (unsigned)B);
  const float rfx = (float)(1.0 / (double)fx), rfy = (float)(1.0 / (double)fy);
  const int vec_ok = (W % 4 == 0) && ((uintptr_t)depth % 16 == 0);
  const InlineSetup in{pos, quat, inv_scale, fx, fy, setup, nullptr, 0};
#define SDFR_LAUNCH_FUSED(RT, SDFG, PCD)                                                                         \
  hipLaunchKernelGGL((render_fused_l1_pc_kernel<RT, SDFG, PCD>), grid, dim3(kBlock), 0, st, sdf, R, sdf_view_stride, W, H,  \
                     ntx, nty, cx, cy, rfx, rfy, threshold, vec_ok, depth, target, in, sdf_grad_mode, g_depth,  \
                     partials, tile_loss, view_cnt, pc_rows, pa)
  // (without d/dSDF the sampler's blocks send no atomics at all: their direct form is simply the shorter one)
  const bool pcd = !g_sdf || B <= 2;
#define SDFR_LAUNCH_FUSED_R(RT)                                                                                  \
  do {                                                                                                           \
    if (!g_sdf) SDFR_LAUNCH_FUSED(RT, false, true);                                                              \
    else if (pcd) SDFR_LAUNCH_FUSED(RT, true, true);                                                             \
    else SDFR_LAUNCH_FUSED(RT, true, false);                                                                     \
  } while (0)
  if (R == 64) SDFR_LAUNCH_FUSED_R(64); else SDFR_LAUNCH_FUSED_R(0);
#undef SDFR_LAUNCH_FUSED_R
#undef SDFR_LAUNCH_FUSED
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
