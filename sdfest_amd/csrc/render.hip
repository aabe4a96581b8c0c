// render.hip -- sphere-tracing depth render of an SDF grid, forward + analytic backward,
// hand-written for MI355X (gfx950, wave64).
//
// What it computes is fixed by the reference (one image per call there, a batch of B here):
//   forward   sdfest/differentiable_renderer/csrc/sdf_renderer_cuda.cu:241-298
//   backward  sdfest/differentiable_renderer/csrc/sdf_renderer_cuda.cu:300-468
// How it computes it is not a translation:
//   * a set-up kernel turns each pose into a 128-byte record (rotation, object-frame camera
//     position, grid-space ray origin, conservative screen rectangle of the bounding cube);
//     the image kernels read it through the scalar cache instead of re-deriving it per pixel;
//   * tiles outside the rectangle only store zeros (forward) or exit without touching memory
//     (backward): in realistic scenes that is most of the frame;
//   * the slab test and the march run in the object frame / in grid coordinates, so one step
//     is 3 FMAs + floor/clamp instead of the reference's scale-normalise-index-denormalise chain;
//   * a wave is an 8x8 pixel patch (neighbouring lanes gather neighbouring voxels, which the
//     64x64x64 grid (1 MiB) serves from L1/L2 -- it cannot live in the 160 KiB LDS);
//   * the 8 pose-gradient sums go wave-shuffle -> LDS -> one 32-byte partial per tile -> a fixed-
//     order reduction kernel (bitwise reproducible); the reference issues 8 same-address float
//     atomics per hit pixel (sdf_renderer_cuda.cu:459-466);
//   * d/dsdf contributions of a tile are pre-summed in an LDS brick around the tile's surface
//     patch before one global float atomic per touched voxel.
#include "common.hpp"

namespace sdfr {
namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// set-up: one thread per view
// ---------------------------------------------------------------------------------------------
__global__ void view_setup_kernel(const float* __restrict__ pos, const float* __restrict__ quat,
                                  const float* __restrict__ inv_scale, int B, int R, int W, int H,
                                  float cx, float cy, float fx, float fy,
                                  ViewSetup* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float x = quat[4 * b], y = quat[4 * b + 1], z = quat[4 * b + 2], w = quat[4 * b + 3];
  const V3 p = mk(pos[3 * b], pos[3 * b + 1], pos[3 * b + 2]);
  const float isc = inv_scale[b];
  const float scale = 1.0f / isc;
  ViewSetup s;
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
  // Conservative screen rectangle of the cube |o_i| <= scale.  A ray can only pass the slab
  // test if its pixel centre lies inside the projection of the cube, which (cube entirely in
  // front of the camera) lies inside the bounding rectangle of the 8 projected corners.
  float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
  bool in_front = true;
  for (int c = 0; c < 8; ++c) {
    const float sx = (c & 1) ? scale : -scale, sy = (c & 2) ? scale : -scale,
                sz = (c & 4) ? scale : -scale;
    const float X = p.x + s.rot[0] * sx + s.rot[1] * sy + s.rot[2] * sz;
    const float Y = p.y + s.rot[3] * sx + s.rot[4] * sy + s.rot[5] * sz;
    const float Z = p.z + s.rot[6] * sx + s.rot[7] * sy + s.rot[8] * sz;
    if (!(Z < -1e-6f)) in_front = false;
    const float iz = 1.0f / fmaxf(-Z, 1e-30f);
    const float u = cx + fx * X * iz;
    const float v = cy - fy * Y * iz;
    umin = fminf(umin, u); umax = fmaxf(umax, u);
    vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
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
  s.rect[0] = x0; s.rect[1] = y0; s.rect[2] = x1; s.rect[3] = y1;
  s.pad[0] = s.pad[1] = s.pad[2] = s.pad[3] = 0.0f;
  out[b] = s;
}

// ---------------------------------------------------------------------------------------------
// shared device pieces
// ---------------------------------------------------------------------------------------------
struct Pixel {
  int row, col;
  bool inside;  // inside the image
};

// 32x8 tile, 4 waves, each wave an 8x8 patch
__device__ __forceinline__ Pixel tile_pixel(int tile_x, int tile_y, int W, int H) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  Pixel px;
  px.col = tile_x * kTileW + wave * 8 + (lane & 7);
  px.row = tile_y * kTileH + (lane >> 3);
  px.inside = (px.col < W) && (px.row < H);
  return px;
}

// unit ray through a pixel centre (camera frame, OpenGL).  sdf_renderer_cuda.cu:137-154.
__device__ __forceinline__ V3 pixel_ray(int row, int col, float cx, float cy, float rfx, float rfy) {
  const float dx = ((float)col + 0.5f - cx) * rfx;
  const float dy = -((float)row + 0.5f - cy) * rfy;
  const float inv_len = __builtin_amdgcn_rsqf(fmaf(dx, dx, fmaf(dy, dy, 1.0f)));
  return mk(dx * inv_len, dy * inv_len, -inv_len);
}

__device__ __forceinline__ V3 rot_t(const ViewSetup& s, V3 v) {  // R^T v
  return mk(fmaf(s.rot[0], v.x, fmaf(s.rot[3], v.y, s.rot[6] * v.z)),
            fmaf(s.rot[1], v.x, fmaf(s.rot[4], v.y, s.rot[7] * v.z)),
            fmaf(s.rot[2], v.x, fmaf(s.rot[5], v.y, s.rot[8] * v.z)));
}
__device__ __forceinline__ V3 rot_f(const ViewSetup& s, V3 v) {  // R v
  return mk(fmaf(s.rot[0], v.x, fmaf(s.rot[1], v.y, s.rot[2] * v.z)),
            fmaf(s.rot[3], v.x, fmaf(s.rot[4], v.y, s.rot[5] * v.z)),
            fmaf(s.rot[6], v.x, fmaf(s.rot[7], v.y, s.rot[8] * v.z)));
}

// z-adjacent corner pair: 8-byte load at 4-byte alignment (global memory allows it)
struct __attribute__((packed, aligned(4))) ZPair {
  float lo, hi;
};

struct Cell {
  float v[8];  // corner values, index 4*ix + 2*iy + iz
  float ox, oy, oz;
  int lin;  // linear index of corner 000
};

// Locate the cell of a grid-space point and gather its 8 corners (4 z-pair loads).
// Same cell choice and the same un-clamped (extrapolating) cell coordinate as
// sdf_renderer_cuda.cu:196-239.
template <int RT>
__device__ __forceinline__ void gather_cell(const float* __restrict__ sdf, int R, float gx,
                                            float gy, float gz, Cell& c) {
  const int Rr = RT > 0 ? RT : R;
  const float top = (float)(Rr - 2);
  const float bx = fminf(fmaxf(floorf(gx), 0.0f), top);
  const float by = fminf(fmaxf(floorf(gy), 0.0f), top);
  const float bz = fminf(fmaxf(floorf(gz), 0.0f), top);
  c.ox = gx - bx; c.oy = gy - by; c.oz = gz - bz;
  const int lin = ((int)bx * Rr + (int)by) * Rr + (int)bz;
  c.lin = lin;
  const float* base = sdf + lin;
  const ZPair p00 = *reinterpret_cast<const ZPair*>(base);
  const ZPair p01 = *reinterpret_cast<const ZPair*>(base + Rr);
  const ZPair p10 = *reinterpret_cast<const ZPair*>(base + Rr * Rr);
  const ZPair p11 = *reinterpret_cast<const ZPair*>(base + Rr * Rr + Rr);
  c.v[0] = p00.lo; c.v[1] = p00.hi; c.v[2] = p01.lo; c.v[3] = p01.hi;
  c.v[4] = p10.lo; c.v[5] = p10.hi; c.v[6] = p11.lo; c.v[7] = p11.hi;
}

// trilinear value, lerp order x, y, z (sdf_renderer_cuda.cu:231-238)
__device__ __forceinline__ float trilerp(const Cell& c) {
  const float ax = 1.0f - c.ox, ay = 1.0f - c.oy, az = 1.0f - c.oz;
  const float c00 = fmaf(c.v[4], c.ox, c.v[0] * ax);
  const float c01 = fmaf(c.v[5], c.ox, c.v[1] * ax);
  const float c10 = fmaf(c.v[6], c.ox, c.v[2] * ax);
  const float c11 = fmaf(c.v[7], c.ox, c.v[3] * ax);
  const float c0 = fmaf(c10, c.oy, c00 * ay);
  const float c1 = fmaf(c11, c.oy, c01 * ay);
  return fmaf(c1, c.oz, c0 * az);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(kBlock) void render_forward_kernel(
    const float* __restrict__ sdf, int R, long long sdf_view_stride,
    const ViewSetup* __restrict__ setup, int W, int H, int ntx, int nty, float cx, float cy,
    float rfx, float rfy, float threshold, float* __restrict__ depth) {
  const int tiles_per_view = ntx * nty;
  const int b = blockIdx.x / tiles_per_view;
  const int t_in_view = blockIdx.x - b * tiles_per_view;
  const int tile_y = t_in_view / ntx, tile_x = t_in_view - tile_y * ntx;
  const ViewSetup& s = setup[b];
  const Pixel px = tile_pixel(tile_x, tile_y, W, H);
  float* out = depth + ((size_t)b * H + px.row) * W + px.col;

  // tile against the cube's screen rectangle (wave-uniform)
  const int x0 = s.rect[0], y0 = s.rect[1], x1 = s.rect[2], y1 = s.rect[3];
  const bool tile_live = (tile_x * kTileW < x1) && (tile_x * kTileW + kTileW > x0) &&
                         (tile_y * kTileH < y1) && (tile_y * kTileH + kTileH > y0);
  if (!tile_live) {
    if (px.inside) *out = 0.0f;
    return;
  }
  float result = 0.0f;
  const bool in_rect = px.inside && px.col >= x0 && px.col < x1 && px.row >= y0 && px.row < y1;
  if (in_rect) {
    const V3 d = pixel_ray(px.row, px.col, cx, cy, rfx, rfy);
    const V3 dobj = rot_t(s, d);
    // slab test in the object frame: axis i of the cube is e_i, the cube centre is at +e
    // relative to the ray origin, f_i = dobj_i.  sdf_renderer_cuda.cu:156-194.
    const float scale = s.scale;
    float t_near = -1e-10f, t_far = 1e10f;
    bool hit_box = true;
    const float dv[3] = {dobj.x, dobj.y, dobj.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float e = s.e[a], f = dv[a];
      if (fabsf(f) > 1e-20f) {
        const float inv = __builtin_amdgcn_rcpf(f);
        const float ta = (e + scale) * inv, tb = (e - scale) * inv;
        t_near = fmaxf(t_near, fminf(ta, tb));
        t_far = fminf(t_far, fmaxf(ta, tb));
      } else if (-e > scale || -e < -scale) {
        hit_box = false;
      }
    }
    hit_box = hit_box && !(t_near > t_far) && !(t_far < 0.0f);
    if (hit_box) {
      const float* vol = sdf + (size_t)b * sdf_view_stride;
      const float k = s.isc * (0.5f * (float)((RT > 0 ? RT : R) - 1));
      const float dgx = dobj.x * k, dgy = dobj.y * k, dgz = dobj.z * k;
      const float ogx = s.og[0], ogy = s.og[1], ogz = s.og[2];
      float t = fmaxf(t_near, 0.0f);
      int n = 0;
      while (t < t_far && n < SDFR_MAX_MARCH_STEPS) {
        Cell c;
        gather_cell<RT>(vol, R, fmaf(t, dgx, ogx), fmaf(t, dgy, ogy), fmaf(t, dgz, ogz), c);
        const float dist = trilerp(c) * scale;
        if (dist < threshold * t) {
          result = -t * d.z;
          break;
        }
        t += dist;
        ++n;
      }
    }
  }
  if (px.inside) *out = result;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

constexpr int kBrick = 4096;  // floats of LDS used to pre-sum a tile's d/dsdf contributions

template <int RT>
__global__ __launch_bounds__(kBlock) void render_backward_kernel(
    const float* __restrict__ grad_depth, const float* __restrict__ depth,
    const float* __restrict__ sdf, int R, long long sdf_view_stride,
    const ViewSetup* __restrict__ setup, int W, int H, int ntx, int nty, float cx, float cy,
    float rfx, float rfy, int sdf_grad_mode, float* __restrict__ g_sdf,
    long long g_sdf_view_stride, float* __restrict__ partials) {
  __shared__ float brick[kBrick];
  __shared__ float wave_part[4][8];
  __shared__ int box_lo[3], box_hi[3];

  const int Rr = RT > 0 ? RT : R;
  const int tiles_per_view = ntx * nty;
  const int b = blockIdx.x / tiles_per_view;
  const int t_in_view = blockIdx.x - b * tiles_per_view;
  const int tile_y = t_in_view / ntx, tile_x = t_in_view - tile_y * ntx;
  const ViewSetup& s = setup[b];
  const int x0 = s.rect[0], y0 = s.rect[1], x1 = s.rect[2], y1 = s.rect[3];
  const bool tile_live = (tile_x * kTileW < x1) && (tile_x * kTileW + kTileW > x0) &&
                         (tile_y * kTileH < y1) && (tile_y * kTileH + kTileH > y0);
  if (!tile_live) return;  // depth is 0 there by construction of the forward: nothing to do

  const int tid = threadIdx.x;
  const Pixel px = tile_pixel(tile_x, tile_y, W, H);
  const size_t pix = ((size_t)b * H + px.row) * W + px.col;
  float z = 0.0f, go = 0.0f;
  if (px.inside) {
    z = depth[pix];
    go = grad_depth[pix];
  }
  const bool hit = z != 0.0f;
  float* part = partials + ((size_t)b * tiles_per_view + t_in_view) * 8;
  if (!__syncthreads_or(hit)) {
    if (tid < 8) part[tid] = 0.0f;
    return;
  }
  if (tid < 3) {
    box_lo[tid] = 1 << 30;
    box_hi[tid] = -1;
  }
  __syncthreads();

  float dz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float wgt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int bx = 0, by = 0, bz = 0;
  if (hit) {
    const V3 d = pixel_ray(px.row, px.col, cx, cy, rfx, rfy);
    const V3 dobj = rot_t(s, d);
    const float h = 0.5f * (float)(Rr - 1);
    const float scale = s.scale, isc = s.isc;
    const float t = z * __builtin_amdgcn_rcpf(-d.z);  // -z / d.z   (cu:339)
    const V3 o = mk(fmaf(t, dobj.x, -s.e[0]), fmaf(t, dobj.y, -s.e[1]), fmaf(t, dobj.z, -s.e[2]));
    const float* vol = sdf + (size_t)b * sdf_view_stride;
    Cell c;
    gather_cell<RT>(vol, R, fmaf(o.x * isc, h, h), fmaf(o.y * isc, h, h), fmaf(o.z * isc, h, h), c);
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
    const float f = scale * adz;      // cu:372
    const float sg = isc * h;         // s = inv_scale / grid_size (cu:391)
    const float kf = f * sg * go;     // common factor of the pose terms, times upstream grad
    // position: dc/dp_j = -s * R[j][:]
    const V3 RG = rot_f(s, G);
    dz[0] = -kf * RG.x; dz[1] = -kf * RG.y; dz[2] = -kf * RG.z;
    // quaternion: dc/dq_k = s * (d/dq_k[Rhom^T v] - 2 q_k o), v = t d - p
    const V3 v = mk(fmaf(t, d.x, -s.p[0]), fmaf(t, d.y, -s.p[1]), fmaf(t, d.z, -s.p[2]));
    const V3 u = mk(s.q[0], s.q[1], s.q[2]);
    const float w = s.q[3];
    const float udv = dot(u, v);
    const V3 uxv = cross(u, v);
    const float Gv = dot(G, v), Go = dot(G, o), Gu = dot(G, u);
    // G . (e_k x v) = (v x G)_k
    const V3 vxG = cross(v, G);
    // d/du_k: -2 u_k v + 2 e_k (u.v) + 2 u v_k - 2 w (e_k x v) - 2 u_k o
    dz[3] = kf * 2.0f * (-u.x * Gv + udv * G.x + v.x * Gu - w * vxG.x - u.x * Go);
    dz[4] = kf * 2.0f * (-u.y * Gv + udv * G.y + v.y * Gu - w * vxG.y - u.y * Go);
    dz[5] = kf * 2.0f * (-u.z * Gv + udv * G.z + v.z * Gu - w * vxG.z - u.z * Go);
    // d/dw: 2 w v - 2 (u x v) - 2 w o
    dz[6] = kf * 2.0f * (w * Gv - dot(G, uxv) - w * Go);
    // inverse scale: dc/ds^-1 = o / g, plus the product rule on scale (cu:439, :457)
    dz[7] = go * (f * h * Go - tri * scale * scale * adz);

    const float gf = go * f;
    const float x1w = c.ox, y1w = c.oy, z1w = c.oz;
    if (sdf_grad_mode == SDFR_SDF_GRAD_EXACT) {
      wgt[0] = ax * ay * az;  wgt[1] = ax * ay * z1w;  wgt[2] = ax * y1w * az;  wgt[3] = ax * y1w * z1w;
      wgt[4] = x1w * ay * az; wgt[5] = x1w * ay * z1w; wgt[6] = x1w * y1w * az; wgt[7] = x1w * y1w * z1w;
    } else {  // the weights the CUDA kernel really adds (cu:373-388)
      wgt[0] = ax * ay * z1w;  wgt[1] = ax * y1w * az;  wgt[2] = ax * y1w * z1w; wgt[3] = x1w * ay * az;
      wgt[4] = x1w * ay * z1w; wgt[5] = x1w * ay * z1w; wgt[6] = x1w * y1w * az; wgt[7] = x1w * y1w * z1w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) wgt[k] *= gf;
    bz = c.lin % Rr;
    by = (c.lin / Rr) % Rr;
    bx = c.lin / (Rr * Rr);
    atomicMin(&box_lo[0], bx); atomicMax(&box_hi[0], bx);
    atomicMin(&box_lo[1], by); atomicMax(&box_hi[1], by);
    atomicMin(&box_lo[2], bz); atomicMax(&box_hi[2], bz);
  }

  // pose sums: wave shuffle -> LDS -> tile partial
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float sk = wave_sum(dz[k]);
    if (lane == 0) wave_part[wave][k] = sk;
  }
  __syncthreads();
  if (tid < 8) part[tid] = (wave_part[0][tid] + wave_part[1][tid]) + (wave_part[2][tid] + wave_part[3][tid]);

  // d/dsdf: pre-sum the tile's contributions in an LDS brick spanning the voxel bounding box of
  // its hit cells (a surface patch is compact), then one global atomic per touched voxel.
  float* gvol = g_sdf + (size_t)b * g_sdf_view_stride;
  const int lx = box_lo[0], ly = box_lo[1], lz = box_lo[2];
  const int nx = box_hi[0] - lx + 2, ny = box_hi[1] - ly + 2, nz = box_hi[2] - lz + 2;
  const int vol_n = nx * ny * nz;
  if (vol_n <= kBrick) {
    for (int i = tid; i < vol_n; i += kBlock) brick[i] = 0.0f;
    __syncthreads();
    if (hit) {
      const int o0 = ((bx - lx) * ny + (by - ly)) * nz + (bz - lz);
      atomicAdd(&brick[o0], wgt[0]);                atomicAdd(&brick[o0 + 1], wgt[1]);
      atomicAdd(&brick[o0 + nz], wgt[2]);           atomicAdd(&brick[o0 + nz + 1], wgt[3]);
      atomicAdd(&brick[o0 + ny * nz], wgt[4]);      atomicAdd(&brick[o0 + ny * nz + 1], wgt[5]);
      atomicAdd(&brick[o0 + ny * nz + nz], wgt[6]); atomicAdd(&brick[o0 + ny * nz + nz + 1], wgt[7]);
    }
    __syncthreads();
    for (int i = tid; i < vol_n; i += kBlock) {
      const float val = brick[i];
      if (val != 0.0f) {
        const int iz = i % nz, iy = (i / nz) % ny, ix = i / (nz * ny);
        atomicAdd(&gvol[((size_t)(lx + ix) * Rr + (ly + iy)) * Rr + (lz + iz)], val);
      }
    }
  } else if (hit) {  // patch too spread out for the brick (rare): straight to global
    float* g0 = gvol + ((size_t)bx * Rr + by) * Rr + bz;
    atomicAdd(g0, wgt[0]);                atomicAdd(g0 + 1, wgt[1]);
    atomicAdd(g0 + Rr, wgt[2]);           atomicAdd(g0 + Rr + 1, wgt[3]);
    atomicAdd(g0 + Rr * Rr, wgt[4]);      atomicAdd(g0 + Rr * Rr + 1, wgt[5]);
    atomicAdd(g0 + Rr * Rr + Rr, wgt[6]); atomicAdd(g0 + Rr * Rr + Rr + 1, wgt[7]);
  }
}

// Fixed-order sum of a view's tile partials: one wave per view.
__global__ __launch_bounds__(64) void pose_reduce_kernel(const float* __restrict__ partials,
                                                         const ViewSetup* __restrict__ setup,
                                                         int ntx, int nty,
                                                         float* __restrict__ g_pos,
                                                         float* __restrict__ g_quat,
                                                         float* __restrict__ g_inv_scale) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const ViewSetup& s = setup[b];
  // live tile range of this view (same predicate as the image kernels)
  const int x0 = s.rect[0], y0 = s.rect[1], x1 = s.rect[2], y1 = s.rect[3];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (x1 > x0 && y1 > y0) {
    const int tx0 = x0 / kTileW, tx1 = (x1 - 1) / kTileW, ty0 = y0 / kTileH, ty1 = (y1 - 1) / kTileH;
    const int nx = tx1 - tx0 + 1, n = nx * (ty1 - ty0 + 1);
    const float* base = partials + (size_t)b * ntx * nty * 8;
    for (int i = lane; i < n; i += 64) {
      const int ty = ty0 + i / nx, tx = tx0 + i % nx;
      const float4* p = reinterpret_cast<const float4*>(base + ((size_t)ty * ntx + tx) * 8);
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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int check_common(int R, int B, int W, int H, float fx, float fy) {
  if (R < 2 || R > 1024) return fail(SDFR_E_INVALID, "R=%d out of range [2,1024]", R);
  if (B < 0 || W < 0 || H < 0) return fail(SDFR_E_INVALID, "negative size B=%d W=%d H=%d", B, W, H);
  if (!(fx != 0.0f) || !(fy != 0.0f)) return fail(SDFR_E_INVALID, "focal length must be non-zero");
  const long long tiles = (long long)tiles_x(W) * tiles_y(H) * (long long)B;
  if (tiles > 0x7fffffffLL) return fail(SDFR_E_INVALID, "B*tiles = %lld exceeds the grid limit", tiles);
  return 0;
}

size_t setup_bytes(int B) { return (size_t)(B > 0 ? B : 0) * sizeof(ViewSetup); }

}  // namespace
}  // namespace sdfr

using namespace sdfr;

extern "C" size_t sdfr_render_forward_workspace_bytes(int B, int W, int H) {
  (void)W; (void)H;
  return setup_bytes(B);
}

extern "C" size_t sdfr_render_backward_workspace_bytes(int B, int W, int H) {
  if (B <= 0 || W <= 0 || H <= 0) return setup_bytes(B);
  return setup_bytes(B) + (size_t)B * tiles_x(W) * tiles_y(H) * 8 * sizeof(float);
}

extern "C" int sdfr_render_forward(const float* sdf, int R, long long sdf_view_stride,
                                   const float* pos, const float* quat, const float* inv_scale,
                                   int B, int W, int H, float cx, float cy, float fx, float fy,
                                   float threshold, float* depth, void* workspace,
                                   size_t workspace_bytes, int device, void* stream) {
  if (int rc = check_common(R, B, W, H, fx, fy)) return rc;
  if (sdf_view_stride != 0 && sdf_view_stride < (long long)R * R * R)
    return fail(SDFR_E_INVALID, "sdf_view_stride must be 0 or >= R^3");
  if (B == 0 || W == 0 || H == 0) return 0;
  if (!sdf || !pos || !quat || !inv_scale || !depth || !workspace)
    return fail(SDFR_E_NULL, "sdfr_render_forward: NULL pointer argument");
  if (workspace_bytes < sdfr_render_forward_workspace_bytes(B, W, H))
    return fail(SDFR_E_WORKSPACE, "sdfr_render_forward: workspace %zu < %zu bytes", workspace_bytes,
                sdfr_render_forward_workspace_bytes(B, W, H));
  if ((uintptr_t)workspace % alignof(ViewSetup))
    return fail(SDFR_E_INVALID, "workspace must be %zu-byte aligned", alignof(ViewSetup));
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  ViewSetup* setup = (ViewSetup*)workspace;
  hipLaunchKernelGGL(view_setup_kernel, dim3((B + 63) / 64), dim3(64), 0, st, pos, quat, inv_scale,
                     B, R, W, H, cx, cy, fx, fy, setup);
  const int ntx = tiles_x(W), nty = tiles_y(H);
  const dim3 grid((unsigned)(ntx * nty * B));
  const float rfx = (float)(1.0 / (double)fx), rfy = (float)(1.0 / (double)fy);
  if (R == 64)
    hipLaunchKernelGGL(render_forward_kernel<64>, grid, dim3(kBlock), 0, st, sdf, R, sdf_view_stride,
                       setup, W, H, ntx, nty, cx, cy, rfx, rfy, threshold, depth);
  else
    hipLaunchKernelGGL(render_forward_kernel<0>, grid, dim3(kBlock), 0, st, sdf, R, sdf_view_stride,
                       setup, W, H, ntx, nty, cx, cy, rfx, rfy, threshold, depth);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_render_backward(const float* grad_depth, const float* depth, const float* sdf,
                                    int R, long long sdf_view_stride, const float* pos,
                                    const float* quat, const float* inv_scale, int B, int W, int H,
                                    float cx, float cy, float fx, float fy, int sdf_grad_mode,
                                    float* g_sdf, long long g_sdf_view_stride, float* g_pos,
                                    float* g_quat, float* g_inv_scale, void* workspace,
                                    size_t workspace_bytes, int device, void* stream) {
  if (int rc = check_common(R, B, W, H, fx, fy)) return rc;
  const long long vox = (long long)R * R * R;
  if (sdf_view_stride != 0 && sdf_view_stride < vox)
    return fail(SDFR_E_INVALID, "sdf_view_stride must be 0 or >= R^3");
  if (g_sdf_view_stride != 0 && g_sdf_view_stride != vox)
    return fail(SDFR_E_INVALID, "g_sdf_view_stride must be 0 or R^3");
  if (sdf_grad_mode != SDFR_SDF_GRAD_EXACT && sdf_grad_mode != SDFR_SDF_GRAD_CUDA_COMPAT)
    return fail(SDFR_E_INVALID, "unknown sdf_grad_mode %d", sdf_grad_mode);
  if (!g_sdf) return fail(SDFR_E_NULL, "sdfr_render_backward: g_sdf is NULL");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const size_t g_bytes = (size_t)vox * sizeof(float) * (g_sdf_view_stride ? (size_t)(B > 0 ? B : 1) : 1);
  SDFR_HIP_TRY(hipMemsetAsync(g_sdf, 0, g_bytes, st));
  if (B == 0) return 0;
  if (!g_pos || !g_quat || !g_inv_scale || !pos || !quat || !inv_scale)
    return fail(SDFR_E_NULL, "sdfr_render_backward: NULL pointer argument");
  if (W == 0 || H == 0) {
    SDFR_HIP_TRY(hipMemsetAsync(g_pos, 0, (size_t)B * 3 * sizeof(float), st));
    SDFR_HIP_TRY(hipMemsetAsync(g_quat, 0, (size_t)B * 4 * sizeof(float), st));
    SDFR_HIP_TRY(hipMemsetAsync(g_inv_scale, 0, (size_t)B * sizeof(float), st));
    return 0;
  }
  if (!grad_depth || !depth || !sdf || !workspace)
    return fail(SDFR_E_NULL, "sdfr_render_backward: NULL pointer argument");
  if (workspace_bytes < sdfr_render_backward_workspace_bytes(B, W, H))
    return fail(SDFR_E_WORKSPACE, "sdfr_render_backward: workspace %zu < %zu bytes", workspace_bytes,
                sdfr_render_backward_workspace_bytes(B, W, H));
  if ((uintptr_t)workspace % alignof(ViewSetup))
    return fail(SDFR_E_INVALID, "workspace must be %zu-byte aligned", alignof(ViewSetup));
  ViewSetup* setup = (ViewSetup*)workspace;
  float* partials = (float*)((char*)workspace + setup_bytes(B));
  hipLaunchKernelGGL(view_setup_kernel, dim3((B + 63) / 64), dim3(64), 0, st, pos, quat, inv_scale,
                     B, R, W, H, cx, cy, fx, fy, setup);
  const int ntx = tiles_x(W), nty = tiles_y(H);
  const dim3 grid((unsigned)(ntx * nty * B));
  const float rfx = (float)(1.0 / (double)fx), rfy = (float)(1.0 / (double)fy);
  if (R == 64)
    hipLaunchKernelGGL(render_backward_kernel<64>, grid, dim3(kBlock), 0, st, grad_depth, depth, sdf,
                       R, sdf_view_stride, setup, W, H, ntx, nty, cx, cy, rfx, rfy, sdf_grad_mode,
                       g_sdf, g_sdf_view_stride, partials);
  else
    hipLaunchKernelGGL(render_backward_kernel<0>, grid, dim3(kBlock), 0, st, grad_depth, depth, sdf,
                       R, sdf_view_stride, setup, W, H, ntx, nty, cx, cy, rfx, rfy, sdf_grad_mode,
                       g_sdf, g_sdf_view_stride, partials);
  hipLaunchKernelGGL(pose_reduce_kernel, dim3(B), dim3(64), 0, st, partials, setup, ntx, nty, g_pos,
                     g_quat, g_inv_scale);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
