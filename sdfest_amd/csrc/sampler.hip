// sampler.hip -- trilinear SDF-grid sampler of the point-cloud loss, forward + VJP, for MI355X.
//
// Replaces the ~40 torch kernels of sdfest/estimation/losses.py:32-135 (pc_loss) and the ~40 more
// autograd replays behind it with one forward and one backward launch for ALL views of an
// optimisation step: points of view v occupy [offsets[v], offsets[v+1]) of the point array and use
// pose v.  Semantics restated from the reference:
//   q^ = q/|q| (gradient flows through the normalisation); R = matrix of q^ in the "1-2(..)" form;
//   o = R^T (P - p); pn = o/scale; c = floor((pn+1)(R-1)/2); a point whose un-clamped cell index
//   leaves [0, R-2] on any axis returns 0 (and gets no gradient); otherwise the value is the
//   trilinear interpolation (lerp order x, y, z) times scale.
// Points come from a depth image in row-major pixel order, so neighbouring threads sample
// neighbouring cells: the d/dsdf contributions of a 256-point block are pre-summed in the same
// fixed-point LDS run-hash as the renderer's backward before touching global memory.
#include "common.hpp"
#include "device.hpp"

namespace sdfr {
namespace {

constexpr int kPts = kSamplerPts;  // points per workgroup
#ifndef SDFR_SAMPLER_PAIR
#define SDFR_SAMPLER_PAIR 1  // 0: plain 4-voxel runs (timing experiments)
#endif
#if SDFR_SAMPLER_PAIR
using SamplerHash = PairRunHash<512>;   // z-pair runs: one LDS add per column of a cell (device.hpp)
#else
using SamplerHash = BatchHash;
#endif

struct PointFrame {
  float qn[4];
  float inv_norm;
  float rot[9];  // R(q^) row-major
  V3 p;
  float scale;
};

__device__ __forceinline__ PointFrame load_frame(const float* __restrict__ pos,
                                                 const float* __restrict__ quat,
                                                 const float* __restrict__ scale, int v) {
  PointFrame f;
  const float x0 = quat[4 * v], y0 = quat[4 * v + 1], z0 = quat[4 * v + 2], w0 = quat[4 * v + 3];
  const float n2 = x0 * x0 + y0 * y0 + z0 * z0 + w0 * w0;
  f.inv_norm = 1.0f / sqrtf(n2);
  const float x = x0 * f.inv_norm, y = y0 * f.inv_norm, z = z0 * f.inv_norm, w = w0 * f.inv_norm;
  f.qn[0] = x; f.qn[1] = y; f.qn[2] = z; f.qn[3] = w;
  f.rot[0] = 1 - 2 * (y * y + z * z); f.rot[1] = 2 * (x * y - w * z);     f.rot[2] = 2 * (x * z + w * y);
  f.rot[3] = 2 * (x * y + w * z);     f.rot[4] = 1 - 2 * (x * x + z * z); f.rot[5] = 2 * (y * z - w * x);
  f.rot[6] = 2 * (x * z - w * y);     f.rot[7] = 2 * (y * z + w * x);     f.rot[8] = 1 - 2 * (x * x + y * y);
  f.p = mk(pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]);
  f.scale = scale[v];
  return f;
}

// object-frame point o = R^T (P - p) and the cell; returns false for a masked (outside) point
template <int RT>
__device__ __forceinline__ bool sample_cell(const PointFrame& f, const float* __restrict__ vol,
                                            int R, V3 P, V3& vrel, V3& o, Cell& c) {
  const int Rr = RT > 0 ? RT : R;
  vrel = P - f.p;
  o = mk(fmaf(f.rot[0], vrel.x, fmaf(f.rot[3], vrel.y, f.rot[6] * vrel.z)),
         fmaf(f.rot[1], vrel.x, fmaf(f.rot[4], vrel.y, f.rot[7] * vrel.z)),
         fmaf(f.rot[2], vrel.x, fmaf(f.rot[5], vrel.y, f.rot[8] * vrel.z)));
  const float h = 0.5f * (float)(Rr - 1);
  const float gx = (o.x / f.scale + 1.0f) * h, gy = (o.y / f.scale + 1.0f) * h,
              gz = (o.z / f.scale + 1.0f) * h;
  const float top = (float)(Rr - 2);
  const float cxf = floorf(gx), cyf = floorf(gy), czf = floorf(gz);
  const bool inside = !(cxf < 0.0f) && !(cyf < 0.0f) && !(czf < 0.0f) && !(cxf > top) && !(cyf > top) &&
                      !(czf > top) && (gx == gx) && (gy == gy) && (gz == gz);
  gather_cell<RT>(vol, R, gx, gy, gz, c);  // clamps the cell, so the loads are always safe
  return inside;
}

template <int RT>
__global__ __launch_bounds__(kPts) void pc_loss_forward_kernel(
    const float* __restrict__ points, const int* __restrict__ offsets, int n_single,
    const float* __restrict__ pos, const float* __restrict__ quat, const float* __restrict__ scale,
    const float* __restrict__ sdf, int R, long long sdf_view_stride, float* __restrict__ out) {
  const int v = blockIdx.y;
  const int begin = offsets ? offsets[v] : 0;
  const int end = offsets ? offsets[v + 1] : n_single;
  const int i = begin + blockIdx.x * kPts + threadIdx.x;
  if (blockIdx.x * kPts >= end - begin) return;
  const PointFrame f = load_frame(pos, quat, scale, v);
  if (i >= end) return;
  const V3 P = mk(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
  V3 vrel, o;
  Cell c;
  const bool inside = sample_cell<RT>(f, sdf + (size_t)v * sdf_view_stride, R, P, vrel, o, c);
  out[i] = inside ? trilerp(c) * f.scale : 0.0f;
}

// L1: the upstream gradient is not read but formed here, for the loss  weight * mean |value|  over
// the view's points (simple_setup.py:144): go = +-weight / M_v by the sign of the point's value, and
// the block's sum of |value| goes to `loss_part` -- the loop then needs neither the sampler's
// forward launch nor the loss launch.
template <int RT, bool L1>
__global__ __launch_bounds__(kPts) void pc_loss_backward_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ points,
    const int* __restrict__ offsets, int n_single, const float* __restrict__ pos,
    const float* __restrict__ quat, const float* __restrict__ scale, const float* __restrict__ sdf,
    int R, long long sdf_view_stride, float* __restrict__ g_sdf, long long g_sdf_view_stride,
    float* __restrict__ partials, int nblk, float l1_weight, float* __restrict__ loss_part) {
  // 4-voxel runs x 512 slots: back-projected depth images are coherent (measured on 64 rendered
  // views, 1.13 M points: 122 -> 99 us against 2 x 1024; uniformly random points 80 -> 83 us)
  __shared__ SamplerHash hash;
  __shared__ float wave_part[kPts / 64][8];
  __shared__ int blk_max_bits;

  const int Rr = RT > 0 ? RT : R;
  const int v = blockIdx.y;
  const int begin = offsets ? offsets[v] : 0;
  const int end = offsets ? offsets[v + 1] : n_single;
  if (blockIdx.x * kPts >= end - begin) return;  // the reducer never reads this block's slot
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = begin + blockIdx.x * kPts + tid;
  const PointFrame f = load_frame(pos, quat, scale, v);
  const float* vol = sdf + (size_t)v * sdf_view_stride;
  float* gvol = g_sdf + (size_t)v * g_sdf_view_stride;

  hash.clear(tid, kPts);
  if (tid == 0) blk_max_bits = 0;

  bool live = false;
  float go = 0.0f, l1_abs = 0.0f;
  V3 vrel = mk(0, 0, 0), o = mk(0, 0, 0);
  Cell c;
  c.lin = 0; c.ox = c.oy = c.oz = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) c.v[k] = 0.0f;
  if (i < end) {
    const V3 P = mk(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
    live = sample_cell<RT>(f, vol, R, P, vrel, o, c);
    if (L1) {
      const float val = live ? trilerp(c) * f.scale : 0.0f;  // losses.py:133-135: masked values are 0
      const float k = l1_weight / (float)(end - begin);       // as pc_l1_kernel (loop.hip)
      go = val > 0.0f ? k : (val < 0.0f ? -k : 0.0f);
      l1_abs = fabsf(val);
    } else {
      go = live ? grad_out[i] : 0.0f;
    }
  }
  __syncthreads();
  const float gmax = wave_max(fabsf(go));
  if (lane == 0) atomicMax(&blk_max_bits, __float_as_int(gmax));
  __syncthreads();

  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (live) {
    const float gsz = 2.0f / (float)(Rr - 1);  // grid size
    const float tri = trilerp(c);
    const float ax = 1.0f - c.ox, ay = 1.0f - c.oy, az = 1.0f - c.oz;
    const float c00 = fmaf(c.v[4], c.ox, c.v[0] * ax), c01 = fmaf(c.v[5], c.ox, c.v[1] * ax);
    const float c10 = fmaf(c.v[6], c.ox, c.v[2] * ax), c11 = fmaf(c.v[7], c.ox, c.v[3] * ax);
    V3 G;  // d tri / d (cell coordinate)
    G.x = ((c.v[4] - c.v[0]) * ay + (c.v[6] - c.v[2]) * c.oy) * az +
          ((c.v[5] - c.v[1]) * ay + (c.v[7] - c.v[3]) * c.oy) * c.oz;
    G.y = (c10 - c00) * az + (c11 - c01) * c.oz;
    G.z = fmaf(c11, c.oy, c01 * ay) - fmaf(c10, c.oy, c00 * ay);
    // value = tri(off) * scale, off = (o/scale - cellpos)/g  =>  d value / d o = G / g
    const V3 dvo = mk(G.x / gsz, G.y / gsz, G.z / gsz);
    // d value / d scale = tri - (dvo . o) / scale
    acc[7] = go * (tri - dot(dvo, o) / f.scale);
    // o = R^T (P - p): d/dp = -R dvo
    acc[0] = -go * fmaf(f.rot[0], dvo.x, fmaf(f.rot[1], dvo.y, f.rot[2] * dvo.z));
    acc[1] = -go * fmaf(f.rot[3], dvo.x, fmaf(f.rot[4], dvo.y, f.rot[5] * dvo.z));
    acc[2] = -go * fmaf(f.rot[6], dvo.x, fmaf(f.rot[7], dvo.y, f.rot[8] * dvo.z));
    // R^T v = (1 - 2|u|^2) v + 2 u (u.v) - 2 w (u x v)  (the matrix form of losses.py:65-77):
    //   d/du_k = -4 u_k v + 2 e_k (u.v) + 2 u v_k - 2 w (e_k x v),   d/dw = -2 (u x v)
    const V3 u = mk(f.qn[0], f.qn[1], f.qn[2]);
    const float w = f.qn[3];
    const float udv = dot(u, vrel), Dv = dot(dvo, vrel), Du = dot(dvo, u);
    const V3 vxD = cross(vrel, dvo);  // dvo . (e_k x v) = (v x dvo)_k
    acc[3] = go * (-4.0f * u.x * Dv + 2.0f * udv * dvo.x + 2.0f * vrel.x * Du - 2.0f * w * vxD.x);
    acc[4] = go * (-4.0f * u.y * Dv + 2.0f * udv * dvo.y + 2.0f * vrel.y * Du - 2.0f * w * vxD.y);
    acc[5] = go * (-4.0f * u.z * Dv + 2.0f * udv * dvo.z + 2.0f * vrel.z * Du - 2.0f * w * vxD.z);
    acc[6] = go * (-2.0f * dot(dvo, cross(u, vrel)));

    // d/dsdf: go * scale * trilinear weight, through the fixed-point run-hash
    const float bound = 2.0f * __int_as_float(blk_max_bits) * fabsf(f.scale);
    int e2;
    (void)frexpf(bound, &e2);
    const bool fixed_ok = (bound > 0.0f) && (bound < 1e30f) && (e2 > -80);
    const float to_fixed = fixed_ok ? ldexpf(1.0f, SamplerHash::kBits - e2) : 0.0f;
    const float gs = go * f.scale;
    const float x0w = ax * gs, x1w = c.ox * gs;
    const float w0 = x0w * ay * az, w1 = x0w * ay * c.oz, w2 = x0w * c.oy * az, w3 = x0w * c.oy * c.oz;
    const float w4 = x1w * ay * az, w5 = x1w * ay * c.oz, w6 = x1w * c.oy * az, w7 = x1w * c.oy * c.oz;
    // (a NaN upstream gradient can be dropped by the block's fmaxf-based maximum: such a lane, like
    // any lane beyond the fixed-point range, adds in float, so NaN/Inf reach g_sdf as in autograd)
    if (fixed_ok && fabsf(gs) * to_fixed < SamplerHash::kWeightLimit) {
      const float wk[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
      hash.add_cell(gvol, c.lin, Rr, wk, to_fixed);
    } else if (go != 0.0f) {
      float* g0 = gvol + c.lin;
      atomicAdd(g0, w0);                atomicAdd(g0 + 1, w1);
      atomicAdd(g0 + Rr, w2);           atomicAdd(g0 + Rr + 1, w3);
      atomicAdd(g0 + Rr * Rr, w4);      atomicAdd(g0 + Rr * Rr + 1, w5);
      atomicAdd(g0 + Rr * Rr + Rr, w6); atomicAdd(g0 + Rr * Rr + Rr + 1, w7);
    }
  }
  {
    const float sk = wave_sum8(acc, lane);
    if ((lane & 7) == 0) wave_part[wave][lane >> 3] = sk;
  }
  __shared__ float wave_abs[kPts / 64];
  if (L1) {
    const float sa = wave_sum(l1_abs);
    if (lane == 0) wave_abs[wave] = sa;
  }
  __syncthreads();
  if (tid < 8) {
    float t = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kPts / 64; ++wv) t += wave_part[wv][tid];
    partials[((size_t)v * nblk + blockIdx.x) * 8 + tid] = t;
  }
  if (L1 && tid == 0) {
    float t = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kPts / 64; ++wv) t += wave_abs[wv];
    loss_part[(size_t)v * nblk + blockIdx.x] = t;
  }
  const float bound = 2.0f * __int_as_float(blk_max_bits) * fabsf(f.scale);
  int e2;
  (void)frexpf(bound, &e2);
  const float from_fixed = ldexpf(1.0f, e2 - SamplerHash::kBits);
  hash.flush(gvol, Rr * Rr * Rr, from_fixed, tid, kPts);
}

// one wave per view: fixed-order sum of the block partials, then the Jacobian of q^ = q/|q|
__global__ __launch_bounds__(64) void pc_loss_reduce_kernel(
    const float* __restrict__ partials, const int* __restrict__ offsets, int n_single, int nblk,
    const float* __restrict__ quat, float* __restrict__ g_pos, float* __restrict__ g_quat,
    float* __restrict__ g_scale, const float* __restrict__ loss_part, float* __restrict__ loss) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const int len = offsets ? offsets[v + 1] - offsets[v] : n_single;
  const int nb = (len + kPts - 1) / kPts;
  if (loss_part) {  // loss[v] = mean |value| (NaN for an empty view, as torch.mean of nothing)
    float sa = 0.0f;
    for (int i = lane; i < nb; i += 64) sa += loss_part[(size_t)v * nblk + i];
    sa = wave_sum(sa);
    if (lane == 0) loss[v] = sa / (float)len;
  }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = lane; i < nb; i += 64) {
    const float4* p = reinterpret_cast<const float4*>(partials + ((size_t)v * nblk + i) * 8);
    const float4 a = p[0], c = p[1];
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    acc[4] += c.x; acc[5] += c.y; acc[6] += c.z; acc[7] += c.w;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
    const float x = quat[4 * v], y = quat[4 * v + 1], z = quat[4 * v + 2], w = quat[4 * v + 3];
    const float inv_norm = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
    const float qn[4] = {x * inv_norm, y * inv_norm, z * inv_norm, w * inv_norm};
    const float d = qn[0] * acc[3] + qn[1] * acc[4] + qn[2] * acc[5] + qn[3] * acc[6];
    g_pos[3 * v] = acc[0]; g_pos[3 * v + 1] = acc[1]; g_pos[3 * v + 2] = acc[2];
    for (int k = 0; k < 4; ++k) g_quat[4 * v + k] = (acc[3 + k] - qn[k] * d) * inv_norm;
    g_scale[v] = acc[7];
  }
}

int check_pc(int R, int B, int max_view_points) {
  if (R < 2 || R > 1024) return fail(SDFR_E_INVALID, "R=%d out of range [2,1024]", R);
  if (B < 0 || B > 65535) return fail(SDFR_E_INVALID, "B=%d out of range [0,65535]", B);
  if (max_view_points < 0) return fail(SDFR_E_INVALID, "max_view_points=%d is negative", max_view_points);
  return 0;
}

}  // namespace
}  // namespace sdfr

using namespace sdfr;

extern "C" int sdfr_pc_loss_forward(const float* points, const int* offsets, int B,
                                    int max_view_points, const float* pos, const float* quat,
                                    const float* scale, const float* sdf, int R,
                                    long long sdf_view_stride, float* out, int device, void* stream) {
  if (int rc = check_pc(R, B, max_view_points)) return rc;
  if (!offsets && B > 1) return fail(SDFR_E_NULL, "offsets may be NULL only for a single view");
  if (sdf_view_stride != 0 && sdf_view_stride < (long long)R * R * R)
    return fail(SDFR_E_INVALID, "sdf_view_stride must be 0 or >= R^3");
  if (B == 0 || max_view_points == 0) return 0;
  if (!points || !pos || !quat || !scale || !sdf || !out)
    return fail(SDFR_E_NULL, "sdfr_pc_loss_forward: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((max_view_points + kPts - 1) / kPts), (unsigned)B);
  if (R == 64)
    hipLaunchKernelGGL(pc_loss_forward_kernel<64>, grid, dim3(kPts), 0, st, points, offsets,
                       max_view_points, pos, quat, scale, sdf, R, sdf_view_stride, out);
  else
    hipLaunchKernelGGL(pc_loss_forward_kernel<0>, grid, dim3(kPts), 0, st, points, offsets,
                       max_view_points, pos, quat, scale, sdf, R, sdf_view_stride, out);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t sdfr_pc_loss_backward_workspace_bytes(int B, int max_view_points) {
  if (B <= 0 || max_view_points <= 0) return 0;
  return (size_t)B * ((max_view_points + kPts - 1) / kPts) * 9 * sizeof(float);  // 8 pose sums + |value| sum
}

namespace {
// loss == nullptr: grad_out is the upstream gradient.  Otherwise the L1 form (grad_out unused).
int pc_backward_impl(const char* fn, const float* grad_out, float l1_weight, float* loss,
                     const float* points, const int* offsets, int B, int max_view_points,
                     const float* pos, const float* quat, const float* scale, const float* sdf, int R,
                     long long sdf_view_stride, float* g_sdf, long long g_sdf_view_stride, float* g_pos,
                     float* g_quat, float* g_scale, void* workspace, size_t workspace_bytes, int device,
                     void* stream, bool accumulate = false) {
  const bool l1 = loss != nullptr;
  if (int rc = check_pc(R, B, max_view_points)) return rc;
  const long long vox = (long long)R * R * R;
  if (!offsets && B > 1) return fail(SDFR_E_NULL, "offsets may be NULL only for a single view");
  if (sdf_view_stride != 0 && sdf_view_stride < vox)
    return fail(SDFR_E_INVALID, "sdf_view_stride must be 0 or >= R^3");
  if (g_sdf_view_stride != 0 && g_sdf_view_stride != vox)
    return fail(SDFR_E_INVALID, "g_sdf_view_stride must be 0 or R^3");
  if (!g_sdf) return fail(SDFR_E_NULL, "%s: g_sdf is NULL", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const size_t g_bytes = (size_t)vox * sizeof(float) * (g_sdf_view_stride ? (size_t)(B > 0 ? B : 1) : 1);
  if (!accumulate) zero_words_async(g_sdf, g_bytes / sizeof(float), st);
  if (B == 0) return 0;
  // deferred: the block partials stay in the workspace for sdfr_views_to_pose_grad_deferred (one launch less)
  const bool deferred = !g_pos && !g_quat && !g_scale;
  if (deferred && max_view_points == 0)
    return fail(SDFR_E_INVALID, "%s: deferred pose gradients need max_view_points > 0", fn);
  if ((!deferred && (!g_pos || !g_quat || !g_scale)) || !pos || !quat || !scale)
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (max_view_points == 0) {
    zero_words_async(g_pos, (size_t)B * 3, st);
    zero_words_async(g_quat, (size_t)B * 4, st);
    zero_words_async(g_scale, (size_t)B, st);
    if (l1) zero_words_async(loss, (size_t)B, st);
    return 0;
  }
  if ((!l1 && !grad_out) || !points || !sdf || !workspace)
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (workspace_bytes < sdfr_pc_loss_backward_workspace_bytes(B, max_view_points))
    return fail(SDFR_E_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, workspace_bytes,
                sdfr_pc_loss_backward_workspace_bytes(B, max_view_points));
  if ((uintptr_t)workspace % 16) return fail(SDFR_E_INVALID, "workspace must be 16-byte aligned");
  const int nblk = (max_view_points + kPts - 1) / kPts;
  const dim3 grid((unsigned)nblk, (unsigned)B);
  float* partials = (float*)workspace;
  float* loss_part = partials + (size_t)B * nblk * 8;
#define SDFR_PC_BWD(RT, L1)                                                                          \
  hipLaunchKernelGGL((pc_loss_backward_kernel<RT, L1>), grid, dim3(kPts), 0, st, grad_out, points,   \
                     offsets, max_view_points, pos, quat, scale, sdf, R, sdf_view_stride, g_sdf,     \
                     g_sdf_view_stride, partials, nblk, l1_weight, loss_part)
  if (R == 64) { if (l1) SDFR_PC_BWD(64, true); else SDFR_PC_BWD(64, false); }
  else { if (l1) SDFR_PC_BWD(0, true); else SDFR_PC_BWD(0, false); }
#undef SDFR_PC_BWD
  if (!deferred)
    hipLaunchKernelGGL(pc_loss_reduce_kernel, dim3(B), dim3(64), 0, st, partials, offsets,
                       max_view_points, nblk, quat, g_pos, g_quat, g_scale,
                       l1 ? (const float*)loss_part : (const float*)nullptr, loss);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int sdfr_pc_loss_backward(const float* grad_out, const float* points, const int* offsets,
                                     int B, int max_view_points, const float* pos, const float* quat,
                                     const float* scale, const float* sdf, int R,
                                     long long sdf_view_stride, float* g_sdf,
                                     long long g_sdf_view_stride, float* g_pos, float* g_quat,
                                     float* g_scale, void* workspace, size_t workspace_bytes,
                                     int device, void* stream) {
  return pc_backward_impl("sdfr_pc_loss_backward", grad_out, 0.0f, nullptr, points, offsets, B,
                          max_view_points, pos, quat, scale, sdf, R, sdf_view_stride, g_sdf,
                          g_sdf_view_stride, g_pos, g_quat, g_scale, workspace, workspace_bytes, device, stream);
}

extern "C" int sdfr_pc_l1_backward(float weight, float* loss, const float* points, const int* offsets,
                                   int B, int max_view_points, const float* pos, const float* quat,
                                   const float* scale, const float* sdf, int R, long long sdf_view_stride,
                                   float* g_sdf, long long g_sdf_view_stride, float* g_pos, float* g_quat,
                                   float* g_scale, void* workspace, size_t workspace_bytes, int device,
                                   void* stream) {
  if (B > 0 && !loss) return fail(SDFR_E_NULL, "sdfr_pc_l1_backward: loss is NULL");
  return pc_backward_impl("sdfr_pc_l1_backward", nullptr, weight, loss, points, offsets, B, max_view_points,
                          pos, quat, scale, sdf, R, sdf_view_stride, g_sdf, g_sdf_view_stride, g_pos, g_quat,
                          g_scale, workspace, workspace_bytes, device, stream);
}

extern "C" int sdfr_pc_l1_backward_accumulate(float weight, float* loss, const float* points, const int* offsets,
                                              int B, int max_view_points, const float* pos, const float* quat,
                                              const float* scale, const float* sdf, int R,
                                              long long sdf_view_stride, float* g_sdf, long long g_sdf_view_stride,
                                              float* g_pos, float* g_quat, float* g_scale, void* workspace,
                                              size_t workspace_bytes, int device, void* stream) {
  if (B > 0 && !loss) return fail(SDFR_E_NULL, "sdfr_pc_l1_backward_accumulate: loss is NULL");
  return pc_backward_impl("sdfr_pc_l1_backward_accumulate", nullptr, weight, loss, points, offsets, B,
                          max_view_points, pos, quat, scale, sdf, R, sdf_view_stride, g_sdf, g_sdf_view_stride, g_pos,
                          g_quat, g_scale, workspace, workspace_bytes, device, stream, true);
}
