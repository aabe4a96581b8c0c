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
#include "sampler_device.hpp"

namespace sdfr {
namespace {

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

// the backward: sampler_device.hpp, pc_backward_block.   grid: (blocks of 256 points, views)
template <int RT, bool L1>
__global__ __launch_bounds__(kPts) void pc_loss_backward_kernel(PcBackwardArgs a) {
  __shared__ PcBackwardLds lds;
  pc_backward_block<RT, L1>(lds, a, blockIdx.x, blockIdx.y);
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
  const PcBackwardArgs pa{grad_out, points, offsets, max_view_points, pos, quat, scale, sdf, R, sdf_view_stride,
                          g_sdf, g_sdf_view_stride, partials, nblk, l1_weight, loss_part};
#define SDFR_PC_BWD(RT, L1) hipLaunchKernelGGL((pc_loss_backward_kernel<RT, L1>), grid, dim3(kPts), 0, st, pa)
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
