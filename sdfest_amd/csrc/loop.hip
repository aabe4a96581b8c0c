// loop.hip -- the glue of one render-and-compare iteration as a handful of tiny kernels, so that
// the whole iteration (decoder, batched render, batched sampler, losses, gradient chain, Adam) is
// a fixed sequence of launches on one stream with no host round trip and can be replayed as a
// hipGraph.  What it restates (sdfest/estimation/simple_setup.py):
//   :411          norm_orientation = orientation / |orientation|
//   :424-430      per view: q_w2c = conj(camera_orientation); position_c = R(q_w2c)(position -
//                 camera_position); orientation_c = q_w2c * norm_orientation
//   :125-131      loss_depth = mean |estimate - input| over (input > 0) & (estimate > 0)
//   :144          loss_pc = mean |pc_loss(...)|
//   :448-454      loss = depth_weight * sum_v loss_depth + pc_weight * sum_v loss_pc
//   :400-406,:458 Adam (torch defaults) with lr 1e-3 / 1e-2 / 1e-3 / 1e-2 for position /
//                 orientation / scale / latent;  :462 orientation /= |orientation|
// and the reverse-mode chain of the first two items (what autograd does in the reference).
#include "common.hpp"
#include "decoder_fc.hpp"
#include "device.hpp"

namespace sdfr {
namespace {

__device__ __forceinline__ void quat_mul(const float* a, const float* b, float* o) {
  const float ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by - ax * bz + ay * bw + az * bx;
  o[2] = aw * bz + ax * by - ay * bx + az * bw;
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
}

// rotation matrix of a unit quaternion (row-major)
__device__ __forceinline__ void quat_matrix(const float* q, float* m) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  m[0] = 1 - 2 * (y * y + z * z); m[1] = 2 * (x * y - w * z);     m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z);     m[4] = 1 - 2 * (x * x + z * z); m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y);     m[7] = 2 * (y * z + w * x);     m[8] = 1 - 2 * (x * x + y * y);
}

// The reference rotates with quaternion_apply = q * (v,0) * conj(q) (quaternion_utils.py:36-54),
// which for a non-unit q scales by |q|^2; camera orientations are unit quaternions, for which
// this equals the matrix product.  The gradient chain below assumes unit camera quaternions.
// (the core takes the view's camera pose by value: the loop's tail has loaded it long before the parameters exist)
__device__ __forceinline__ void pose_to_view_core(int v, const float* position, const float* orientation,
                                                  const float* scale, const float (&cp)[3], const float (&cq)[4],
                                                  float* __restrict__ pos_c, float* __restrict__ quat_c,
                                                  float* __restrict__ inv_scale, float* __restrict__ scale_v) {
  const float q[4] = {orientation[0], orientation[1], orientation[2], orientation[3]};
  const float inv_n = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float nq[4] = {q[0] * inv_n, q[1] * inv_n, q[2] * inv_n, q[3] * inv_n};
  const float qc[4] = {-cq[0], -cq[1], -cq[2], cq[3]};
  float m[9];
  quat_matrix(qc, m);
  const float dx = position[0] - cp[0], dy = position[1] - cp[1], dz = position[2] - cp[2];
  pos_c[3 * v] = m[0] * dx + m[1] * dy + m[2] * dz;
  pos_c[3 * v + 1] = m[3] * dx + m[4] * dy + m[5] * dz;
  pos_c[3 * v + 2] = m[6] * dx + m[7] * dy + m[8] * dz;
  quat_mul(qc, nq, quat_c + 4 * v);
  inv_scale[v] = 1.0f / scale[0];
  scale_v[v] = scale[0];
}
__device__ __forceinline__ void pose_to_view(int v, const float* position, const float* orientation,
                                             const float* scale, const float* __restrict__ cam_pos,
                                             const float* __restrict__ cam_quat,
                                             float* __restrict__ pos_c, float* __restrict__ quat_c,
                                             float* __restrict__ inv_scale, float* __restrict__ scale_v) {
  const float cp[3] = {cam_pos[3 * v], cam_pos[3 * v + 1], cam_pos[3 * v + 2]};
  const float cq[4] = {cam_quat[4 * v], cam_quat[4 * v + 1], cam_quat[4 * v + 2], cam_quat[4 * v + 3]};
  pose_to_view_core(v, position, orientation, scale, cp, cq, pos_c, quat_c, inv_scale, scale_v);
}
__global__ void pose_to_views_kernel(const float* __restrict__ position,
                                     const float* __restrict__ orientation,
                                     const float* __restrict__ scale,
                                     const float* __restrict__ cam_pos,
                                     const float* __restrict__ cam_quat, int V,
                                     float* __restrict__ pos_c, float* __restrict__ quat_c,
                                     float* __restrict__ inv_scale, float* __restrict__ scale_v) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) pose_to_view(v, position, orientation, scale, cam_pos, cam_quat, pos_c, quat_c, inv_scale, scale_v);
}

// the same for K estimates (rows of `params`, `n` floats each: position 3 | orientation 4 | scale 1 | ...) seen from
// ONE camera list: the launch's view k V + v is object k from camera v (sdfr_loop_tail_objects' layout)
__global__ void pose_to_views_objects_kernel(const float* __restrict__ params, int n, int K,
                                             const float* __restrict__ cam_pos, const float* __restrict__ cam_quat,
                                             int V, float* __restrict__ pos_c, float* __restrict__ quat_c,
                                             float* __restrict__ inv_scale, float* __restrict__ scale_v,
                                             float* __restrict__ latents) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * V) return;
  const int k = i / V, v = i - k * V;
  const float* p = params + (size_t)k * n;
  if (latents && v == 0)   // the object's latent, packed [K][n - 8]: what a batched decode reads
    for (int j = 8; j < n; ++j) latents[(size_t)k * (n - 8) + j - 8] = p[j];
  const float cp[3] = {cam_pos[3 * v], cam_pos[3 * v + 1], cam_pos[3 * v + 2]};
  const float cq[4] = {cam_quat[4 * v], cam_quat[4 * v + 1], cam_quat[4 * v + 2], cam_quat[4 * v + 3]};
  pose_to_view_core(i, p, p + 3, p + 7, cp, cq, pos_c, quat_c, inv_scale, scale_v);
}

// one thread: sum the per-view gradients back to the world-frame parameters
__global__ void views_to_pose_grad_kernel(const float* __restrict__ orientation,
                                          const float* __restrict__ scale,
                                          const float* __restrict__ cam_quat, int V,
                                          const float* __restrict__ ga_pos,
                                          const float* __restrict__ ga_quat,
                                          const float* __restrict__ ga_inv_scale,
                                          const float* __restrict__ gb_pos,
                                          const float* __restrict__ gb_quat,
                                          const float* __restrict__ gb_scale,
                                          float* __restrict__ g_position,
                                          float* __restrict__ g_orientation,
                                          float* __restrict__ g_scale) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const float q[4] = {orientation[0], orientation[1], orientation[2], orientation[3]};
  const float inv_n = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float nq[4] = {q[0] * inv_n, q[1] * inv_n, q[2] * inv_n, q[3] * inv_n};
  float gp[3] = {0, 0, 0}, gn[4] = {0, 0, 0, 0}, gs = 0.0f;
  const float s = scale[0];
  for (int v = 0; v < V; ++v) {
    const float a[4] = {-cam_quat[4 * v], -cam_quat[4 * v + 1], -cam_quat[4 * v + 2], cam_quat[4 * v + 3]};
    float m[9];
    quat_matrix(a, m);
    float gpc[3], gqc[4];
    for (int k = 0; k < 3; ++k) gpc[k] = (ga_pos ? ga_pos[3 * v + k] : 0.0f) + (gb_pos ? gb_pos[3 * v + k] : 0.0f);
    for (int k = 0; k < 4; ++k) gqc[k] = (ga_quat ? ga_quat[4 * v + k] : 0.0f) + (gb_quat ? gb_quat[4 * v + k] : 0.0f);
    // position_c = M (p - c):  g_p += M^T g
    gp[0] += m[0] * gpc[0] + m[3] * gpc[1] + m[6] * gpc[2];
    gp[1] += m[1] * gpc[0] + m[4] * gpc[1] + m[7] * gpc[2];
    gp[2] += m[2] * gpc[0] + m[5] * gpc[1] + m[8] * gpc[2];
    // orientation_c = a * nq = L(a) nq:  g_nq += L(a)^T g
    const float ax = a[0], ay = a[1], az = a[2], aw = a[3];
    gn[0] += aw * gqc[0] + az * gqc[1] - ay * gqc[2] - ax * gqc[3];
    gn[1] += -az * gqc[0] + aw * gqc[1] + ax * gqc[2] - ay * gqc[3];
    gn[2] += ay * gqc[0] - ax * gqc[1] + aw * gqc[2] - az * gqc[3];
    gn[3] += ax * gqc[0] + ay * gqc[1] + az * gqc[2] + aw * gqc[3];
    // inv_scale = 1/scale (render), scale itself (sampler)
    gs += (ga_inv_scale ? -ga_inv_scale[v] / (s * s) : 0.0f) + (gb_scale ? gb_scale[v] : 0.0f);
  }
  const float d = nq[0] * gn[0] + nq[1] * gn[1] + nq[2] * gn[2] + nq[3] * gn[3];
  for (int k = 0; k < 3; ++k) g_position[k] = gp[k];
  for (int k = 0; k < 4; ++k) g_orientation[k] = (gn[k] - nq[k] * d) * inv_n;
  g_scale[0] = gs;
}

// The same chain with the two per-view reductions in front of it folded in (pose_reduce_kernel of render.hip,
// pc_loss_reduce_kernel of sampler.hip: same order of additions, so the same numbers): one launch instead of
// three.  One workgroup; wave w reduces views w, w + 4, ... into LDS, thread 0 then runs the chain.
//
// One wave: the 16 sums of view v -- dst[0..7] the renderer's (pos, quat, inv_scale: its tile partials in the order
// pose_reduce_kernel adds them), dst[8..15] the sampler's (pos, quat through the Jacobian of q^ = q / |q|, scale: its
// block partials as pc_loss_reduce_kernel) -- and, with pc_loss_part, the view's point-cloud loss to *pc_loss_out.
// Every load whose address does not depend on another load is issued first (the rectangle, the length of the point
// set, the quaternion), and the sampler's first 64 blocks are in flight together with the renderer's first 64 tiles:
// three dependent round trips instead of six -- the sums, and the order of their additions, are the same.
// the depth term of sdfr_render_step_fused_l1_pc (render.hip): unscaled tile sums beside the view's overlap count
struct FusedDepth {
  float* view_cnt = nullptr;          // [views], read and reset here
  const float* tile_loss = nullptr;   // [views][tiles][kLossRec]: (sum |est - obs|, count, ...)
  float weight = 0.0f;
  float* depth_loss = nullptr;        // [views] out (nullable)
};
__device__ __forceinline__ void reduce_view_wave(
    int v, int lane, const ViewSetup* __restrict__ setup, const float* __restrict__ tile_part, int W, int H,
    int ntx_all, int nty_all, int tile_w_all, int tile_h_all, int stride, const float* __restrict__ pc_part,
    const float* __restrict__ pc_loss_part, const int* __restrict__ offsets, int n_single, int nblk,
    const float* __restrict__ quat_c, float* __restrict__ pc_loss_out, float* __restrict__ dst,
    const FusedDepth& fd = FusedDepth{}) {
  // (the one-launch render step: the view's count -- on its way while the tiles are summed; reset below, once it has
  // arrived: the render launch that added to it is complete, the next one comes after this launch)
  const float cnt_l = fd.view_cnt ? fd.view_cnt[v] : 0.0f;
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0, big_flag = 0;
  if (tile_part) {
    const ViewSetup& s = setup[v];
    x0 = s.rect[0]; y0 = s.rect[1]; x1 = s.rect[2]; y1 = s.rect[3];
    big_flag = s.bwd_big;
  }
  int len = n_single;
  float qx = 0.0f, qy = 0.0f, qz = 0.0f, qw = 1.0f;
  if (pc_part) {
    if (offsets) len = offsets[v + 1] - offsets[v];
    qx = quat_c[4 * v]; qy = quat_c[4 * v + 1]; qz = quat_c[4 * v + 2]; qw = quat_c[4 * v + 3];
  }
  const int nb = pc_part ? (len + kSamplerPts - 1) / kSamplerPts : 0;
  // the sampler's first block of this lane
  float4 pa0 = make_float4(0, 0, 0, 0), pc0 = pa0;
  float sl0 = 0.0f;
  if (lane < nb) {
    const float4* p = reinterpret_cast<const float4*>(pc_part + ((size_t)v * nblk + lane) * 8);
    pa0 = p[0]; pc0 = p[1];
    if (pc_loss_part) sl0 = pc_loss_part[(size_t)v * nblk + lane];
  }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float l_sum = 0.0f;
  if (tile_part) {
    // the view's tiling, as pose_reduce_kernel (render.hip): a batch backward picks it per view
    int ntx = ntx_all, nty = nty_all, tile_w = tile_w_all, tile_h = tile_h_all;
    size_t first = (size_t)v * ntx * nty;
    const bool big = stride > 0 && big_flag;
    if (stride > 0) first = (size_t)v * stride;
    if (big) { tile_w = kBwdBigTile.w(); tile_h = kBwdBigTile.h(); }
    (void)nty;
    if (x1 > x0 && y1 > y0) {
      const int tx0 = x0 / tile_w, tx1 = (x1 - 1) / tile_w, ty0 = y0 / tile_h, ty1 = (y1 - 1) / tile_h;
      const int nx = tx1 - tx0 + 1, n = nx * (ty1 - ty0 + 1);
      const float* base = tile_part + first * 8;
      for (int i = lane; i < n; i += 64) {
        const int ty = ty0 + i / nx, tx = tx0 + i % nx;
        const size_t rec = big ? (size_t)backward_big_record(tx, ty, W) : (size_t)ty * ntx + tx;
        const float4* p = reinterpret_cast<const float4*>(base + rec * 8);
        const float4 a = p[0], c = p[1];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += c.x; acc[5] += c.y; acc[6] += c.z; acc[7] += c.w;
        if (fd.tile_loss) l_sum += fd.tile_loss[(first + rec) * kLossRec];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = wave_sum(acc[k]);
    if (fd.view_cnt) {
      // the tiles' sums were formed with the bare sign of (est - obs) as their upstream gradient: times
      // k = weight / count, the expression of the two-launch form's backward tiles (render.hip, backward_tile)
      const float cnt = cnt_l;
      const float kk = cnt > 0.0f ? fd.weight / cnt : 0.0f;
      if (lane == 0) fd.view_cnt[v] = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] *= kk;
      l_sum = wave_sum(l_sum);
      if (lane == 0 && fd.depth_loss) fd.depth_loss[v] = l_sum / cnt;   // (no overlap: 0 / 0, as loss_reduce_kernel)
    }
  }
  if (lane < 8) {
    float r = acc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) r = (lane == k) ? acc[k] : r;
    dst[lane] = r;
  }
  float pcs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pc_part) {
    if (pc_loss_part) {
      float sa = 0.0f;
      if (lane < nb) sa += sl0;
      for (int i = lane + 64; i < nb; i += 64) sa += pc_loss_part[(size_t)v * nblk + i];
      sa = wave_sum(sa);
      if (lane == 0) *pc_loss_out = sa / (float)len;
    }
    if (lane < nb) {
      pcs[0] += pa0.x; pcs[1] += pa0.y; pcs[2] += pa0.z; pcs[3] += pa0.w;
      pcs[4] += pc0.x; pcs[5] += pc0.y; pcs[6] += pc0.z; pcs[7] += pc0.w;
    }
    for (int i = lane + 64; i < nb; i += 64) {
      const float4* p = reinterpret_cast<const float4*>(pc_part + ((size_t)v * nblk + i) * 8);
      const float4 a = p[0], c = p[1];
      pcs[0] += a.x; pcs[1] += a.y; pcs[2] += a.z; pcs[3] += a.w;
      pcs[4] += c.x; pcs[5] += c.y; pcs[6] += c.z; pcs[7] += c.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) pcs[k] = wave_sum(pcs[k]);
    // Jacobian of q^ = q / |q| on the view's quaternion, as pc_loss_reduce_kernel
    const float x = qx, y = qy, z = qz, w = qw;
    const float inv_norm = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
    const float qn[4] = {x * inv_norm, y * inv_norm, z * inv_norm, w * inv_norm};
    const float dq = qn[0] * pcs[3] + qn[1] * pcs[4] + qn[2] * pcs[5] + qn[3] * pcs[6];
#pragma unroll
    for (int k = 0; k < 4; ++k) pcs[3 + k] = (pcs[3 + k] - qn[k] * dq) * inv_norm;
  }
  if (lane < 8) {
    float r = pcs[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) r = (lane == k) ? pcs[k] : r;
    dst[8 + lane] = r;
  }
}

// One thread: the V views' 16 sums (rec + v * rec_stride; [0..7] renderer, [8..15] sampler) back to the world-frame
// parameters -- the reverse of pose_to_view -- and through the normalisation of the orientation.  Views are added
// in index order: the result depends on the records alone, not on where they were computed.
__device__ __forceinline__ void pose_chain(const float* orientation, const float* scale,
                                           const float* cam_quat, int V, const float* rec,
                                           int rec_stride, bool use_a, bool use_b, float* g_position,
                                           float* g_orientation, float* g_scale) {
  const float q[4] = {orientation[0], orientation[1], orientation[2], orientation[3]};
  const float inv_n = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float nq[4] = {q[0] * inv_n, q[1] * inv_n, q[2] * inv_n, q[3] * inv_n};
  float gp[3] = {0, 0, 0}, gn[4] = {0, 0, 0, 0}, gs = 0.0f;
  const float s = scale[0];
  for (int v = 0; v < V; ++v) {
    const float a[4] = {-cam_quat[4 * v], -cam_quat[4 * v + 1], -cam_quat[4 * v + 2], cam_quat[4 * v + 3]};
    float m[9];
    quat_matrix(a, m);
    const float* ga = rec + (size_t)v * rec_stride;
    const float* gb = ga + 8;
    float gpc[3], gqc[4];
    for (int k = 0; k < 3; ++k) gpc[k] = (use_a ? ga[k] : 0.0f) + (use_b ? gb[k] : 0.0f);
    for (int k = 0; k < 4; ++k) gqc[k] = (use_a ? ga[3 + k] : 0.0f) + (use_b ? gb[3 + k] : 0.0f);
    gp[0] += m[0] * gpc[0] + m[3] * gpc[1] + m[6] * gpc[2];
    gp[1] += m[1] * gpc[0] + m[4] * gpc[1] + m[7] * gpc[2];
    gp[2] += m[2] * gpc[0] + m[5] * gpc[1] + m[8] * gpc[2];
    const float ax = a[0], ay = a[1], az = a[2], aw = a[3];
    gn[0] += aw * gqc[0] + az * gqc[1] - ay * gqc[2] - ax * gqc[3];
    gn[1] += -az * gqc[0] + aw * gqc[1] + ax * gqc[2] - ay * gqc[3];
    gn[2] += ay * gqc[0] - ax * gqc[1] + aw * gqc[2] - az * gqc[3];
    gn[3] += ax * gqc[0] + ay * gqc[1] + az * gqc[2] + aw * gqc[3];
    gs += (use_a ? -ga[7] / (s * s) : 0.0f) + (use_b ? gb[7] : 0.0f);
  }
  const float d = nq[0] * gn[0] + nq[1] * gn[1] + nq[2] * gn[2] + nq[3] * gn[3];
  for (int k = 0; k < 3; ++k) g_position[k] = gp[k];
  for (int k = 0; k < 4; ++k) g_orientation[k] = (gn[k] - nq[k] * d) * inv_n;
  g_scale[0] = gs;
}

// pose_chain by the whole workgroup, for view lists of any length (the sharded loop's tail: V is the number of views
// of ALL ranks): thread t adds the views t, t + 256, ... in that order, the 256 partial sums meet in a fixed tree.  The
// result depends on the records and on V alone -- not on which rank computed which record.  Every thread calls it.
__device__ __forceinline__ void pose_chain_block(const float* orientation, const float* scale,
                                                 const float* __restrict__ cam_quat, int V, const float* rec,
                                                 int rec_stride, float* g_position,
                                                 float* g_orientation, float* g_scale) {
  __shared__ float part[8][256];
  const int tid = threadIdx.x;
  float gp[3] = {0, 0, 0}, gn[4] = {0, 0, 0, 0}, gs = 0.0f;
  const float s = scale[0];
  for (int v = tid; v < V; v += 256) {
    const float a[4] = {-cam_quat[4 * v], -cam_quat[4 * v + 1], -cam_quat[4 * v + 2], cam_quat[4 * v + 3]};
    float m[9];
    quat_matrix(a, m);
    const float* ga = rec + (size_t)v * rec_stride;
    const float* gb = ga + 8;
    float gpc[3], gqc[4];
    for (int k = 0; k < 3; ++k) gpc[k] = ga[k] + gb[k];
    for (int k = 0; k < 4; ++k) gqc[k] = ga[3 + k] + gb[3 + k];
    gp[0] += m[0] * gpc[0] + m[3] * gpc[1] + m[6] * gpc[2];
    gp[1] += m[1] * gpc[0] + m[4] * gpc[1] + m[7] * gpc[2];
    gp[2] += m[2] * gpc[0] + m[5] * gpc[1] + m[8] * gpc[2];
    const float ax = a[0], ay = a[1], az = a[2], aw = a[3];
    gn[0] += aw * gqc[0] + az * gqc[1] - ay * gqc[2] - ax * gqc[3];
    gn[1] += -az * gqc[0] + aw * gqc[1] + ax * gqc[2] - ay * gqc[3];
    gn[2] += ay * gqc[0] - ax * gqc[1] + aw * gqc[2] - az * gqc[3];
    gn[3] += ax * gqc[0] + ay * gqc[1] + az * gqc[2] + aw * gqc[3];
    gs += -ga[7] / (s * s) + gb[7];
  }
  part[0][tid] = gp[0]; part[1][tid] = gp[1]; part[2][tid] = gp[2];
  part[3][tid] = gn[0]; part[4][tid] = gn[1]; part[5][tid] = gn[2]; part[6][tid] = gn[3];
  part[7][tid] = gs;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int k = 0; k < 8; ++k) part[k][tid] += part[k][tid + off];
    }
    __syncthreads();
  }
  if (tid != 0) return;   // (every thread has passed the last barrier)
  const float q[4] = {orientation[0], orientation[1], orientation[2], orientation[3]};
  const float inv_n = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float nq[4] = {q[0] * inv_n, q[1] * inv_n, q[2] * inv_n, q[3] * inv_n};
  const float d = nq[0] * part[3][0] + nq[1] * part[4][0] + nq[2] * part[5][0] + nq[3] * part[6][0];
  for (int k = 0; k < 3; ++k) g_position[k] = part[k][0];
  for (int k = 0; k < 4; ++k) g_orientation[k] = (part[3 + k][0] - nq[k] * d) * inv_n;
  g_scale[0] = part[7][0];
}

// (The tail keeps wave 0 busy with the decoder's Linear stack first; which wave reduces which view changes nothing.)
__device__ __forceinline__ void deferred_chain(
    const float* orientation, const float* scale, const float* __restrict__ cam_quat,
    int V, const ViewSetup* __restrict__ setup, const float* __restrict__ tile_part, int W, int H, int ntx_all,
    int nty_all, int tile_w_all, int tile_h_all, int stride, const float* __restrict__ pc_part,
    const float* __restrict__ pc_loss_part,
    const int* __restrict__ offsets, int n_single, int nblk, const float* __restrict__ quat_c,
    float* __restrict__ pc_loss, float* g_position, float* g_orientation, float* g_scale, int view_base = 0,
    const FusedDepth& fd = FusedDepth{}) {
  // view_base (sdfr_loop_tail_objects): the V views of THIS object are the launch's views view_base .. view_base + V - 1
  // (their set-up records, tile partials, point blocks, quaternions, losses); the cameras are the object's own list
  __shared__ float view_g[kDeferredMaxViews][16];  // [0..7] renderer: pos, quat, inv_scale; [8..15] sampler
  __shared__ float view_cq[kDeferredMaxViews][4];  // the views' camera orientations (the chain's thread reads them here)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // view v is wave (v + 1) % 4's: wave 0, which may arrive late, takes the fourth, eighth, ... view only
  for (int v = (wave + 3) & 3; v < V; v += 4) {
    const float cq = lane < 4 ? cam_quat[4 * v + lane] : 0.0f;
    reduce_view_wave(view_base + v, lane, setup, tile_part, W, H, ntx_all, nty_all, tile_w_all, tile_h_all, stride,
                     pc_part, pc_loss_part, offsets, n_single, nblk, quat_c,
                     pc_loss ? pc_loss + view_base + v : nullptr, view_g[v], fd);
    if (lane < 4) view_cq[v][lane] = cq;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;   // (callers that go on afterwards: every thread reaches the barrier above)
  pose_chain(orientation, scale, &view_cq[0][0], V, &view_g[0][0], 16, tile_part != nullptr, pc_part != nullptr,
             g_position, g_orientation, g_scale);
}

// The sharded loop's exchange record of one view (include/sdfr.h, sdfr_loop_view_records): one wave per view of the
// WHOLE list; a view of another rank's shard gets zeros, so that the sum over ranks is every view's own record.
constexpr int kViewRecord = SDFR_VIEW_RECORD_FLOATS;
static_assert(kViewRecord >= 18 && kViewRecord % 4 == 0 && kViewRecord <= 64, "record: 16 sums + 2 losses, 16-byte multiple");
__global__ __launch_bounds__(64) void view_records_kernel(
    const ViewSetup* __restrict__ setup, const float* __restrict__ tile_part, int W, int H, int ntx_all, int nty_all,
    int tile_w_all, int tile_h_all, int stride, const float* __restrict__ pc_part,
    const float* __restrict__ pc_loss_part, const int* __restrict__ offsets, int n_single, int nblk,
    const float* __restrict__ quat_c, const float* __restrict__ loss_depth, int view_begin, int V_local,
    float* __restrict__ records) {
  const int v = blockIdx.x, lane = threadIdx.x;
  float* rec = records + (size_t)v * kViewRecord;
  const int lv = v - view_begin;
  if (lv < 0 || lv >= V_local) {
    if (lane < kViewRecord) rec[lane] = 0.0f;
    return;
  }
  if (lane == 0) {
    rec[16] = loss_depth ? loss_depth[lv] : 0.0f;
    if (!pc_loss_part) rec[17] = 0.0f;
  }
  if (lane >= 18 && lane < kViewRecord) rec[lane] = 0.0f;
  reduce_view_wave(lv, lane, setup, tile_part, W, H, ntx_all, nty_all, tile_w_all, tile_h_all, stride, pc_part,
                   pc_loss_part, offsets, n_single, nblk, quat_c, rec + 17, rec);
}
__global__ __launch_bounds__(256) void views_to_pose_grad_deferred_kernel(
    const float* __restrict__ orientation, const float* __restrict__ scale, const float* __restrict__ cam_quat,
    int V, const ViewSetup* __restrict__ setup, const float* __restrict__ tile_part, int W, int H, int ntx_all,
    int nty_all, int tile_w_all, int tile_h_all, int stride, const float* __restrict__ pc_part,
    const float* __restrict__ pc_loss_part,
    const int* __restrict__ offsets, int n_single, int nblk, const float* __restrict__ quat_c,
    float* __restrict__ pc_loss, float* __restrict__ g_position, float* __restrict__ g_orientation,
    float* __restrict__ g_scale) {
  deferred_chain(orientation, scale, cam_quat, V, setup, tile_part, W, H, ntx_all, nty_all, tile_w_all, tile_h_all,
                 stride, pc_part, pc_loss_part, offsets, n_single, nblk, quat_c, pc_loss, g_position, g_orientation,
                 g_scale);
}

// ---------------------------------------------------------------------------------------------
// What the loop does besides the two image losses (simple_setup.py):
//   :164-175  point constraint: weight * | quaternion_apply(orientation, source) - target |
//             (losses.py:138-153) on the UN-normalised orientation parameter -- quaternion_apply is
//             q (v,0) conj(q) (quaternion_utils.py:36-54), which scales by |q|^2;
//   :177-211  inlier ratio of the LAST view's input / estimate (the loop variables that survive the
//             `for` over views, :463-470) and the best-so-far bookkeeping.
// ---------------------------------------------------------------------------------------------
// r = q (s,0) conj(q) = (w^2 - u.u) s + 2 (u.s) u + 2 w (u x s);  loss = weight |r - t|;  the gradient
// w.r.t. q is ADDED to g_orientation (the image terms have been written there before).
__device__ __forceinline__ void point_constraint_one(const float* q, const float* src, const float* tgt, float weight,
                                                     float* __restrict__ loss, float* g_orientation) {
  const V3 u = mk(q[0], q[1], q[2]), s = mk(src[0], src[1], src[2]);
  const float w = q[3];
  const float us = dot(u, s), uu = dot(u, u);
  const V3 uxs = cross(u, s);
  const V3 r = (w * w - uu) * s + (2.0f * us) * u + (2.0f * w) * uxs;
  const V3 d = r - mk(tgt[0], tgt[1], tgt[2]);
  const float len = sqrtf(dot(d, d));
  if (loss) loss[0] = weight * len;
  if (!g_orientation) return;
  const float k = len > 0.0f ? weight / len : 0.0f;  // torch: the norm's gradient at 0 is 0
  const V3 n = k * d;
  const float ns = dot(n, s), nu = dot(n, u);
  const V3 sxn = cross(s, n);
  g_orientation[0] += -2.0f * u.x * ns + 2.0f * s.x * nu + 2.0f * us * n.x + 2.0f * w * sxn.x;
  g_orientation[1] += -2.0f * u.y * ns + 2.0f * s.y * nu + 2.0f * us * n.y + 2.0f * w * sxn.y;
  g_orientation[2] += -2.0f * u.z * ns + 2.0f * s.z * nu + 2.0f * us * n.z + 2.0f * w * sxn.z;
  g_orientation[3] += 2.0f * w * ns + 2.0f * dot(n, uxs);
}
__global__ void point_constraint_kernel(const float* __restrict__ q, const float* __restrict__ src,
                                        const float* __restrict__ tgt, float weight,
                                        float* __restrict__ loss, float* __restrict__ g_orientation) {
  if (blockIdx.x == 0 && threadIdx.x == 0) point_constraint_one(q, src, tgt, weight, loss, g_orientation);
}

// counts[0] += #(|in - est| / in < thr), counts[1] += #(in != 0) over one chunk of pixels.  The
// expression is the reference's (:182-186): a zero input gives inf or NaN, neither is below thr.
__global__ __launch_bounds__(256) void inlier_count_kernel(const float* __restrict__ depth_in,
                                                           const float* __restrict__ depth_est, int npix,
                                                           float rel_thr, int* __restrict__ counts) {
  int inl = 0, val = 0;
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < npix && i < (blockIdx.x + 1) * 1024; i += 256) {
    const float a = depth_in[i], e = depth_est[i];
    inl += (fabsf(a - e) / a < rel_thr) ? 1 : 0;
    val += (a != 0.0f) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    inl += __shfl_xor(inl, off, 64);
    val += __shfl_xor(val, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (inl) atomicAdd(&counts[0], inl);
    if (val) atomicAdd(&counts[1], val);
  }
}

// ratio = inliers / valid; history[step - 1] = ratio; a strictly better ratio (or the first) takes
// the current parameters as the best estimate (:203-211).  state = {best ratio, its iteration
// (1-based), has_best}; counts are cleared for the next iteration.
// counts_rec (the sharded loop): the two counts as floats in the last view's exchanged record (exact: < 2^24)
__global__ void inlier_update_kernel(int* __restrict__ counts, const int* __restrict__ step,
                                     float* __restrict__ history, int max_history,
                                     float* __restrict__ state, const float* __restrict__ params,
                                     int n_params, float* __restrict__ best_params,
                                     const float* __restrict__ counts_rec) {
  __shared__ int better;
  if (threadIdx.x == 0) {
    const float ratio = counts_rec ? counts_rec[0] / counts_rec[1] : (float)counts[0] / (float)counts[1];
    const int it = step[0];  // steps taken so far: Adam has run for this iteration already
    if (history && it >= 1 && it <= max_history) history[it - 1] = ratio;
    better = (state[2] == 0.0f) || (ratio > state[0]);
    if (better) { state[0] = ratio; state[1] = (float)it; state[2] = 1.0f; }
    if (counts) { counts[0] = 0; counts[1] = 0; }
  }
  __syncthreads();
  if (better && best_params)
    for (int i = threadIdx.x; i < n_params; i += blockDim.x) best_params[i] = params[i];
}
__global__ void inlier_counts_to_record_kernel(int* __restrict__ counts, float* __restrict__ rec) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    rec[0] = (float)counts[0]; rec[1] = (float)counts[1];
    counts[0] = 0; counts[1] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// losses.nn_loss (losses.py:8-29): squared distance of every point of `from` to its nearest point
// of `to`, evaluated as the reference does (-2 a.b + |a|^2 + |b|^2, negatives clamped to 0).  One
// thread per `from` point, `to` staged through LDS in blocks of 256.  D = 3.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nn_loss_forward_kernel(const float* __restrict__ from, int N,
                                                              const float* __restrict__ to, int M,
                                                              float* __restrict__ dist,
                                                              int* __restrict__ nearest) {
  __shared__ float tile[256][4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < N;
  const float ax = live ? from[3 * i] : 0.0f, ay = live ? from[3 * i + 1] : 0.0f, az = live ? from[3 * i + 2] : 0.0f;
  const float a2 = (ax * ax + ay * ay) + az * az;
  float best = 3.0e38f;
  int arg = -1;
  for (int j0 = 0; j0 < M; j0 += 256) {
    const int j = j0 + threadIdx.x;
    if (j < M) {
      const float bx = to[3 * j], by = to[3 * j + 1], bz = to[3 * j + 2];
      tile[threadIdx.x][0] = bx; tile[threadIdx.x][1] = by; tile[threadIdx.x][2] = bz;
      tile[threadIdx.x][3] = (bx * bx + by * by) + bz * bz;
    }
    __syncthreads();
    const int n = min(256, M - j0);
    for (int k = 0; k < n; ++k) {
      const float ab = (ax * tile[k][0] + ay * tile[k][1]) + az * tile[k][2];
      float d = (-2.0f * ab + a2) + tile[k][3];
      d = d < 0.0f ? 0.0f : d;
      if (d < best) { best = d; arg = j0 + k; }  // first minimum, like torch.min
    }
    __syncthreads();
  }
  if (live) { dist[i] = best; nearest[i] = arg; }
}

// VJP of the above through the selected pair: d/da = 2 (a - b_j), d/db_j = -2 (a - b_j); a clamped
// distance (exactly 0) passes no gradient (the in-place d[d < 0] = 0).
__global__ __launch_bounds__(256) void nn_loss_backward_kernel(const float* __restrict__ grad_dist,
                                                               const float* __restrict__ from, int N,
                                                               const float* __restrict__ to,
                                                               const float* __restrict__ dist,
                                                               const int* __restrict__ nearest,
                                                               float* __restrict__ g_from,
                                                               float* __restrict__ g_to) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int j = nearest[i];
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  if (j >= 0 && dist[i] > 0.0f) {
    const float g = 2.0f * grad_dist[i];
    gx = g * (from[3 * i] - to[3 * j]); gy = g * (from[3 * i + 1] - to[3 * j + 1]); gz = g * (from[3 * i + 2] - to[3 * j + 2]);
    if (g_to) { atomicAdd(g_to + 3 * j, -gx); atomicAdd(g_to + 3 * j + 1, -gy); atomicAdd(g_to + 3 * j + 2, -gz); }
  }
  if (g_from) { g_from[3 * i] = gx; g_from[3 * i + 1] = gy; g_from[3 * i + 2] = gz; }
}

// ---------------------------------------------------------------------------------------------
// generated_dataset.py:234-245 -- the mask perturbation of the synthetic-view generator: a random
// affine map of the exact mask (depth != 0), nearest neighbour, zeros outside.  The reference calls
// torchvision's RandomAffine, which for tensors is an inverse affine matrix about the image centre
// applied to the pixel-centre grid and torch's grid_sample(mode = nearest, align_corners = False):
// output pixel (row i, col j) takes the input pixel nearest to m (j + 0.5 - W/2, i + 0.5 - H/2, 1)
// + (W/2 - 0.5, H/2 - 0.5), rounding half to even.  One thread per output pixel, grid.y = view.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_mask_kernel(const float* __restrict__ depth, int W, int H,
                                                          const float* __restrict__ matrices,
                                                          unsigned char* __restrict__ mask) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const float* m = matrices + 6 * b;
  const float xo = (float)(i % W) + 0.5f - 0.5f * (float)W, yo = (float)(i / W) + 0.5f - 0.5f * (float)H;
  const float xs = fmaf(m[0], xo, fmaf(m[1], yo, m[2])) + (0.5f * (float)W - 0.5f);
  const float ys = fmaf(m[3], xo, fmaf(m[4], yo, m[5])) + (0.5f * (float)H - 0.5f);
  const float cx = rintf(xs), cy = rintf(ys);
  bool on = false;
  if (cx >= 0.0f && cx <= (float)(W - 1) && cy >= 0.0f && cy <= (float)(H - 1))
    on = depth[(size_t)b * W * H + (size_t)cy * W + (size_t)cx] != 0.0f;
  mask[(size_t)b * W * H + i] = on ? 1 : 0;
}

constexpr int kLossChunk = 4096;  // pixels per workgroup of the depth-loss reduction

// pass 1: per (view, chunk) the sum of |est - tgt| and the count over the overlap mask
__global__ __launch_bounds__(256) void depth_l1_partial_kernel(const float* __restrict__ est,
                                                               const float* __restrict__ tgt,
                                                               int npix, int nchunk,
                                                               float* __restrict__ partial) {
  __shared__ float ssum[4], scnt[4];
  const int v = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)v * npix;
  float sum = 0.0f, cnt = 0.0f;
  for (int i = chunk * kLossChunk + tid; i < min(npix, (chunk + 1) * kLossChunk); i += 256) {
    const float e = est[base + i], t = tgt[base + i];
    if (t > 0.0f && e > 0.0f) {
      sum += fabsf(e - t);
      cnt += 1.0f;
    }
  }
  sum = wave_sum(sum);
  cnt = wave_sum(cnt);
  if ((tid & 63) == 0) { ssum[tid >> 6] = sum; scnt[tid >> 6] = cnt; }
  __syncthreads();
  if (tid == 0) {
    partial[((size_t)v * nchunk + chunk) * 2] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
    partial[((size_t)v * nchunk + chunk) * 2 + 1] = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
  }
}

// pass 2: loss[v] = sum / count (NaN for an empty overlap, as torch.mean of an empty selection);
// grad_est = weight * sign(est - tgt) / count on the overlap, 0 elsewhere
__global__ __launch_bounds__(256) void depth_l1_grad_kernel(const float* __restrict__ est,
                                                            const float* __restrict__ tgt, int npix,
                                                            int nchunk,
                                                            const float* __restrict__ partial,
                                                            float weight, float* __restrict__ loss,
                                                            float* __restrict__ grad) {
  const int v = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  float sum = 0.0f, cnt = 0.0f;
  for (int c = 0; c < nchunk; ++c) {  // same order in every thread: one value per view
    sum += partial[((size_t)v * nchunk + c) * 2];
    cnt += partial[((size_t)v * nchunk + c) * 2 + 1];
  }
  if (chunk == 0 && tid == 0) loss[v] = sum / cnt;
  const float k = cnt > 0.0f ? weight / cnt : 0.0f;
  const size_t base = (size_t)v * npix;
  for (int i = chunk * kLossChunk + tid; i < min(npix, (chunk + 1) * kLossChunk); i += 256) {
    const float e = est[base + i], t = tgt[base + i];
    float g = 0.0f;
    if (t > 0.0f && e > 0.0f) g = (e > t) ? k : ((e < t) ? -k : 0.0f);
    grad[base + i] = g;
  }
}

// one workgroup per view: loss[v] = mean |val| over the view's points; grad = weight*sign/M_v
__global__ __launch_bounds__(256) void pc_l1_kernel(const float* __restrict__ val,
                                                    const int* __restrict__ offsets, int n_single,
                                                    float weight, float* __restrict__ loss,
                                                    float* __restrict__ grad) {
  __shared__ float ssum[4];
  const int v = blockIdx.x, tid = threadIdx.x;
  const int begin = offsets ? offsets[v] : 0, end = offsets ? offsets[v + 1] : n_single;
  float sum = 0.0f;
  for (int i = begin + tid; i < end; i += 256) sum += fabsf(val[i]);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) ssum[tid >> 6] = sum;
  __syncthreads();
  const float m = (float)(end - begin);
  if (tid == 0) loss[v] = ((ssum[0] + ssum[1]) + (ssum[2] + ssum[3])) / m;
  const float k = weight / m;
  for (int i = begin + tid; i < end; i += 256) {
    const float x = val[i];
    grad[i] = x > 0.0f ? k : (x < 0.0f ? -k : 0.0f);
  }
}

__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] += b[i];
}

// Adam with torch.optim.Adam's defaults (betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad):
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// One thread per scalar of the four groups laid out [position 3 | orientation 4 | scale 1 | latent L];
// `step` lives on the device so that a captured graph advances it on every replay.
// (every thread of the workgroup calls it: two barriers inside)
// (the core takes thread i's parameter, moments and the step count by value and returns the new ones: the loop's tail
// loads the old ones at its very start, long before the gradients exist, and stores the new ones at its very end --
// a global store in front of a workgroup barrier is a store round trip the barrier waits for.  Optionally leaves a
// copy of the new parameters in LDS.)
__device__ __forceinline__ void adam_step_core(const float* grads, int n, float lr_pos, float lr_quat, float lr_scale,
                                               float lr_latent, int update_latent, float p_old, float m_old,
                                               float v_old, int step_old, float* params_lds, float& p_out,
                                               float& m_out, float& v_out) {
  __shared__ float qnorm2;
  const int i = threadIdx.x;
  const int t = step_old + 1;
  if (i == 0) qnorm2 = 0.0f;
  __syncthreads();
  float p = 0.0f;
  m_out = m_old;
  v_out = v_old;
  if (i < n && (i < 8 || update_latent)) {
    const float lr = i < 3 ? lr_pos : (i < 7 ? lr_quat : (i < 8 ? lr_scale : lr_latent));
    const float g = grads[i];
    const float mi = 0.9f * m_old + 0.1f * g;
    const float vi = 0.999f * v_old + 0.001f * g * g;
    m_out = mi;
    v_out = vi;
    const float bc1 = 1.0f - powf(0.9f, (float)t), bc2 = 1.0f - powf(0.999f, (float)t);
    p = p_old - (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + 1e-8f);
    if (i >= 3 && i < 7) atomicAdd(&qnorm2, p * p);
  } else if (i < n) {
    p = p_old;
  }
  __syncthreads();
  p_out = (i >= 3 && i < 7) ? p / sqrtf(qnorm2) : p;  // :462 renormalise the quaternion
  if (i < n && params_lds) params_lds[i] = p_out;
}
__device__ __forceinline__ void adam_step_block(float* __restrict__ params, const float* __restrict__ grads,
                                                float* __restrict__ m, float* __restrict__ v,
                                                int* __restrict__ step, int n, float lr_pos, float lr_quat,
                                                float lr_scale, float lr_latent, int update_latent) {
  const int i = threadIdx.x;
  const bool mine = i < n;
  const int step_old = step[0];
  float p, mi, vi;
  adam_step_core(grads, n, lr_pos, lr_quat, lr_scale, lr_latent, update_latent, mine ? params[i] : 0.0f,
                 mine ? m[i] : 0.0f, mine ? v[i] : 0.0f, step_old, nullptr, p, mi, vi);
  if (mine) { params[i] = p; m[i] = mi; v[i] = vi; }
  if (i == 0) step[0] = step_old + 1;
}
__global__ void adam_step_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                 float* __restrict__ m, float* __restrict__ v,
                                 int* __restrict__ step, int n, float lr_pos, float lr_quat,
                                 float lr_scale, float lr_latent, int update_latent) {
  adam_step_block(params, grads, m, v, step, n, lr_pos, lr_quat, lr_scale, lr_latent, update_latent);
}

// The tail of one iteration of the captured loop in ONE launch (one workgroup): the gradient chain with the two
// per-view reductions (deferred_chain), the point constraint, Adam on the 8 + L parameters, and the camera-frame
// poses of the NEXT iteration (pose_to_view) -- four launches of ~5.7 us each in the replayed graph, none of which
// can fill more than one workgroup.  Same functions, same order, same numbers as the separate launches.
struct LoopTailArgs {
  float* params; float* grads; float* m; float* v; int* step; int n;
  float lr_pos, lr_quat, lr_scale, lr_latent; int update_latent;
  const float* cam_pos; const float* cam_quat; int V;
  const ViewSetup* setup; const float* tile_part; int W, H, ntx, nty, tile_w, tile_h, stride;
  const float* pc_part; const float* pc_loss_part; const int* offsets; int n_single, nblk;
  float* pos_c; float* quat_c; float* inv_scale; float* scale_v; float* pc_loss;
  const float* con_source; const float* con_target; float con_weight; float* con_loss;
  const float* dec_params; const float* t_mid; FcDesc fc;   // t_mid != NULL: the decoder VJP's last stage runs here
  // records != NULL (sdfr_loop_tail_records): the chain runs over the V_all exchanged view records instead of this
  // rank's partials; cam_pos / cam_quat / V above are then this rank's shard (the next iteration's view poses)
  const float* records; int V_all; const float* cam_quat_all;
  int fc_one_wave;   // the Linear stack is narrow enough for fc_stack_backward_one_wave (decoder_fc_one_wave)
  // sdfr_loop_tail_objects: workgroup k is object k -- its own parameters, moments, step counter and V views (the
  // launch's views k V .. k V + V - 1), the same cameras for every object; 0 / 1: the single estimate of sdfr_loop_tail
  int n_obj;
  float* latents;   // objects: the updated latents, packed [n_obj][n - 8] (the next iteration's batched decode), or NULL
  FusedDepth fd;    // sdfr_loop_tail_fused: the render step was ONE launch (its depth term comes unscaled)
  // sdfr_loop_tail_fused(decoder_tape): gridDim.x workgroups, each leaves 256 outputs of the decoder's wide Linear layer
  // for the NEXT iteration's latent in fc_next; `arrivals` counts the workgroups that have read the state
  float* fc_next;
  unsigned* arrivals;
};
constexpr int kTailArrivalPolls = 1 << 16;   // bound of workgroup 0's wait (~50 ms: a schedule that never comes)
static_assert(kFcBlock == 256, "the tail's workgroup runs the decoder's Linear-stack backward");
#ifdef SDFR_TAIL_STAMPS   // timing experiment (tools/microbench): where the tail's time goes, in 10 ns ticks
__device__ unsigned long long g_tail_stamps[8];
#define SDFR_STAMP(k) do { if (threadIdx.x == 0) g_tail_stamps[k] = wall_clock64(); } while (0)
#else
#define SDFR_STAMP(k) do { } while (0)
#endif
// A dependent chain by construction -- what it pays is memory round trips (~0.5 us each), not arithmetic: 14.7 us as
// first written (tools/microbench/tail_stamps.py: 8.0 us in the Linear stack's backward, 4.6 us in the per-view
// reductions and the chain, 1.4 us Adam, 0.6 us the next poses).  Hence (a) everything that depends on nothing
// computed here is loaded FIRST (parameters, Adam's moments and step count, the first 256 views' camera poses, the
// constraint's points); (b) a narrow Linear stack is one wave's work out of LDS (fc_stack_backward_one_wave), and the
// other three waves reduce the views meanwhile; (c) gradients and the updated parameters pass between the stages in
// LDS (g_l, p_new) instead of through global memory.  Same arithmetic, same order: the numbers do not change.
__global__ __launch_bounds__(256) void loop_tail_kernel(LoopTailArgs a_in) {
  __shared__ float g_l[256];      // the gradients, laid out as a.grads (copied there at the end)
  __shared__ float p_cur[256];    // the parameters of this iteration ...
  __shared__ float p_new[256];    // ... and after the Adam step
  __shared__ FcWaveLds fc_lds;    // the decoder's leading Linear layers (one-wave forms)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const LoopTailArgs& a = a_in;
  // the next iteration's Linear stack in this launch (fc_next): workgroup 0 is the tail proper, the others repeat the
  // latent's share of it -- gradient, Adam step: same inputs, same arithmetic -- and every workgroup forms its 256
  // outputs of the wide layer from the updated latent.  Nothing passes between workgroups; the one thing to keep apart
  // is workgroup 0's stores of the new state and the others' loads of the old one (arrivals).
  const bool helper = a.fc_next != nullptr && blockIdx.x != 0;
  // (the object's own slices as plain locals: a modified copy of the argument block would live in scratch memory)
  const int view_base = a.n_obj > 1 ? (int)blockIdx.x * a.V : 0;
  const size_t obj_o = a.n_obj > 1 ? (size_t)blockIdx.x * a.n : 0;
  float* const params = a.params + obj_o;
  float* const grads = a.grads + obj_o;
  float* const mom1 = a.m + obj_o;
  float* const mom2 = a.v + obj_o;
  int* const step = a.step + (a.n_obj > 1 ? blockIdx.x : 0);
  float* const pos_c = a.pos_c + 3 * view_base;
  float* const quat_c = a.quat_c + 4 * view_base;
  float* const inv_scale = a.inv_scale + view_base;
  float* const scale_v = a.scale_v + view_base;
  SDFR_STAMP(0);
  const bool mine = tid < a.n;
  const float p_old = mine ? params[tid] : 0.0f, m_old = mine ? mom1[tid] : 0.0f, v_old = mine ? mom2[tid] : 0.0f;
  const int step_old = step[0];
  const float g_given = (mine && tid >= 8 && !a.t_mid) ? grads[tid] : 0.0f;   // the latent's gradient is the caller's
  float cp[3] = {0, 0, 0}, cq[4] = {0, 0, 0, 1};
  if (tid < a.V) {
    cp[0] = a.cam_pos[3 * tid]; cp[1] = a.cam_pos[3 * tid + 1]; cp[2] = a.cam_pos[3 * tid + 2];
    cq[0] = a.cam_quat[4 * tid]; cq[1] = a.cam_quat[4 * tid + 1]; cq[2] = a.cam_quat[4 * tid + 2];
    cq[3] = a.cam_quat[4 * tid + 3];
  }
  float con[6] = {0, 0, 0, 0, 0, 0};
  if (tid == 0 && a.con_source) {
    for (int k = 0; k < 3; ++k) { con[k] = a.con_source[k]; con[3 + k] = a.con_target[k]; }
  }
  const bool fc_wave = a.t_mid && a.fc_one_wave;
  // (objects: the batched VJP left one row per object)
  const float* const t_mid = a.t_mid ? a.t_mid + (a.n_obj > 1 ? (size_t)blockIdx.x * a.fc.width[a.fc.n_fc - 1] : 0) : nullptr;
  if (a.t_mid && !fc_wave) {   // d loss / d latent (g[8 ...]) from the gradient w.r.t. the wide Linear layer's input
    fc_stack_backward_sample(a.dec_params, a.fc, params + 8, t_mid, g_l + 8);
    __syncthreads();
  }
  // (the leading layers' parameters staged by all four waves behind a barrier instead of by wave 0 alone: measured,
  // the tail 10.3 -> 11.9 us -- the barrier costs the chain more than the shorter copy gives back)
  if (fc_wave && wave == 0) fc_stack_backward_one_wave(fc_lds, a.dec_params, a.fc, params + 8, t_mid, g_l + 8, lane);
  // (only now: an LDS store of a loaded value waits for the load, and the Linear stack's loads should not queue
  // behind that wait)
  if (mine) p_cur[tid] = p_old;
  if (mine && tid >= 8 && !a.t_mid) g_l[tid] = g_given;
  SDFR_STAMP(1);
  if (helper) {
    if (tid < 8) g_l[tid] = 0.0f;   // (the pose's share of the step is workgroup 0's)
    __builtin_amdgcn_s_waitcnt(0);  // this thread's loads of the state have landed (moments, step count: registers only)
    __syncthreads();                // ... and every other thread's
    if (tid == 0) atomicAdd(a.arrivals, 1u);
  } else if (a.records) {
    pose_chain_block(params + 3, params + 7, a.cam_quat_all, a.V_all, a.records, kViewRecord, g_l, g_l + 3,
                     g_l + 7);
  } else {
    // (the chain indexes the LAUNCH's views: the un-advanced quat_c / pc_loss and the object's first view)
    deferred_chain(p_cur + 3, p_cur + 7, a.cam_quat, a.V, a.setup, a.tile_part, a.W, a.H, a.ntx, a.nty,
                   a.tile_w, a.tile_h, a.stride, a.pc_part, a.pc_loss_part, a.offsets, a.n_single, a.nblk,
                   a.quat_c, a.pc_loss, g_l, g_l + 3, g_l + 7, view_base, a.fd);
  }
  SDFR_STAMP(2);
  if (!helper && tid == 0 && a.con_source)   // (p_cur: thread 0 has passed a barrier of the chain above since it was written)
    point_constraint_one(p_cur + 3, con, con + 3, a.con_weight, a.con_loss, g_l + 3);
  __syncthreads();   // the gradients (thread 0's, wave 0's) are visible to the Adam threads
  SDFR_STAMP(3);
  float p_i, m_i, v_i;
  adam_step_core(g_l, a.n, a.lr_pos, a.lr_quat, a.lr_scale, a.lr_latent, a.update_latent, p_old, m_old, v_old,
                 step_old, p_new, p_i, m_i, v_i);
  __syncthreads();   // the updated parameters are visible to the pose chain
  SDFR_STAMP(4);
  if (a.fc_next) {
    // the updated latent through the leading layers (wave 0, out of the LDS copy the backward staged), then this
    // workgroup's slice of the wide layer: fc_stack_kernel's arithmetic
    unsigned seen = 0;
    const unsigned want = (unsigned)(step_old + 1) * (gridDim.x - 1);   // (the word counts up over a run's iterations)
    if (!helper && tid == 0) seen = __hip_atomic_load(a.arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      if (lane < a.fc.width[0]) fc_lds.a[0][lane] = p_new[8 + lane];
      __builtin_amdgcn_wave_barrier();
      fc_narrow_forward_one_wave(fc_lds, a.fc, lane);
    }
    __syncthreads();
    // (the thread's column of the wide layer's weights preloaded into 64 registers -- at the kernel's start, or behind
    // the chain's last load, under the Adam step -- was measured: the tail 10.3 -> 13.6 / 13.9 us)
    fc_wide_slice(a.dec_params, a.fc, fc_lds.a[a.fc.n_fc - 1], (int)blockIdx.x, a.fc_next);
    if (helper) return;
    // workgroup 0 stores the new state below: not before every other workgroup has read the old one
    if (tid == 0) {
      for (int polls = 0; seen < want && polls < kTailArrivalPolls; ++polls)
        seen = __hip_atomic_load(a.arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (tid < a.V)
    pose_to_view_core(tid, p_new, p_new + 3, p_new + 7, cp, cq, pos_c, quat_c, inv_scale, scale_v);
  for (int v = tid + 256; v < a.V; v += 256)
    pose_to_view(v, p_new, p_new + 3, p_new + 7, a.cam_pos, a.cam_quat, pos_c, quat_c, inv_scale, scale_v);
  // the stores nothing in this launch waits for
  if (mine) {
    grads[tid] = g_l[tid];
    params[tid] = p_i;
    mom1[tid] = m_i;
    mom2[tid] = v_i;
    if (a.latents && tid >= 8) a.latents[(size_t)blockIdx.x * (a.n - 8) + tid - 8] = p_i;
  }
  if (tid == 0) step[0] = step_old + 1;
  SDFR_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// depth image(s) -> packed point set (pointset_utils.depth_to_pointcloud :57-77, convention
// "opengl", pixel-centre-0 intrinsics): every non-zero pixel, view-major, row-major.
// Two passes over the images instead of torch.nonzero + gathers: count per block of 1024 pixels,
// then a stable in-block compaction at the block's offset.
// ---------------------------------------------------------------------------------------------
constexpr int kCompactPix = 1024;  // pixels per workgroup: 256 threads x 4 consecutive pixels

// The order in which the pixels are enumerated (and so the order of a view's points).  Row-major is the
// reference's (torch.nonzero).  TILED: the image in tiles of 64 x 16 pixels (one workgroup each), a tile in four
// sub-tiles of 16 x 16, a sub-tile row-major -- 256 consecutive points are then a compact patch of the surface,
// and the sampler's backward, which pre-sums the d/dSDF of 256 consecutive points in LDS, meets each voxel in fewer
// blocks: 0.55 -> ~0.3 global atomics per point on back-projected depth images.  For consumers that only SUM over a
// view's points (the loss-fused loop); the thread's 4 consecutive pixels are 4 columns of one row in both orders.
struct PixelOrder {
  int W, H, tiled, tiles_x;
  __host__ __device__ int count() const {
    return tiled ? tiles_x * ((H + 15) / 16) * kCompactPix : W * H;
  }
  // first of the thread's 4 pixels -> (row, col); the 4 are (row, col .. col + 3)
  __device__ __forceinline__ void at(int e, int& row, int& col) const {
    if (!tiled) { row = e / W; col = e - row * W; return; }
    const int tile = e >> 10, in = e & 1023;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    row = ty * 16 + ((in >> 4) & 15);
    col = tx * 64 + (in >> 8) * 16 + (in & 15);
  }
};
inline PixelOrder pixel_order(int W, int H, int tiled) { return PixelOrder{W, H, tiled, (W + 63) / 64}; }

// grid (nblk, V): block_count[v][blk] and (atomically) count[v]
// the thread's 4 depth values: row-major order may run over the end of a row (4 consecutive LINEAR pixels), the
// tiled order stays in its row
__device__ __forceinline__ void load_four(const float* __restrict__ img, const PixelOrder& po, int p0, float (&z)[4],
                                          int& row, int& col) {
  if (!po.tiled) {
    const int npix = po.W * po.H;
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = (p0 + k < npix) ? img[p0 + k] : 0.0f;
    row = col = 0;
    return;
  }
  po.at(p0, row, col);
#pragma unroll
  for (int k = 0; k < 4; ++k) z[k] = (row < po.H && col + k < po.W) ? img[(size_t)row * po.W + col + k] : 0.0f;
}

// SUMS: also the block's sums of the back-projected coordinates (block_sum [v][blk][4]: x, y, z, unused), for the
// view's centroid (generated_dataset.py:318-320 normalises the point set by its mean) -- fixed order: lanes, waves.
template <bool SUMS>
__global__ __launch_bounds__(256) void depth_count_kernel(const float* __restrict__ depth, PixelOrder po, int nblk,
                                                          int* __restrict__ block_count,
                                                          int* __restrict__ count, float rfx, float rfy, float cx0,
                                                          float cy0, float* __restrict__ block_sum) {
  __shared__ int wsum[4];
  __shared__ float wxyz[4][3];
  const int v = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const float* img = depth + (size_t)v * po.W * po.H;
  const int p0 = blk * kCompactPix + tid * 4;
  int c = 0;
  float z4[4];
  int row, col;
  load_four(img, po, p0, z4, row, col);
#pragma unroll
  for (int k = 0; k < 4; ++k) c += (z4[k] != 0.0f) ? 1 : 0;
  if (SUMS) {
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (z4[k] == 0.0f) continue;
      const int p = p0 + k;
      const int r = po.tiled ? row : p / po.W, cc = po.tiled ? col + k : p - (p / po.W) * po.W;
      sx += ((float)cc - cx0) * z4[k] * rfx;     // the expressions of depth_compact_kernel
      sy += -((float)r - cy0) * z4[k] * rfy;
      sz += -z4[k];
    }
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    if ((tid & 63) == 0) { wxyz[tid >> 6][0] = sx; wxyz[tid >> 6][1] = sy; wxyz[tid >> 6][2] = sz; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) {
    const int t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    block_count[(size_t)v * nblk + blk] = t;
    if (t && count) atomicAdd(&count[v], t);
  }
  if (SUMS && tid < 3)
    block_sum[((size_t)v * nblk + blk) * 4 + tid] = (wxyz[0][tid] + wxyz[1][tid]) + (wxyz[2][tid] + wxyz[3][tid]);
}

// one wave: offsets[v] = number of points of the views before v (exclusive prefix of the counts), 64 views per round
__global__ __launch_bounds__(64) void count_prefix_kernel(const int* __restrict__ count, int V, int* __restrict__ offsets) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int v0 = 0; v0 < V; v0 += 64) {
    const int v = v0 + lane;
    const int c = v < V ? count[v] : 0;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (v < V) offsets[v] = base + incl - c;
    base += __shfl(incl, 63, 64);
  }
}

// The same without a host in between (sdfr_depth_to_points_resident): ONE workgroup; wave w sums the block counts of the
// views w, w + 4, ... (integers: any order) into count[v], wave 0 then writes the exclusive prefix offsets[0 .. V]
// (V + 1 entries: offsets[V] = all points), 64 views per round.
__global__ __launch_bounds__(256) void resident_offsets_kernel(const int* __restrict__ block_count, int nblk, int V,
                                                               int* __restrict__ count, int* __restrict__ offsets) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int v = wave; v < V; v += 4) {
    int s = 0;
    for (int i = lane; i < nblk; i += 64) s += block_count[(size_t)v * nblk + i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) count[v] = s;
  }
  __threadfence_block();
  __syncthreads();
  if (wave != 0) return;
  int base = 0;
  for (int v0 = 0; v0 < V; v0 += 64) {
    const int v = v0 + lane;
    const int c = v < V ? count[v] : 0;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (v < V) offsets[v] = base + incl - c;
    base += __shfl(incl, 63, 64);
  }
  if (lane == 0) offsets[V] = base;
}

// SDFPipeline._preprocess_depth (simple_setup.py:671-693), in place: depth[~mask] = 0, then (has_far) depth[depth >
// far_field] = 0 -- a NaN depth is not "> far_field" and stays, as in torch.  Four pixels per thread; a value is stored
// only where it changes.  copy_to (nullable): the preprocessed images also go there (the loop's own target buffer).
__global__ __launch_bounds__(256) void preprocess_depth_kernel(float* __restrict__ depth,
                                                               const unsigned char* __restrict__ mask, size_t n,
                                                               float far_field, int has_far, int vec,
                                                               float* __restrict__ copy_to) {
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  float z[4];
  unsigned char m[4];
  const bool full = vec && i0 + 4 <= n;
  if (full) {
    const float4 q = *reinterpret_cast<const float4*>(depth + i0);
    const uchar4 mm = *reinterpret_cast<const uchar4*>(mask + i0);
    z[0] = q.x; z[1] = q.y; z[2] = q.z; z[3] = q.w;
    m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[k] = (i0 + k < n) ? depth[i0 + k] : 0.0f;
      m[k] = (i0 + k < n) ? mask[i0 + k] : (unsigned char)1;
    }
  }
  float o[4];
  bool changed = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t = m[k] ? z[k] : 0.0f;
    if (has_far && t > far_field) t = 0.0f;
    o[k] = t;
    changed = changed || (__float_as_uint(t) != __float_as_uint(z[k]));
  }
  if (full) {
    const float4 q = make_float4(o[0], o[1], o[2], o[3]);
    if (changed) *reinterpret_cast<float4*>(depth + i0) = q;
    if (copy_to) *reinterpret_cast<float4*>(copy_to + i0) = q;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k >= n) break;
      if (__float_as_uint(o[k]) != __float_as_uint(z[k])) depth[i0 + k] = o[k];
      if (copy_to) copy_to[i0 + k] = o[k];
    }
  }
}

// one wave per view: centroid[v] = (sum of the block sums, lane-strided then a butterfly) / count; 0 for an empty view
__global__ __launch_bounds__(64) void centroid_reduce_kernel(const float* __restrict__ block_sum, int nblk,
                                                             const int* __restrict__ count,
                                                             float* __restrict__ centroid) {
  const int v = blockIdx.x, lane = threadIdx.x;
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  for (int i = lane; i < nblk; i += 64) {
    const float4 q = *reinterpret_cast<const float4*>(block_sum + ((size_t)v * nblk + i) * 4);
    sx += q.x; sy += q.y; sz += q.z;
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  if (lane == 0) {
    const int n = count[v];
    const float inv = n > 0 ? 1.0f / (float)n : 0.0f;
    centroid[3 * v] = n > 0 ? sx / (float)n : 0.0f;
    centroid[3 * v + 1] = n > 0 ? sy / (float)n : 0.0f;
    centroid[3 * v + 2] = n > 0 ? sz / (float)n : 0.0f;
    (void)inv;
  }
}

// grid (nblk, V): points[offsets[v] + rank of the pixel within its view] = back-projection
__global__ __launch_bounds__(256) void depth_compact_kernel(const float* __restrict__ depth, PixelOrder po,
                                                            int nblk, const int* __restrict__ block_count,
                                                            const int* __restrict__ offsets, float rfx,
                                                            float rfy, float cx0, float cy0,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ noise,
                                                            float* __restrict__ points) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int v = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* bc = block_count + (size_t)v * nblk;
  if (bc[blk] == 0) return;  // workgroup-uniform
  // pixels of this view in earlier blocks (one wave sums the block counts in a fixed order)
  if (wave == 0) {
    int s = 0;
    for (int i = lane; i < blk; i += 64) s += bc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) base_s = offsets[v] + s;
  }
  const int W = po.W;
  const float* img = depth + (size_t)v * po.W * po.H;
  const int p0 = blk * kCompactPix + tid * 4;
  float z[4];
  int c = 0, trow, tcol;
  load_four(img, po, p0, z, trow, tcol);
#pragma unroll
  for (int k = 0; k < 4; ++k) c += (z[k] != 0.0f) ? 1 : 0;
  // exclusive scan of c over the workgroup: within the wave, then across the four waves
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int before = incl - c;
  for (int w = 0; w < wave; ++w) before += wsum[w];
  float* out = points + 3 * (size_t)(base_s + before);
  // (x - 0.0f == x bit for bit, so the plain form is this one with a zero shift)
  const V3 sh = shift ? mk(shift[3 * v], shift[3 * v + 1], shift[3 * v + 2]) : mk(0.0f, 0.0f, 0.0f);
  // (noise: a SECOND rounding, as the reference's `pointset -= centroid; pointset += noise`, generated_dataset.py:314-326)
  const V3 nz = noise ? mk(noise[3 * v], noise[3 * v + 1], noise[3 * v + 2]) : mk(0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // (torch divides by a scalar as "* (1 / scalar)" on the GPU.  Every operation rounds on its own, as torch's
    // separate kernels do -- no contraction into fused multiply-adds -- so the points, their centred and their
    // noised forms are the reference's bit for bit)
#pragma clang fp contract(off)
    if (z[k] == 0.0f) continue;
    const int p = p0 + k;
    const int row = po.tiled ? trow : p / W, col = po.tiled ? tcol + k : p - (p / W) * W;
    float x = (((float)col - cx0) * z[k]) * rfx;
    float y = (-((float)row - cy0) * z[k]) * rfy;
    float zz = -z[k];
    x = x - sh.x; y = y - sh.y; zz = zz - sh.z;
    if (noise) { x = x + nz.x; y = y + nz.y; zz = zz + nz.z; }
    out[0] = x; out[1] = y; out[2] = zz;
    out += 3;
  }
}

}  // namespace
}  // namespace sdfr

using namespace sdfr;

extern "C" int sdfr_pose_to_views(const float* position, const float* orientation, const float* scale,
                                  const float* cam_pos, const float* cam_quat, int V, float* pos_c,
                                  float* quat_c, float* inv_scale, float* scale_v, int device,
                                  void* stream) {
  if (V < 0) return fail(SDFR_E_INVALID, "V=%d is negative", V);
  if (V == 0) return 0;
  if (!position || !orientation || !scale || !cam_pos || !cam_quat || !pos_c || !quat_c || !inv_scale || !scale_v)
    return fail(SDFR_E_NULL, "sdfr_pose_to_views: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(pose_to_views_kernel, dim3((V + 63) / 64), dim3(64), 0, (hipStream_t)stream, position,
                     orientation, scale, cam_pos, cam_quat, V, pos_c, quat_c, inv_scale, scale_v);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_pose_to_views_objects(const float* params, int n_params, int n_objects, const float* cam_pos,
                                          const float* cam_quat, int V, float* pos_c, float* quat_c, float* inv_scale,
                                          float* scale_v, float* latents, int device, void* stream) {
  if (V < 0 || n_objects < 0 || n_params < 8 || (long long)V * n_objects > 0x7fffffffLL)
    return fail(SDFR_E_INVALID, "sdfr_pose_to_views_objects: bad sizes V=%d objects=%d n_params=%d", V, n_objects, n_params);
  if (V == 0 || n_objects == 0) return 0;
  if (!params || !cam_pos || !cam_quat || !pos_c || !quat_c || !inv_scale || !scale_v)
    return fail(SDFR_E_NULL, "sdfr_pose_to_views_objects: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  const int total = V * n_objects;
  hipLaunchKernelGGL(pose_to_views_objects_kernel, dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, params,
                     n_params, n_objects, cam_pos, cam_quat, V, pos_c, quat_c, inv_scale, scale_v, latents);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_views_to_pose_grad(const float* orientation, const float* scale,
                                       const float* cam_quat, int V, const float* ga_pos,
                                       const float* ga_quat, const float* ga_inv_scale,
                                       const float* gb_pos, const float* gb_quat,
                                       const float* gb_scale, float* g_position,
                                       float* g_orientation, float* g_scale, int device,
                                       void* stream) {
  if (V < 0) return fail(SDFR_E_INVALID, "V=%d is negative", V);
  if (!orientation || !scale || !cam_quat || !g_position || !g_orientation || !g_scale)
    return fail(SDFR_E_NULL, "sdfr_views_to_pose_grad: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(views_to_pose_grad_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, orientation,
                     scale, cam_quat, V, ga_pos, ga_quat, ga_inv_scale, gb_pos, gb_quat, gb_scale,
                     g_position, g_orientation, g_scale);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_views_to_pose_grad_deferred(const float* orientation, const float* scale, const float* cam_quat,
                                                int V, const void* render_workspace, int W, int H,
                                                const void* pc_workspace, const int* offsets, int max_view_points,
                                                const float* quat_c, float* pc_loss, float* g_position,
                                                float* g_orientation, float* g_scale, int device, void* stream) {
  if (V < 0 || V > kDeferredMaxViews)
    return fail(SDFR_E_INVALID, "sdfr_views_to_pose_grad_deferred: V=%d out of range [0,%d]", V, kDeferredMaxViews);
  if (!orientation || !scale || !cam_quat || !g_position || !g_orientation || !g_scale)
    return fail(SDFR_E_NULL, "sdfr_views_to_pose_grad_deferred: NULL pointer argument");
  if (render_workspace && (W <= 0 || H <= 0))
    return fail(SDFR_E_INVALID, "sdfr_views_to_pose_grad_deferred: W=%d H=%d", W, H);
  if (pc_workspace && (max_view_points <= 0 || !quat_c || (!offsets && V > 1)))
    return fail(SDFR_E_INVALID, "sdfr_views_to_pose_grad_deferred: bad sampler arguments");
  if (render_workspace && (uintptr_t)render_workspace % alignof(ViewSetup))
    return fail(SDFR_E_INVALID, "render_workspace must be %zu-byte aligned", alignof(ViewSetup));
  SDFR_HIP_TRY(hipSetDevice(device));
  // the layouts sdfr_render_backward* and sdfr_pc_*_backward* leave behind (render.hip backward_impl,
  // sampler.hip pc_backward_impl)
  const ViewSetup* setup = (const ViewSetup*)render_workspace;
  const float* tile_part = render_workspace ? (const float*)((const char*)render_workspace + scratch_offset_bytes(V, H)) : nullptr;
  const TileGeom geom = render_workspace ? backward_geom(V, W, H) : kSmallTile;
  const int ntx = render_workspace ? geom.nx(W) : 0, nty = render_workspace ? geom.ny(H) : 0;
  const int stride = (render_workspace && geom.sx * geom.sy > 1) ? backward_tile_stride(W, H) : 0;
  const int nblk = pc_workspace ? (max_view_points + kSamplerPts - 1) / kSamplerPts : 0;
  const float* pc_part = (const float*)pc_workspace;
  const float* pc_loss_part = (pc_workspace && pc_loss) ? pc_part + (size_t)V * nblk * 8 : nullptr;
  hipLaunchKernelGGL(views_to_pose_grad_deferred_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, orientation,
                     scale, cam_quat, V, setup, tile_part, W, H, ntx, nty, geom.w(), geom.h(), stride, pc_part,
                     pc_loss_part,
                     offsets, max_view_points, nblk, quat_c, pc_loss, g_position, g_orientation, g_scale);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

namespace {
int loop_tail_impl(const char* fn, float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step, int n_params,
                   float lr_position, float lr_orientation, float lr_scale, float lr_latent,
                   int update_latent, const float* cam_pos, const float* cam_quat, int V,
                   const void* render_workspace, size_t render_partials_offset, int W, int H,
                   const void* pc_workspace, const int* offsets, int max_view_points, float* pos_c,
                   float* quat_c, float* inv_scale, float* scale_v, float* pc_loss,
                   const float* con_source, const float* con_target, float con_weight, float* con_loss,
                   const sdfr_decoder* decoder, const float* decoder_t_mid, int device, void* stream,
                   const FusedDepth& fd, float* decoder_tape = nullptr, unsigned* arrivals = nullptr) {
  if ((decoder != nullptr) != (decoder_t_mid != nullptr))
    return fail(SDFR_E_NULL, "%s: decoder and decoder_t_mid go together", fn);
  if (V < 1 || V > kDeferredMaxViews) return fail(SDFR_E_INVALID, "%s: V=%d out of range [1,%d]", fn, V, kDeferredMaxViews);
  if (n_params < 8 || n_params > 256) return fail(SDFR_E_INVALID, "%s: n_params=%d out of range [8,256]", fn, n_params);
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step || !cam_pos || !cam_quat || !pos_c || !quat_c ||
      !inv_scale || !scale_v)
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (render_workspace && (W <= 0 || H <= 0)) return fail(SDFR_E_INVALID, "%s: W=%d H=%d", fn, W, H);
  if (pc_workspace && (max_view_points <= 0 || (!offsets && V > 1)))
    return fail(SDFR_E_INVALID, "%s: bad sampler arguments", fn);
  if (con_source && !con_target) return fail(SDFR_E_NULL, "%s: constraint target is NULL", fn);
  if (render_workspace && ((uintptr_t)render_workspace % alignof(ViewSetup) || render_partials_offset % 16))
    return fail(SDFR_E_INVALID, "%s: render_workspace / partials offset misaligned", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  // (the one-launch render step always works in 32 x 8 tiles)
  const TileGeom geom = (render_workspace && !fd.view_cnt) ? backward_geom(V, W, H) : kSmallTile;
  const int nblk = pc_workspace ? (max_view_points + kSamplerPts - 1) / kSamplerPts : 0;
  const float* pc_part = (const float*)pc_workspace;
  LoopTailArgs a{};
  a.fd = fd;
  a.params = params; a.grads = grads; a.m = exp_avg; a.v = exp_avg_sq; a.step = step; a.n = n_params;
  a.lr_pos = lr_position; a.lr_quat = lr_orientation; a.lr_scale = lr_scale; a.lr_latent = lr_latent;
  a.update_latent = update_latent;
  a.cam_pos = cam_pos; a.cam_quat = cam_quat; a.V = V;
  a.setup = (const ViewSetup*)render_workspace;
  a.tile_part = render_workspace ? (const float*)((const char*)render_workspace + render_partials_offset) : nullptr;
  a.W = W; a.H = H;
  a.ntx = render_workspace ? geom.nx(W) : 0; a.nty = render_workspace ? geom.ny(H) : 0;
  a.tile_w = geom.w(); a.tile_h = geom.h();
  a.stride = (render_workspace && geom.sx * geom.sy > 1) ? backward_tile_stride(W, H) : 0;
  a.pc_part = pc_part;
  a.pc_loss_part = (pc_workspace && pc_loss) ? pc_part + (size_t)V * nblk * 8 : nullptr;
  a.offsets = offsets; a.n_single = max_view_points; a.nblk = nblk;
  a.pos_c = pos_c; a.quat_c = quat_c; a.inv_scale = inv_scale; a.scale_v = scale_v; a.pc_loss = pc_loss;
  a.con_source = con_source; a.con_target = con_target; a.con_weight = con_weight; a.con_loss = con_loss;
  if (decoder) {
    decoder_fc_desc(decoder, &a.fc, &a.dec_params, nullptr);
    a.fc_one_wave = decoder_fc_one_wave(decoder, a.fc) ? 1 : 0;
    if (a.fc.width[0] != n_params - 8)
      return fail(SDFR_E_INVALID, "%s: the decoder's latent has %d entries, the parameter vector %d", fn, a.fc.width[0],
                  n_params - 8);
    a.t_mid = decoder_t_mid;
  }
  a.n_obj = 1;
  unsigned nwg = 1;
  if (decoder_tape || arrivals) {
    if (!decoder_tape || !arrivals) return fail(SDFR_E_NULL, "%s: decoder_tape and arrivals go together", fn);
    if (!decoder || !update_latent || !a.fc_one_wave)
      return fail(SDFR_E_INVALID, "%s: the next Linear stack in this launch needs a decoder whose leading layers are "
                  "at most %d wide (and its one-wave form switched on), and update_latent", fn, kFcWaveWidth);
    size_t tape_fc_off = 0;
    decoder_fc_desc(decoder, &a.fc, &a.dec_params, &tape_fc_off);
    a.fc_next = decoder_tape + tape_fc_off;
    a.arrivals = arrivals;
    nwg = (unsigned)((a.fc.width[a.fc.n_fc] + kFcBlock - 1) / kFcBlock);
  }
  hipLaunchKernelGGL(loop_tail_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, a);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int sdfr_loop_tail(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step, int n_params,
                              float lr_position, float lr_orientation, float lr_scale, float lr_latent,
                              int update_latent, const float* cam_pos, const float* cam_quat, int V,
                              const void* render_workspace, size_t render_partials_offset, int W, int H,
                              const void* pc_workspace, const int* offsets, int max_view_points, float* pos_c,
                              float* quat_c, float* inv_scale, float* scale_v, float* pc_loss,
                              const float* con_source, const float* con_target, float con_weight, float* con_loss,
                              const sdfr_decoder* decoder, const float* decoder_t_mid, int device, void* stream) {
  return loop_tail_impl("sdfr_loop_tail", params, grads, exp_avg, exp_avg_sq, step, n_params, lr_position,
                        lr_orientation, lr_scale, lr_latent, update_latent, cam_pos, cam_quat, V, render_workspace,
                        render_partials_offset, W, H, pc_workspace, offsets, max_view_points, pos_c, quat_c, inv_scale,
                        scale_v, pc_loss, con_source, con_target, con_weight, con_loss, decoder, decoder_t_mid, device,
                        stream, FusedDepth{});
}

extern "C" int sdfr_loop_tail_fused(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step,
                                    int n_params, float lr_position, float lr_orientation, float lr_scale,
                                    float lr_latent, int update_latent, const float* cam_pos, const float* cam_quat,
                                    int V, void* render_workspace, size_t render_partials_offset,
                                    size_t view_count_offset, size_t tile_loss_offset, float depth_weight,
                                    float* depth_loss, int W, int H, const void* pc_workspace, const int* offsets,
                                    int max_view_points, float* pos_c, float* quat_c, float* inv_scale, float* scale_v,
                                    float* pc_loss, const float* con_source, const float* con_target, float con_weight,
                                    float* con_loss, const sdfr_decoder* decoder, const float* decoder_t_mid,
                                    float* decoder_tape, unsigned* arrivals, int device, void* stream) {
  const char* fn = "sdfr_loop_tail_fused";
  if (!render_workspace) return fail(SDFR_E_NULL, "%s: render_workspace is NULL", fn);
  if (view_count_offset % 4 || tile_loss_offset % 16)
    return fail(SDFR_E_INVALID, "%s: view count / tile loss offsets misaligned", fn);
  FusedDepth fd;
  fd.view_cnt = (float*)((char*)render_workspace + view_count_offset);
  fd.tile_loss = (const float*)((const char*)render_workspace + tile_loss_offset);
  fd.weight = depth_weight;
  fd.depth_loss = depth_loss;
  return loop_tail_impl(fn, params, grads, exp_avg, exp_avg_sq, step, n_params, lr_position, lr_orientation, lr_scale,
                        lr_latent, update_latent, cam_pos, cam_quat, V, render_workspace, render_partials_offset, W, H,
                        pc_workspace, offsets, max_view_points, pos_c, quat_c, inv_scale, scale_v, pc_loss, con_source,
                        con_target, con_weight, con_loss, decoder, decoder_t_mid, device, stream, fd, decoder_tape,
                        arrivals);
}

extern "C" int sdfr_loop_tail_objects(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step,
                                      int n_params, int n_objects, float lr_position, float lr_orientation,
                                      float lr_scale, float lr_latent, int update_latent, const float* cam_pos,
                                      const float* cam_quat, int V, const void* render_workspace,
                                      size_t render_partials_offset, int W, int H, const void* pc_workspace,
                                      const int* offsets, int max_view_points, float* pos_c, float* quat_c,
                                      float* inv_scale, float* scale_v, float* pc_loss, const sdfr_decoder* decoder,
                                      const float* decoder_t_mid, float* latents, int device, void* stream) {
  const char* fn = "sdfr_loop_tail_objects";
  if ((decoder != nullptr) != (decoder_t_mid != nullptr))
    return fail(SDFR_E_NULL, "%s: decoder and decoder_t_mid go together", fn);
  if (n_objects < 1 || n_objects > 65535) return fail(SDFR_E_INVALID, "%s: n_objects=%d", fn, n_objects);
  if (V < 1 || V > kDeferredMaxViews) return fail(SDFR_E_INVALID, "%s: V=%d out of range [1,%d]", fn, V, kDeferredMaxViews);
  if ((long long)n_objects * V > 65535) return fail(SDFR_E_INVALID, "%s: %d objects x %d views exceed a launch", fn, n_objects, V);
  if (n_params < 8 || n_params > 256) return fail(SDFR_E_INVALID, "%s: n_params=%d out of range [8,256]", fn, n_params);
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step || !cam_pos || !cam_quat || !pos_c || !quat_c ||
      !inv_scale || !scale_v)
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (render_workspace && (W <= 0 || H <= 0)) return fail(SDFR_E_INVALID, "%s: W=%d H=%d", fn, W, H);
  const int B = n_objects * V;
  if (pc_workspace && (max_view_points <= 0 || (!offsets && B > 1)))
    return fail(SDFR_E_INVALID, "%s: bad sampler arguments", fn);
  if (render_workspace && ((uintptr_t)render_workspace % alignof(ViewSetup) || render_partials_offset % 16))
    return fail(SDFR_E_INVALID, "%s: render_workspace / partials offset misaligned", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  const TileGeom geom = render_workspace ? backward_geom(B, W, H) : kSmallTile;   // the tiling of the launch of ALL views
  const int nblk = pc_workspace ? (max_view_points + kSamplerPts - 1) / kSamplerPts : 0;
  const float* pc_part = (const float*)pc_workspace;
  LoopTailArgs a{};
  a.params = params; a.grads = grads; a.m = exp_avg; a.v = exp_avg_sq; a.step = step; a.n = n_params;
  a.lr_pos = lr_position; a.lr_quat = lr_orientation; a.lr_scale = lr_scale; a.lr_latent = lr_latent;
  a.update_latent = update_latent;
  a.cam_pos = cam_pos; a.cam_quat = cam_quat; a.V = V;
  a.setup = (const ViewSetup*)render_workspace;
  a.tile_part = render_workspace ? (const float*)((const char*)render_workspace + render_partials_offset) : nullptr;
  a.W = W; a.H = H;
  a.ntx = render_workspace ? geom.nx(W) : 0; a.nty = render_workspace ? geom.ny(H) : 0;
  a.tile_w = geom.w(); a.tile_h = geom.h();
  a.stride = (render_workspace && geom.sx * geom.sy > 1) ? backward_tile_stride(W, H) : 0;
  a.pc_part = pc_part;
  a.pc_loss_part = (pc_workspace && pc_loss) ? pc_part + (size_t)B * nblk * 8 : nullptr;
  a.offsets = offsets; a.n_single = max_view_points; a.nblk = nblk;
  a.pos_c = pos_c; a.quat_c = quat_c; a.inv_scale = inv_scale; a.scale_v = scale_v; a.pc_loss = pc_loss;
  a.n_obj = n_objects;
  a.latents = latents;
  if (decoder) {   // the batched VJP's last stage (sdfr_decoder_backward_latent_deferred_batch), one row per object
    decoder_fc_desc(decoder, &a.fc, &a.dec_params, nullptr);
    if (a.fc.width[0] != n_params - 8)
      return fail(SDFR_E_INVALID, "%s: n_params=%d does not hold the decoder's latent (%d)", fn, n_params, a.fc.width[0]);
    a.fc_one_wave = decoder_fc_one_wave(decoder, a.fc) ? 1 : 0;
    a.t_mid = decoder_t_mid;
  }
  hipLaunchKernelGGL(loop_tail_kernel, dim3((unsigned)n_objects), dim3(256), 0, (hipStream_t)stream, a);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_loop_view_records(const void* render_workspace, size_t render_partials_offset, int W, int H,
                                      int sdf_grad_mode, const void* pc_workspace, int with_pc_loss, const int* offsets,
                                      int max_view_points, const float* quat_c, const float* loss_depth, int view_begin,
                                      int V_local, int V_total, float* records, int device, void* stream) {
  const char* fn = "sdfr_loop_view_records";
  if (V_total < 1 || V_total > 65535 || V_local < 0 || view_begin < 0 || view_begin + V_local > V_total)
    return fail(SDFR_E_INVALID, "%s: shard [%d, %d) of %d views", fn, view_begin, view_begin + V_local, V_total);
  if (!records) return fail(SDFR_E_NULL, "%s: records is NULL", fn);
  if (render_workspace && (W <= 0 || H <= 0)) return fail(SDFR_E_INVALID, "%s: W=%d H=%d", fn, W, H);
  if (pc_workspace && (max_view_points <= 0 || !quat_c || (!offsets && V_local > 1)))
    return fail(SDFR_E_INVALID, "%s: bad sampler arguments", fn);
  if (render_workspace && ((uintptr_t)render_workspace % alignof(ViewSetup) || render_partials_offset % 16))
    return fail(SDFR_E_INVALID, "%s: render_workspace / partials offset misaligned", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  const bool have = render_workspace && V_local > 0;
  const TileGeom geom = !have ? kSmallTile
                        : (sdf_grad_mode & SDFR_BWD_SMALL_TILES) ? kSmallTile : backward_geom(V_local, W, H);
  const int nblk = pc_workspace ? (max_view_points + kSamplerPts - 1) / kSamplerPts : 0;
  const float* pc_part = V_local > 0 ? (const float*)pc_workspace : nullptr;
  const float* tile_part = have ? (const float*)((const char*)render_workspace + render_partials_offset) : nullptr;
  const int stride = (have && geom.sx * geom.sy > 1) ? backward_tile_stride(W, H) : 0;
  hipLaunchKernelGGL(view_records_kernel, dim3(V_total), dim3(64), 0, (hipStream_t)stream,
                     (const ViewSetup*)render_workspace, tile_part, W, H, have ? geom.nx(W) : 0, have ? geom.ny(H) : 0,
                     geom.w(), geom.h(), stride, pc_part,
                     (pc_part && with_pc_loss) ? pc_part + (size_t)V_local * nblk * 8 : nullptr, offsets,
                     max_view_points, nblk, quat_c, loss_depth, view_begin, V_local, records);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_loop_tail_records(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int* step,
                                      int n_params, float lr_position, float lr_orientation, float lr_scale,
                                      float lr_latent, int update_latent, const float* cam_pos, const float* cam_quat,
                                      int V_total, int view_begin, int V_local, const float* records, float* pos_c,
                                      float* quat_c, float* inv_scale, float* scale_v, const float* con_source,
                                      const float* con_target, float con_weight, float* con_loss,
                                      const sdfr_decoder* decoder, const float* decoder_t_mid, int device, void* stream) {
  const char* fn = "sdfr_loop_tail_records";
  if ((decoder != nullptr) != (decoder_t_mid != nullptr))
    return fail(SDFR_E_NULL, "%s: decoder and decoder_t_mid go together", fn);
  if (V_total < 1 || V_total > 65535 || V_local < 0 || view_begin < 0 || view_begin + V_local > V_total)
    return fail(SDFR_E_INVALID, "%s: shard [%d, %d) of %d views", fn, view_begin, view_begin + V_local, V_total);
  if (n_params < 8 || n_params > 256) return fail(SDFR_E_INVALID, "%s: n_params=%d out of range [8,256]", fn, n_params);
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step || !cam_pos || !cam_quat || !records ||
      (V_local > 0 && (!pos_c || !quat_c || !inv_scale || !scale_v)))
    return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (con_source && !con_target) return fail(SDFR_E_NULL, "%s: constraint target is NULL", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  LoopTailArgs a{};
  a.params = params; a.grads = grads; a.m = exp_avg; a.v = exp_avg_sq; a.step = step; a.n = n_params;
  a.lr_pos = lr_position; a.lr_quat = lr_orientation; a.lr_scale = lr_scale; a.lr_latent = lr_latent;
  a.update_latent = update_latent;
  a.cam_pos = cam_pos + 3 * (size_t)view_begin; a.cam_quat = cam_quat + 4 * (size_t)view_begin; a.V = V_local;
  a.records = records; a.V_all = V_total; a.cam_quat_all = cam_quat;
  a.pos_c = pos_c; a.quat_c = quat_c; a.inv_scale = inv_scale; a.scale_v = scale_v;
  a.con_source = con_source; a.con_target = con_target; a.con_weight = con_weight; a.con_loss = con_loss;
  if (decoder) {
    decoder_fc_desc(decoder, &a.fc, &a.dec_params, nullptr);
    a.fc_one_wave = decoder_fc_one_wave(decoder, a.fc) ? 1 : 0;
    if (a.fc.width[0] != n_params - 8)
      return fail(SDFR_E_INVALID, "%s: the decoder's latent has %d entries, the parameter vector %d", fn, a.fc.width[0],
                  n_params - 8);
    a.t_mid = decoder_t_mid;
  }
  hipLaunchKernelGGL(loop_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t sdfr_depth_l1_workspace_bytes(int V, int W, int H) {
  if (V <= 0 || W <= 0 || H <= 0) return 0;
  const int nchunk = (W * H + kLossChunk - 1) / kLossChunk;
  return (size_t)V * nchunk * 2 * sizeof(float);
}

extern "C" int sdfr_depth_l1_loss(const float* estimate, const float* target, int V, int W, int H,
                                  float weight, float* loss, float* grad_estimate, void* workspace,
                                  size_t workspace_bytes, int device, void* stream) {
  if (V < 0 || W < 0 || H < 0 || V > 65535 || (long long)W * H > 0x7fffffffLL)
    return fail(SDFR_E_INVALID, "sdfr_depth_l1_loss: bad sizes");
  if (V == 0 || W == 0 || H == 0) return 0;
  if (!estimate || !target || !loss || !grad_estimate || !workspace)
    return fail(SDFR_E_NULL, "sdfr_depth_l1_loss: NULL pointer argument");
  if (workspace_bytes < sdfr_depth_l1_workspace_bytes(V, W, H))
    return fail(SDFR_E_WORKSPACE, "sdfr_depth_l1_loss: workspace too small");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const int npix = W * H, nchunk = (npix + kLossChunk - 1) / kLossChunk;
  hipLaunchKernelGGL(depth_l1_partial_kernel, dim3(nchunk, V), dim3(256), 0, st, estimate, target, npix,
                     nchunk, (float*)workspace);
  hipLaunchKernelGGL(depth_l1_grad_kernel, dim3(nchunk, V), dim3(256), 0, st, estimate, target, npix,
                     nchunk, (const float*)workspace, weight, loss, grad_estimate);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_pc_l1_loss(const float* values, const int* offsets, int V, int max_view_points,
                               float weight, float* loss, float* grad_values, int device,
                               void* stream) {
  if (V < 0 || V > 65535 || max_view_points < 0) return fail(SDFR_E_INVALID, "sdfr_pc_l1_loss: bad sizes");
  if (!offsets && V > 1) return fail(SDFR_E_NULL, "offsets may be NULL only for a single view");
  if (V == 0) return 0;
  if (!loss || (max_view_points > 0 && (!values || !grad_values)))
    return fail(SDFR_E_NULL, "sdfr_pc_l1_loss: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(pc_l1_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, values, offsets,
                     max_view_points, weight, loss, grad_values);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_add_inplace(float* a, const float* b, size_t n, int device, void* stream) {
  if (n == 0) return 0;
  if (!a || !b) return fail(SDFR_E_NULL, "sdfr_add_inplace: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a, b, n);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              int* step, int n_params, float lr_position, float lr_orientation,
                              float lr_scale, float lr_latent, int update_latent, int device,
                              void* stream) {
  if (n_params < 8 || n_params > 1024) return fail(SDFR_E_INVALID, "n_params=%d out of range [8,1024]", n_params);
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step)
    return fail(SDFR_E_NULL, "sdfr_adam_step: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(adam_step_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, step, n_params, lr_position, lr_orientation, lr_scale, lr_latent,
                     update_latent);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_point_constraint(const float* orientation, const float* source, const float* target,
                                     float weight, float* loss, float* g_orientation, int device,
                                     void* stream) {
  if (!orientation || !source || !target) return fail(SDFR_E_NULL, "sdfr_point_constraint: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(point_constraint_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, orientation, source,
                     target, weight, loss, g_orientation);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_inlier_ratio(const float* depth_input, const float* depth_estimate, int W, int H,
                                 float relative_threshold, const int* step, int* counts, float* history,
                                 int max_history, float* state, const float* params, int n_params,
                                 float* best_params, int device, void* stream) {
  if (W < 0 || H < 0 || (long long)W * H > 0x7fffffffLL || n_params < 0 || max_history < 0)
    return fail(SDFR_E_INVALID, "sdfr_inlier_ratio: bad sizes");
  if (!depth_input || !depth_estimate || !step || !counts || !state || (n_params > 0 && best_params && !params))
    return fail(SDFR_E_NULL, "sdfr_inlier_ratio: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const int npix = W * H;
  if (npix > 0)
    hipLaunchKernelGGL(inlier_count_kernel, dim3((npix + 1023) / 1024), dim3(256), 0, st, depth_input,
                       depth_estimate, npix, relative_threshold, counts);
  hipLaunchKernelGGL(inlier_update_kernel, dim3(1), dim3(256), 0, st, counts, step, history, max_history, state,
                     params, n_params, best_params, (const float*)nullptr);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_inlier_counts_record(const float* depth_input, const float* depth_estimate, int W, int H,
                                         float relative_threshold, int* counts, float* record, int device,
                                         void* stream) {
  if (W < 0 || H < 0 || (long long)W * H >= (1 << 24)) return fail(SDFR_E_INVALID, "sdfr_inlier_counts_record: bad sizes");
  if (!depth_input || !depth_estimate || !counts || !record)
    return fail(SDFR_E_NULL, "sdfr_inlier_counts_record: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  const int npix = W * H;
  if (npix > 0)
    hipLaunchKernelGGL(inlier_count_kernel, dim3((npix + 1023) / 1024), dim3(256), 0, st, depth_input,
                       depth_estimate, npix, relative_threshold, counts);
  hipLaunchKernelGGL(inlier_counts_to_record_kernel, dim3(1), dim3(64), 0, st, counts, record + 18);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_inlier_update_record(const float* record, const int* step, float* history, int max_history,
                                         float* state, const float* params, int n_params, float* best_params,
                                         int device, void* stream) {
  if (n_params < 0 || max_history < 0) return fail(SDFR_E_INVALID, "sdfr_inlier_update_record: bad sizes");
  if (!record || !step || !state || (n_params > 0 && best_params && !params))
    return fail(SDFR_E_NULL, "sdfr_inlier_update_record: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(inlier_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (int*)nullptr, step, history,
                     max_history, state, params, n_params, best_params, record + 18);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_nn_loss_forward(const float* points_from, int N, const float* points_to, int M,
                                    float* dist, int* nearest, int device, void* stream) {
  if (N < 0 || M < 0) return fail(SDFR_E_INVALID, "sdfr_nn_loss_forward: negative size");
  if (N == 0) return 0;
  if (M == 0) return fail(SDFR_E_INVALID, "sdfr_nn_loss_forward: empty target set (torch: min over an empty dimension)");
  if (!points_from || !points_to || !dist || !nearest)
    return fail(SDFR_E_NULL, "sdfr_nn_loss_forward: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(nn_loss_forward_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     points_from, N, points_to, M, dist, nearest);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_nn_loss_backward(const float* grad_dist, const float* points_from, int N,
                                     const float* points_to, int M, const float* dist, const int* nearest,
                                     float* g_from, float* g_to, int device, void* stream) {
  if (N < 0 || M < 0) return fail(SDFR_E_INVALID, "sdfr_nn_loss_backward: negative size");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  if (g_to) zero_words_async(g_to, (size_t)M * 3, st);
  if (N == 0) return 0;
  if (!grad_dist || !points_from || !points_to || !dist || !nearest)
    return fail(SDFR_E_NULL, "sdfr_nn_loss_backward: NULL pointer argument");
  hipLaunchKernelGGL(nn_loss_backward_kernel, dim3((N + 255) / 256), dim3(256), 0, st, grad_dist, points_from, N,
                     points_to, dist, nearest, g_from, g_to);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_affine_mask(const float* depth, int B, int W, int H, const float* matrices,
                                unsigned char* mask, int device, void* stream) {
  if (B < 0 || W < 0 || H < 0 || B > 65535 || (long long)W * H > 0x7fffffffLL)
    return fail(SDFR_E_INVALID, "sdfr_affine_mask: bad sizes");
  if (B == 0 || W == 0 || H == 0) return 0;
  if (!depth || !matrices || !mask) return fail(SDFR_E_NULL, "sdfr_affine_mask: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(affine_mask_kernel, dim3((unsigned)((W * H + 255) / 256), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, depth, W, H, matrices, mask);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t sdfr_depth_points_workspace_bytes(int V, int W, int H) {
  if (V <= 0 || W <= 0 || H <= 0) return 0;
  // (either pixel order: the tiled one pads the image to whole 64 x 16 tiles)
  const long long nblk = std::max(((long long)W * H + kCompactPix - 1) / kCompactPix,
                                  (long long)((W + 63) / 64) * ((H + 15) / 16));
  return (size_t)V * (size_t)nblk * sizeof(int);
}

extern "C" int sdfr_depth_count(const float* depth, int V, int W, int H, int* counts, void* workspace,
                                size_t workspace_bytes, int device, void* stream) {
  return sdfr_depth_count_ordered(depth, V, W, H, 0, counts, workspace, workspace_bytes, device, stream);
}

extern "C" int sdfr_depth_to_points(const float* depth, int V, int W, int H, float rfx, float rfy, float cx0,
                                    float cy0, const int* offsets, const void* workspace, float* points,
                                    int device, void* stream) {
  return sdfr_depth_to_points_ordered(depth, V, W, H, 0, rfx, rfy, cx0, cy0, offsets, workspace, points, device, stream);
}

extern "C" int sdfr_depth_count_ordered(const float* depth, int V, int W, int H, int order, int* counts,
                                        void* workspace, size_t workspace_bytes, int device, void* stream) {
  if (order != SDFR_POINT_ORDER_ROW_MAJOR && order != SDFR_POINT_ORDER_TILED)
    return fail(SDFR_E_INVALID, "sdfr_depth_count: unknown point order %d", order);
  if (V < 0 || W < 0 || H < 0 || V > 65535 || (long long)(W + 63) * (H + 15) > 0x7fffffffLL)
    return fail(SDFR_E_INVALID, "sdfr_depth_count: bad sizes");
  if (V == 0) return 0;
  if (!counts) return fail(SDFR_E_NULL, "sdfr_depth_count: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  zero_words_async(reinterpret_cast<float*>(counts), (size_t)V, st);
  if (W == 0 || H == 0) return 0;
  if (!depth || !workspace) return fail(SDFR_E_NULL, "sdfr_depth_count: NULL pointer argument");
  if (workspace_bytes < sdfr_depth_points_workspace_bytes(V, W, H))
    return fail(SDFR_E_WORKSPACE, "sdfr_depth_count: workspace too small");
  const PixelOrder po = pixel_order(W, H, order);
  const int nblk = (po.count() + kCompactPix - 1) / kCompactPix;
  hipLaunchKernelGGL(depth_count_kernel<false>, dim3(nblk, V), dim3(256), 0, st, depth, po, nblk, (int*)workspace,
                     counts, 0.0f, 0.0f, 0.0f, 0.0f, (float*)nullptr);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" size_t sdfr_depth_centroid_workspace_bytes(int V, int W, int H) {
  // [block counts][block sums, 16 bytes each, 16-byte aligned]
  const size_t n = (sdfr_depth_points_workspace_bytes(V, W, H) + 15) & ~(size_t)15;
  return n + 4 * n;
}

extern "C" int sdfr_depth_count_centroid(const float* depth, int V, int W, int H, int order, float rfx, float rfy,
                                         float cx0, float cy0, int* counts, int* offsets, float* centroid,
                                         void* workspace, size_t workspace_bytes, int device, void* stream) {
  const char* fn = "sdfr_depth_count_centroid";
  if (order != SDFR_POINT_ORDER_ROW_MAJOR && order != SDFR_POINT_ORDER_TILED)
    return fail(SDFR_E_INVALID, "%s: unknown point order %d", fn, order);
  if (V < 0 || W < 0 || H < 0 || V > 65535 || (long long)(W + 63) * (H + 15) > 0x7fffffffLL)
    return fail(SDFR_E_INVALID, "%s: bad sizes", fn);
  if (V == 0) return 0;
  if (!counts || !centroid) return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  zero_words_async(reinterpret_cast<float*>(counts), (size_t)V, st);
  if (W == 0 || H == 0) {
    zero_words_async(centroid, (size_t)V * 3, st);
    if (offsets) zero_words_async(reinterpret_cast<float*>(offsets), (size_t)V, st);
    return 0;
  }
  if (!depth || !workspace) return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (workspace_bytes < sdfr_depth_centroid_workspace_bytes(V, W, H) || (uintptr_t)workspace % 16)
    return fail(SDFR_E_WORKSPACE, "%s: workspace too small or not 16-byte aligned", fn);
  const PixelOrder po = pixel_order(W, H, order);
  const int nblk = (po.count() + kCompactPix - 1) / kCompactPix;
  float* sums = (float*)((char*)workspace + ((sdfr_depth_points_workspace_bytes(V, W, H) + 15) & ~(size_t)15));
  hipLaunchKernelGGL(depth_count_kernel<true>, dim3(nblk, V), dim3(256), 0, st, depth, po, nblk, (int*)workspace,
                     counts, rfx, rfy, cx0, cy0, sums);
  hipLaunchKernelGGL(centroid_reduce_kernel, dim3(V), dim3(64), 0, st, sums, nblk, counts, centroid);
  if (offsets) hipLaunchKernelGGL(count_prefix_kernel, dim3(1), dim3(64), 0, st, counts, V, offsets);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_depth_to_points_ordered(const float* depth, int V, int W, int H, int order, float rfx, float rfy,
                                            float cx0, float cy0, const int* offsets, const void* workspace,
                                            float* points, int device, void* stream) {
  return sdfr_depth_to_points_shifted(depth, V, W, H, order, rfx, rfy, cx0, cy0, offsets, workspace, nullptr, nullptr,
                                      points, device, stream);
}

extern "C" int sdfr_depth_to_points_shifted(const float* depth, int V, int W, int H, int order, float rfx, float rfy,
                                            float cx0, float cy0, const int* offsets, const void* workspace,
                                            const float* shift, const float* noise, float* points, int device,
                                            void* stream) {
  if (order != SDFR_POINT_ORDER_ROW_MAJOR && order != SDFR_POINT_ORDER_TILED)
    return fail(SDFR_E_INVALID, "sdfr_depth_to_points: unknown point order %d", order);
  if (V < 0 || W < 0 || H < 0 || V > 65535 || (long long)(W + 63) * (H + 15) > 0x7fffffffLL)
    return fail(SDFR_E_INVALID, "sdfr_depth_to_points: bad sizes");
  if (V == 0 || W == 0 || H == 0) return 0;
  if (!depth || !offsets || !workspace || !points)
    return fail(SDFR_E_NULL, "sdfr_depth_to_points: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  const PixelOrder po = pixel_order(W, H, order);
  const int nblk = (po.count() + kCompactPix - 1) / kCompactPix;
  hipLaunchKernelGGL(depth_compact_kernel, dim3(nblk, V), dim3(256), 0, (hipStream_t)stream, depth, po,
                     nblk, (const int*)workspace, offsets, rfx, rfy, cx0, cy0, shift, noise, points);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_depth_to_points_resident(const float* depth, int V, int W, int H, int order, float rfx, float rfy,
                                             float cx0, float cy0, int* counts, int* offsets, void* workspace,
                                             size_t workspace_bytes, float* points, int device, void* stream) {
  const char* fn = "sdfr_depth_to_points_resident";
  if (order != SDFR_POINT_ORDER_ROW_MAJOR && order != SDFR_POINT_ORDER_TILED)
    return fail(SDFR_E_INVALID, "%s: unknown point order %d", fn, order);
  if (V < 0 || W < 0 || H < 0 || V > 65535 || (long long)(W + 63) * (H + 15) > 0x7fffffffLL ||
      (long long)V * W * H > 0x7fffffffLL)
    return fail(SDFR_E_INVALID, "%s: bad sizes", fn);
  if (V == 0) return 0;
  if (!counts || !offsets) return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  if (W == 0 || H == 0) {
    zero_words_async(reinterpret_cast<float*>(counts), (size_t)V, st);
    zero_words_async(reinterpret_cast<float*>(offsets), (size_t)V + 1, st);
    return 0;
  }
  if (!depth || !workspace || !points) return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (workspace_bytes < sdfr_depth_points_workspace_bytes(V, W, H))
    return fail(SDFR_E_WORKSPACE, "%s: workspace too small", fn);
  const PixelOrder po = pixel_order(W, H, order);
  const int nblk = (po.count() + kCompactPix - 1) / kCompactPix;
  hipLaunchKernelGGL(depth_count_kernel<false>, dim3(nblk, V), dim3(256), 0, st, depth, po, nblk, (int*)workspace,
                     (int*)nullptr, 0.0f, 0.0f, 0.0f, 0.0f, (float*)nullptr);
  hipLaunchKernelGGL(resident_offsets_kernel, dim3(1), dim3(256), 0, st, (const int*)workspace, nblk, V, counts,
                     offsets);
  hipLaunchKernelGGL(depth_compact_kernel, dim3(nblk, V), dim3(256), 0, st, depth, po, nblk, (const int*)workspace,
                     offsets, rfx, rfy, cx0, cy0, (const float*)nullptr, (const float*)nullptr, points);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_preprocess_depth(float* depth, const unsigned char* mask, int V, int W, int H, float far_field,
                                     int has_far_field, float* copy_to, int device, void* stream) {
  const char* fn = "sdfr_preprocess_depth";
  if (V < 0 || W < 0 || H < 0) return fail(SDFR_E_INVALID, "%s: negative size V=%d W=%d H=%d", fn, V, W, H);
  const size_t n = (size_t)V * (size_t)W * (size_t)H;
  if (n == 0) return 0;
  if (n > ((size_t)1 << 40)) return fail(SDFR_E_INVALID, "%s: image batch too large", fn);
  if (!depth || !mask) return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  const int vec = ((uintptr_t)depth % 16 == 0) && ((uintptr_t)mask % 4 == 0) && (!copy_to || (uintptr_t)copy_to % 16 == 0);
  hipLaunchKernelGGL(preprocess_depth_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                     depth, mask, n, far_field, has_far_field ? 1 : 0, vec, copy_to);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

#ifdef SDFR_TAIL_STAMPS
extern "C" __attribute__((visibility("default"))) int sdfr_debug_tail_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sdfr::g_tail_stamps), 8 * sizeof(unsigned long long));
}
#endif
