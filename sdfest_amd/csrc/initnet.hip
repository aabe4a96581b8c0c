// initnet.hip -- forward of the single-shot initialisation network that starts the render-and-compare
// loop: VanillaPointNet backbone + SDFPoseHead, inference mode.
//   sdfest/initialization/pointnet.py:7-96          per-point MLP (Linear -> BatchNorm1d -> ReLU), optional
//                                                   dense links (the set maximum is concatenated to every
//                                                   point) and residual links, then max over the points
//   sdfest/initialization/sdf_pose_network.py:9-115 MLP on the set feature, final Linear ->
//                                                   (latent, position, scale, orientation)
//   sdfest/estimation/simple_setup.py:795-812       softmax of the orientation logits, optional prior
//                                                   adjustment (:978-1009), argmax
// Not a translation of the torch graph:
//   * the per-point layers are GEMMs (points x channels) on the matrix cores in exact fp32
//     (v_mfma_f32_16x16x4_f32), with bias, the folded BatchNorm, ReLU, the residual sum and the column
//     maximum over the set in the epilogue -- the (M, 2C) concatenation of a dense link never exists: its
//     second half is the same vector for every point, so W[:, C:] . max goes into the layer's bias;
//   * the last backbone layer (M x 1024) is never written: only its column maximum leaves the kernel.
#include "common.hpp"

#include <algorithm>

namespace sdfr {
namespace {

// torch.relu keeps NaN (fmaxf(NaN, 0) would return 0 and hide bad weights or points); the NaN returned is the
// canonical positive one, which the bit-pattern maximum of the set pooling carries to the output
__device__ __forceinline__ float relu_nan(float v) { return (v != v) ? __int_as_float(0x7fc00000) : fmaxf(v, 0.0f); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
// (K = 128, the mug backbone's inner layers, as ONE chunk -- 66 KB of dynamic LDS, a run-time row stride -- was
// measured: 13.8 -> 16.7 us per layer; the second chunk's four columns cost less than that)
constexpr int kPtsPerBlock = 64, kColsPerBlock = 64, kChunk = 124;

// Y = relu((X W^T + c) * s + t), optionally F_out = F_res + Y, and colmax[col] = max over the points.
//   x [M][ldx] (K = cin columns used), w [cout][ldw] (first cin columns used), c / s / t [cout]
// grid (ceil(M / 64), cout / 64); a wave = 16 points x 64 columns (4 accumulator tiles).
__global__ __launch_bounds__(256) void pointnet_layer_kernel(
    const float* __restrict__ x, int M, int cin, int ldx, const float* __restrict__ w, int ldw,
    const float* __restrict__ cvec, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
    const float* __restrict__ resid, float* __restrict__ y, int ldy, int cout, int* __restrict__ colmax,
    const int* __restrict__ row_count) {
  // K runs in chunks of kChunk columns staged in LDS (X tile and W tile, rows padded by one float: a row
  // stride of a multiple of 32 floats would put the 16 rows a wave-instruction reads into one bank)
  __shared__ float xs[kPtsPerBlock * (kChunk + 1)];
  __shared__ float ws[kColsPerBlock * (kChunk + 1)];
  __shared__ int wave_max[4][kColsPerBlock];
  constexpr int ld = kChunk + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // row_count (sdfr_pointnet_layer_counted): the number of points lives on the device and M is the buffers' capacity;
  // the grid's row blocks stride over the real rows (workgroup-uniform loop)
  if (row_count) M = min(max(row_count[0], 0), M);
  const int c0 = blockIdx.y * kColsPerBlock;
  const int row = lane & 15, kq = lane >> 4;
  // (rows of X and W 16 bytes apart and aligned: the staging's vector form -- workgroup-uniform)
  const bool vec_ok = ((ldx | ldw) & 3) == 0 && (((uintptr_t)x | (uintptr_t)w) & 15) == 0;
  for (int p0 = blockIdx.x * kPtsPerBlock; p0 < M; p0 += gridDim.x * kPtsPerBlock) {
  if (p0 != (int)blockIdx.x * kPtsPerBlock) __syncthreads();   // (the previous block's LDS reads are done)
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const float* xa = xs + (wave * 16 + row) * ld + kq;
  const float* wb = ws + row * ld + kq;
  for (int kc = 0; kc < cin; kc += kChunk) {
    const int kn = min(kChunk, cin - kc), kp = (kn + 3) & ~3;  // this chunk, padded to the MFMA's 4
    if (kc) __syncthreads();
    if (vec_ok && (kn & 3) == 0) {
      // 16-byte loads, no division: thread -> (row tid / 4, vectors tid % 4, + 4, ...): four threads walk a row's 512
      // bytes 64 at a time, and eight loads (four of X, four of W) are in flight per thread before the first LDS store
      // (the element-wise form below: 124 dependent iterations of a divide, two 4-byte loads and two stores per thread
      // -- 27 us per 128 -> 128 layer over 21 k points, most of it here)
      const int r = tid >> 2, sub = tid & 3, kv = kn >> 2;
      const bool xr = p0 + r < M, wr = c0 + r < cout;
      const f32x4* xg = reinterpret_cast<const f32x4*>(x + (size_t)(p0 + r) * ldx + kc);
      const f32x4* wg = reinterpret_cast<const f32x4*>(w + (size_t)(c0 + r) * ldw + kc);
      const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int v0 = sub; v0 < kv; v0 += 16) {
        f32x4 xv[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int v = v0 + 4 * u;
          xv[u] = (xr && v < kv) ? xg[v] : zero4;
          wv[u] = (wr && v < kv) ? wg[v] : zero4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int v = v0 + 4 * u;
          if (v < kv) {
            float* xd = xs + r * ld + 4 * v;
            float* wd = ws + r * ld + 4 * v;
            xd[0] = xv[u].x; xd[1] = xv[u].y; xd[2] = xv[u].z; xd[3] = xv[u].w;
            wd[0] = wv[u].x; wd[1] = wv[u].y; wd[2] = wv[u].z; wd[3] = wv[u].w;
          }
        }
      }
    } else {
      for (int i = tid; i < kPtsPerBlock * kp; i += 256) {
        const int r = i / kp, k = i - r * kp;
        xs[r * ld + k] = (p0 + r < M && k < kn) ? x[(size_t)(p0 + r) * ldx + kc + k] : 0.0f;
        ws[r * ld + k] = (c0 + r < cout && k < kn) ? w[(size_t)(c0 + r) * ldw + kc + k] : 0.0f;
      }
    }
    __syncthreads();
    for (int k0 = 0; k0 < kp; k0 += 4) {
      const float a = xa[k0];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[j * 16 * ld + k0], acc[j], 0, 0, 0);
    }
  }
  // epilogue: the accumulator holds D[point 4 * kq + r][column row] of each tile
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = c0 + j * 16 + row;
    const bool col_ok = col < cout;
    const float cv = col_ok ? cvec[col] : 0.0f, s = col_ok ? bn_scale[col] : 0.0f, t = col_ok ? bn_shift[col] : 0.0f;
    // ReLU output is >= 0 or NaN (torch's relu and max propagate NaN: pointnet.py:64-96); non-negative floats
    // order as ints and the canonical positive NaN lies above +inf, so the set maximum is taken on the bit patterns
    int vmax = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = p0 + wave * 16 + kq * 4 + r;
      const float v = relu_nan(fmaf(acc[j][r] + cv, s, t));
      if (p < M && col_ok) {
        vmax = max(vmax, __float_as_int(v));
        if (y) y[(size_t)p * ldy + col] = resid ? resid[(size_t)p * ldy + col] + v : v;
      }
    }
    vmax = max(vmax, __shfl_xor(vmax, 16, 64));
    vmax = max(vmax, __shfl_xor(vmax, 32, 64));
    if (kq == 0) wave_max[wave][j * 16 + row] = vmax;
  }
  __syncthreads();
  if (tid < kColsPerBlock && c0 + tid < cout)
    atomicMax(&colmax[c0 + tid], max(max(wave_max[0][tid], wave_max[1][tid]), max(wave_max[2][tid], wave_max[3][tid])));
  }
}

// simple_setup.py:790-838 for ONE view, on the device: the head's output row -> the estimate in the world frame,
// written straight into the loop's parameter vector [position 3 | orientation 4 | scale 1 | latent L].
//   head [L + 4 + C] (discretised: latent, position, scale, C orientation logits) or [L + 8] (quaternion)
//   grid_quats [C][4] + index (the argmax of sdfr_orientation_posterior): the orientation of the chosen cell
//   centroid (nullable): added to the position (`position += centroid`, :793-794)
//   cam_pos [3], cam_quat [4]: camera in the world (:819-825): p_w = q_c (p, 0) q_c* + t_c, q_w = q_c q
//   take_if_better: 0 = "first" (always written); 1 = "best": written only if post_max[0] > best[0], which it then
//   becomes (:835-838; best[0] starts at 0)
__global__ void init_estimate_kernel(const float* __restrict__ head, int L, const float* __restrict__ grid_quats,
                                     const int* __restrict__ index, const float* __restrict__ centroid,
                                     const float* __restrict__ cam_pos, const float* __restrict__ cam_quat,
                                     int mean_shape, int take_if_better, const float* __restrict__ post_max,
                                     float* __restrict__ best, float* __restrict__ params) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (take_if_better) {
    if (!(post_max[0] > best[0])) return;
    best[0] = post_max[0];
  }
  float p[3] = {head[L], head[L + 1], head[L + 2]};
  if (centroid) { p[0] += centroid[0]; p[1] += centroid[1]; p[2] += centroid[2]; }
  float q[4];
  if (grid_quats) {
    const int i = index[0];
    for (int k = 0; k < 4; ++k) q[k] = grid_quats[4 * i + k];
  } else {   // sdf_pose_network.py:97-101: the head normalises its quaternion
    const float* o = head + L + 4;
    const float n = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    for (int k = 0; k < 4; ++k) q[k] = o[k] / n;
  }
  const float c[4] = {cam_quat[0], cam_quat[1], cam_quat[2], cam_quat[3]};
  // quaternion_multiply (quaternion_utils.py:12-33), scalar last
  auto mul = [](const float* a, const float* b, float* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  };
  const float pq[4] = {p[0], p[1], p[2], 0.0f}, ci[4] = {-c[0], -c[1], -c[2], c[3]};
  float t[4], r[4], qw[4];
  mul(c, pq, t);      // quaternion_apply (:36-54): q (p, 0) q^-1
  mul(t, ci, r);
  mul(c, q, qw);
  params[0] = r[0] + cam_pos[0]; params[1] = r[1] + cam_pos[1]; params[2] = r[2] + cam_pos[2];
  for (int k = 0; k < 4; ++k) params[3 + k] = qw[k];
  params[7] = head[L + 3];
  for (int k = 0; k < L; ++k) params[8 + k] = mean_shape ? 0.0f : head[k];
}

// y[col] = act((W[col][koff .. koff + k) . x + bias[col]) * s[col] + t[col]); one wave per column.
__global__ __launch_bounds__(256) void linear_vec_kernel(const float* __restrict__ w, int ldw, int koff,
                                                         const float* __restrict__ x, int k,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ bn_scale,
                                                         const float* __restrict__ bn_shift, int relu,
                                                         float* __restrict__ y, int cout) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (col >= cout) return;
  const float* wr = w + (size_t)col * ldw + koff;
  float acc = 0.0f;
  for (int i = lane; i < k; i += 64) acc = fmaf(wr[i], x[i], acc);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    float v = acc + (bias ? bias[col] : 0.0f);
    if (bn_scale) v = fmaf(v, bn_scale[col], bn_shift[col]);
    y[col] = relu ? relu_nan(v) : v;
  }
}

// simple_setup.py:795-812: posterior = softmax(logits); with a prior: posterior * prior / train_prior,
// L1-normalised (:1001-1008); out_index = argmax (first maximum), out_max = its probability.  One workgroup.
__global__ __launch_bounds__(256) void orientation_posterior_kernel(const float* __restrict__ logits, int C,
                                                                    const float* __restrict__ prior,
                                                                    const float* __restrict__ train_prior,
                                                                    float* __restrict__ posterior,
                                                                    int* __restrict__ out_index,
                                                                    float* __restrict__ out_max) {
  __shared__ float red[256];
  __shared__ int red_i[256];
  const int tid = threadIdx.x;
  float m = -3.0e38f;
  for (int i = tid; i < C; i += 256) m = fmaxf(m, logits[i]);
  red[tid] = m;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
  m = red[0];
  __syncthreads();
  float sum = 0.0f;
  for (int i = tid; i < C; i += 256) sum += expf(logits[i] - m);
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  const float inv = 1.0f / red[0];
  __syncthreads();
  float l1 = 0.0f;
  for (int i = tid; i < C; i += 256) {
    float p = expf(logits[i] - m) * inv;
    if (prior) {
      p *= prior[i];
      if (train_prior) p /= train_prior[i];
      l1 += fabsf(p);
    }
    posterior[i] = p;
  }
  if (prior) {  // torch.nn.functional.normalize(p = 1): x / max(|x|_1, 1e-12)
    red[tid] = l1;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float k = 1.0f / fmaxf(red[0], 1e-12f);
    __syncthreads();
    for (int i = tid; i < C; i += 256) posterior[i] *= k;
  }
  __syncthreads();
  float best = -1.0f;
  int arg = 0x7fffffff;
  for (int i = tid; i < C; i += 256) {
    const float p = posterior[i];
    if (p > best) { best = p; arg = i; }
  }
  red[tid] = best; red_i[tid] = arg;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s) {
      const float o = red[tid + s];
      const int oi = red_i[tid + s];
      if (o > red[tid] || (o == red[tid] && oi < red_i[tid])) { red[tid] = o; red_i[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0) { out_index[0] = red_i[0]; out_max[0] = red[0]; }
}

}  // namespace
}  // namespace sdfr

using namespace sdfr;

namespace {
int pointnet_layer_impl(const char* fn, const float* x, const int* row_count, int M, int cin, int ldx, const float* w,
                        int ldw, const float* cvec, const float* bn_scale, const float* bn_shift, const float* resid,
                        float* y, int ldy, int cout, float* colmax, int device, void* stream) {
  if (M < 1 || cin < 1 || cout < 1 || ldx < cin || ldw < cin || (y && ldy < cout))
    return fail(SDFR_E_INVALID, "%s: bad sizes M=%d cin=%d cout=%d", fn, M, cin, cout);
  if (!x || !w || !cvec || !bn_scale || !bn_shift || !colmax) return fail(SDFR_E_NULL, "%s: NULL pointer argument", fn);
  if (resid && !y) return fail(SDFR_E_NULL, "%s: a residual needs an output", fn);
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  zero_words_async(colmax, (size_t)cout, st);
  // (a counted call does not know its rows: enough row blocks to fill the chip, striding over the real ones)
  const int row_blocks = (M + kPtsPerBlock - 1) / kPtsPerBlock;
  const int col_blocks = (cout + kColsPerBlock - 1) / kColsPerBlock;
  // (1024 workgroups: the point set of one mug view is ~330 row blocks, and workgroups that find no row cost the
  // dispatcher their launch all the same)
  const int gx = row_count ? std::min(row_blocks, std::max(1, 1024 / col_blocks)) : row_blocks;
  hipLaunchKernelGGL(pointnet_layer_kernel, dim3((unsigned)gx, (unsigned)col_blocks), dim3(256), 0, st, x, M, cin, ldx, w,
                     ldw, cvec, bn_scale, bn_shift, resid, y, ldy, cout, reinterpret_cast<int*>(colmax), row_count);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int sdfr_pointnet_layer(const float* x, int M, int cin, int ldx, const float* w, int ldw,
                                   const float* cvec, const float* bn_scale, const float* bn_shift,
                                   const float* resid, float* y, int ldy, int cout, float* colmax, int device,
                                   void* stream) {
  return pointnet_layer_impl("sdfr_pointnet_layer", x, nullptr, M, cin, ldx, w, ldw, cvec, bn_scale, bn_shift, resid, y,
                             ldy, cout, colmax, device, stream);
}

extern "C" int sdfr_pointnet_layer_counted(const float* x, const int* row_count, int M_capacity, int cin, int ldx,
                                           const float* w, int ldw, const float* cvec, const float* bn_scale,
                                           const float* bn_shift, const float* resid, float* y, int ldy, int cout,
                                           float* colmax, int device, void* stream) {
  if (!row_count) return fail(SDFR_E_NULL, "sdfr_pointnet_layer_counted: row_count is NULL");
  return pointnet_layer_impl("sdfr_pointnet_layer_counted", x, row_count, M_capacity, cin, ldx, w, ldw, cvec, bn_scale,
                             bn_shift, resid, y, ldy, cout, colmax, device, stream);
}

extern "C" int sdfr_init_estimate(const float* head, int latent, const float* grid_quats, const int* index,
                                  const float* centroid, const float* cam_pos, const float* cam_quat, int mean_shape,
                                  int take_if_better, const float* posterior_max, float* best, float* params, int device,
                                  void* stream) {
  if (latent < 0 || latent > 1024) return fail(SDFR_E_INVALID, "sdfr_init_estimate: latent=%d", latent);
  if (!head || !cam_pos || !cam_quat || !params || (grid_quats && !index) || (take_if_better && (!posterior_max || !best)))
    return fail(SDFR_E_NULL, "sdfr_init_estimate: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(init_estimate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, head, latent, grid_quats, index,
                     centroid, cam_pos, cam_quat, mean_shape, take_if_better, posterior_max, best, params);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_linear_vec(const float* w, int ldw, int koff, const float* x, int k, const float* bias,
                               const float* bn_scale, const float* bn_shift, int relu, float* y, int cout,
                               int device, void* stream) {
  if (cout < 1 || k < 0 || koff < 0 || ldw < koff + k) return fail(SDFR_E_INVALID, "sdfr_linear_vec: bad sizes");
  if (!w || !y || (k > 0 && !x) || ((bn_scale == nullptr) != (bn_shift == nullptr)))
    return fail(SDFR_E_NULL, "sdfr_linear_vec: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(linear_vec_kernel, dim3((unsigned)((cout + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, ldw,
                     koff, x, k, bias, bn_scale, bn_shift, relu, y, cout);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sdfr_orientation_posterior(const float* logits, int C, const float* prior, const float* train_prior,
                                          float* posterior, int* out_index, float* out_max, int device,
                                          void* stream) {
  if (C < 1) return fail(SDFR_E_INVALID, "sdfr_orientation_posterior: C=%d", C);
  if (!logits || !posterior || !out_index || !out_max || (train_prior && !prior))
    return fail(SDFR_E_NULL, "sdfr_orientation_posterior: NULL pointer argument");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(orientation_posterior_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, C, prior,
                     train_prior, posterior, out_index, out_max);
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
