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

namespace sdfr {
namespace {

// torch.relu keeps NaN (fmaxf(NaN, 0) would return 0 and hide bad weights or points); the NaN returned is the
// canonical positive one, which the bit-pattern maximum of the set pooling carries to the output
__device__ __forceinline__ float relu_nan(float v) { return (v != v) ? __int_as_float(0x7fc00000) : fmaxf(v, 0.0f); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kPtsPerBlock = 64, kColsPerBlock = 64, kChunk = 124;

// Y = relu((X W^T + c) * s + t), optionally F_out = F_res + Y, and colmax[col] = max over the points.
//   x [M][ldx] (K = cin columns used), w [cout][ldw] (first cin columns used), c / s / t [cout]
// grid (ceil(M / 64), cout / 64); a wave = 16 points x 64 columns (4 accumulator tiles).
__global__ __launch_bounds__(256) void pointnet_layer_kernel(
    const float* __restrict__ x, int M, int cin, int ldx, const float* __restrict__ w, int ldw,
    const float* __restrict__ cvec, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
    const float* __restrict__ resid, float* __restrict__ y, int ldy, int cout, int* __restrict__ colmax) {
  // K runs in chunks of kChunk columns staged in LDS (X tile and W tile, rows padded by one float: a row
  // stride of a multiple of 32 floats would put the 16 rows a wave-instruction reads into one bank)
  __shared__ float xs[kPtsPerBlock * (kChunk + 1)];
  __shared__ float ws[kColsPerBlock * (kChunk + 1)];
  __shared__ int wave_max[4][kColsPerBlock];
  constexpr int ld = kChunk + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = blockIdx.x * kPtsPerBlock, c0 = blockIdx.y * kColsPerBlock;
  const int row = lane & 15, kq = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const float* xa = xs + (wave * 16 + row) * ld + kq;
  const float* wb = ws + row * ld + kq;
  for (int kc = 0; kc < cin; kc += kChunk) {
    const int kn = min(kChunk, cin - kc), kp = (kn + 3) & ~3;  // this chunk, padded to the MFMA's 4
    if (kc) __syncthreads();
    for (int i = tid; i < kPtsPerBlock * kp; i += 256) {
      const int r = i / kp, k = i - r * kp;
      xs[r * ld + k] = (p0 + r < M && k < kn) ? x[(size_t)(p0 + r) * ldx + kc + k] : 0.0f;
      ws[r * ld + k] = (c0 + r < cout && k < kn) ? w[(size_t)(c0 + r) * ldw + kc + k] : 0.0f;
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

extern "C" int sdfr_pointnet_layer(const float* x, int M, int cin, int ldx, const float* w, int ldw,
                                   const float* cvec, const float* bn_scale, const float* bn_shift,
                                   const float* resid, float* y, int ldy, int cout, float* colmax, int device,
                                   void* stream) {
  if (M < 1 || cin < 1 || cout < 1 || ldx < cin || ldw < cin || (y && ldy < cout))
    return fail(SDFR_E_INVALID, "sdfr_pointnet_layer: bad sizes M=%d cin=%d cout=%d", M, cin, cout);
  if (!x || !w || !cvec || !bn_scale || !bn_shift || !colmax)
    return fail(SDFR_E_NULL, "sdfr_pointnet_layer: NULL pointer argument");
  if (resid && !y) return fail(SDFR_E_NULL, "sdfr_pointnet_layer: a residual needs an output");
  SDFR_HIP_TRY(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  zero_words_async(colmax, (size_t)cout, st);
  hipLaunchKernelGGL(pointnet_layer_kernel, dim3((unsigned)((M + kPtsPerBlock - 1) / kPtsPerBlock),
                                                 (unsigned)((cout + kColsPerBlock - 1) / kColsPerBlock)),
                     dim3(256), 0, st, x, M, cin, ldx, w, ldw, cvec, bn_scale, bn_shift, resid, y, ldy, cout,
                     reinterpret_cast<int*>(colmax));
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
