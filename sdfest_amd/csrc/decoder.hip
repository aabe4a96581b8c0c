// decoder.hip -- forward of the SDF VAE decoder (sdfest/vae/sdf_vae.py:171-259) on MI355X.
//
//   z (N, L) -> [Linear + ReLU]* -> view (C, s, s, s)
//            -> per conv layer: [trilinear resize to in_size] -> Conv3d (valid) -> [ReLU]
//            -> [trilinear resize to the volume size] -> [clamp to +-tsdf]
//
// MI355X mapping
//   * the whole Linear stack is ONE launch: every workgroup recomputes the small leading layers
//     in LDS and produces a 256-wide slice of the last (wide) layer from a transposed weight copy
//     (coalesced 8192 x 50 GEMV);
//   * each Conv3d is an im2col contraction on the matrix cores with the exact-fp32 MFMA
//     (v_mfma_f32_16x16x4_f32: same rounding as an fmaf chain, so the 1e-4 parity budget is not
//     spent here): a wave owns 16 output voxels x 16 output channels, K = Cin*k^3 runs four taps
//     per MFMA; the weight matrix (K x 16, zero padded) and the tap->offset table live in LDS;
//     bias and ReLU are applied to the accumulator;
//   * the resize is its own element-wise kernel with ATen's align_corners=False index arithmetic
//     (the output of a layer is at most 4 MiB and stays in L2 / Infinity Cache between launches).
//   * batches take different kernels where the single-decode ones are bound by their operand
//     fetch: conv3d_direct_kernel (packed-fp32 FMAs, LDS input patch, scalar-register weights) for
//     3x3x3 layers with 4 / 8 / 16 output channels, resize3_tiled_kernel for up-sampling;
//   * 1x1 layers: a 1x1x1 convolution commutes with the trilinear resize in front of it (both are
//     linear, one across channels, one across space; the interpolation weights sum to 1, so the
//     bias passes through).  When Cout <= Cin such a layer runs conv -> resize instead of
//     resize -> conv: for the mug decoder (resize 30^3 -> 64^3 of 4 channels, then 4 -> 1) the big
//     tensor shrinks from 4 x 64^3 to 1 x 64^3 per sample -- batched decode 16.5 -> ~6 us per
//     latent.  Same result up to fp32 rounding order (~1e-7); ReLU, if any, is applied after the
//     resize, where the reference applies it.
// A single decode is latency-, not FLOP-bound: what matters there is 8 launches instead of the
// reference's ~20 eager ops, no host synchronisation, and that everything can be graph-captured.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "decoder_fc.hpp"

struct sdfr_decoder {
  int device;
  int latent, n_fc, n_conv, volume;
  float tsdf;
  std::vector<int> fc_out, conv_in_size, conv_cin, conv_cout, conv_k, conv_relu, conv_kpad;
  std::vector<int> conv_swap, conv_prev;   // see "1x1 layers" below; size of the tensor entering layer l
  // device copies
  float* d_params = nullptr;               // everything below lives in this one allocation
  std::vector<size_t> fc_w_off, fc_b_off;  // fc weights: [out][in] except the last: [in][out]
  std::vector<size_t> conv_w_off, conv_b_off, conv_tab_off;  // conv weights: [Kpad][16]
  size_t max_act = 0;                      // floats of the largest intermediate tensor
  // backward (VJP to the latent): transposed/flipped conv matrices, their tap tables, zero bias
  std::vector<size_t> bwd_w_off, bwd_tab_off;
  std::vector<int> bwd_kpad;
  size_t zero_bias_off = 0;
  // transposed-resize tables of the fused VJP stages (vjp_stage_kernel): for the resize in FRONT of layer l (conv_prev[l]
  // -> conv_in_size[l]), a row of 16 words per coarse index: first source (int), taps (int), 12 weights; 0: none
  std::vector<size_t> rs_tab_off;
  // z-grouped contraction for layers with few output channels (see conv3d_mfma_kernel): per layer,
  // forward and data-gradient
  struct ZPlan {
    int zg = 1, kpad = 0;
    size_t w_off = 0, tab_off = 0;
  };
  std::vector<ZPlan> fwd_z, bwd_z;
  // direct (VALU) convolution for batches, conv3d_direct_kernel: weights [ci][a][b][c][co], 0 = none
  std::vector<size_t> fwd_direct_off, bwd_direct_off;
  size_t max_bwd = 0;                      // floats of the largest gradient tensor (incl. padding)
  // tape: post-ReLU outputs kept by a forward that will be differentiated
  size_t tape_fc_off = 0;                  // per-sample float offsets
  std::vector<size_t> tape_conv_off;
  size_t tape_floats = 0;                  // per sample
  // which of two equivalent kernel forms a call takes (sdfr_decoder_set_option; the defaults are the measured-faster
  // ones, results are the same bit for bit): per HANDLE, so that nothing one caller selects reaches another's decoder
  mutable std::atomic<int> opt_fused_resize{1}, opt_tiled_vjp{1}, opt_fc_one_wave{1};
  // bits: 1 resize + convolution (conv3d_mfma_up_kernel), 2 Linear stack + first convolution (fc_conv_kernel),
  // 4 / 8 transposed resize + transposed convolution of the VJP (vjp_stage_kernel): its first stage / the stages
  // behind a fused one -- few latents only
  mutable std::atomic<int> opt_fused_single{5};   // (the pairs measured faster: C5 0.1200 -> 0.1120 ms per iteration)
};

namespace sdfr {
namespace {

constexpr long long kZGroupMinRows = 64 * 1024;  // z-grouped conv only when it still fills the chip

typedef float f32x4 __attribute__((ext_vector_type(4)));


// grid (ceil(out_last / 256), N)
// NARROW (fc_one_wave_ok: every leading layer at most 64 wide, the mug decoder's 8 -> 20 -> 50): the leading layers'
// parameters are copied to LDS first, all loads in flight at once, and wave 0 runs the layers out of LDS -- the general
// form pays a runtime-length loop of dependent global loads per layer (1.0 + 1.8 us of the single decode's 6.7 us
// launch; tools/microbench/tail_stamps.py measured the same loops in the backward).  Same fmaf chains, same numbers.
template <bool NARROW>
__global__ __launch_bounds__(kFcBlock) void fc_stack_kernel(const float* __restrict__ params,
                                                            FcDesc d, const float* __restrict__ z,
                                                            float* __restrict__ out) {
  __shared__ float act[2][NARROW ? kFcWaveWidth : kMaxHidden];
  const int tid = threadIdx.x, n = blockIdx.y;
  int cur = 0;
  if (NARROW) {
    __shared__ float p_lds[kFcWaveSpan];
    const long long base = d.w_off[0];
    const int span = (int)fc_wave_span(d);
    const float z_t = tid < d.width[0] ? z[(size_t)n * d.width[0] + tid] : 0.0f;
    {
      constexpr int U = kFcWaveSpan / kFcBlock;   // 24 loads per thread cover the largest span
      float r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = u * kFcBlock + tid;
        r[u] = e < span ? params[base + e] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = u * kFcBlock + tid;
        if (e < span) p_lds[e] = r[u];
      }
    }
    if (tid < d.width[0]) act[0][tid] = z_t;
    __syncthreads();
    if (tid < 64) {
      for (int l = 0; l < d.n_fc - 1; ++l) {
        const int win = d.width[l], wout = d.width[l + 1];
        if (tid < wout) {
          float acc = p_lds[d.b_off[l] - base + tid];
          const float* w = p_lds + (d.w_off[l] - base) + tid * win;
#pragma unroll 8
          for (int i = 0; i < win; ++i) acc = fmaf(w[i], act[cur][i], acc);
          act[cur ^ 1][tid] = fmaxf(acc, 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
        cur ^= 1;
      }
    } else {
      cur = (d.n_fc - 1) & 1;
    }
    __syncthreads();
  } else {
    for (int i = tid; i < d.width[0]; i += kFcBlock) act[0][i] = z[(size_t)n * d.width[0] + i];
    __syncthreads();
    for (int l = 0; l < d.n_fc - 1; ++l) {
      const int win = d.width[l], wout = d.width[l + 1];
      const float* w = params + d.w_off[l];
      const float* b = params + d.b_off[l];
      for (int o = tid; o < wout; o += kFcBlock) {
        float acc = b[o];
        for (int i = 0; i < win; ++i) acc = fmaf(w[(size_t)o * win + i], act[cur][i], acc);
        act[cur ^ 1][o] = fmaxf(acc, 0.0f);
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  const int l = d.n_fc - 1;
  const int win = d.width[l], wout = d.width[l + 1];
  const int o = blockIdx.x * kFcBlock + tid;
  if (o < wout) {
    const float* wt = params + d.w_off[l];  // transposed: [in][out]
    float acc = params[d.b_off[l] + o];
    // (sixteen of the column's weights in flight at a time: the chain is bound by their round trips)
#pragma unroll 16
    for (int i = 0; i < win; ++i) acc = fmaf(wt[(size_t)i * wout + o], act[cur][i], acc);
    out[(size_t)n * wout + o] = fmaxf(acc, 0.0f);
  }
}

// The same stack for batches: a workgroup takes S samples and 256 outputs of the last layer.  The per-sample form
// above walks four dependent round trips (z -> w0 -> w1 -> wt) per 256 outputs of ONE sample: 8192 workgroups, 36 us
// per 256 mug latents, all of it latency.  Here the hidden layers of S samples are formed together and every weight
// of the last layer is loaded once per S samples (the activations come from LDS as broadcasts, [unit][sample]):
// the same fmaf chains (bias first, inputs in ascending order), so bit-identical to the per-sample form.
// grid (ceil(out_last / 256), ceil(N / S));  LDS: 2 * hid * S floats, hid = widest layer input
// NARROW (fc_one_wave_ok): the leading layers' parameters come from an LDS copy made with all loads in flight at once
// (as fc_stack_kernel<true>), not from a runtime-length loop of dependent global loads per layer.
template <int S, bool NARROW>
__global__ __launch_bounds__(kFcBlock) void fc_stack_batch_kernel(const float* __restrict__ params, FcDesc d,
                                                                  const float* __restrict__ z, int N, int hid,
                                                                  float* __restrict__ out) {
  extern __shared__ float fc_act[];   // [2][hid][S]
  __shared__ float p_lds[NARROW ? kFcWaveSpan : 1];
  const int tid = threadIdx.x, n0 = blockIdx.y * S, ns = min(S, N - n0);
  float* cur = fc_act;
  float* nxt = fc_act + (size_t)hid * S;
  const long long base = d.w_off[0];
  if (NARROW) {
    const int span = (int)fc_wave_span(d);
    constexpr int U = kFcWaveSpan / kFcBlock;
    float r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = u * kFcBlock + tid;
      r[u] = e < span ? params[base + e] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = u * kFcBlock + tid;
      if (e < span) p_lds[e] = r[u];
    }
  }
  {
    const int w0 = d.width[0];
    for (int e = tid; e < S * w0; e += kFcBlock) {
      const int i = e / S, sm = e - i * S;
      cur[e] = sm < ns ? z[(size_t)(n0 + sm) * w0 + i] : 0.0f;
    }
  }
  __syncthreads();
  for (int l = 0; l < d.n_fc - 1; ++l) {
    const int win = d.width[l], wout = d.width[l + 1];
    const float* w = NARROW ? p_lds + (d.w_off[l] - base) : params + d.w_off[l];
    const float* b = NARROW ? p_lds + (d.b_off[l] - base) : params + d.b_off[l];
    for (int e = tid; e < S * wout; e += kFcBlock) {
      const int o = e / S, sm = e - o * S;
      float acc = b[o];
#pragma unroll 8
      for (int i = 0; i < win; ++i) acc = fmaf(w[(size_t)o * win + i], cur[i * S + sm], acc);
      nxt[e] = fmaxf(acc, 0.0f);
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
  const int l = d.n_fc - 1, win = d.width[l], wout = d.width[l + 1];
  const int o = blockIdx.x * kFcBlock + tid;
  if (o < wout) {
    const float* wt = params + d.w_off[l];  // transposed: [in][out]
    float acc[S];
    const float b = params[d.b_off[l] + o];
#pragma unroll
    for (int sm = 0; sm < S; ++sm) acc[sm] = b;
#pragma unroll 16
    for (int i = 0; i < win; ++i) {
      const float w = wt[(size_t)i * wout + o];
#pragma unroll
      for (int sm = 0; sm < S; ++sm) acc[sm] = fmaf(w, cur[i * S + sm], acc[sm]);
    }
#pragma unroll
    for (int sm = 0; sm < S; ++sm)
      if (sm < ns) out[(size_t)(n0 + sm) * wout + o] = fmaxf(acc[sm], 0.0f);
  }
}

// trilinear resize, ATen upsample_trilinear3d semantics with align_corners=False:
//   src = max(ratio * (dst + 0.5) - 0.5, 0), ratio = float(in) / out; i0 = int(src);
//   i1 = i0 + (i0 < in - 1); l1 = src - i0; l0 = 1 - l1
__device__ __forceinline__ void resize_axis(int d, float ratio, int n_in, int& i0, int& i1, float& l1) {
  float src = fmaf(ratio, (float)d + 0.5f, -0.5f);
  src = src < 0.0f ? 0.0f : src;
  i0 = min((int)src, n_in - 1);
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

// w0 * a + w1 * b with the rounding spelled out (one product rounded, then one fma): the three resize kernels
// must agree bit for bit, which "a * b + c * d" under -ffp-contract=fast does not promise.
__device__ __forceinline__ float blend(float w0, float a, float w1, float b) { return fmaf(w0, a, w1 * b); }

// grid: ceil(C * n_out^3 / 256) x N;  relu: max(., 0);  clamp > 0 clamps the result to [-clamp, clamp]
__global__ __launch_bounds__(256) void resize3_kernel(const float* __restrict__ in, int C, int n_in,
                                                      int n_out, int relu, float clamp,
                                                      float* __restrict__ out) {
  const size_t vo = (size_t)n_out * n_out * n_out, vi = (size_t)n_in * n_in * n_in;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)C * vo) return;
  const int n = blockIdx.y;
  const int c = (int)(idx / vo);
  const int r = (int)(idx - (size_t)c * vo);
  const int z = r % n_out, y = (r / n_out) % n_out, x = r / (n_out * n_out);
  const float ratio = (float)n_in / (float)n_out;
  int x0, x1, y0, y1, z0, z1;
  float lx, ly, lz;
  resize_axis(x, ratio, n_in, x0, x1, lx);
  resize_axis(y, ratio, n_in, y0, y1, ly);
  resize_axis(z, ratio, n_in, z0, z1, lz);
  const float* p = in + ((size_t)n * C + c) * vi;
#define AT(ix, iy, iz) p[((size_t)(ix) * n_in + (iy)) * n_in + (iz)]
  const float wx0 = 1.0f - lx, wy0 = 1.0f - ly, wz0 = 1.0f - lz;
  float v = blend(wx0, blend(wy0, blend(wz0, AT(x0, y0, z0), lz, AT(x0, y0, z1)),
                             ly, blend(wz0, AT(x0, y1, z0), lz, AT(x0, y1, z1))),
                  lx, blend(wy0, blend(wz0, AT(x1, y0, z0), lz, AT(x1, y0, z1)),
                            ly, blend(wz0, AT(x1, y1, z0), lz, AT(x1, y1, z1))));
#undef AT
  if (relu) v = fmaxf(v, 0.0f);
  if (clamp > 0.0f) v = fminf(fmaxf(v, -clamp), clamp);
  out[((size_t)n * C + c) * vo + r] = v;
}

// x / d == umulhi(x, magic_of(d)) for x < 2^16 and 2 <= d < 2^16; d == 1 has no 32-bit reciprocal: div_by() tests for it
__device__ __forceinline__ unsigned magic_of(int d) { return 0xffffffffu / (unsigned)d + 1u; }
__device__ __forceinline__ int div_by(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }   // (magic_of(1) == 0)

// (The swapped last layer's 1x1 mix inside this kernel's column loads -- conv1x1 launch and the tensor in between
// saved -- was built and measured for batches: 36 + 71 us as two launches, 140 us fused, four gathers per staged
// element with the tiles' 2.5x overlap; not kept.)
// The same resize for up-sampling (ratio <= 1), tiled through LDS.  resize3_kernel issues 8 gathers
// per output and is bound by the per-CU gather rate (0.8 TB/s of stores at 30^3 -> 64^3); here a
// workgroup owns an 8 x 8 (x, y) patch of output columns, loads the few input columns under it
// once (coalesced z-rows), interpolates them along z into a second LDS array and blends four of
// those per output -- the same expression tree, z innermost, as resize3_kernel.  Stores are
// whole z-rows.   grid: (tiles_x * tiles_y, C, N);  LDS: cols * (n_in + n_out) floats,
// cols = max_rx * max_ry input columns under a tile (computed by the host with the same arithmetic).
constexpr int kResizeTile = 8;
// tensors up to this many elements take the launch-count-saving fused forms (single decodes, up to 8 latents)
constexpr size_t kFewElements = (size_t)1 << 21;
// ... and the forward's resize with the 1x1x1 mix inside (resize3_mix_kernel, eight gathers per output) up to this many
constexpr size_t kFewElementsMix = (size_t)1 << SDFR_FEW_MIX_LOG2;
// A workgroup takes its tile of `ipw` consecutive (sample, channel) volumes one after the other (round 4): the tables,
// where each thread's staged elements come from and its x / y / z terms are formed ONCE -- they were two thirds of the
// 36 VALU instructions per output of a kernel that is VALU-bound (0.98 busy, profiles/r04_decoder_batched_pmc.md) --
// and the next volume's columns are in flight while this one is interpolated.   grid: (tiles^2, ceil(items / ipw))
// VEC: a thread owns FOUR consecutive z of its columns -- 16-byte LDS reads and 16-byte stores (a wave's store is 1 KB of
// one z-row instead of 256 bytes), a quarter of the address arithmetic; needs `out` 16-byte aligned.
template <int LOG_NO, bool VEC>  // n_out = 1 << LOG_NO (16 .. 128): index splits are shifts, z is fixed per thread
__global__ __launch_bounds__(256) void resize3_tiled_kernel(const float* __restrict__ in, int items, int ipw, int n_in,
                                                            int relu, float clamp, int max_cols,
                                                            float* __restrict__ out) {
  constexpr int n_out = 1 << LOG_NO, zmask = n_out - 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // per-axis interpolation tables of this tile: entries 0..7 x, 8..15 y (indices relative to a0 / b0)
  __shared__ int t_i0[16], t_i1[16], z_i0[n_out], z_i1[n_out];
  __shared__ float t_l[16], z_l[n_out];
  const int tid = threadIdx.x;
  constexpr int tiles = (n_out + kResizeTile - 1) / kResizeTile;
  const int tx0 = (blockIdx.x / tiles) * kResizeTile, ty0 = (blockIdx.x % tiles) * kResizeTile;
  const int it0 = (int)blockIdx.y * ipw, it1 = min(it0 + ipw, items);
  const float ratio = (float)n_in / (float)n_out;
  int a0, a1, b0, b1, t0, t1;
  float fl;
  resize_axis(tx0, ratio, n_in, a0, t1, fl);
  resize_axis(min(tx0 + kResizeTile, n_out) - 1, ratio, n_in, t0, a1, fl);
  resize_axis(ty0, ratio, n_in, b0, t1, fl);
  resize_axis(min(ty0 + kResizeTile, n_out) - 1, ratio, n_in, t0, b1, fl);
  const int rx = a1 - a0 + 1, ry = b1 - b0 + 1, cols = rx * ry;  // input columns x in [a0,a1], y in [b0,b1]
  if (tid < n_out) {
    resize_axis(tid, ratio, n_in, t0, t1, fl);
    z_i0[tid] = t0; z_i1[tid] = t1; z_l[tid] = fl;
  }
  if (tid >= 128 && tid < 144) {
    const int j = tid - 128, dd = min(((j < 8) ? tx0 : ty0) + (j & 7), n_out - 1);
    resize_axis(dd, ratio, n_in, t0, t1, fl);
    t_i0[j] = t0 - ((j < 8) ? a0 : b0); t_i1[j] = t1 - ((j < 8) ? a0 : b0); t_l[j] = fl;
  }
  float* col_in = lds;                            // [cols][n_in]
  float* col_z = lds + (((size_t)max_cols * n_in + 3) & ~(size_t)3);   // [cols][n_out], interpolated along z (16-byte aligned)
  const size_t vin = (size_t)n_in * n_in * n_in, vout = (size_t)1 << (3 * LOG_NO);
  // where this thread's staged elements come from (the first kStage x 256 of the tile's cols * n_in; more: the loop below)
  constexpr int kStage = 4;
  const int total = cols * n_in;
  const unsigned m_nin = magic_of(n_in), m_ry = magic_of(ry);
  auto source = [&](int i) {
    const int col = div_by(i, m_nin), z = i - col * n_in;
    const int cx = div_by(col, m_ry), cy = col - cx * ry;
    return ((a0 + cx) * n_in + (b0 + cy)) * n_in + z;
  };
  int soff[kStage];
#pragma unroll
  for (int j = 0; j < kStage; ++j) soff[j] = tid + 256 * j < total ? source(tid + 256 * j) : -1;
  float pre[kStage];
  auto fetch = [&](int it) {
    const float* src = in + (size_t)it * vin;
#pragma unroll
    for (int j = 0; j < kStage; ++j) pre[j] = soff[j] >= 0 ? src[soff[j]] : 0.0f;
  };
  fetch(it0);
  __syncthreads();   // tables
  const int z = tid & zmask;
  const int z0 = z_i0[z], z1 = z_i1[z];
  const float lz = z_l[z], wz0 = 1.0f - lz;
  // a thread's outputs: its z (VEC: its four z), the tile's x, and the y = sub, sub + P, ... of its wave part (P threads
  // share a z): the y terms once per y, the x terms once per thread
  constexpr int LOG_Q = VEC ? LOG_NO - 2 : LOG_NO;                           // threads per z-row
  constexpr int P = 256 >> LOG_Q, PY = P < kResizeTile ? P : kResizeTile;    // distinct y per pass
  constexpr int XS = P / PY;                                                 // x handled side by side (P > 8)
  static_assert(XS <= kResizeTile, "a tile row per thread at least");
  constexpr int NX = kResizeTile / XS, NY = kResizeTile / PY;
  const int sub = tid >> LOG_Q, jy0 = sub % PY, jx0 = sub / PY;
  const int ze = VEC ? (tid & ((1 << LOG_Q) - 1)) << 2 : z;                  // first z of the thread's outputs
  static_assert(n_out % kResizeTile == 0, "tiles are whole: no bounds checks below");
  float lxs[NX], wxs[NX];
  int r0s[NX], r1s[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int jx = jx0 + XS * i;
    lxs[i] = t_l[jx];
    wxs[i] = 1.0f - lxs[i];
    r0s[i] = (t_i0[jx] * ry) << LOG_NO;
    r1s[i] = (t_i1[jx] * ry) << LOG_NO;
  }
  float lys[NY], wys[NY];
  int c0s[NY], c1s[NY];
#pragma unroll
  for (int h = 0; h < NY; ++h) {
    const int jy = jy0 + PY * h;
    lys[h] = t_l[8 + jy];
    wys[h] = 1.0f - lys[h];
    c0s[h] = (t_i0[8 + jy] << LOG_NO) + ze;
    c1s[h] = (t_i1[8 + jy] << LOG_NO) + ze;
  }
  for (int it = it0; it < it1; ++it) {
#pragma unroll
    for (int j = 0; j < kStage; ++j)
      if (soff[j] >= 0) col_in[tid + 256 * j] = pre[j];
    if (total > 256 * kStage) {
      const float* src = in + (size_t)it * vin;
      for (int i = tid + 256 * kStage; i < total; i += 256) col_in[i] = src[source(i)];
    }
    __syncthreads();   // (and: every thread has left the previous volume's store loop, which reads col_z)
    if (it + 1 < it1) fetch(it + 1);
    for (int col = tid >> LOG_NO; col < cols; col += 256 >> LOG_NO)
      col_z[(col << LOG_NO) + z] = blend(wz0, col_in[col * n_in + z0], lz, col_in[col * n_in + z1]);
    __syncthreads();
    // (relu / clamp: workgroup-uniform branches around the whole store loop, not selects per output)
    float* dst_tile = out + (size_t)it * vout + ((size_t)tx0 << (2 * LOG_NO));
    auto emit = [&](auto post) {
#pragma unroll
      for (int h = 0; h < NY; ++h) {
        const float ly = lys[h], wy0 = wys[h];
        const float* cz0 = col_z + c0s[h];
        const float* cz1 = col_z + c1s[h];
        const unsigned at = ((unsigned)(ty0 + jy0 + PY * h) << LOG_NO) + (unsigned)ze;   // 32-bit offsets from a uniform base
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const unsigned jx = (unsigned)(jx0 + XS * i);
          if (VEC) {
            const f32x4 q00 = *reinterpret_cast<const f32x4*>(cz0 + r0s[i]), q01 = *reinterpret_cast<const f32x4*>(cz1 + r0s[i]);
            const f32x4 q10 = *reinterpret_cast<const f32x4*>(cz0 + r1s[i]), q11 = *reinterpret_cast<const f32x4*>(cz1 + r1s[i]);
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              v[k] = post(blend(wxs[i], blend(wy0, q00[k], ly, q01[k]), lxs[i], blend(wy0, q10[k], ly, q11[k])));
            *reinterpret_cast<f32x4*>(dst_tile + at + (jx << (2 * LOG_NO))) = v;
          } else {
            const float v = blend(wxs[i], blend(wy0, cz0[r0s[i]], ly, cz1[r0s[i]]), lxs[i],
                                  blend(wy0, cz0[r1s[i]], ly, cz1[r1s[i]]));
            dst_tile[at + (jx << (2 * LOG_NO))] = post(v);
          }
        }
      }
    };
    if (clamp > 0.0f) {
      if (relu) emit([clamp](float v) { return fminf(fmaxf(fmaxf(v, 0.0f), -clamp), clamp); });
      else emit([clamp](float v) { return fminf(fmaxf(v, -clamp), clamp); });
    } else if (relu) {
      emit([](float v) { return fmaxf(v, 0.0f); });
    } else {
      emit([](float v) { return v; });
    }
  }
}

// 1x1x1 convolution with few output channels, element-wise: one thread per voxel, the input read
// once (coalesced per channel), the same fmaf chain over ci and "+ bias" last as the MFMA path.
// wmat: the layer's [Kpad][16] matrix (co_tile 0).   grid: (ceil(vox / 256), N)
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_kernel(const float* __restrict__ in,
                                                      const float* __restrict__ wmat,
                                                      const float* __restrict__ bias, int Cin, int vox,
                                                      int relu, float* __restrict__ out) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= vox) return;
  const float* src = in + (size_t)blockIdx.y * Cin * vox + v;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
  for (int ci = 0; ci < Cin; ++ci) {
    const float a = src[(size_t)ci * vox];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = fmaf(a, wmat[ci * 16 + co], acc[co]);
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float r = acc[co] + bias[co];
    if (relu) r = fmaxf(r, 0.0f);
    out[((size_t)blockIdx.y * COUT + co) * vox + v] = r;
  }
}

// The same, four voxels per thread with 16-byte loads and stores (batches whose voxel count is a multiple of 4:
// 36 -> ~25 us per 256 latents of 4 x 30^3; the per-voxel chain is unchanged).   grid: (ceil(vox / 1024), N)
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_vec4_kernel(const float* __restrict__ in,
                                                           const float* __restrict__ wmat,
                                                           const float* __restrict__ bias, int Cin, int vox,
                                                           int relu, float* __restrict__ out) {
  const int v = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (v >= vox) return;
  const float* src = in + (size_t)blockIdx.y * Cin * vox + v;
  f32x4 acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
  for (int ci = 0; ci < Cin; ++ci) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)ci * vox);
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float w = wmat[ci * 16 + co];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[co][e] = fmaf(a[e], w, acc[co][e]);
    }
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r[e] = acc[co][e] + bias[co];
      if (relu) r[e] = fmaxf(r[e], 0.0f);
    }
    *reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.y * COUT + co) * vox + v) = r;
  }
}

// The swapped last layer of a single decode in one launch (one launch less per iteration of the captured loop):
// out = resize(conv1x1(in)), the 1x1 mix (same fmaf chain over ci and "+ bias" last as conv1x1_kernel) formed
// at each of the 8 corners, then resize3_kernel's expression tree -- bit-identical to the two launches.
// grid: (ceil(n_out^3 / 256), N)
template <int COUT>
__global__ __launch_bounds__(256) void resize3_mix_kernel(const float* __restrict__ in, const float* __restrict__ wmat,
                                                          const float* __restrict__ bias, int Cin, int n_in,
                                                          int n_out, int relu, float clamp,
                                                          float* __restrict__ out) {
  const size_t vo = (size_t)n_out * n_out * n_out, vi = (size_t)n_in * n_in * n_in;
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= vo) return;
  const int n = blockIdx.y;
  const int z = (int)(r % n_out), y = (int)((r / n_out) % n_out), x = (int)(r / ((size_t)n_out * n_out));
  const float ratio = (float)n_in / (float)n_out;
  int xi[2], yi[2], zi[2];
  float lx, ly, lz;
  resize_axis(x, ratio, n_in, xi[0], xi[1], lx);
  resize_axis(y, ratio, n_in, yi[0], yi[1], ly);
  resize_axis(z, ratio, n_in, zi[0], zi[1], lz);
  const float* p = in + (size_t)n * Cin * vi;
  float corner[8][COUT];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const size_t off = ((size_t)xi[k >> 2] * n_in + yi[(k >> 1) & 1]) * n_in + zi[k & 1];
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
    for (int ci = 0; ci < Cin; ++ci) {
      const float a = p[(size_t)ci * vi + off];
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] = fmaf(a, wmat[ci * 16 + co], acc[co]);
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) corner[k][co] = acc[co] + bias[co];
  }
  const float wx0 = 1.0f - lx, wy0 = 1.0f - ly, wz0 = 1.0f - lz;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float v = blend(wx0, blend(wy0, blend(wz0, corner[0][co], lz, corner[1][co]),
                               ly, blend(wz0, corner[2][co], lz, corner[3][co])),
                    lx, blend(wy0, blend(wz0, corner[4][co], lz, corner[5][co]),
                              ly, blend(wz0, corner[6][co], lz, corner[7][co])));
    if (relu) v = fmaxf(v, 0.0f);
    if (clamp > 0.0f) v = fminf(fmaxf(v, -clamp), clamp);
    out[((size_t)n * COUT + co) * vo + r] = v;
  }
}

// grid: ceil(count / 256);  clamp a tensor in place (only when the last layer already has the
// volume size and enforce_tsdf is set)
__global__ void clamp_kernel(float* __restrict__ x, size_t count, float clamp) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) x[i] = fminf(fmaxf(x[i], -clamp), clamp);
}


// Copy n floats (n % 4 == 0, both 16-byte aligned) global -> LDS with a whole workgroup: 16-byte
// pieces, eight loads in flight per thread before the first LDS store (a plain `lds[i] = g[i]`
// loop waits for every load before it issues the next).
// (eight in flight since round 6: a 432-tap layer's 27 KB are 6.75 vectors per thread, and a second pass of the loop is
// a second memory round trip in front of everything else the workgroup does -- ~1 us of a 5.6 us launch)
__device__ __forceinline__ void stage_to_lds(float* __restrict__ dst, const float* __restrict__ src, int n,
                                             int tid) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  const int n4 = n >> 2;
  constexpr int kIn = 8;
  for (int i = tid; i < n4; i += kIn * 256) {
    f32x4 v[kIn];
#pragma unroll
    for (int u = 0; u < kIn; ++u) v[u] = s4[min(i + u * 256, n4 - 1)];
#pragma unroll
    for (int u = 0; u < kIn; ++u)
      if (i + u * 256 < n4) d4[i + u * 256] = v[u];
  }
}


// Valid 3-D convolution as an im2col contraction on v_mfma_f32_16x16x4_f32.
//   wmat  [Kpad][16]  weight matrix, wmat[kk][co] = W[co][ci][a][b][c], kk = ci*k^3 + (a*k+b)*k+c,
//                     zero for kk >= K or co >= Cout_tile
//   taps  [Kpad]      input offset of tap kk: ci*n^3 + (a*n + b)*n + c  (0 for padding taps)
// One wave = 16 consecutive output positions x 16 columns; lane l supplies
// A[position l&15][tap l>>4] and B[tap l>>4][column l&15]; the accumulator holds
// D[position 4*(l>>4)+r][column l&15], r = 0..3.   grid: (ceil(tiles / (4*TPW)), co_tiles, N)
//
// Z-grouping (zg > 1): with Cout <= 8 most of the 16 columns would be padding, and the kernel is
// bound by its gathers (one per lane and MFMA), not by the matrix cores.  So zg consecutive z
// outputs share a row: position = (x, y, z-group), column = (dz, co), and the contraction runs
// over K' = Cin*k*k*(k+zg-1) taps with
//   wmat[(ci,a,b,c')][(dz,co)] = W[co][ci][a][b][c'-dz]   (0 outside 0 <= c'-dz < k).
// For the 8 -> 4 channel layer (zg = 3): 12 of 16 columns used, 120 instead of 216 gathers and
// 1.9 instead of 3.4 MFMAs per output voxel.
// (An LDS-staged variant -- input patch of a 4 x 4 column tile in LDS, operands from ds_read --
// was built and measured on the same layer: 772 us vs 860 us for the plain gather kernel; patch
// load 319 + MFMA 220 + stores 58 + staging 120, serialised by 2 workgroups per CU.  Not kept.)
// RESIDENT (batches of small layers, e.g. 16 x 8^3 -> 16 x 6^3: 14 tiles per sample): one workgroup per (sample,
// column tile) stages the sample's whole input in LDS beside the weights and walks over all of the sample's tiles, the
// A operands read from LDS.  The plain form staged 27 KB of weights per FOUR tiles and gathered every A operand from
// global memory: 27.6 us per 256 mug latents where the MFMAs need 5; RESIDENT: see DESIGN 9.5.  Same MFMA sequence
// per tile: bit-identical.   grid: (1, co_tiles, N);  LDS: kpad * 17 + Cin * n^3 floats (Cin * n^3 % 4 == 0, aligned: host)
template <bool RESIDENT>
__global__ __launch_bounds__(256) void conv3d_mfma_kernel(
    const float* __restrict__ in, const float* __restrict__ wmat, const int* __restrict__ taps,
    const float* __restrict__ bias, float* __restrict__ out, int Cin, int Cout, int n, int m,
    int kpad, int relu, int tiles_per_wave, int zg, int split_k) {
  extern __shared__ float lds[];
  float* w_l = lds;                                       // [kpad][16]
  int* tap_l = reinterpret_cast<int*>(lds + (size_t)kpad * 16);  // [kpad]
  float* red = lds + (size_t)kpad * 17;                   // split_k: [4 waves][64 lanes][4]
  float* in_l = lds + (size_t)kpad * 17;                  // RESIDENT (never with split_k): [Cin][n][n][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co_tile = blockIdx.y, nb = blockIdx.z;
  const float* wsrc = wmat + (size_t)co_tile * kpad * 16;
  stage_to_lds(w_l, wsrc, kpad * 16, tid);
  stage_to_lds(reinterpret_cast<float*>(tap_l), reinterpret_cast<const float*>(taps), kpad, tid);
  if (RESIDENT) stage_to_lds(in_l, in + (size_t)nb * Cin * n * n * n, Cin * n * n * n, tid);
  __syncthreads();

  const int zgn = m / zg, mrows = m * m * zgn;  // positions: (x, y, z-group), z-group fastest
  const size_t mv = (size_t)m * m * m, nv = (size_t)n * n * n;
  const float* src = in + (size_t)nb * Cin * nv;
  const int n_tiles = (mrows + 15) / 16;
  const int row = lane & 15, kq = lane >> 4;
  // this lane's column (C/D column = lane & 15)
  const int dz = zg > 1 ? row / Cout : 0;
  const int co = zg > 1 ? row % Cout : co_tile * 16 + row;
  const bool col_ok = zg > 1 ? row < zg * Cout : co < Cout;
  const float bv = col_ok ? bias[co] : 0.0f;
  if (split_k) {
    // Single latents: a layer has fewer tiles than the chip has SIMDs and a wave's K loop is a chain of
    // kpad / 32 gather round trips (8 us of a 10 us launch for the 16 -> 16 layer).  The four waves of a
    // workgroup share ONE tile, each taking every fourth 32-tap chunk, and wave 0 adds the four partial
    // accumulators, (w0 + w1) + (w2 + w3).   grid: (tiles, co_tiles, N)
    const int t = blockIdx.x;
    const int pos = min(t * 16 + row, mrows - 1);
    const int xy = pos / zgn, g = pos - xy * zgn, x = xy / m, y = xy - x * m;
    const float* base = src + ((size_t)x * n + y) * n + g * zg;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int full = kpad / 32;
    for (int c = wave; c < full; c += 4) {
      const int kk0 = c * 32;
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = kk0 + 4 * u + kq;
        a[u] = base[tap_l[kk]];
        b[u] = w_l[kk * 16 + row];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    if (wave == (full & 3)) {  // the wave with the fewest chunks takes the tail
      for (int kk0 = full * 32; kk0 < kpad; kk0 += 4) {
        const int kk = kk0 + kq;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[tap_l[kk]], w_l[kk * 16 + row], acc, 0, 0, 0);
      }
    }
    *reinterpret_cast<f32x4*>(red + (size_t)tid * 4) = acc;
    __syncthreads();
    if (wave != 0 || !col_ok) return;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(red + (size_t)lane * 4);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(red + (size_t)(64 + lane) * 4);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(red + (size_t)(128 + lane) * 4);
    const f32x4 a3 = *reinterpret_cast<const f32x4*>(red + (size_t)(192 + lane) * 4);
    float* dst = out + ((size_t)nb * Cout + co) * mv + dz;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = t * 16 + kq * 4 + r;
      if (p >= mrows) continue;
      const int pxy = p / zgn, pg = p - pxy * zgn;
      float v = ((a0[r] + a1[r]) + (a2[r] + a3[r])) + bv;
      if (relu) v = fmaxf(v, 0.0f);
      dst[(size_t)pxy * m + pg * zg] = v;
    }
    return;
  }
  if (RESIDENT) {
    // two tiles per wave and pass: two independent accumulator chains (a wave alone on its SIMD otherwise waits out
    // every MFMA's latency), the weight operand read once for both
    for (int t = wave; t < n_tiles; t += 8) {
      const int t1 = t + 4;
      const float* base[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pos = min((q ? t1 : t) * 16 + row, mrows - 1);
        const int xy = pos / zgn, g = pos - xy * zgn, x = xy / m, y = xy - x * m;
        base[q] = in_l + ((size_t)x * n + y) * n + g * zg;
      }
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
      // (fetching the next group of operands while this group's MFMAs run -- written out as a two-stage loop -- was
      // measured: 19.4 -> 23.8 us; the compiler's waits serialise it)
      int kk0 = 0;
      for (; kk0 + 16 <= kpad; kk0 += 16) {
        float a0[4], a1[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = kk0 + 4 * u + kq, tp = tap_l[kk];
          a0[u] = base[0][tp];
          a1[u] = base[1][tp];
          b[u] = w_l[kk * 16 + row];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b[u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b[u], acc1, 0, 0, 0);
        }
      }
      for (; kk0 < kpad; kk0 += 4) {
        const int kk = kk0 + kq, tp = tap_l[kk];
        const float b = w_l[kk * 16 + row];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(base[0][tp], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(base[1][tp], b, acc1, 0, 0, 0);
      }
      if (col_ok) {
        float* dst = out + ((size_t)nb * Cout + co) * mv + dz;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int p = (q ? t1 : t) * 16 + kq * 4 + r;
            if (p >= mrows) continue;
            const int pxy = p / zgn, pg = p - pxy * zgn;
            float v = (q ? acc1[r] : acc0[r]) + bv;
            if (relu) v = fmaxf(v, 0.0f);
            dst[(size_t)pxy * m + pg * zg] = v;
          }
      }
    }
    return;
  }
  const int first = (blockIdx.x * 4 + wave) * tiles_per_wave;
  for (int t = first; t < first + tiles_per_wave && t < n_tiles; ++t) {
    const int pos = min(t * 16 + row, mrows - 1);
    const int xy = pos / zgn, g = pos - xy * zgn, x = xy / m, y = xy - x * m;
    const float* base = src + ((size_t)x * n + y) * n + g * zg;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // the gathers of 8 K-steps are issued together and only then consumed: the loop is a chain of
    // dependent (LDS tap -> global gather -> MFMA) round trips otherwise
    int kk0 = 0;
    for (; kk0 + 32 <= kpad; kk0 += 32) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = kk0 + 4 * u + kq;
        a[u] = base[tap_l[kk]];
        b[u] = w_l[kk * 16 + row];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    for (; kk0 < kpad; kk0 += 4) {
      const int kk = kk0 + kq;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[tap_l[kk]], w_l[kk * 16 + row], acc, 0, 0, 0);
    }
    if (col_ok) {
      float* dst = out + ((size_t)nb * Cout + co) * mv + dz;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = t * 16 + kq * 4 + r;
        if (p >= mrows) continue;
        const int pxy = p / zgn, pg = p - pxy * zgn;  // (zg == 1: pxy * m + pg == p)
        float v = acc[r] + bv;
        if (relu) v = fmaxf(v, 0.0f);
        dst[(size_t)pxy * m + pg * zg] = v;
      }
    }
  }
}


// The direct convolutions' epilogue: bias, ReLU, store of a thread's ZR x COUT outputs -- and (mix_out != NULL) the
// 1x1x1 layer that follows, applied while they are in registers (see conv3d_direct_kernel).
template <int COUT, int ZR>
__device__ __forceinline__ void direct_epilogue(float (&acc)[ZR][COUT], const float* __restrict__ bias, int relu,
                                                float* __restrict__ out, const float* __restrict__ mix_w,
                                                const float* __restrict__ mix_b, int mix_co, float* __restrict__ mix_out,
                                                int nb, size_t mv, int x, int y, int z0, int m) {
  if (x < m && y < m && z0 < m) {
    const size_t col = ((size_t)x * m + y) * m + z0;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float bv = bias[co];
#pragma unroll
      for (int z = 0; z < ZR; ++z) {
        float r = acc[z][co] + bv;
        if (relu) r = fmaxf(r, 0.0f);
        acc[z][co] = r;
      }
      if (out) {
        float* dst = out + ((size_t)nb * COUT + co) * mv + col;
#pragma unroll
        for (int z = 0; z < ZR; ++z)
          if (z0 + z < m) dst[z] = acc[z][co];
      }
    }
    if (mix_out) {
      for (int c2 = 0; c2 < mix_co; ++c2) {
        float* dst = mix_out + ((size_t)nb * mix_co + c2) * mv + col;
        const float b2 = mix_b[c2];
#pragma unroll
        for (int z = 0; z < ZR; ++z) {
          float a2 = 0.0f;
#pragma unroll
          for (int co = 0; co < COUT; ++co) a2 = fmaf(acc[z][co], mix_w[co * 16 + c2], a2);
          if (z0 + z < m) dst[z] = a2 + b2;
        }
      }
    }
  }
}

// Batched 3x3x3 convolution on the vector ALUs.  For the decoder's narrow layers (4 .. 16 output
// channels) the matrix cores are the wrong tool: a 16-column MFMA tile is mostly padding, and both
// MFMA forms above are bound by how they fetch their operands (560 us gather-bound, 772 us LDS-staged
// for the 8 -> 4 layer at 256 samples; its 6 GMAC are 76 us of packed-fp32 FMA).  Here:
//   * a workgroup owns TX x TY output columns (all of z) and stages the input patch of CK channels
//     at a time in LDS (coalesced runs, as many workgroups per CU as fit: loads of one overlap the
//     FMAs of another);
//   * a thread owns one column and ZR = 4 consecutive z: 4 x COUT accumulators in registers;
//   * per (ci, a, b) it reads its 6-float z-run once (ds_read_b128 + b64) and issues 3 x 4 x COUT FMAs
//     whose weight operands are wave-uniform -> scalar registers (no operand traffic at all).
// wd: [Cin][3][3][3][COUT].   grid: (tiles_x * tiles_y, 1, N), block 256 = TX*TY columns x ZC chunks
template <int COUT, bool VEC4>
__global__ __launch_bounds__(256) void conv3d_direct_kernel(
    const float* __restrict__ in, const float* __restrict__ wd, const float* __restrict__ bias,
    float* __restrict__ out, int Cin, int n, int pz, int m, int relu, int TX, int TY, int ZC, int CK,
    const float* __restrict__ mix_w, const float* __restrict__ mix_b, int mix_co, float* __restrict__ mix_out) {
  // (pz: floats between z-rows of `in`, >= n -- the padded tensors of the backward pass have their rows 16 bytes apart)
  // mix_out != NULL: the 1x1x1 layer that FOLLOWS this one (mix_co <= 4 output channels, its [Kpad][16] matrix and
  // bias; no ReLU: it is a layer whose resize comes after it) is applied to this layer's outputs while they are in
  // registers -- conv1x1_kernel's chain, channel by channel, "+ bias" last, so the same numbers -- and written to
  // mix_out [N][mix_co][m^3]: the conv1x1 launch and its read of this layer's output are saved (22 us and 110 MB per
  // 256 mug latents); `out` may then be NULL (a forward without a tape needs this layer's output nowhere else).
  constexpr int K = 3, ZR = 4;
  constexpr int kUnrollB = COUT >= 8 ? 1 : K;  // keep a step's weights within the scalar registers
  extern __shared__ float tile[];  // [CK][IX][IY][np4]
  const int tid = threadIdx.x;
  // ZC threads per column (a power of two), TX * TY <= 256 / ZC columns: threads beyond them only help to load
  const int IX = TX + K - 1, IY = TY + K - 1;
  const int tiles_y = (m + TY - 1) / TY;
  const int tx0 = (blockIdx.x / tiles_y) * TX, ty0 = (blockIdx.x % tiles_y) * TY;
  const int nb = blockIdx.z;
  const size_t nv = (size_t)n * n * pz, mv = (size_t)m * m * m;
  const float* src = in + (size_t)nb * Cin * nv;
  const int col = tid / ZC, zc = tid - col * ZC;
  const int lx = col / TY, ly = col - lx * TY, z0 = col < TX * TY ? zc * ZR : m;   // (z0 = m: no outputs)
  float acc[ZR][COUT];
#pragma unroll
  for (int z = 0; z < ZR; ++z)
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[z][co] = 0.0f;
  const int run = IY * pz, y_valid = (n - ty0) * pz, np4 = (pz + 3) & ~3;   // np4: LDS row pitch
  // VEC4: 16-byte loads -- every run starts and ends on a 16-byte boundary (rows pz % 4 == 0 floats apart; host).
  // 16-byte path: the patch of chunk i+1 is fetched into registers while chunk i is being
  // convolved (<= 6 loads per thread: a chunk is <= 24 KB), so a workgroup waits for memory once.
  // (Where each of a thread's <= 6 vectors of a chunk comes from does not depend on the chunk: channel within the
  // chunk and offset are formed once, packed into one register -- two integer divisions per vector per chunk were
  // 1/3 of the kernel's VALU instructions, and the kernel is VALU-bound.)
  constexpr bool vec4 = VEC4;
  const int run4 = run >> 2;
  constexpr int kPre = VEC4 ? 6 : 1;
  f32x4 pre[kPre];
  int pf[kPre];   // (channel within the chunk << 24) | float offset (0xffffff: zeros); channel 127: not part of any chunk
  if (VEC4) {
    const unsigned m_run = magic_of(run4), m_ix = magic_of(IX);
    const int e4_full = CK * IX * run4;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int e = tid + 256 * j;
      const int slab = div_by(min(e, e4_full - 1), m_run), off = (e - slab * run4) << 2;
      const int ci = div_by(slab, m_ix), x = tx0 + slab - ci * IX;
      const int where = (x < n && off < y_valid) ? (int)(ci * nv) + (x * n + ty0) * pz + off : 0xffffff;
      pf[j] = ((e < e4_full ? ci : 127) << 24) | where;
    }
  }
  auto prefetch = [&](int c0) {
    const int ck = min(CK, Cin - c0);
    const float* base = src + (size_t)c0 * nv;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int where = pf[j] & 0xffffff;
      pre[j] = ((pf[j] >> 24) < ck && where != 0xffffff) ? *reinterpret_cast<const f32x4*>(base + where)
                                                         : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  if (vec4) prefetch(0);
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    const int ck = min(CK, Cin - c0), n_slab = ck * IX;
    __syncthreads();  // the previous chunk has been consumed
    if (vec4) {
#pragma unroll
      for (int j = 0; j < kPre; ++j)
        if ((pf[j] >> 24) < ck) reinterpret_cast<f32x4*>(tile)[tid + 256 * j] = pre[j];
    } else {
      // (LDS rows are np4 = n rounded up to 4 floats apart, so that the z-runs below are 16-byte aligned whatever n)
      const unsigned m_n = magic_of(pz);
      for (int off = tid; off < run; off += 256) {
        const int l_off = off + div_by(off, m_n) * (np4 - pz);
        for (int s0 = 0; s0 < n_slab; s0 += 8) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int slab = min(s0 + u, n_slab - 1), ci = slab / IX, x = tx0 + slab - ci * IX;
            const float* row = src + (size_t)(c0 + ci) * nv + ((size_t)min(x, n - 1) * n + ty0) * pz;
            v[u] = (x < n && off < y_valid) ? row[off] : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (s0 + u < n_slab) tile[(s0 + u) * IY * np4 + l_off] = v[u];
        }
      }
    }
    __syncthreads();
    if (vec4 && c0 + CK < Cin) prefetch(c0 + CK);
    if (z0 < m) {
      for (int ci = 0; ci < ck; ++ci) {
        // (a is not unrolled: the 27 * COUT weights of a channel do not fit the scalar registers,
        // and spilled scalars come back through v_readlane)
#pragma unroll 1
        for (int a = 0; a < K; ++a)
#pragma unroll kUnrollB
          for (int b = 0; b < K; ++b) {
            const float* p = tile + ((ci * IX + lx + a) * IY + ly + b) * np4 + z0;
            // the 6-float z-run as one 16-byte and one 8-byte LDS read (the runs of a wave then cover
            // all banks evenly; six 4-byte reads hit 8 banks 8 ways each: 3.7x slower, measured).  Values
            // past the end of a row (the next row's, or the unwritten floats between n and np4) belong to
            // outputs z >= m, which are never stored.
            float v[ZR + K - 1];
            {
              const f32x4 q = *reinterpret_cast<const f32x4*>(p);
              const float2 r = *reinterpret_cast<const float2*>(p + 4);
              v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; v[4] = r.x; v[5] = r.y;
            }
            const float* w = wd + ((((size_t)(c0 + ci) * K + a) * K + b) * K) * COUT;  // wave-uniform
#pragma unroll
            for (int c = 0; c < K; ++c)
#pragma unroll
              for (int z = 0; z < ZR; ++z)
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[z][co] = fmaf(v[z + c], w[c * COUT + co], acc[z][co]);
          }
      }
    }
  }
  direct_epilogue<COUT, ZR>(acc, bias, relu, out, mix_w, mix_b, mix_co, mix_out, nb, mv, tx0 + lx, ty0 + ly, z0, m);
}

// conv3d_direct_kernel with the trilinear resize IN FRONT of the layer folded into its patch load (round 4): the
// up-sampled tensor (8 x 32^3 floats per sample for the mug decoder's third layer -- 268 MB per 256 latents, written
// by resize3_tiled_kernel and read back here) never exists.  A workgroup stages, per chunk of CK channels, the few
// COARSE columns under its patch (all of z; prefetched into registers during the previous chunk's FMAs) and forms the
// fine patch from them in LDS with the expression tree of resize3_kernel -- z innermost, then y, then x, every blend
// one product and one fma -- so the layer's output is bit for bit the two-launch result.
//   thread <-> (fine z, patch row py): its z and y terms live in registers for the whole kernel; it walks the patch's
//   x, holding the y- and z-blended values of the two coarse x-columns around it and advancing them as x crosses a
//   coarse cell (the advance is workgroup-uniform): per fine element 2 LDS table reads, 2 VALU, 1 LDS write, and per
//   coarse column 4 LDS reads + 6 VALU.
// n (the fine size) is a power of two <= 64 (np4 == n; 256 / n patch rows are served side by side).
//   in: [N][Cin][ni^3];  LDS (dynamic): tile [CK][IX][IY][n] + 8 | coarse [CK][max_cols][ni];  grid as conv3d_direct_kernel
template <int COUT>
__global__ __launch_bounds__(256) void conv3d_direct_up_kernel(
    const float* __restrict__ in, int ni, const float* __restrict__ wd, const float* __restrict__ bias,
    float* __restrict__ out, int Cin, int n, int log_n, int m, int relu, int TX, int TY, int ZC, int CK, int max_cols) {
  constexpr int K = 3, ZR = 4;
  constexpr int kUnrollB = COUT >= 8 ? 1 : K;
  constexpr int kMaxPatch = 40;   // IX, IY <= 34 (host)
  extern __shared__ float tile[];  // [CK][IX][IY][n] (+ 8), then the coarse columns
  __shared__ int x_i0[kMaxPatch], x_i1[kMaxPatch], y_i0[kMaxPatch], y_i1[kMaxPatch];
  __shared__ float x_l[kMaxPatch], y_l[kMaxPatch];
  const int tid = threadIdx.x;
  const int IX = TX + K - 1, IY = TY + K - 1;
  const int tiles_y = (m + TY - 1) / TY;
  const int tx0 = (blockIdx.x / tiles_y) * TX, ty0 = (blockIdx.x % tiles_y) * TY;
  const int nb = blockIdx.z;
  const size_t vi = (size_t)ni * ni * ni, mv = (size_t)m * m * m;
  const float* src = in + (size_t)nb * Cin * vi;
  const int col = tid / ZC, zc = tid - col * ZC;
  const int lx = col / TY, ly = col - lx * TY, z0 = col < TX * TY ? zc * ZR : m;
  float acc[ZR][COUT];
#pragma unroll
  for (int z = 0; z < ZR; ++z)
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[z][co] = 0.0f;
  const float ratio = (float)ni / (float)n;
  // coarse columns under the patch: x in [a0, a1], y in [b0, b1] (patch pixels beyond the tensor are zeros)
  int a0, a1, b0, b1, t0, t1;
  float fl;
  resize_axis(tx0, ratio, ni, a0, t1, fl);
  resize_axis(min(tx0 + IX, n) - 1, ratio, ni, t0, a1, fl);
  resize_axis(ty0, ratio, ni, b0, t1, fl);
  resize_axis(min(ty0 + IY, n) - 1, ratio, ni, t0, b1, fl);
  const int rx = a1 - a0 + 1, ry = b1 - b0 + 1, cols = rx * ry;
  if (tid < IX) {
    resize_axis(min(tx0 + tid, n - 1), ratio, ni, t0, t1, fl);
    x_i0[tid] = tx0 + tid < n ? t0 - a0 : -1;   // -1: outside the tensor
    x_i1[tid] = t1 - a0;
    x_l[tid] = fl;
  } else if (tid >= 64 && tid < 64 + IY) {
    const int j = tid - 64;
    resize_axis(min(ty0 + j, n - 1), ratio, ni, t0, t1, fl);
    y_i0[j] = ty0 + j < n ? t0 - b0 : -1;
    y_i1[j] = t1 - b0;
    y_l[j] = fl;
  }
  float* coarse = tile + (size_t)CK * IX * IY * n + 8;   // [CK][max_cols][ni]
  // this thread's fine z and its terms
  const int fz = tid & (n - 1), slot = tid >> log_n, slots = 256 >> log_n;
  int cz0, cz1;
  float lz;
  resize_axis(fz, ratio, ni, cz0, cz1, lz);
  const float wz0 = 1.0f - lz;
  // the coarse values of a chunk: <= kPre per thread, where from does not depend on the chunk
  constexpr int kPre = 8;
  float pre[kPre];
  int pf[kPre];   // (channel within the chunk << 24) | float offset within the channel; channel 127: none
  {
    const unsigned m_cn = magic_of(cols * ni), m_ni = magic_of(ni), m_ry = magic_of(ry);
    const int e_full = CK * cols * ni;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int e = tid + 256 * j;
      const int ch = div_by(min(e, e_full - 1), m_cn), r = min(e, e_full - 1) - ch * cols * ni;
      const int cc = div_by(r, m_ni), z = r - cc * ni;
      const int cx = div_by(cc, m_ry), cy = cc - cx * ry;
      pf[j] = ((e < e_full ? ch : 127) << 24) | (((a0 + cx) * ni + (b0 + cy)) * ni + z);
    }
  }
  auto prefetch = [&](int c0) {
    const int ck = min(CK, Cin - c0);
    const float* base = src + (size_t)c0 * vi;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int ch = pf[j] >> 24;
      pre[j] = ch < ck ? base[(size_t)ch * vi + (pf[j] & 0xffffff)] : 0.0f;
    }
  };
  prefetch(0);
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    const int ck = min(CK, Cin - c0);
    __syncthreads();  // the previous chunk has been consumed (and, first time round, the tables are written)
#pragma unroll
    for (int j = 0; j < kPre; ++j)
      if ((pf[j] >> 24) < ck) {
        // [ch][cc][z] with max_cols columns per channel
        const int e = tid + 256 * j, ch = pf[j] >> 24;
        coarse[(ch * max_cols) * ni + (e - ch * cols * ni)] = pre[j];
      }
    __syncthreads();
    if (c0 + CK < Cin) prefetch(c0 + CK);
    // the fine patch of the chunk
    for (int py = slot; py < IY; py += slots) {
      const int yi0 = y_i0[py], yi1 = y_i1[py];
      const float lyv = y_l[py], wy0 = 1.0f - lyv;
      const bool y_in = yi0 >= 0;
      // the four corner offsets of this thread within a coarse x-column (clamped for rows outside the tensor)
      const int o00 = max(yi0, 0) * ni + cz0, o01 = max(yi0, 0) * ni + cz1, o10 = yi1 * ni + cz0, o11 = yi1 * ni + cz1;
      const int xstride = ry * ni;
      for (int ch = 0; ch < ck; ++ch) {
        const float* cc = coarse + ch * max_cols * ni;
        float* row = tile + (ch * IX * IY + py) * n + fz;
        // value of the coarse x-column cx at (this y, this z): z innermost, then y (resize3_kernel's tree)
        auto column = [&](int cx) {
          const float* p = cc + cx * xstride;   // (cx: scalar)
          return blend(wy0, blend(wz0, p[o00], lz, p[o01]), lyv, blend(wz0, p[o10], lz, p[o11]));
        };
        int cur = -2;   // scalar: the x tables are the same for every thread
        float ya = 0.0f, yb = 0.0f;
        for (int px = 0; px < IX; ++px) {
          const int xi0 = __builtin_amdgcn_readfirstlane(x_i0[px]), xi1 = __builtin_amdgcn_readfirstlane(x_i1[px]);
          float v = 0.0f;
          if (xi0 >= 0) {
            if (xi0 != cur) {
              ya = (xi0 == cur + 1) ? yb : column(xi0);
              cur = xi0;
              yb = column(min(xi0 + 1, rx - 1));
            }
            const float lxv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x_l[px])));
            v = blend(1.0f - lxv, ya, lxv, xi1 == xi0 ? ya : yb);
          }
          row[px * IY * n] = y_in ? v : 0.0f;
        }
      }
    }
    __syncthreads();
    if (z0 < m) {
      for (int ci = 0; ci < ck; ++ci) {
#pragma unroll 1
        for (int a = 0; a < K; ++a)
#pragma unroll kUnrollB
          for (int b = 0; b < K; ++b) {
            const float* p = tile + ((ci * IX + lx + a) * IY + ly + b) * n + z0;
            float v[ZR + K - 1];
            {
              const f32x4 q = *reinterpret_cast<const f32x4*>(p);
              const float2 r = *reinterpret_cast<const float2*>(p + 4);
              v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; v[4] = r.x; v[5] = r.y;
            }
            const float* w = wd + ((((size_t)(c0 + ci) * K + a) * K + b) * K) * COUT;  // wave-uniform
#pragma unroll
            for (int c = 0; c < K; ++c)
#pragma unroll
              for (int z = 0; z < ZR; ++z)
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[z][co] = fmaf(v[z + c], w[c * COUT + co], acc[z][co]);
          }
      }
    }
  }
  direct_epilogue<COUT, ZR>(acc, bias, relu, out, nullptr, nullptr, 0, nullptr, nb, mv, tx0 + lx, ty0 + ly, z0, m);
}

// ---- backward (VJP to the latent; weights are constants) -------------------------------------

// out[N][C][np][np][pz] = zero-padded (by `pad` on every side) copy of g[N][C][m^3], multiplied by the
// ReLU mask of the layer's forward output when `act` is given.  The padded tensor turns the
// data-gradient of a valid convolution into another valid convolution (with flipped weights).
__global__ __launch_bounds__(256) void pad_mask_kernel(const float* __restrict__ g,
                                                       const float* __restrict__ act, int C, int m,
                                                       int pad, int pz, float* __restrict__ out) {
  const int np = m + 2 * pad;   // rows are pz >= np floats apart (floats np .. pz - 1 of a row: zeros)
  const size_t vp = (size_t)np * np * pz, vm = (size_t)m * m * m;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)C * vp) return;
  const int n = blockIdx.y;
  const int c = (int)(idx / vp);
  const int r = (int)(idx - (size_t)c * vp);
  const int z = r % pz - pad, y = (r / pz) % np - pad, x = r / (pz * np) - pad;
  float v = 0.0f;
  if (x >= 0 && x < m && y >= 0 && y < m && z >= 0 && z < m) {
    const size_t src = ((size_t)n * C + c) * vm + ((size_t)x * m + y) * m + z;
    v = g[src];
    if (act && !(act[src] > 0.0f)) v = 0.0f;
  }
  out[((size_t)n * C + c) * vp + r] = v;
}

// weight of output index d on input index i along one axis of the trilinear resize
__device__ __forceinline__ float resize_weight(int d, int i, float ratio, int n_in) {
  int i0, i1;
  float l1;
  resize_axis(d, ratio, n_in, i0, i1, l1);
  return (i0 == i ? 1.0f - l1 : 0.0f) + (i1 == i ? l1 : 0.0f);
}

// Transpose of resize3_kernel, one axis per launch (the interpolation is separable), in gather
// form: deterministic, no atomics.  (All three axes in one launch -- to save two ~5 us launches per
// resize in the captured loop -- was measured: the <= 10^3-term gather per voxel costs more than it
// saves, 182 -> 221 us for a forward + VJP; the three passes run on an LDS-staged block in one launch,
// bit-identical: 182 -> 196 us, and 2.4 -> 5.3 ms for 256 latents -- the per-term weight arithmetic on few,
// fat workgroups loses to three thin, fully parallel launches.)  The tensor is viewed as [outer][n_out][inner] -> [outer][n_in][inner]:
//   g_in[o][i][r] = sum over the few d with i0(d) == i or i1(d) == i of w(d -> i) * g_out[o][d][r]
__global__ __launch_bounds__(256) void resize_axis_backward_kernel(const float* __restrict__ g_out,
                                                                   size_t outer, int n_in, int n_out,
                                                                   size_t inner,
                                                                   float* __restrict__ g_in) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= outer * n_in * inner) return;
  const size_t r = idx % inner;
  const int i = (int)((idx / inner) % n_in);
  const size_t o = idx / (inner * n_in);
  const float ratio = (float)n_in / (float)n_out, inv = (float)n_out / (float)n_in;
  int d0 = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
  int d1 = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  d0 = max(d0, 0);
  d1 = min(d1, n_out - 1);
  const float* p = g_out + (o * n_out) * inner + r;
  float acc = 0.0f;
  for (int d = d0; d <= d1; ++d) {
    const float w = resize_weight(d, i, ratio, n_in);
    if (w != 0.0f) acc = fmaf(w, p[(size_t)d * inner], acc);
  }
  g_in[idx] = acc;
}

constexpr int kZyTaps = 10;  // longest source range resize_zy_backward_kernel takes (a 2x resize: 8)
// range of output indices d that can carry weight on input index i (a superset; zero weights are skipped)
__device__ __forceinline__ void resize_sources(int i, int n_in, int n_out, int& d0, int& d1) {
  const float inv = (float)n_out / (float)n_in;
  d0 = max((int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1, 0);
  d1 = min((int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1, n_out - 1);
}

// The z pass and the y pass of the transposed resize in one launch (single latents: the captured loop is bound
// by its launch count; a 2x resize has <= 4 x 4 weighted terms per voxel).  [outer][n_out][n_out] ->
// [outer][n_in][n_in]; the inner chain (over z) and the outer chain (over y) are the two passes' fmaf chains in
// their order, so the result is bit-identical to resize_axis_backward_kernel run twice.
__global__ __launch_bounds__(256) void resize_zy_backward_kernel(const float* __restrict__ g_out, size_t outer,
                                                                 int n_in, int n_out, float* __restrict__ g_in) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= outer * n_in * n_in) return;
  const int iz = (int)(idx % n_in), iy = (int)((idx / n_in) % n_in);
  const size_t o = idx / ((size_t)n_in * n_in);
  const float ratio = (float)n_in / (float)n_out;
  int y0, y1, z0, z1;
  resize_sources(iy, n_in, n_out, y0, y1);
  resize_sources(iz, n_in, n_out, z0, z1);
  // the z weights do not depend on the row: form them once; a tap past z1 re-reads z1 with weight 0
  // (the host takes this kernel only when no range is longer than kZyTaps)
  float wz[kZyTaps];
  int oz[kZyTaps];
#pragma unroll
  for (int k = 0; k < kZyTaps; ++k) {
    const int dz = min(z0 + k, z1);
    wz[k] = (z0 + k <= z1) ? resize_weight(dz, iz, ratio, n_in) : 0.0f;
    oz[k] = dz;
  }
  const float* p = g_out + o * n_out * n_out;
  float acc = 0.0f;
  for (int dy = y0; dy <= y1; ++dy) {
    const float wy = resize_weight(dy, iy, ratio, n_in);
    if (wy == 0.0f) continue;
    const float* row = p + (size_t)dy * n_out;
    float v[kZyTaps];
#pragma unroll
    for (int k = 0; k < kZyTaps; ++k) v[k] = row[oz[k]];   // independent loads, all in flight
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < kZyTaps; ++k)
      if (wz[k] != 0.0f) t = fmaf(wz[k], v[k], t);
    acc = fmaf(wy, t, acc);
  }
  g_in[idx] = acc;
}

// The x pass of the transposed resize followed by the transposed 1x1 layer that was swapped with it (Cin -> COUT
// channels, conv1x1_kernel's chain and "+ 0" bias), ReLU-masked and zero-padded for the transposed convolution
// below: three launches of the captured loop in one, same arithmetic.   grid: (ceil(np^3 / 256), N)
template <int COUT>
__global__ __launch_bounds__(256) void resize_x_backward_mix_pad_kernel(
    const float* __restrict__ g_out, int Cin, int n_in, int n_out, const float* __restrict__ wmat,
    const float* __restrict__ bias, const float* __restrict__ act, int pad, int pz, float* __restrict__ out) {
  const int np = n_in + 2 * pad;   // rows of `out` are pz >= np floats apart (floats np .. pz - 1: zeros)
  const size_t vp = (size_t)np * np * pz, vin = (size_t)n_in * n_in * n_in;
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= vp) return;
  const int n = blockIdx.y;
  const int z = (int)(r % pz) - pad, y = (int)((r / pz) % np) - pad, i = (int)(r / ((size_t)pz * np)) - pad;
  float res[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) res[co] = 0.0f;
  if (i >= 0 && i < n_in && y >= 0 && y < n_in && z >= 0 && z < n_in) {
    const size_t inner = (size_t)n_in * n_in, rr = (size_t)y * n_in + z;
    const float ratio = (float)n_in / (float)n_out;
    int d0, d1;
    resize_sources(i, n_in, n_out, d0, d1);
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
    for (int ci = 0; ci < Cin; ++ci) {
      const float* p = g_out + (((size_t)n * Cin + ci) * n_out) * inner + rr;
      float a = 0.0f;
      for (int d = d0; d <= d1; ++d) {
        const float w = resize_weight(d, i, ratio, n_in);
        if (w != 0.0f) a = fmaf(w, p[(size_t)d * inner], a);
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] = fmaf(a, wmat[ci * 16 + co], acc[co]);
    }
    const size_t src = (size_t)i * inner + rr;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float v = acc[co] + bias[co];
      res[co] = (!act || act[((size_t)n * COUT + co) * vin + src] > 0.0f) ? v : 0.0f;
    }
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) out[((size_t)n * COUT + co) * vp + r] = res[co];
}

// The x pass of the transposed resize (outer = N*C volumes, inner = n_in^2) writing straight into the
// zero-padded, ReLU-masked tensor the next transposed convolution reads (= the pass followed by
// pad_mask_kernel, one launch less per layer in the captured loop):
//   out[o][x + pad][y + pad][z + pad] = (act[o][x][y][z] > 0 or no act) ? sum_d w(d -> x) g_out[o][d][y][z] : 0
__global__ __launch_bounds__(256) void resize_x_backward_pad_kernel(const float* __restrict__ g_out, size_t outer,
                                                                    int n_in, int n_out,
                                                                    const float* __restrict__ act, int pad, int pz,
                                                                    float* __restrict__ out) {
  const int np = n_in + 2 * pad;   // rows of `out` are pz >= np floats apart (floats np .. pz - 1: zeros)
  const size_t vp = (size_t)np * np * pz;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= outer * vp) return;
  const size_t o = idx / vp;
  const int r = (int)(idx - o * vp);
  const int z = r % pz - pad, y = (r / pz) % np - pad, i = r / (pz * np) - pad;
  float acc = 0.0f;
  if (i >= 0 && i < n_in && y >= 0 && y < n_in && z >= 0 && z < n_in) {
    const size_t inner = (size_t)n_in * n_in, rr = (size_t)y * n_in + z;
    const size_t src = (o * n_in + i) * inner + rr;
    if (!act || act[src] > 0.0f) {
      const float ratio = (float)n_in / (float)n_out, inv = (float)n_out / (float)n_in;
      int d0 = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
      int d1 = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
      d0 = max(d0, 0);
      d1 = min(d1, n_out - 1);
      const float* p = g_out + (o * n_out) * inner + rr;
      for (int d = d0; d <= d1; ++d) {
        const float w = resize_weight(d, i, ratio, n_in);
        if (w != 0.0f) acc = fmaf(w, p[(size_t)d * inner], acc);
      }
    }
  }
  out[idx] = acc;
}

// The three passes of the transposed resize in ONE launch for BATCHES (round 3).  As three gather launches the
// transposed resizes were the largest item of a batched VJP (945 of ~2 200 us per 256 latents: every pass re-reads
// what the one before wrote, and forms the weight of every term with the full source-index arithmetic).  Here a
// workgroup owns a tc x tc tile of coarse (x, y) columns and walks over CHANNELS with it: the three axes' weight
// tables (first source index + up to kBtTaps weights per coarse index) and every address are formed once, then per
// channel the fine rows that feed the tile are staged in LDS (16-byte loads, the NEXT channel's rows already in
// flight into registers while this channel's passes run -- a workgroup is a chain of load -> barrier -> z -> barrier
// -> y -> barrier -> x, and with two or three workgroups per CU nothing else hides that chain), and z-, y- and x-pass
// run on the staged block: each term one LDS read and one fma, the same terms in the same order as the three
// launches (zero weights skipped), so the result is bit-identical to them.  The z pass writes its results over the
// head of each fine row IN PLACE (a row is read and written by one wave in one step).  The last pass applies the
// ReLU mask and writes straight into the zero-padded tensor the next transposed convolution reads, like
// resize_x_backward_pad_kernel; with COUT > 0 it also runs the transposed 1x1 layer that was swapped with the resize
// (one source channel -> COUT channels, conv1x1_kernel's chain and "+ 0" bias), like
// resize_x_backward_mix_pad_kernel (conv1x1 and pad_mask launches and their traffic saved).
//   g_out [nc][no][no][no] -> out [nc (* COUT)][np][np][np], np = ni + 2 pad (pad 0, act NULL: the plain transpose)
// Tiles are laid over the COARSE index range (tiles = ceil(ni / tc) per axis); the first and last tile of an axis
// also write that side's padding.  Grid: 1-D, tiles^2 * slots workgroups, slots % 8 == 0; workgroup L works for
// channel slot (L / 8 / tiles^2) * 8 + L % 8 on tile (L / 8) % tiles^2 -- workgroups are dealt to the 8 XCDs
// round-robin, so all tiles of a channel (whose fine blocks overlap by the taps' reach) run on ONE XCD, at the same
// time, and the overlap is served by that XCD's L2.  Slot s takes channels s, s + slots, ...
// LDS: max_f^2 * no + max_f * tc * ni floats (+ tables).
constexpr int kBtTaps = 12, kBtLoads = SDFR_BT_LOADS, kBtOut = SDFR_BT_OUT, kBtMaxTile = 16, kBtMaxThreads = 1024;
// exact range of fine indices d that carry weight on coarse index i (empty: d0 > d1)
__device__ __forceinline__ void resize_sources_exact(int i, float ratio, int n_in, int n_out, int& d0, int& d1) {
  resize_sources(i, n_in, n_out, d0, d1);
  while (d0 <= d1 && resize_weight(d0, i, ratio, n_in) == 0.0f) ++d0;
  while (d1 >= d0 && resize_weight(d1, i, ratio, n_in) == 0.0f) --d1;
}
// (index arithmetic: a lane is a z index in the passes, several rows per wave when the coarse row is short; divisions
// are multiplications by a reciprocal formed once, and everything that does not depend on the channel -- addresses
// of the loads, of the stores, of the mask -- is formed once per workgroup.  TAPS >= the longest source range of the
// resize: the tap loops are unrolled, ALL of a step's LDS reads are issued before the first fma (read one tap, wait,
// test its weight, read the next: the passes were chains of LDS latencies, 3/4 of the kernel's time); a tap beyond a
// row's range has weight 0 and re-reads the row's last source.  The mask values of a channel are loaded before its
// z pass and used after its y pass.)
// The exact source range [d0, d0 + nt) of coarse index i (i < 0: empty) and the weight of its tap k, by the SIXTEEN lanes
// k = 0 .. 15 of a 16-lane group (all of them active; group_shift = the group's first lane within the wave): every lane
// weighs one candidate of resize_sources' superset range (<= 16 long: host), a ballot finds the first and the last
// non-zero one.  resize_sources_exact + a loop over the taps are ~17 dependent evaluations of resize_weight on one
// lane -- 4 us of set-up per launch of the kernel below (stamped: 1.5 us for the tile's bounds, 2.5 us for the
// tables), a third of a single latent's launch; this is two evaluations deep.
__device__ __forceinline__ void resize_row16(int i, int k, int group_shift, float ratio, int n_in, int n_out, int& d0,
                                             int& nt, float& w_k) {
  int d0s = 0, d1s = -1;
  if (i >= 0) resize_sources(i, n_in, n_out, d0s, d1s);
  const int d = d0s + k;
  const float w = (i >= 0 && d <= d1s) ? resize_weight(d, i, ratio, n_in) : 0.0f;
  const unsigned m16 = (unsigned)(__ballot(w != 0.0f) >> group_shift) & 0xffffu;
  d0 = d0s;
  nt = 0;
  if (m16) {
    const int first = __ffs(m16) - 1, last = 31 - __clz(m16);
    d0 = d0s + first;
    nt = last - first + 1;
  }
  w_k = (k < nt) ? resize_weight(d0 + k, i, ratio, n_in) : 0.0f;
}
template <int COUT, int TAPS>
__global__ __launch_bounds__(kBtMaxThreads) void resize3_backward_tiled_kernel(
    const float* __restrict__ g_out, int n_in, int n_out, const float* __restrict__ act, int pad, int pz, int tc,
    int max_f, int nc, int slots, const float* __restrict__ wmat, const float* __restrict__ bias, float* __restrict__ out) {
  constexpr int CO = COUT > 0 ? COUT : 1;
  extern __shared__ float lds[];
#ifdef SDFR_BT_STAMPS   // timing experiment (tools/microbench/tail_stamps.py prints them): the stages of a workgroup, 10 ns ticks
  unsigned long long bts[8]; int btk = 0;
#define BTS() do { if (btk < 8) bts[btk++] = wall_clock64(); } while (0)
#else
#define BTS() do { } while (0)
#endif
  BTS();
  __shared__ float w_tab[2 * kBtMaxTile + 64][kBtTaps];   // rows: x (tc), y (tc), z (n_in <= 64)
  __shared__ int d_tab[2 * kBtMaxTile + 64];              // first source index, relative to the tile's first fine row / 0 for z
  __shared__ int f_rng[4];
  const int tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, nw = nthr >> 6, lane = tid & 63;
  const int np = n_in + 2 * pad, tiles = (n_in + tc - 1) / tc, tt = tiles * tiles;
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int tile = idx % tt, slot0 = (idx / tt) * 8 + xcd;
  if (slot0 >= nc) return;
  const int tile_x = tile / tiles, tile_y = tile - tile_x * tiles;
  const float ratio = (float)n_in / (float)n_out;
  // coarse columns of this tile, and the columns of the padded tensor it writes
  const int cx0 = tile_x * tc, cx1 = min(cx0 + tc, n_in) - 1, cy0 = tile_y * tc, cy1 = min(cy0 + tc, n_in) - 1;
  const int ncx = cx1 - cx0 + 1, ncy = cy1 - cy0 + 1;
  const int ox0 = tile_x == 0 ? 0 : cx0 + pad, ox1 = tile_x == tiles - 1 ? np : cx1 + 1 + pad;
  const int oy0 = tile_y == 0 ? 0 : cy0 + pad, oy1 = tile_y == tiles - 1 ? np : cy1 + 1 + pad;
  const int k16 = tid & 15, g16 = lane & 48;   // (16-lane groups: resize_row16)
  if (tid < 64) {   // the tile's fine rows: the first wave's four groups find one bound each
    const int q = tid >> 4;
    int d0, nt;
    float w;
    resize_row16(q == 0 ? cx0 : (q == 1 ? cx1 : (q == 2 ? cy0 : cy1)), k16, g16, ratio, n_in, n_out, d0, nt, w);
    if (k16 == 0) f_rng[q] = (q & 1) ? d0 + nt - 1 : d0;
  }
  __syncthreads();
  BTS();
  const int fx0 = f_rng[0], fx1 = f_rng[1], fy0 = f_rng[2], fy1 = f_rng[3];
  const int fnx = fx1 - fx0 + 1, fny = fy1 - fy0 + 1;
  float* F = lds;                                      // [fnx][fny][n_out], heads overwritten by the z pass
  float* Y = lds + (size_t)max_f * max_f * n_out;      // [fnx][ncy][n_in]
  // a thread's vectors of the fine block (n_out % 4 == 0, <= kBtLoads * nthr vectors: host)
  const int q4 = n_out >> 2, total4 = fnx * fny * q4;
  int off[kBtLoads];
  {
    const unsigned m_q = magic_of(q4), m_y = magic_of(fny);
#pragma unroll
    for (int j = 0; j < kBtLoads; ++j) {
      const int e = min(tid + nthr * j, total4 - 1);
      const int row = div_by(e, m_q), q = e - row * q4;
      const int fx = div_by(row, m_y), fy = row - fx * fny;
      off[j] = ((fx0 + fx) * n_out + (fy0 + fy)) * n_out + 4 * q;
    }
  }
  const size_t fine_vol = (size_t)n_out * n_out * n_out, coarse_vol = (size_t)n_in * n_in * n_in, pad_vol = (size_t)np * np * pz;
  f32x4 pre[kBtLoads];
  {
    const float* src = g_out + (size_t)slot0 * fine_vol;
#pragma unroll
    for (int j = 0; j < kBtLoads; ++j)
      if (nthr * j < total4) pre[j] = *reinterpret_cast<const f32x4*>(src + off[j]);
  }
  // weight tables, a row per 16-lane group (while the block's first vectors are on their way)
  for (int row = tid >> 4; row < 2 * kBtMaxTile + n_in; row += nthr >> 4) {
    int i, base;
    if (row < kBtMaxTile) { i = row < ncx ? cx0 + row : -1; base = fx0; }
    else if (row < 2 * kBtMaxTile) { i = row - kBtMaxTile < ncy ? cy0 + row - kBtMaxTile : -1; base = fy0; }
    else { i = row - 2 * kBtMaxTile; base = 0; }
    int d0, nt;
    float w;
    resize_row16(i, k16, g16, ratio, n_in, n_out, d0, nt, w);
    if (k16 == 0) d_tab[row] = d0 - base;
    if (k16 < kBtTaps) w_tab[row][k16] = w;
  }
  // a thread's elements of the tile's part of the padded tensor (<= kBtOut * nthr: host): where each goes, its mask
  // value, its Y column and its x weights (-1: padding, a zero)
  const int ocx = ox1 - ox0, ocy = oy1 - oy0, n_store = ocx * ocy * pz;   // (rows pz >= np floats apart, the rest zeros)
  int st_r[kBtOut], st_a[kBtOut], st_y[kBtOut];
  {
    const unsigned m_np = magic_of(pz), m_ocy = magic_of(ocy);
#pragma unroll
    for (int j = 0; j < kBtOut; ++j) {
      const int e = min(tid + nthr * j, n_store - 1);
      const int col = div_by(e, m_np), zp = e - col * pz;
      const int jx = div_by(col, m_ocy), jy = col - jx * ocy;
      const int xp = ox0 + jx, yp = oy0 + jy;
      const int ix = xp - pad, iy = yp - pad, izz = zp - pad;
      const bool inside = ix >= cx0 && ix <= cx1 && iy >= cy0 && iy <= cy1 && izz >= 0 && izz < n_in;
      st_r[j] = (xp * np + yp) * pz + zp;
      st_a[j] = inside ? (ix * n_in + iy) * n_in + izz : -1;
      st_y[j] = inside ? (((iy - cy0) * n_in + izz) << 4) | (ix - cx0) : 0;   // (Y column, x-table row < 16)
    }
  }
  __syncthreads();   // tables
  BTS();
  // lane -> (slot, iz): 64 / nl rows per wave-step, nl = the power of two that holds a coarse row
  const int nl = n_in <= 8 ? 8 : (n_in <= 16 ? 16 : (n_in <= 32 ? 32 : 64));
  const int spw = 64 / nl, slot = lane / nl, iz = lane - slot * nl;
  const bool zok = iz < n_in;
  const int dz = zok ? d_tab[2 * kBtMaxTile + iz] : 0;
  float wz[TAPS];
  int oz[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; ++k) {
    wz[k] = zok ? w_tab[2 * kBtMaxTile + iz][k] : 0.0f;
    oz[k] = min(dz + k, n_out - 1);
  }
  const unsigned m_ncy = magic_of(ncy);
  const int y_stride = ncy * n_in;
  for (int o = slot0; o < nc; o += slots) {
#pragma unroll
    for (int j = 0; j < kBtLoads; ++j)
      if (tid + nthr * j < total4) reinterpret_cast<f32x4*>(F)[tid + nthr * j] = pre[j];
    __syncthreads();
    BTS();
    // this channel's mask values, then the next channel's block: both in flight during this channel's passes
    float mask[kBtOut][CO];
    if (act) {
#pragma unroll
      for (int j = 0; j < kBtOut; ++j)
#pragma unroll
        for (int co = 0; co < CO; ++co)
          mask[j][co] = (nthr * j < n_store && st_a[j] >= 0) ? act[((size_t)o * CO + co) * coarse_vol + st_a[j]] : 0.0f;
    }
    if (o + slots < nc) {
      const float* src = g_out + (size_t)(o + slots) * fine_vol;
#pragma unroll
      for (int j = 0; j < kBtLoads; ++j)
        if (nthr * j < total4) pre[j] = *reinterpret_cast<const f32x4*>(src + off[j]);
    }
    {   // z pass, in place
      const int rows = fnx * fny;
      for (int row = wave * spw + slot; row < rows; row += nw * spw) {
        const float* f = F + row * n_out;
        float v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) v[k] = f[oz[k]];
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (wz[k] != 0.0f) ? fmaf(wz[k], v[k], acc) : acc;
        if (zok) F[row * n_out + iz] = acc;
      }
    }
    __syncthreads();
    BTS();
    {   // y pass: pairs (fx, jy), jy fastest
      const int pairs = fnx * ncy;
      for (int pr = wave * spw + slot; pr < pairs; pr += nw * spw) {
        const int fx = div_by(pr, m_ncy), jy = pr - fx * ncy;
        const int d0 = d_tab[kBtMaxTile + jy];
        const float* zc = F + (size_t)fx * fny * n_out + (zok ? iz : 0);
        float w[TAPS], v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
          w[k] = w_tab[kBtMaxTile + jy][k];
          v[k] = zc[min(d0 + k, fny - 1) * n_out];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (w[k] != 0.0f) ? fmaf(w[k], v[k], acc) : acc;
        if (zok) Y[pr * n_in + iz] = acc;
      }
    }
    __syncthreads();
    BTS();
    // x pass, (mix,) mask, store: the tile's columns of the padded tensor, zeros in the padding
#pragma unroll
    for (int j = 0; j < kBtOut; ++j) {
      if (tid + nthr * j >= n_store) break;
      float acc = 0.0f;
      const bool inside = st_a[j] >= 0;
      if (inside) {
        const int xr = st_y[j] & 15, d0 = d_tab[xr];
        const float* yc = Y + (st_y[j] >> 4);
        float w[TAPS], v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
          w[k] = w_tab[xr][k];
          v[k] = yc[min(d0 + k, fnx - 1) * y_stride];
        }
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (w[k] != 0.0f) ? fmaf(w[k], v[k], acc) : acc;
      }
      if (COUT == 0) {
        if (act && !(mask[j][0] > 0.0f)) acc = 0.0f;   // (padding: acc is 0 already)
        out[(size_t)o * pad_vol + st_r[j]] = acc;
      } else {
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          const float v = fmaf(acc, wmat[co], 0.0f) + bias[co];
          out[((size_t)o * CO + co) * pad_vol + st_r[j]] = (inside && (!act || mask[j][co] > 0.0f)) ? v : 0.0f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // (one element's taps at a time: six elements' worth of reads in flight cost 40 VGPRs)
    }
    BTS();
#ifdef SDFR_BT_STAMPS
    if (tid == 0 && blockIdx.x == 0 && o == slot0)
      printf("rbt n_out %d threads %d: bounds %.2f tables %.2f stage %.2f z %.2f y %.2f x+store %.2f us\n", n_out, nthr,
             (double)(bts[1] - bts[0]) * 0.01, (double)(bts[2] - bts[1]) * 0.01, (double)(bts[3] - bts[2]) * 0.01,
             (double)(bts[4] - bts[3]) * 0.01, (double)(bts[5] - bts[4]) * 0.01, (double)(bts[6] - bts[5]) * 0.01);
#endif
  }
}

// g += k g2,  k = weight / cnt[0] (0 for a zero count): the scaled form's incoming gradients as one volume, for the
// decoders whose first VJP launch is not the fused stage (which adds them on load)
// (nv volumes g2 [nv][n] with their counts: views 0, 1, ... in that order)
__global__ __launch_bounds__(256) void scaled_combine_kernel(float* __restrict__ g, const float* __restrict__ g2, int nv,
                                                             const float* __restrict__ cnt, float weight, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = g[i];
  for (int v = 0; v < nv; ++v) {
    const float c = cnt[v];
    const float k = c > 0.0f ? weight / c : 0.0f;
    acc = fmaf(k, g2[(size_t)v * n + i], acc);
  }
  g[i] = acc;
}

// Backward of the last (wide) Linear layer: t[n][i] = sum_o Wt[i][o] * g_last[n][o], one workgroup
// per (i, sample) -- a 50 x 8192 GEMV spread over 50 workgroups instead of one.
// `act`: the layer's forward output; its ReLU' is applied to g_last on the fly (no mask launch).
__global__ __launch_bounds__(kFcBlock) void fc_last_backward_kernel(const float* __restrict__ params,
                                                                    FcDesc d,
                                                                    const float* __restrict__ g_last,
                                                                    const float* __restrict__ act,
                                                                    float* __restrict__ t_out,
                                                                    float* __restrict__ clear_a,
                                                                    float* __restrict__ clear_b, int clear_vec,
                                                                    int clear_b_volumes) {
  __shared__ float red[kFcBlock / 64];
  const int i = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  // sdfr_decoder_backward_latent_deferred_scaled: the VJP's incoming gradient volumes were consumed by its first
  // launch -- this one, its last with a grid, zero-fills them for their producer (clear_vec 16-byte vectors each)
  if (clear_a) {
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int nthreads = (int)(gridDim.x * gridDim.y) * kFcBlock, me = ((int)blockIdx.y * (int)gridDim.x + i) * kFcBlock + tid;
    for (int e = me; e < clear_vec; e += nthreads) reinterpret_cast<f32x4*>(clear_a)[e] = zero4;
    for (int e = me; e < clear_vec * clear_b_volumes; e += nthreads) reinterpret_cast<f32x4*>(clear_b)[e] = zero4;
  }
  const int l = d.n_fc - 1, win = d.width[l], wout = d.width[l + 1];
  const float* wt = params + d.w_off[l] + (size_t)i * wout;
  const float* g = g_last + (size_t)n * wout;
  const float* a = act + (size_t)n * wout;
  float part = 0.0f;
  if ((wout & 3) == 0 && (((uintptr_t)wt | (uintptr_t)g | (uintptr_t)a) & 15) == 0) {
    // 16-byte loads, all of a thread's loads in flight at once (8192 outputs: 8 per array): 15.9 -> 4.6 us in the loop
    const float4 *w4 = reinterpret_cast<const float4*>(wt), *g4 = reinterpret_cast<const float4*>(g),
                 *a4 = reinterpret_cast<const float4*>(a);
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll 8
    for (int o = tid; o < wout / 4; o += kFcBlock) {
      const float4 w = w4[o], gv = g4[o], av = a4[o];
      p0 = fmaf(w.x, (av.x > 0.0f) ? gv.x : 0.0f, p0);
      p1 = fmaf(w.y, (av.y > 0.0f) ? gv.y : 0.0f, p1);
      p2 = fmaf(w.z, (av.z > 0.0f) ? gv.z : 0.0f, p2);
      p3 = fmaf(w.w, (av.w > 0.0f) ? gv.w : 0.0f, p3);
    }
    part = (p0 + p1) + (p2 + p3);
  } else {
    for (int o = tid; o < wout; o += kFcBlock) part = fmaf(wt[o], (a[o] > 0.0f) ? g[o] : 0.0f, part);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) {
    float t = 0.0f;
    for (int k = 0; k < kFcBlock / 64; ++k) t += red[k];
    t_out[(size_t)n * win + i] = t;
  }
}

// The same for batches: a workgroup takes IB input units x S samples, so that a sample's g / act row is read once per
// IB units and a weight row once per S samples (20.8 instead of 96 KB per (unit, sample) pair at 5 x 4).  Every pair
// keeps the per-sample form's thread -> column map, four partial sums and reduction tree: bit-identical to it.
// 16-byte form only (wout % 4 == 0, aligned: host).   grid (ceil(win / IB), ceil(N / S))
template <int IB, int S>
__global__ __launch_bounds__(kFcBlock) void fc_last_backward_batch_kernel(const float* __restrict__ params, FcDesc d,
                                                                          const float* __restrict__ g_last,
                                                                          const float* __restrict__ act, int N,
                                                                          float* __restrict__ t_out) {
  __shared__ float red[IB * S][kFcBlock / 64];
  const int i0 = blockIdx.x * IB, n0 = blockIdx.y * S, tid = threadIdx.x;
  const int l = d.n_fc - 1, win = d.width[l], wout = d.width[l + 1];
  const float4* w4[IB];
  const float4 *g4[S], *a4[S];
#pragma unroll
  for (int k = 0; k < IB; ++k) w4[k] = reinterpret_cast<const float4*>(params + d.w_off[l] + (size_t)min(i0 + k, win - 1) * wout);
#pragma unroll
  for (int sm = 0; sm < S; ++sm) {
    g4[sm] = reinterpret_cast<const float4*>(g_last + (size_t)min(n0 + sm, N - 1) * wout);
    a4[sm] = reinterpret_cast<const float4*>(act + (size_t)min(n0 + sm, N - 1) * wout);
  }
  float p[IB][S][4];
#pragma unroll
  for (int k = 0; k < IB; ++k)
#pragma unroll
    for (int sm = 0; sm < S; ++sm)
#pragma unroll
      for (int c = 0; c < 4; ++c) p[k][sm][c] = 0.0f;
#pragma unroll 2
  for (int o = tid; o < wout / 4; o += kFcBlock) {
    float4 m[S];
#pragma unroll
    for (int sm = 0; sm < S; ++sm) {
      const float4 gv = g4[sm][o], av = a4[sm][o];
      m[sm] = float4{(av.x > 0.0f) ? gv.x : 0.0f, (av.y > 0.0f) ? gv.y : 0.0f, (av.z > 0.0f) ? gv.z : 0.0f,
                     (av.w > 0.0f) ? gv.w : 0.0f};
    }
#pragma unroll
    for (int k = 0; k < IB; ++k) {
      const float4 w = w4[k][o];
#pragma unroll
      for (int sm = 0; sm < S; ++sm) {
        p[k][sm][0] = fmaf(w.x, m[sm].x, p[k][sm][0]);
        p[k][sm][1] = fmaf(w.y, m[sm].y, p[k][sm][1]);
        p[k][sm][2] = fmaf(w.z, m[sm].z, p[k][sm][2]);
        p[k][sm][3] = fmaf(w.w, m[sm].w, p[k][sm][3]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < IB; ++k)
#pragma unroll
    for (int sm = 0; sm < S; ++sm) {
      float part = (p[k][sm][0] + p[k][sm][1]) + (p[k][sm][2] + p[k][sm][3]);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
      if ((tid & 63) == 0) red[k * S + sm][tid >> 6] = part;
    }
  __syncthreads();
  if (tid < IB * S) {
    const int k = tid / S, sm = tid - k * S;
    if (i0 + k < win && n0 + sm < N) {
      float t = 0.0f;
      for (int q = 0; q < kFcBlock / 64; ++q) t += red[tid][q];
      t_out[(size_t)(n0 + sm) * win + i0 + k] = t;
    }
  }
}

// Backward of the small leading layers, one workgroup per sample (decoder_fc.hpp: fc_stack_backward_sample).
__global__ __launch_bounds__(kFcBlock) void fc_stack_backward_kernel(const float* __restrict__ params,
                                                                     FcDesc d,
                                                                     const float* __restrict__ z,
                                                                     const float* __restrict__ t_in,
                                                                     float* __restrict__ g_z) {
  const int n = blockIdx.x, l = d.n_fc - 1;
  fc_stack_backward_sample(params, d, z + (size_t)n * d.width[0], t_in + (size_t)n * d.width[l],
                           g_z + (size_t)n * d.width[0]);
}
// ... one wave per sample when the leading layers are narrow (fc_one_wave_ok): the same numbers
__global__ __launch_bounds__(64) void fc_stack_backward_wave_kernel(const float* __restrict__ params, FcDesc d,
                                                                    const float* __restrict__ z,
                                                                    const float* __restrict__ t_in,
                                                                    float* __restrict__ g_z) {
  __shared__ FcWaveLds fc_lds;
  const int n = blockIdx.x, l = d.n_fc - 1;
  fc_stack_backward_one_wave(fc_lds, params, d, z + (size_t)n * d.width[0], t_in + (size_t)n * d.width[l],
                             g_z + (size_t)n * d.width[0], threadIdx.x);
}

#include "decoder_fused.hpp"   // the single decode's layer pairs as one launch each (round 6)

}  // namespace
}  // namespace sdfr

using namespace sdfr;

namespace {
// conv3d_mfma_kernel's split-K form: for launches with so few tiles that a wave's K loop is the critical path
#ifndef SDFR_SPLITK_MAX_TILES
#define SDFR_SPLITK_MAX_TILES 2048
#endif
int use_split_k(bool zgrp, int n_tiles, int co_tiles, int N, int kpad) {
  // (the choice does not depend on N up to 16 samples: small batches decode bit-identically to single latents)
  return (!zgrp && (long long)n_tiles * co_tiles <= SDFR_SPLITK_MAX_TILES && N <= SDFR_SPLITK_MAX_LATENTS && kpad >= 128) ? 1 : 0;
}

// The direct convolution is for batches (enough tiles to fill the chip); returns false when the
// layer / batch does not qualify and the caller falls back to the MFMA kernel.
// conv3d_mfma_kernel in the form the layer / batch takes (plain, z-grouped, split-K, input resident in LDS)
void launch_mfma(const float* src, const float* w, const int* tab, const float* bias, float* dst, int cin, int cout,
                 int n, int m, int kp, int relu, int nt, int co_tiles, int zg, int split, int N, hipStream_t st) {
  const int tw = nt >= 32768 ? 4 : 1;
  const size_t in_floats = (size_t)cin * n * n * n, lds_res = ((size_t)kp * 17 + in_floats) * sizeof(float);
  // (resident: the whole grid is then N * co_tiles workgroups -- only where that still fills the chip)
  const bool resident = !split && zg == 1 && N * co_tiles >= 128 && nt >= 8 && nt <= 64 && lds_res <= 120 * 1024 &&
                        (in_floats & 3) == 0 && ((uintptr_t)src & 15) == 0;
  if (resident) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_mfma_kernel<true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (attr == hipSuccess) {
      hipLaunchKernelGGL(conv3d_mfma_kernel<true>, dim3(1, co_tiles, N), dim3(256), lds_res, st, src, w, tab, bias, dst,
                         cin, cout, n, m, kp, relu, 1, 1, 0);
      return;
    }
  }
  hipLaunchKernelGGL(conv3d_mfma_kernel<false>, dim3(split ? nt : (nt + 4 * tw - 1) / (4 * tw), co_tiles, N), dim3(256),
                     (size_t)kp * 17 * sizeof(float) + (split ? 4096 : 0), st, src, w, tab, bias, dst, cin, cout, n, m,
                     kp, relu, tw, zg, split);
}

// Does a layer / batch qualify for it?  (n: input size, m: output size)
bool direct_ok(size_t w_off, int n, int m, int N, int* tx = nullptr, int* ty = nullptr, int* zc = nullptr) {
  if (w_off == 0 || n > 64 || m < 4) return false;
  const int ZC = (m + 3) / 4;                       // z chunks of 4 outputs per column
  int zc_pow = 1;
  while (zc_pow < ZC) zc_pow <<= 1;                 // threads per column (power of two <= 16)
  if (zc_pow > 16) return false;
  const int cols = 256 / zc_pow;                    // columns per workgroup, at most
  // TX x TY <= cols columns per workgroup: the fewest tiles, then the smallest staged patch (30^3 outputs: 5 x 6
  // columns cover them exactly in 30 tiles of 7 x 8 patch columns; the power-of-two 4 x 8 takes 32 tiles of 6 x 10)
  int TX = 1, TY = cols;
  long long best = -1;
  for (int a = 1; a <= std::min(cols, m); ++a) {
    const int b = std::min(cols / a, m);
    const long long tiles = (long long)((m + a - 1) / a) * ((m + b - 1) / b);
    const long long cost = tiles * 4096 + (a + 2) * (b + 2);
    if (best < 0 || cost < best) { best = cost; TX = a; TY = b; }
  }
  const int tiles = ((m + TX - 1) / TX) * ((m + TY - 1) / TY);
  // enough workgroups to fill the chip -- or, for the handful of latents of a multi-object frame (8 ... 31: K objects
  // side by side), a layer of at least 15 tiles from 240 workgroups on: decoder N = 16 forward 162 -> 97 us, the 8- and
  // 16-object loops +8 % / +24 % objects per second; single latents, small layers and big batches keep their forms
  // (measured with the plain 240 threshold: N = 1 VJP 123 -> 160 us, N = 256 403 -> 453 us)
  const long long wgs = (long long)tiles * N;
  if (wgs < SDFR_DIRECT_MIN_TILES && !(N >= SDFR_DIRECT_MIN_LATENTS_FEW && tiles >= 15 && wgs >= SDFR_DIRECT_MIN_TILES_FEW)) return false;
  if (tx) *tx = TX;
  if (ty) *ty = TY;
  if (zc) *zc = zc_pow;
  return true;
}
// pz: floats between the z-rows of src (n, or more: the padded tensors of the backward pass)
// mix_*: the 1x1x1 layer applied in the epilogue (conv3d_direct_kernel), or mix_out == NULL
bool launch_direct(const sdfr_decoder* d, size_t w_off, const float* src, const float* bias, float* dst,
                   int cin, int cout, int n, int pz, int m, int relu, int N, hipStream_t st,
                   const float* mix_w = nullptr, const float* mix_b = nullptr, int mix_co = 0, float* mix_out = nullptr) {
  int TX, TY, ZC;
  if (!direct_ok(w_off, n, m, N, &TX, &TY, &ZC)) return false;
  const int tiles = ((m + TX - 1) / TX) * ((m + TY - 1) / TY);
  // channels per LDS chunk: patch of CK channels <= 24 KB
  const int per_ch = (TX + 2) * (TY + 2) * ((pz + 3) & ~3);   // (LDS rows are padded to 16 bytes)
  int CK = std::max(1, std::min(cin, (24 * 1024 / 4) / per_ch));
  const size_t lds = ((size_t)CK * per_ch + 8) * sizeof(float);  // + the over-read of the last z-run
  const dim3 grid(tiles, 1, N);
  const float* w = d->d_params + w_off;
  // 16-byte loads: runs start and end on 16-byte boundaries, offsets and channel fit the packed word of the prefetch
  const bool vec4 = (pz & 3) == 0 && ((uintptr_t)src & 15) == 0 && (size_t)CK * n * n * pz < 0xffffff && CK < 127;
  if (!vec4 && pz != n) return false;   // (the scalar-load form takes dense rows only; callers pad rows only to 16 bytes)
#define SDFR_DIRECT(CO)                                                                                          \
  if (vec4) hipLaunchKernelGGL((conv3d_direct_kernel<CO, true>), grid, dim3(256), lds, st, src, w, bias, dst,   \
                               cin, n, pz, m, relu, TX, TY, ZC, CK, mix_w, mix_b, mix_co, mix_out);             \
  else hipLaunchKernelGGL((conv3d_direct_kernel<CO, false>), grid, dim3(256), lds, st, src, w, bias, dst, cin,  \
                          n, pz, m, relu, TX, TY, ZC, CK, mix_w, mix_b, mix_co, mix_out)
  if (cout == 4) { SDFR_DIRECT(4); } else if (cout == 8) { SDFR_DIRECT(8); } else { SDFR_DIRECT(16); }
#undef SDFR_DIRECT
  return true;
}

// resize ni -> n (up-sampling) folded into the direct convolution that follows it (conv3d_direct_up_kernel): true if
// launched.  Same tiling as launch_direct; the chunk is also bounded by the coarse columns a thread prefetches.
bool launch_direct_up(const sdfr_decoder* d, size_t w_off, const float* src, int ni, const float* bias, float* dst,
                      int cin, int cout, int n, int m, int relu, int N, hipStream_t st) {
  const int mode = d->opt_fused_resize.load(std::memory_order_relaxed);
  if (!mode) return false;
  if (n < 16 || n > 64 || (n & (n - 1)) != 0 || ni > n || ni < 2) return false;
  // Measured on 256 mug latents (profiles/r04_decoder_fused_resize.txt): 6 -> 16 in front of the 16 -> 8 layer,
  // 112 us folded against 96 + 26 as two launches; 14 -> 32 in front of the 8 -> 4 layer, 268 us folded against
  // 170 + 77 -- there the patch's 1.87x overlap makes every workgroup interpolate what its neighbours interpolate
  // too, in a kernel that is VALU-bound already.  So: folded up to a fine size of 16 (mode 2, the tests': always).
  if (mode == 1 && n > 16) return false;
  int TX, TY, ZC;
  if (!direct_ok(w_off, n, m, N, &TX, &TY, &ZC)) return false;
  const int IX = TX + 2, IY = TY + 2;
  if (IX > 34 || IY > 34) return false;
  // coarse columns under a patch, worst tile (the kernel's float arithmetic)
  const float ratio = (float)ni / (float)n;
  auto first_last = [&](int dd, int& i0, int& i1) {
    float sp = fmaf(ratio, (float)dd + 0.5f, -0.5f);
    sp = sp < 0.0f ? 0.0f : sp;
    i0 = std::min((int)sp, ni - 1);
    i1 = i0 + (i0 < ni - 1 ? 1 : 0);
  };
  auto span = [&](int T, int I) {
    int worst = 0;
    for (int t0 = 0; t0 < m; t0 += T) {
      int lo, hi, tmp;
      first_last(t0, lo, tmp);
      first_last(std::min(t0 + I, n) - 1, tmp, hi);
      worst = std::max(worst, hi - lo + 1);
    }
    return worst;
  };
  const int max_cols = span(TX, IX) * span(TY, IY);
  const int per_ch = IX * IY * n;
  int CK = std::max(1, std::min(cin, (24 * 1024 / 4) / per_ch));
  CK = std::min(CK, 2048 / (max_cols * ni));      // <= 8 prefetched values per thread
  if (CK < 1 || CK >= 127 || (size_t)CK * ni * ni * ni >= 0xffffff) return false;
  const size_t lds = ((size_t)CK * per_ch + 8 + (size_t)CK * max_cols * ni) * sizeof(float);
  if (lds > 60 * 1024) return false;
  int log_n = 0;
  while ((1 << log_n) < n) ++log_n;
  const int tiles = ((m + TX - 1) / TX) * ((m + TY - 1) / TY);
  const dim3 grid(tiles, 1, N);
  const float* w = d->d_params + w_off;
#define SDFR_DIRECT_UP(CO)                                                                                           \
  hipLaunchKernelGGL((conv3d_direct_up_kernel<CO>), grid, dim3(256), lds, st, src, ni, w, bias, dst, cin, n, log_n, \
                     m, relu, TX, TY, ZC, CK, max_cols)
  if (cout == 4) { SDFR_DIRECT_UP(4); } else if (cout == 8) { SDFR_DIRECT_UP(8); } else if (cout == 16) { SDFR_DIRECT_UP(16); }
  else return false;
#undef SDFR_DIRECT_UP
  return true;
}

// ---- few latents: layer pairs as one launch each (decoder_fused.hpp) --------------------------------------------------
constexpr size_t kFusedLdsMax = 150 * 1024;

// the kernel's dynamic-LDS limit raised once (above 64 KiB it must be; `bytes` + the kernel's static LDS <= 160 KiB);
// false: the runtime refused, take the unfused form
bool fused_lds_limit(const void* fn, size_t bytes = kFusedLdsMax) {
  static std::map<const void*, bool> done;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find(fn);
  if (it == done.end()) {
    const bool ok = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
    if (!ok) (void)hipGetLastError();   // (not an error of the call that asked: it takes the other form)
    it = done.emplace(fn, ok).first;
  }
  return it->second;
}

// resize_axis on the host (the kernels' float arithmetic)
void host_resize_axis(int dd, float ratio, int ni, int& i0, int& i1, float& l1) {
  float sp = fmaf(ratio, (float)dd + 0.5f, -0.5f);
  sp = sp < 0.0f ? 0.0f : sp;
  i0 = std::min((int)sp, ni - 1);
  i1 = i0 + (i0 < ni - 1 ? 1 : 0);
  l1 = sp - (float)i0;
}

// resize ni -> n + the 3x3x3 convolution n -> m behind it in one launch (conv3d_mfma_up_kernel): true if launched.
// Only where the unfused convolution takes its split-K form (the arithmetic this kernel reproduces).
bool launch_mfma_up(const sdfr_decoder* d, size_t direct_off, const float* src, int ni, const float* w, const float* bias,
                    float* dst, int cin, int cout, int n, int m, int kpad, int relu, int N, hipStream_t st,
                    const float* mix_w = nullptr, const float* mix_b = nullptr, int mix_co = 0, float* mix_out = nullptr) {
  if (!(d->opt_fused_single.load(std::memory_order_relaxed) & 1)) return false;
  if (ni > n || ni < 1 || m != n - 2 || m < 1 || direct_ok(direct_off, n, m, N)) return false;
  const int co_tiles = (cout + 15) / 16, ZT = (m + 15) / 16;
  if (ZT > 4 || !use_split_k(false, (m * m * m + 15) / 16, co_tiles, N, kpad)) return false;
  // columns per workgroup: the smallest tile that gives every workgroup a CU of its own (<= 16 MFMA tiles each)
  static const int kShapes[][2] = {{1, 1}, {1, 2}, {2, 2}, {2, 3}, {3, 3}, {2, 4}, {3, 4}, {4, 4}};
  int TX = 1, TY = 1;
  for (const auto& sh : kShapes) {
    if (sh[0] * sh[1] * ZT > 16) break;
    TX = sh[0]; TY = sh[1];
    if ((long long)((m + TX - 1) / TX) * ((m + TY - 1) / TY) * co_tiles * N <= 256) break;
  }
  const int tiles = TX * TY * ZT, wpt = 4 * tiles <= 16 ? 4 : 1;
  const int threads = std::max(256, 64 * wpt * tiles);
  const int IX = TX + 2, IY = TY + 2;
  const float ratio = (float)ni / (float)n;
  int CX = 1;
  for (int pass = 0; pass < 2; ++pass) {
    const int T = pass ? TY : TX, I = pass ? IY : IX;
    for (int t0 = 0; t0 < m; t0 += T) {
      int lo, hi, t;
      float f;
      host_resize_axis(t0, ratio, ni, lo, t, f);
      host_resize_axis(std::min(t0 + I, n) - 1, ratio, ni, t, hi, f);
      CX = std::max(CX, hi - lo + 1);
    }
  }
  const int PZ = 16 * ZT + 2, patch_n = cin * IX * IY * PZ;
  const size_t zc_n = (size_t)cin * CX * CX * PZ;
  if (zc_n >= 65536 || patch_n >= 65536 || threads / PZ < 1 || threads / (IX * IY) < 1) return false;   // (reciprocal divisions)
  const size_t lds = ((size_t)kpad * 17 + (wpt == 4 ? 4 * threads : 0) + ((patch_n + 3) & ~3) + zc_n) * sizeof(float);
  const void* fn = wpt == 4 ? reinterpret_cast<const void*>(&conv3d_mfma_up_kernel<4>) : reinterpret_cast<const void*>(&conv3d_mfma_up_kernel<1>);
  if (lds > kFusedLdsMax || (lds > 64 * 1024 && !fused_lds_limit(fn))) return false;
  const dim3 grid(((m + TX - 1) / TX) * ((m + TY - 1) / TY), co_tiles, N);
  if (mix_out && co_tiles != 1) return false;
  if (wpt == 4) hipLaunchKernelGGL(conv3d_mfma_up_kernel<4>, grid, dim3(threads), lds, st, src, ni, w, bias, dst, cin, cout, n,
                                   m, kpad, relu, CX, ZT, TX, TY, mix_w, mix_b, mix_co, mix_out);
  else hipLaunchKernelGGL(conv3d_mfma_up_kernel<1>, grid, dim3(threads), lds, st, src, ni, w, bias, dst, cin, cout, n, m,
                          kpad, relu, CX, ZT, TX, TY, mix_w, mix_b, mix_co, mix_out);
  return true;
}

// the Linear stack + the first convolution in one launch (fc_conv_kernel): true if launched
bool launch_fc_conv(const sdfr_decoder* d, const FcDesc& fd, const float* z, float* fc_out, size_t direct_off,
                    const float* w, const float* bias, float* dst, int cin, int cout, int n, int kpad, int relu, int N,
                    hipStream_t st) {
  if (!(d->opt_fused_single.load(std::memory_order_relaxed) & 2)) return false;
  const int m = n - 2;
  if (m < 1 || !decoder_fc_one_wave(d, fd) || direct_ok(direct_off, n, m, N)) return false;
  const int co_tiles = (cout + 15) / 16, ZT = (m + 15) / 16;
  if (ZT > 4 || !use_split_k(false, (m * m * m + 15) / 16, co_tiles, N, kpad)) return false;
  const int PZ = 16 * ZT + 2, patch_n = cin * 9 * PZ, items = cin * 9 * n;
  if (patch_n >= 65536 || fd.width[fd.n_fc - 1] > kFcWaveWidth) return false;
  // a row of the wide layer per thread where the workgroup size allows
  int threads = 256 * ZT;
  while (threads < 1024 && threads < items) threads *= 2;
  const size_t lds = ((size_t)kpad * 17 + 4 * threads + patch_n) * sizeof(float);
  const void* fn = reinterpret_cast<const void*>(&fc_conv_kernel);
  if (lds > 100 * 1024 || (lds > 38 * 1024 && !fused_lds_limit(fn, 100 * 1024))) return false;   // (+ 25 KB static)
  hipLaunchKernelGGL(fc_conv_kernel, dim3(m * m, co_tiles, N), dim3(threads), lds, st, d->d_params, fd, z, fc_out, w, bias,
                     dst, cin, cout, n, m, kpad, relu, ZT);
  return true;
}

// One stage of the VJP for few latents in one launch (vjp_stage_kernel): the transposed resize n_out -> n_in of C channels
// (+ ReLU mask, + the swapped 1x1x1 layer C = 1 -> mix_cout channels, + zero padding) and the transposed convolution of
// layer lc that reads it.  The plan, or ok == false where the pair keeps its two launches.  zin: the stage runs the z pass
// itself (g is the fine gradient); 0: its producer's epilogue has (g is coarse along z already).
struct VjpPlan {
  bool ok = false;
  int taps = 0, mode = 0, CK = 0, FX = 0, ZT = 0, TX = 1, TY = 1, threads = 0, zin = 1;
  size_t lds = 0;
};
VjpPlan vjp_stage_plan(const sdfr_decoder* d, int lc, int C, int n_in, int n_out, int mix_cout, int N, int zin,
                       const float* g, bool first) {
  VjpPlan p;
  p.zin = zin;
  // bit 4: the FIRST stage of the VJP (the one that reads the decoder's incoming gradient); bit 8: the stages behind a
  // fused one as well, chained through the z-pass epilogue (measured: no faster than their two launches each)
  const int opt = d->opt_fused_single.load(std::memory_order_relaxed);
  if (!(opt & (first ? 4 : 8))) return p;
  const int k = d->conv_k[lc], ci_n = d->conv_cin[lc], co_n = d->conv_cout[lc];
  const int CP = mix_cout > 0 ? mix_cout : C;
  if (k != 3 || d->conv_swap[lc] || CP != co_n || (mix_cout > 0 && (C != 1 || mix_cout > 4 || !zin))) return p;
  if (n_in != d->conv_in_size[lc] - k + 1 || n_in > n_out || n_in > 64) return p;
  if (zin && ((n_out & 3) || ((uintptr_t)g & 15))) return p;
  // (the tables of this resize: built at creation for the resize in front of layer lc + 1)
  if (lc + 1 >= d->n_conv || d->rs_tab_off[lc + 1] == 0 || d->conv_prev[lc + 1] != n_in || d->conv_in_size[lc + 1] != n_out) return p;
  const int pad = k - 1, np = n_in + 2 * pad, nc = np - 2, kpad = d->bwd_kpad[lc], ci_tiles = (ci_n + 15) / 16;
  // the form the unfused transposed convolution takes: plain or split-K only (not the direct, not the z-grouped one)
  if (N > SDFR_SPLITK_MAX_LATENTS || direct_ok(d->bwd_direct_off[lc], np, nc, N)) return p;
  const int split = use_split_k(false, (nc * nc * nc + 15) / 16, ci_tiles, N, kpad);
  const sdfr_decoder::ZPlan& zp = d->bwd_z[lc];
  if (!split && zp.zg > 1 && (long long)nc * nc * (nc / zp.zg) * N >= kZGroupMinRows) return p;
  p.ZT = (nc + 15) / 16;
  if (p.ZT > 4) return p;
  // source ranges (resize_sources / resize_weight on the host)
  const float ratio = (float)n_in / (float)n_out, inv = (float)n_out / (float)n_in;
  auto weight = [&](int dd, int i) {
    int i0, i1;
    float l1;
    host_resize_axis(dd, ratio, n_in, i0, i1, l1);
    return (i0 == i ? 1.0f - l1 : 0.0f) + (i1 == i ? l1 : 0.0f);
  };
  std::vector<int> lo(n_in), hi(n_in);
  int max_taps = 1, max_span = 1;
  for (int i = 0; i < n_in; ++i) {
    int d0 = std::max((int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1, 0);
    int d1 = std::min((int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1, n_out - 1);
    max_span = std::max(max_span, d1 - d0 + 1);
    while (d0 <= d1 && weight(d0, i) == 0.0f) ++d0;
    while (d1 >= d0 && weight(d1, i) == 0.0f) --d1;
    if (d0 > d1) return p;
    lo[i] = d0; hi[i] = d1;
    max_taps = std::max(max_taps, d1 - d0 + 1);
  }
  if (max_span > 16 || max_taps > kBtTaps) return p;
  p.taps = max_taps <= 6 ? 6 : 12;
  // columns per workgroup: the smallest tile that gives every workgroup a CU of its own
  static const int kShapes[][2] = {{1, 1}, {1, 2}, {2, 2}, {2, 3}, {3, 3}, {2, 4}, {3, 4}, {4, 4}};
  for (const auto& sh : kShapes) {
    if (sh[0] * sh[1] * p.ZT > 16) break;
    p.TX = sh[0]; p.TY = sh[1];
    if ((long long)((nc + p.TX - 1) / p.TX) * ((nc + p.TY - 1) / p.TY) * ci_tiles * N <= 256) break;
  }
#ifdef SDFR_VJP_TUNE   // timing experiments: tile shape and workgroup size from the environment
  if (const char* e = getenv("SDFR_VJP_TX")) p.TX = atoi(e);
  if (const char* e = getenv("SDFR_VJP_TY")) p.TY = atoi(e);
#endif
  const int tiles = p.TX * p.TY * p.ZT, max_thr = zin ? 512 : 1024;   // (vjp_stage_kernel's launch bounds)
  if (64 * tiles > max_thr) return p;
  p.mode = !split ? 0 : (256 * tiles <= max_thr ? 4 : 1);
  const int IX = p.TX + 2, IY = p.TY + 2;
  for (int pass = 0; pass < 2; ++pass) {
    const int T = pass ? p.TY : p.TX, I = pass ? IY : IX;
    for (int t0 = 0; t0 < nc; t0 += T)
      p.FX = std::max(p.FX, hi[std::min(t0 - pad + I - 1, n_in - 1)] - lo[std::max(t0 - pad, 0)] + 1);
  }
  // channels per round and workgroup size: the whole tensor in one round where the LDS allows
  const int PZ = 16 * p.ZT + 2, patch_n = CP * IX * IY * PZ, RL = zin ? n_out : n_in;
  const int min_thr = std::max(256, p.mode == 4 ? 256 * tiles : 64 * tiles);
  const size_t epi = (size_t)p.TX * p.TY * 16 * nc;   // (the epilogue's rows reuse F)
  for (int CK = C; CK >= 1; --CK) {
    const size_t f_n = std::max((size_t)CK * p.FX * p.FX * RL, epi);
    const size_t units = (size_t)CK * p.FX * p.FX * (zin ? RL / 4 : RL);
    int threads = min_thr;
    while (threads < max_thr && (size_t)threads * 8 < units) threads *= 2;
#ifdef SDFR_VJP_TUNE
    if (const char* e = getenv("SDFR_VJP_THREADS")) threads = std::max(min_thr, std::min(max_thr, atoi(e)));
#endif
    const size_t fixed = (size_t)kpad * 17 + (p.mode == 4 ? 4 * threads : 0) + ((patch_n + 3) & ~3);
    const size_t lds = (fixed + (((size_t)CK * p.FX * IY * n_in + 3) & ~(size_t)3) + f_n) * sizeof(float);
    if (lds > kFusedLdsMax - 8 * 1024 || units >= 65536 || (size_t)CK * IX * IY * n_in >= 65536) continue;   // (8 KB static)
    p.CK = CK;
    p.lds = lds;
    p.threads = threads;
    break;
  }
  if (!p.CK || patch_n >= 65536) return p;
  p.ok = true;
  return p;
}

// e_nin > 0: the stage's epilogue applies the z pass of the NEXT stage's transposed resize (nc -> e_nin)
struct ScaledGrad {   // sdfr_decoder_backward_latent_deferred_scaled
  float* g2;          // [n][volume]
  const float* cnt;   // [n]
  float weight;
  int n;
};
bool launch_vjp_stage(const sdfr_decoder* d, const VjpPlan& p, int lc, const float* g, int C, int n_in, int n_out,
                      const float* act, const float* mix_w, int mix_cout, int e_nin, float* dst, int N, hipStream_t st,
                      const ScaledGrad* sg = nullptr) {
  const int ci_n = d->conv_cin[lc], ci_tiles = (ci_n + 15) / 16, pad = d->conv_k[lc] - 1, nc = n_in + 2 * pad - 2;
  const float* zb = d->d_params + d->zero_bias_off;
  VjpStage s;
  s.g = g; s.act = act; s.mix_w = mix_w; s.mix_b = zb;
  s.wmat = d->d_params + d->bwd_w_off[lc]; s.bias = zb; s.out = dst;
  s.C = C; s.n_in = n_in; s.n_out = n_out; s.pad = pad; s.Cc = ci_n; s.kpad = d->bwd_kpad[lc];
  s.CK = p.CK; s.FX = p.FX; s.ZT = p.ZT; s.TX = p.TX; s.TY = p.TY; s.zin = p.zin; s.e_nin = e_nin;
  s.tab = d->d_params + d->rs_tab_off[lc + 1];
  s.e_tab = e_nin > 0 ? d->d_params + d->rs_tab_off[lc] : nullptr;
  if (sg) {
    if (!p.zin || ((uintptr_t)sg->g2 & 15)) return false;
    s.g2 = sg->g2; s.cnt = sg->cnt; s.weight = sg->weight;
  }
  const dim3 grid(((nc + p.TX - 1) / p.TX) * ((nc + p.TY - 1) / p.TY), ci_tiles, N), block(p.threads);
#define SDFR_VS(CO, TAPS, MODE, ZIN)                                                               \
  do {                                                                                           \
    const void* fn = reinterpret_cast<const void*>(&vjp_stage_kernel<CO, TAPS, MODE, ZIN>);      \
    if (p.lds > 56 * 1024 && !fused_lds_limit(fn, kFusedLdsMax - 8 * 1024)) return false;        \
    hipLaunchKernelGGL((vjp_stage_kernel<CO, TAPS, MODE, ZIN>), grid, block, p.lds, st, s);      \
  } while (0)
#define SDFR_VS_M(CO, TAPS, ZIN) { if (p.mode == 0) SDFR_VS(CO, TAPS, 0, ZIN); else if (p.mode == 4) SDFR_VS(CO, TAPS, 4, ZIN); else SDFR_VS(CO, TAPS, 1, ZIN); }
#define SDFR_VS_T(CO, ZIN) { if (p.taps == 6) SDFR_VS_M(CO, 6, ZIN) else SDFR_VS_M(CO, 12, ZIN) }
  if (mix_cout == 0) { if (p.zin) SDFR_VS_T(0, true) else SDFR_VS_T(0, false) }
  else if (mix_cout == 1) SDFR_VS_T(1, true)
  else if (mix_cout == 2) SDFR_VS_T(2, true)
  else if (mix_cout == 3) SDFR_VS_T(3, true)
  else SDFR_VS_T(4, true)
#undef SDFR_VS_T
#undef SDFR_VS_M
#undef SDFR_VS
  return true;
}
}  // namespace

extern "C" int sdfr_decoder_set_option(sdfr_decoder* d, int option, int value) {
  if (!d) return fail(SDFR_E_NULL, "sdfr_decoder_set_option: decoder is NULL");
  switch (option) {
    case SDFR_DECODER_OPT_FUSED_RESIZE:
      return d->opt_fused_resize.exchange(value < 0 ? 0 : (value > 2 ? 2 : value), std::memory_order_relaxed);
    case SDFR_DECODER_OPT_TILED_VJP:
      return d->opt_tiled_vjp.exchange(value ? 1 : 0, std::memory_order_relaxed);
    case SDFR_DECODER_OPT_FC_ONE_WAVE:
      return d->opt_fc_one_wave.exchange(value ? 1 : 0, std::memory_order_relaxed);
    case SDFR_DECODER_OPT_FUSED_SINGLE:
      return d->opt_fused_single.exchange(value & 15, std::memory_order_relaxed);
    default:
      return fail(SDFR_E_INVALID, "sdfr_decoder_set_option: unknown option %d", option);
  }
}

extern "C" int sdfr_decoder_create(const float* h_params, size_t n_params, int latent, int n_fc,
                                   const int* fc_out, int n_conv, const int* conv_in_size,
                                   const int* conv_cin, const int* conv_cout, const int* conv_k,
                                   const int* conv_relu, int volume, float tsdf, int device,
                                   sdfr_decoder** out_handle) {
  if (!h_params || !fc_out || !conv_in_size || !conv_cin || !conv_cout || !conv_k || !conv_relu ||
      !out_handle)
    return fail(SDFR_E_NULL, "sdfr_decoder_create: NULL pointer argument");
  if (latent < 1 || latent > kMaxHidden || n_fc < 1 || n_fc > 8 || n_conv < 1 || n_conv > 16 || volume < 1)
    return fail(SDFR_E_INVALID, "sdfr_decoder_create: unsupported layer counts / sizes");
  // consistency checks of SDFDecoder.sanity_check (sdf_vae.py:207-215) + parameter count
  size_t need = 0;
  int width = latent;
  for (int l = 0; l < n_fc; ++l) {
    if (fc_out[l] < 1) return fail(SDFR_E_INVALID, "fc layer %d has no outputs", l);
    if (l < n_fc - 1 && fc_out[l] > kMaxHidden)
      return fail(SDFR_E_INVALID, "hidden fc layer %d wider than %d", l, kMaxHidden);
    need += (size_t)fc_out[l] * width + fc_out[l];
    width = fc_out[l];
  }
  if ((long long)conv_cin[0] * conv_in_size[0] * conv_in_size[0] * conv_in_size[0] != width)
    return fail(SDFR_E_INVALID, "last fc layer (%d) does not match the first conv input", width);
  for (int l = 0; l < n_conv; ++l) {
    const int k = conv_k[l];
    if (k < 1 || conv_in_size[l] < k || conv_cin[l] < 1 || conv_cout[l] < 1)
      return fail(SDFR_E_INVALID, "conv layer %d has an invalid shape", l);
    if (l + 1 < n_conv && conv_cout[l] != conv_cin[l + 1])
      return fail(SDFR_E_INVALID, "conv layer %d out_channels != next in_channels", l);
    need += (size_t)conv_cout[l] * conv_cin[l] * k * k * k + conv_cout[l];
    const size_t kpad = ((size_t)conv_cin[l] * k * k * k + 3) / 4 * 4;
    if (kpad * 17 * sizeof(float) > 64 * 1024)
      return fail(SDFR_E_INVALID, "conv layer %d: Cin*k^3 = %zu too large for the LDS-resident weight tile", l, kpad);
  }
  if (conv_cout[n_conv - 1] != 1) return fail(SDFR_E_INVALID, "last conv layer must have one output channel");
  if (need != n_params)
    return fail(SDFR_E_INVALID, "parameter count %zu does not match the layer description (%zu)", n_params, need);

  sdfr_decoder* d = new sdfr_decoder();
  d->device = device; d->latent = latent; d->n_fc = n_fc; d->n_conv = n_conv; d->volume = volume;
  d->tsdf = tsdf;
  d->fc_out.assign(fc_out, fc_out + n_fc);
  d->conv_in_size.assign(conv_in_size, conv_in_size + n_conv);
  d->conv_cin.assign(conv_cin, conv_cin + n_conv);
  d->conv_cout.assign(conv_cout, conv_cout + n_conv);
  d->conv_k.assign(conv_k, conv_k + n_conv);
  d->conv_relu.assign(conv_relu, conv_relu + n_conv);
  for (int l = 0, prev = conv_in_size[0]; l < n_conv; ++l) {
    d->conv_prev.push_back(prev);
    d->conv_swap.push_back(conv_k[l] == 1 && prev != conv_in_size[l] && conv_cout[l] <= conv_cin[l]);
    prev = conv_in_size[l] - conv_k[l] + 1;
  }

  // device image: fc weights (last one transposed), conv weight matrices [co_tile][Kpad][16],
  // tap tables, biases
  std::vector<float> img;
  auto align = [&]() { while (img.size() % 64) img.push_back(0.0f); };
  const float* p = h_params;
  width = latent;
  for (int l = 0; l < n_fc; ++l) {
    const int wo = fc_out[l];
    align();
    d->fc_w_off.push_back(img.size());
    if (l < n_fc - 1) {
      img.insert(img.end(), p, p + (size_t)wo * width);
    } else {
      for (int i = 0; i < width; ++i)
        for (int o = 0; o < wo; ++o) img.push_back(p[(size_t)o * width + i]);
    }
    p += (size_t)wo * width;
    align();
    d->fc_b_off.push_back(img.size());
    img.insert(img.end(), p, p + wo);
    p += wo;
    d->max_act = std::max(d->max_act, (size_t)wo);
    width = wo;
  }
  for (int l = 0; l < n_conv; ++l) {
    // a swapped 1x1 layer convolves the tensor as it arrives (size conv_prev), before the resize
    const int k = conv_k[l], ci_n = conv_cin[l], co_n = conv_cout[l];
    const int n = d->conv_swap[l] ? d->conv_prev[l] : conv_in_size[l];
    const int K = ci_n * k * k * k, kpad = (K + 3) / 4 * 4, co_tiles = (co_n + 15) / 16;
    d->conv_kpad.push_back(kpad);
    align();
    d->conv_w_off.push_back(img.size());
    for (int ct = 0; ct < co_tiles; ++ct)
      for (int kk = 0; kk < kpad; ++kk)
        for (int j = 0; j < 16; ++j) {
          const int co = ct * 16 + j;
          img.push_back((kk < K && co < co_n) ? p[(size_t)co * K + kk] : 0.0f);
        }
    p += (size_t)co_n * K;
    align();
    d->conv_b_off.push_back(img.size());
    img.insert(img.end(), p, p + co_n);
    p += co_n;
    align();
    d->conv_tab_off.push_back(img.size());
    for (int kk = 0; kk < kpad; ++kk) {
      int off = 0;
      if (kk < K) {
        const int ci = kk / (k * k * k), r = kk % (k * k * k);
        const int a = r / (k * k), b = (r / k) % k, c = r % k;
        off = ci * n * n * n + (a * n + b) * n + c;
      }
      float f;
      static_assert(sizeof(int) == sizeof(float), "tap table is stored in the float image");
      memcpy(&f, &off, sizeof(f));
      img.push_back(f);
    }
    const int m = conv_in_size[l] - k + 1;
    d->max_act = std::max(d->max_act, (size_t)ci_n * n * n * n);
    d->max_act = std::max(d->max_act, (size_t)co_n * n * n * n);
    d->max_act = std::max(d->max_act, (size_t)co_n * m * m * m);
  }
  d->max_act = std::max(d->max_act, (size_t)volume * volume * volume);
  // backward: data-gradient of conv l = valid conv of the zero-padded output gradient with
  //   Wb[kk = co*k^3 + (a*k+b)*k + c][ci] = W[co][ci][k-1-a][k-1-b][k-1-c]
  {
    const float* q = h_params;
    int wdt = latent;
    for (int l = 0; l < n_fc; ++l) { q += (size_t)fc_out[l] * wdt + fc_out[l]; wdt = fc_out[l]; }
    int prev_n = conv_in_size[0];
    d->max_bwd = (size_t)volume * volume * volume;
    size_t tape = 0;
    d->tape_fc_off = tape;
    tape += (size_t)fc_out[n_fc - 1];
    for (int l = 0; l < n_conv; ++l) {
      const int k = conv_k[l], ci_n = conv_cin[l], co_n = conv_cout[l], n = conv_in_size[l];
      const int m = n - k + 1, k3 = k * k * k;
      // (swapped 1x1 layer: the transposed conv runs at the size of the tensor that entered it)
      const int np = d->conv_swap[l] ? d->conv_prev[l] : m + 2 * (k - 1);
      const int Kb = co_n * k3, kpad = (Kb + 3) / 4 * 4, ci_tiles = (ci_n + 15) / 16;
      if ((size_t)kpad * 17 * sizeof(float) > 64 * 1024) {
        delete d;
        return fail(SDFR_E_INVALID, "conv layer %d: Cout*k^3 too large for the backward weight tile", l);
      }
      d->bwd_kpad.push_back(kpad);
      align();
      d->bwd_w_off.push_back(img.size());
      for (int ct = 0; ct < ci_tiles; ++ct)
        for (int kk = 0; kk < kpad; ++kk)
          for (int j = 0; j < 16; ++j) {
            const int ci = ct * 16 + j;
            float v = 0.0f;
            if (kk < Kb && ci < ci_n) {
              const int co = kk / k3, r = kk % k3, a = r / (k * k), b = (r / k) % k, c = r % k;
              v = q[(((size_t)co * ci_n + ci) * k + (k - 1 - a)) * k * k + (size_t)(k - 1 - b) * k + (k - 1 - c)];
            }
            img.push_back(v);
          }
      align();
      d->bwd_tab_off.push_back(img.size());
      for (int kk = 0; kk < kpad; ++kk) {
        int off = 0;
        if (kk < Kb) {
          const int co = kk / k3, r = kk % k3, a = r / (k * k), b = (r / k) % k, c = r % k;
          off = co * np * np * np + (a * np + b) * np + c;
        }
        float f;
        memcpy(&f, &off, sizeof(f));
        img.push_back(f);
      }
      q += (size_t)co_n * ci_n * k3 + co_n;
      d->max_bwd = std::max(d->max_bwd, (size_t)co_n * np * np * ((np + 3) & ~3));   // (z pitch padded to 16 bytes)
      d->max_bwd = std::max(d->max_bwd, (size_t)co_n * n * n * n);
      d->max_bwd = std::max(d->max_bwd, (size_t)ci_n * n * n * n);
      d->max_bwd = std::max(d->max_bwd, (size_t)ci_n * prev_n * prev_n * prev_n);
      d->tape_conv_off.push_back(tape);
      tape += (size_t)co_n * m * m * m;
      prev_n = m;
    }
    d->max_bwd = (d->max_bwd + 3) & ~(size_t)3;   // both halves of the backward workspace 16-byte aligned
    d->tape_floats = tape;
    align();
    d->zero_bias_off = img.size();
    int max_c = 16;
    for (int l = 0; l < n_conv; ++l) max_c = std::max(max_c, conv_cin[l]);
    img.insert(img.end(), (size_t)max_c, 0.0f);
  }
  // transposed-resize tables (resize_row16's results, the kernels' float arithmetic): see rs_tab_off
  d->rs_tab_off.assign(n_conv, 0);
  for (int l = 1; l < n_conv; ++l) {
    const int n_in = d->conv_prev[l], n_out = conv_in_size[l];
    if (n_in == n_out || n_in > n_out || n_in > 64) continue;
    const float ratio = (float)n_in / (float)n_out, inv = (float)n_out / (float)n_in;
    auto weight = [&](int dd, int i) {
      float sp = fmaf(ratio, (float)dd + 0.5f, -0.5f);
      sp = sp < 0.0f ? 0.0f : sp;
      const int i0 = std::min((int)sp, n_in - 1), i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
      const float l1 = sp - (float)i0;
      return (i0 == i ? 1.0f - l1 : 0.0f) + (i1 == i ? l1 : 0.0f);
    };
    std::vector<float> tab((size_t)n_in * 16, 0.0f);
    bool ok = true;
    for (int i = 0; i < n_in && ok; ++i) {
      int d0 = std::max((int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1, 0);
      int d1 = std::min((int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1, n_out - 1);
      if (d1 - d0 + 1 > 16) ok = false;
      while (d0 <= d1 && weight(d0, i) == 0.0f) ++d0;
      while (d1 >= d0 && weight(d1, i) == 0.0f) --d1;
      const int nt = d1 - d0 + 1;
      if (nt < 1 || nt > 12) { ok = false; break; }
      memcpy(&tab[(size_t)i * 16], &d0, sizeof(int));
      memcpy(&tab[(size_t)i * 16 + 1], &nt, sizeof(int));
      for (int kk = 0; kk < nt; ++kk) tab[(size_t)i * 16 + 2 + kk] = weight(d0 + kk, i);
    }
    if (!ok) continue;
    align();
    d->rs_tab_off[l] = img.size();
    img.insert(img.end(), tab.begin(), tab.end());
  }
  // z-grouped plans.  wfun(co, ci, a, b, c): weight of input channel ci at tap (a,b,c) for output
  // channel co, in the correlation sense out[x] = sum w(a,b,c) in[x + (a,b,c)].
  auto plan_z = [&](int cin, int cout, int k, int n, auto wfun) {
    sdfr_decoder::ZPlan zp;
    const int m = n - k + 1;
    if (k < 2 || m < 1) return zp;
    for (int g = 4; g >= 2; --g)
      if (g * cout <= 16 && m % g == 0) { zp.zg = g; break; }
    if (zp.zg == 1) return zp;
    const int kz = k + zp.zg - 1, K = cin * k * k * kz;
    zp.kpad = (K + 3) / 4 * 4;
    if ((size_t)zp.kpad * 17 * sizeof(float) > 64 * 1024) { zp.zg = 1; return zp; }
    align();
    zp.w_off = img.size();
    for (int kk = 0; kk < zp.kpad; ++kk)
      for (int j = 0; j < 16; ++j) {
        float v = 0.0f;
        if (kk < K) {
          const int ci = kk / (k * k * kz), r = kk % (k * k * kz);
          const int a = r / (k * kz), b = (r / kz) % k, cz = r % kz;
          const int dz = j / cout, co = j % cout, c = cz - dz;
          if (dz < zp.zg && c >= 0 && c < k) v = wfun(co, ci, a, b, c);
        }
        img.push_back(v);
      }
    align();
    zp.tab_off = img.size();
    for (int kk = 0; kk < zp.kpad; ++kk) {
      int off = 0;
      if (kk < K) {
        const int ci = kk / (k * k * kz), r = kk % (k * k * kz);
        const int a = r / (k * kz), b = (r / kz) % k, cz = r % kz;
        off = ci * n * n * n + (a * n + b) * n + cz;
      }
      float f;
      memcpy(&f, &off, sizeof(f));
      img.push_back(f);
    }
    return zp;
  };
  {
    const float* q = h_params;
    int wdt = latent;
    for (int l = 0; l < n_fc; ++l) { q += (size_t)fc_out[l] * wdt + fc_out[l]; wdt = fc_out[l]; }
    for (int l = 0; l < n_conv; ++l) {
      const int k = conv_k[l], ci_n = conv_cin[l], co_n = conv_cout[l], n = conv_in_size[l];
      const int k3 = k * k * k, m = n - k + 1;
      const float* W = q;  // [co][ci][k][k][k]
      d->fwd_z.push_back(plan_z(ci_n, co_n, k, n, [&](int co, int ci, int a, int b, int c) {
        return W[((size_t)co * ci_n + ci) * k3 + (a * k + b) * k + c];
      }));
      // data gradient: channels swap roles, taps are flipped, the input is the zero-padded gradient
      d->bwd_z.push_back(plan_z(co_n, ci_n, k, m + 2 * (k - 1), [&](int co, int ci, int a, int b, int c) {
        return W[((size_t)ci * ci_n + co) * k3 + ((k - 1 - a) * k + (k - 1 - b)) * k + (k - 1 - c)];
      }));
      // direct-convolution weights [ci][a][b][c][co] (k = 3, 4 / 8 / 16 output channels)
      auto direct = [&](int cin, int cout, auto wfun) -> size_t {
        if (k != 3 || !(cout == 4 || cout == 8 || cout == 16)) return 0;
        align();
        const size_t off = img.size();
        for (int ci = 0; ci < cin; ++ci)
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
              for (int c = 0; c < 3; ++c)
                for (int co = 0; co < cout; ++co) img.push_back(wfun(co, ci, a, b, c));
        return off;
      };
      d->fwd_direct_off.push_back(direct(ci_n, co_n, [&](int co, int ci, int a, int b, int c) {
        return W[((size_t)co * ci_n + ci) * k3 + (a * k + b) * k + c];
      }));
      d->bwd_direct_off.push_back(direct(co_n, ci_n, [&](int co, int ci, int a, int b, int c) {
        return W[((size_t)ci * ci_n + co) * k3 + ((k - 1 - a) * k + (k - 1 - b)) * k + (k - 1 - c)];
      }));
      q += (size_t)co_n * ci_n * k3 + co_n;
    }
  }
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc((void**)&d->d_params, img.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(d->d_params, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (d->d_params) (void)hipFree(d->d_params);
    delete d;
    return hip_fail(e, "sdfr_decoder_create (device image)");
  }
  *out_handle = d;
  return 0;
}

extern "C" void sdfr_decoder_destroy(sdfr_decoder* d) {
  if (!d) return;
  if (d->d_params) {
    (void)hipSetDevice(d->device);
    (void)hipFree(d->d_params);
  }
  delete d;
}

extern "C" size_t sdfr_decoder_workspace_bytes(const sdfr_decoder* d, int N) {
  if (!d || N <= 0) return 0;
  return 2 * (size_t)N * d->max_act * sizeof(float) + 512;
}

extern "C" size_t sdfr_decoder_tape_bytes(const sdfr_decoder* d, int N) {
  if (!d || N <= 0) return 0;
  return (size_t)N * d->tape_floats * sizeof(float);
}

namespace {
// stages: 1 the Linear stack only (its output into the tape's slot), 2 the convolutional part from that slot, 3 both
int decoder_forward_impl(const sdfr_decoder* d, const float* z, int N, int enforce_tsdf,
                         float* out, float* tape, void* workspace,
                         size_t workspace_bytes, void* stream, int stages) {
  if (!d) return fail(SDFR_E_NULL, "sdfr_decoder_forward: NULL decoder");
  if (N < 0 || N > 65535) return fail(SDFR_E_INVALID, "N=%d out of range", N);
  if (stages < 1 || stages > 3) return fail(SDFR_E_INVALID, "sdfr_decoder_forward_stage: stages=%d", stages);
  if (N == 0) return 0;
  if (stages != 3 && !tape) return fail(SDFR_E_NULL, "sdfr_decoder_forward_stage: the stages meet in the tape");
  if (((stages & 1) && !z) || ((stages & 2) && !out) || !workspace)
    return fail(SDFR_E_NULL, "sdfr_decoder_forward: NULL pointer argument");
  if (workspace_bytes < sdfr_decoder_workspace_bytes(d, N))
    return fail(SDFR_E_WORKSPACE, "sdfr_decoder_forward: workspace %zu < %zu bytes", workspace_bytes,
                sdfr_decoder_workspace_bytes(d, N));
  SDFR_HIP_TRY(hipSetDevice(d->device));
  hipStream_t st = (hipStream_t)stream;
  uintptr_t wsp = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  float* buf[2] = {(float*)wsp, (float*)wsp + (size_t)N * d->max_act};
  int cur = 0;

  FcDesc fd;
  fd.n_fc = d->n_fc;
  fd.width[0] = d->latent;
  for (int l = 0; l < d->n_fc; ++l) {
    fd.width[l + 1] = d->fc_out[l];
    fd.w_off[l] = (long long)d->fc_w_off[l];
    fd.b_off[l] = (long long)d->fc_b_off[l];
  }
  const int last = d->fc_out[d->n_fc - 1];
  // with a tape, every ReLU'd layer output goes to its own slot (and is read from there)
  const float* act_in;
  int l_first = 0;   // the first conv layer the loop below still has to run
  // few latents: the Linear stack and the first convolution as one launch (fc_conv_kernel) -- a 3x3x3 layer that is not
  // the last, with a layer behind it that is not a 1x1x1 one swapped with its resize
  if (stages == 2) {
    act_in = tape + (size_t)N * d->tape_fc_off;   // (left there by stage 1 -- or by the loop's tail, sdfr_loop_tail_fused)
    l_first = -1;
  }
  if (stages == 3 && d->n_conv >= 2 && !d->conv_swap[0] && d->conv_k[0] == 3 && !(d->conv_swap[1] && d->conv_k[1] == 1) &&
      N < 32 && d->conv_in_size[0] - 2 != d->volume) {
    float* fc_dst = tape ? tape + (size_t)N * d->tape_fc_off : nullptr;
    float* dst0 = tape ? tape + (size_t)N * d->tape_conv_off[0] : buf[cur ^ 1];
    if (launch_fc_conv(d, fd, z, fc_dst, d->fwd_direct_off[0], d->d_params + d->conv_w_off[0], d->d_params + d->conv_b_off[0], dst0,
                       d->conv_cin[0], d->conv_cout[0], d->conv_in_size[0], d->conv_kpad[0], d->conv_relu[0], N, st)) {
      if (dst0 == buf[cur ^ 1]) cur ^= 1;
      act_in = dst0;
      l_first = 1;
    }
  }
  if (l_first == 0) {
    float* fc_dst = tape ? tape + (size_t)N * d->tape_fc_off : buf[cur];
    int hid = d->latent;   // widest input of a layer
    for (int l = 0; l + 1 < d->n_fc; ++l) hid = std::max(hid, d->fc_out[l]);
    constexpr int kFcSamples = 8;
    const size_t fc_lds = 2 * (size_t)hid * kFcSamples * sizeof(float);
    if (N >= 4 * kFcSamples && fc_lds <= 32 * 1024 && decoder_fc_one_wave(d, fd))   // batches (small ones keep the form of a single decode)
      hipLaunchKernelGGL((fc_stack_batch_kernel<kFcSamples, true>),
                         dim3((last + kFcBlock - 1) / kFcBlock, (N + kFcSamples - 1) / kFcSamples), dim3(kFcBlock),
                         fc_lds, st, d->d_params, fd, z, N, hid, fc_dst);
    else if (N >= 4 * kFcSamples && fc_lds <= 48 * 1024)
      hipLaunchKernelGGL((fc_stack_batch_kernel<kFcSamples, false>),
                         dim3((last + kFcBlock - 1) / kFcBlock, (N + kFcSamples - 1) / kFcSamples), dim3(kFcBlock),
                         fc_lds, st, d->d_params, fd, z, N, hid, fc_dst);
    else if (decoder_fc_one_wave(d, fd))
      hipLaunchKernelGGL(fc_stack_kernel<true>, dim3((last + kFcBlock - 1) / kFcBlock, N), dim3(kFcBlock), 0, st,
                         d->d_params, fd, z, fc_dst);
    else
      hipLaunchKernelGGL(fc_stack_kernel<false>, dim3((last + kFcBlock - 1) / kFcBlock, N), dim3(kFcBlock), 0, st,
                         d->d_params, fd, z, fc_dst);
    act_in = fc_dst;
  }
  if (l_first < 0) l_first = 0;
  if (stages == 1) {
    SDFR_HIP_TRY(hipGetLastError());
    return 0;
  }

  const float clampv = (enforce_tsdf && d->tsdf > 0.0f) ? d->tsdf : 0.0f;
  int c = d->conv_cin[0], n = d->conv_in_size[0];
  if (l_first == 1) { c = d->conv_cout[0]; n = d->conv_in_size[0] - d->conv_k[0] + 1; }
  const size_t vox = (size_t)d->volume * d->volume * d->volume;
  auto resize = [&](const float* src, int C, int ni, int no, int relu, float clamp, float* dst) {
    // input columns under an 8 x 8 output patch, worst tile (same float arithmetic as the kernel)
    int max_r = 0;
    const float ratio = (float)ni / (float)no;
    auto first_last = [&](int dd, int& i0, int& i1) {
      float sp = fmaf(ratio, (float)dd + 0.5f, -0.5f);
      sp = sp < 0.0f ? 0.0f : sp;
      i0 = std::min((int)sp, ni - 1);
      i1 = i0 + (i0 < ni - 1 ? 1 : 0);
    };
    for (int t0 = 0; t0 < no; t0 += kResizeTile) {
      int lo, hi, tmp;
      first_last(t0, lo, tmp);
      first_last(std::min(t0 + kResizeTile, no) - 1, tmp, hi);
      max_r = std::max(max_r, hi - lo + 1);
    }
    const size_t lds = ((size_t)max_r * max_r * (ni + no) + 3) * sizeof(float);   // (+3: the second array starts 16-byte aligned)
    const int tiles = (no + kResizeTile - 1) / kResizeTile;
    const bool pow2 = no >= 16 && no <= 128 && (no & (no - 1)) == 0;
    // (a single decode has too few tiles to fill the chip: the gather kernel is faster there)
    const long long tile_items = (long long)tiles * tiles * C * N;
    if (pow2 && ni <= no && lds <= 48 * 1024 && (long long)C * N < (1ll << 30) && tile_items >= SDFR_RESIZE_TILED_MIN_ITEMS) {
      // volumes per workgroup: as many as still leave ~8 workgroups per CU
      const int items = C * N;
      // (256 mug latents, us: 14 -> 32 x 8 channels 68.6 / 52.9 / 55.4 / 53.3 at 1 / 2 / 4 / 8 volumes per workgroup,
      // 30 -> 64 x 1 channel 68.5 / 60.3 / 55.8 / 63.6 -- profiles/r04_decoder_resize_ipw.txt; before this form 70.8 / 66.7)
      const int ipw = std::max((int)std::max<long long>(1, std::min<long long>(4, tile_items / 2048)), (items + 65534) / 65535);
      const dim3 grid(tiles * tiles, (items + ipw - 1) / ipw, 1);
      const bool vec = ((uintptr_t)dst & 15) == 0;
#define SDFR_RESIZE_T(LOG)                                                                                              \
  do {                                                                                                                  \
    if (vec) hipLaunchKernelGGL((resize3_tiled_kernel<LOG, true>), grid, dim3(256), lds, st, src, items, ipw, ni, relu, \
                                clamp, max_r * max_r, dst);                                                             \
    else hipLaunchKernelGGL((resize3_tiled_kernel<LOG, false>), grid, dim3(256), lds, st, src, items, ipw, ni, relu,    \
                            clamp, max_r * max_r, dst);                                                                 \
  } while (0)
      if (no == 16) SDFR_RESIZE_T(4); else if (no == 32) SDFR_RESIZE_T(5); else if (no == 64) SDFR_RESIZE_T(6); else SDFR_RESIZE_T(7);
#undef SDFR_RESIZE_T
      return;
    }
    const size_t cnt = (size_t)C * no * no * no;
    hipLaunchKernelGGL(resize3_kernel, dim3((unsigned)((cnt + 255) / 256), N), dim3(256), 0, st, src, C,
                       ni, no, relu, clamp, dst);
  };
  bool premixed = false;   // the previous layer's epilogue has applied this (1x1x1) layer already: act_in is its output
  for (int l = l_first; l < d->n_conv; ++l) {
    const bool swap = d->conv_swap[l] != 0, is_last = (l == d->n_conv - 1);
    const int k = d->conv_k[l], co_n = d->conv_cout[l], kpad = d->conv_kpad[l];
    // (batches: an up-sampling resize in front of a 3x3x3 layer is folded into that layer's patch load --
    // conv3d_direct_up_kernel -- and the up-sampled tensor is never written)
    bool fused_up = false, fused_mixed = false;
    if (!swap && n != d->conv_in_size[l] && k == 3 && d->fwd_direct_off[l] != 0) {
      const int nf = d->conv_in_size[l], mf = nf - k + 1;
      const bool to_out_f = is_last && mf == d->volume && clampv == 0.0f && !(tape && d->conv_relu[l]);
      float* ldst = to_out_f ? out : (tape ? tape + (size_t)N * d->tape_conv_off[l] : nullptr);
      float* cdst = ldst ? ldst : buf[cur ^ 1];
      fused_up = launch_direct_up(d, d->fwd_direct_off[l], act_in, n, d->d_params + d->conv_b_off[l], cdst, c, co_n,
                                  nf, mf, d->conv_relu[l], N, st);
      if (fused_up) n = nf;   // (act_in stays the coarse tensor: the launch has consumed it)
    }
    // few latents: the resize inside the split-K MFMA convolution's operand fetch (conv3d_mfma_up_kernel)
    if (!fused_up && !swap && n != d->conv_in_size[l] && k == 3) {
      const int nf = d->conv_in_size[l], mf = nf - k + 1;
      const bool to_out_f = is_last && mf == d->volume && clampv == 0.0f && !(tape && d->conv_relu[l]);
      float* ldst = to_out_f ? out : (tape ? tape + (size_t)N * d->tape_conv_off[l] : nullptr);
      float* cdst = ldst ? ldst : buf[cur ^ 1];
      // ... and where a swapped 1x1x1 layer follows (the mug decoder's last: 4 -> 1 channels behind the resize to the
      // volume), its channel mix in this launch's epilogue: the resize behind it gathers one channel, not four
      // (resize3_kernel instead of resize3_mix_kernel: bit for bit the same corner values)
      const bool mix2 = (d->opt_fused_single.load(std::memory_order_relaxed) & 1) && l + 1 < d->n_conv && !to_out_f &&
                        d->conv_swap[l + 1] && d->conv_k[l + 1] == 1 && d->conv_cout[l + 1] <= 4 && co_n <= 16 &&
                        d->conv_in_size[l + 1] != mf;
      fused_up = launch_mfma_up(d, d->fwd_direct_off[l], act_in, n, d->d_params + d->conv_w_off[l],
                                d->d_params + d->conv_b_off[l], mix2 ? ldst : cdst, c, co_n, nf, mf, kpad, d->conv_relu[l],
                                N, st, mix2 ? d->d_params + d->conv_w_off[l + 1] : nullptr,
                                mix2 ? d->d_params + d->conv_b_off[l + 1] : nullptr, mix2 ? d->conv_cout[l + 1] : 0,
                                mix2 ? buf[cur ^ 1] : nullptr);
      if (fused_up) {
        n = nf;
        fused_mixed = mix2;
      }
    }
    if (!swap && n != d->conv_in_size[l]) {
      resize(act_in, c, n, d->conv_in_size[l], 0, 0.0f, buf[cur ^ 1]);
      cur ^= 1;
      act_in = buf[cur];
      n = d->conv_in_size[l];
    }
    const int m = n - k + 1;                     // swapped: k == 1, the conv keeps the incoming size
    const int m_out = d->conv_in_size[l] - k + 1;  // size of the layer's output tensor
    // where the layer's output goes: straight to `out`, to its tape slot, or to the other buffer
    // (a ReLU'd last layer of a taped forward goes to its tape slot first -- the VJP reads its mask there -- and is
    // copied out: until round 6 it went straight to `out` and the VJP masked with whatever the slot held)
    const bool to_out = is_last && m_out == d->volume && (swap || clampv == 0.0f) && !(tape && d->conv_relu[l]);
    float* layer_dst = to_out ? out : (tape ? tape + (size_t)N * d->tape_conv_off[l] : nullptr);
    float* conv_dst = (!swap && layer_dst) ? layer_dst : buf[cur ^ 1];
    const int conv_relu = swap ? 0 : d->conv_relu[l];
    const float* wm = d->d_params + d->conv_w_off[l];
    const float* bs = d->d_params + d->conv_b_off[l];
    if (!premixed && swap && k == 1 && co_n <= 4 && m_out != n &&
        (size_t)N * co_n * m_out * m_out * m_out <= kFewElementsMix) {
      // single latents: the 1x1 mix inside the resize (one launch)
      float* dst = layer_dst ? layer_dst : buf[cur ^ 1];
      const size_t vo = (size_t)m_out * m_out * m_out;
      const dim3 gm((unsigned)((vo + 255) / 256), N);
      const float clampm = to_out ? clampv : 0.0f;
#define SDFR_MIXF(CO) hipLaunchKernelGGL((resize3_mix_kernel<CO>), gm, dim3(256), 0, st, act_in, wm, bs, c, n, m_out, d->conv_relu[l], clampm, dst)
      if (co_n == 1) SDFR_MIXF(1); else if (co_n == 2) SDFR_MIXF(2); else if (co_n == 3) SDFR_MIXF(3); else SDFR_MIXF(4);
#undef SDFR_MIXF
      if (dst == buf[cur ^ 1]) cur ^= 1;
      act_in = dst;
      c = co_n;
      n = m_out;
      if (is_last && act_in != out) {
        if (n != d->volume) {
          resize(act_in, 1, n, d->volume, 0, clampv, out);
        } else {
          copy_words_async(out, act_in, (size_t)N * vox, st);
          if (clampv > 0.0f)
            hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)(((size_t)N * vox + 255) / 256)), dim3(256), 0,
                               st, out, (size_t)N * vox, clampv);
        }
      }
      continue;
    }
    // batches: a swapped 1x1x1 layer behind this one is applied in this layer's epilogue (conv3d_direct_kernel)
    bool mix_next = false;
    if (!swap && !fused_up && k == 3 && l + 1 < d->n_conv && d->conv_swap[l + 1] && d->conv_k[l + 1] == 1 &&
        d->conv_cout[l + 1] <= 4 && d->fwd_direct_off[l] != 0) {
      const int mo = d->conv_in_size[l + 1], co2 = d->conv_cout[l + 1];
      const bool few_next = mo != m && (size_t)N * co2 * mo * mo * mo <= kFewElementsMix;   // (takes resize3_mix_kernel)
      float* act_dst = tape ? tape + (size_t)N * d->tape_conv_off[l] : nullptr;
      mix_next = !few_next && launch_direct(d, d->fwd_direct_off[l], act_in, bs, act_dst, c, co_n, n, n, m, conv_relu,
                                            N, st, d->d_params + d->conv_w_off[l + 1], d->d_params + d->conv_b_off[l + 1],
                                            co2, buf[cur ^ 1]);
      if (mix_next) conv_dst = buf[cur ^ 1];
    }
    if (premixed) {
      conv_dst = const_cast<float*>(act_in);   // (nothing to launch, no buffer taken)
      premixed = false;
    } else if (mix_next) {
      premixed = true;
    } else if (fused_up) {
      // (launched above)
      if (fused_mixed) {
        premixed = true;
        conv_dst = buf[cur ^ 1];
      }
    } else if (k == 1 && co_n <= 4) {
      const int voxn = n * n * n;
      const dim3 g1((voxn + 255) / 256, N);
      const bool v4 = N >= 32 && (voxn & 3) == 0 && (((uintptr_t)act_in | (uintptr_t)conv_dst) & 15) == 0;
      const dim3 g4((voxn / 4 + 255) / 256, N);
#define SDFR_CONV1(CO)                                                                                                \
  if (v4) hipLaunchKernelGGL((conv1x1_vec4_kernel<CO>), g4, dim3(256), 0, st, act_in, wm, bs, c, voxn, conv_relu,    \
                             conv_dst);                                                                              \
  else hipLaunchKernelGGL((conv1x1_kernel<CO>), g1, dim3(256), 0, st, act_in, wm, bs, c, voxn, conv_relu, conv_dst)
      if (co_n == 1) { SDFR_CONV1(1); } else if (co_n == 2) { SDFR_CONV1(2); } else if (co_n == 3) { SDFR_CONV1(3); } else { SDFR_CONV1(4); }
#undef SDFR_CONV1
    } else if (!swap && launch_direct(d, d->fwd_direct_off[l], act_in, bs, conv_dst, c, co_n, n, n, m, conv_relu, N, st)) {
      // (batched: direct VALU convolution)
    } else {
      const sdfr_decoder::ZPlan& zp = d->fwd_z[l];
      // (a single decode is latency-bound: there the finer grid of the ungrouped form wins)
      // (up to 16 samples the split-K form decides, so that small batches equal single decodes bit for bit)
      const int split = use_split_k(false, (m * m * m + 15) / 16, (co_n + 15) / 16, N, kpad);
      const bool zgrp = !split && zp.zg > 1 && !swap && (long long)m * m * (m / zp.zg) * N >= kZGroupMinRows;
      const int kp = zgrp ? zp.kpad : kpad, rows = m * m * (m / (zgrp ? zp.zg : 1));
      const int nt = (rows + 15) / 16;
      launch_mfma(act_in, zgrp ? d->d_params + zp.w_off : wm,
                  reinterpret_cast<const int*>(d->d_params + (zgrp ? zp.tab_off : d->conv_tab_off[l])), bs, conv_dst, c,
                  co_n, n, m, kp, conv_relu, nt, zgrp ? 1 : (co_n + 15) / 16, zgrp ? zp.zg : 1, split, N, st);
    }
    if (conv_dst == buf[cur ^ 1]) cur ^= 1;
    act_in = conv_dst;
    c = co_n;
    n = m;
    if (swap) {  // ... then the resize (and the layer's ReLU, and the final clamp when this is it)
      float* dst = layer_dst ? layer_dst : buf[cur ^ 1];
      resize(act_in, c, n, m_out, d->conv_relu[l], to_out ? clampv : 0.0f, dst);
      if (dst == buf[cur ^ 1]) cur ^= 1;
      act_in = dst;
      n = m_out;
    }
    if (is_last && act_in != out) {
      if (n != d->volume) {
        resize(act_in, 1, n, d->volume, 0, clampv, out);
      } else {
        copy_words_async(out, act_in, (size_t)N * vox, st);
        if (clampv > 0.0f)
          hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)(((size_t)N * vox + 255) / 256)), dim3(256), 0,
                             st, out, (size_t)N * vox, clampv);
      }
    }
  }
  SDFR_HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int sdfr_decoder_forward(const sdfr_decoder* d, const float* z, int N, int enforce_tsdf,
                                    float* out, float* tape, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return decoder_forward_impl(d, z, N, enforce_tsdf, out, tape, workspace, workspace_bytes, stream, 3);
}

extern "C" int sdfr_decoder_forward_stage(const sdfr_decoder* d, const float* z, int N, int enforce_tsdf,
                                          float* out, float* tape, void* workspace,
                                          size_t workspace_bytes, void* stream, int stages) {
  return decoder_forward_impl(d, z, N, enforce_tsdf, out, tape, workspace, workspace_bytes, stream, stages);
}

namespace sdfr {
bool decoder_fc_one_wave(const sdfr_decoder* dec, const FcDesc& d) {
  return dec->opt_fc_one_wave.load(std::memory_order_relaxed) != 0 && fc_one_wave_ok(d);
}
}

extern "C" size_t sdfr_decoder_backward_workspace_bytes(const sdfr_decoder* d, int N) {
  if (!d || N <= 0) return 0;
  return 2 * (size_t)N * d->max_bwd * sizeof(float) + 512;
}

namespace sdfr {
void decoder_fc_desc(const sdfr_decoder* d, FcDesc* out, const float** d_params, size_t* tape_fc_off) {
  out->n_fc = d->n_fc;
  out->width[0] = d->latent;
  for (int l = 0; l < d->n_fc; ++l) {
    out->width[l + 1] = d->fc_out[l];
    out->w_off[l] = (long long)d->fc_w_off[l];
    out->b_off[l] = (long long)d->fc_b_off[l];
  }
  if (d_params) *d_params = d->d_params;
  if (tape_fc_off) *tape_fc_off = d->tape_fc_off;
}
}  // namespace sdfr

namespace {
// t_mid_out != nullptr: the last launch (the small leading layers, fc_stack_backward_kernel) is left to the caller --
// *t_mid_out is where the gradient w.r.t. the input of the wide layer lies ([N][width], inside the workspace).
// sg != nullptr (N = 1): the incoming gradient is  grad_out + k sg->g2  (ScaledGrad), and both volumes are zero-filled
// once they have been read.
int decoder_backward_impl(const sdfr_decoder* d, const float* z, const float* tape, const float* grad_out, int N,
                          float* g_z, void* workspace, size_t workspace_bytes, void* stream,
                          const float** t_mid_out, const ScaledGrad* sg = nullptr) {
  if (!d) return fail(SDFR_E_NULL, "sdfr_decoder_backward_latent: NULL decoder");
  if (N < 0 || N > 65535) return fail(SDFR_E_INVALID, "N=%d out of range", N);
  if (N == 0) return 0;
  if (!z || !tape || !grad_out || (!g_z && !t_mid_out) || !workspace)
    return fail(SDFR_E_NULL, "sdfr_decoder_backward_latent: NULL pointer argument");
  if (workspace_bytes < sdfr_decoder_backward_workspace_bytes(d, N))
    return fail(SDFR_E_WORKSPACE, "sdfr_decoder_backward_latent: workspace %zu < %zu bytes",
                workspace_bytes, sdfr_decoder_backward_workspace_bytes(d, N));
  SDFR_HIP_TRY(hipSetDevice(d->device));
  hipStream_t st = (hipStream_t)stream;
  uintptr_t wsp = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  float* buf[2] = {(float*)wsp, (float*)wsp + (size_t)N * d->max_bwd};
  int cur = 0;
  const float* g = grad_out;  // gradient w.r.t. the current tensor, [N][c][n^3]
  const size_t vox_in = (size_t)d->volume * d->volume * d->volume;
  if (sg && (N != 1 || sg->n < 1 || sg->n > 64 || (vox_in & 3) || ((uintptr_t)grad_out & 15) || ((uintptr_t)sg->g2 & 15) ||
             !sg->g2 || !sg->cnt))
    return fail(SDFR_E_INVALID, "sdfr_decoder_backward_latent_deferred_scaled: one latent, 16-byte aligned volumes of a "
                "multiple of 4 voxels");

  // sizes of the tensor each conv layer produces
  std::vector<int> out_n(d->n_conv);
  for (int l = 0; l < d->n_conv; ++l) out_n[l] = d->conv_in_size[l] - d->conv_k[l] + 1;
  // transpose of a trilinear resize of [N*C] volumes n_out^3 -> n_in^3: z, then y, then x
  // pad >= 0: the last pass also applies the ReLU mask `act` (may be null) and the zero padding of the
  // transposed convolution that follows (resize_x_backward_pad_kernel)
  // Single latents (the captured loop) are bound by the launch count: there the z and y passes share a launch.
  // mix_w != nullptr: the x pass also applies the transposed 1x1 layer C -> mix_cout channels
  // (resize_x_backward_mix_pad_kernel; needs pad >= 0).
  // pz: floats between the z-rows of the padded tensor written (pad >= 0), >= n_in + 2 pad
  auto resize_backward = [&](int C, int n_in, int n_out, int pad = -1, const float* act = nullptr,
                             const float* mix_w = nullptr, int mix_cout = 0, int pz = 0) {
    if (pz == 0) pz = n_in + 2 * std::max(pad, 0);
    const size_t nc = (size_t)N * C;
    bool few = nc * n_out * n_out * n_out <= kFewElements;
    if (few) {  // longest source range, same arithmetic as resize_sources
      const float inv = (float)n_out / (float)n_in;
      for (int i = 0; i < n_in && few; ++i) {
        const int d0 = std::max((int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1, 0);
        const int d1 = std::min((int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1, n_out - 1);
        few = d1 - d0 + 1 <= kZyTaps;
      }
    }
    // (n_in <= n_out: the z pass writes a row's n_in results over the row's own n_out sources)
    // (single latents too: one launch of ~9 us instead of two of 6 + 5 in the captured loop, C5 0.144 -> 0.1375 ms)
    if (d->opt_tiled_vjp.load(std::memory_order_relaxed) && (!mix_w || (C == 1 && pad >= 0)) && n_in <= 64 &&
        n_in <= n_out && n_out <= 1024 && nc < (1u << 24)) {
      // the three passes in one launch on an LDS-staged block (resize3_backward_tiled_kernel)
      const float ratio = (float)n_in / (float)n_out;
      auto weight = [&](int dd, int i) {   // resize_weight on the host, same float arithmetic
        float sp = fmaf(ratio, (float)dd + 0.5f, -0.5f);
        sp = sp < 0.0f ? 0.0f : sp;
        const int i0 = std::min((int)sp, n_in - 1), i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
        const float l1 = sp - (float)i0;
        return (i0 == i ? 1.0f - l1 : 0.0f) + (i1 == i ? l1 : 0.0f);
      };
      const float inv = (float)n_out / (float)n_in;
      auto exact = [&](int i, int& d0, int& d1) {
        d0 = std::max((int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1, 0);
        d1 = std::min((int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1, n_out - 1);
        while (d0 <= d1 && weight(d0, i) == 0.0f) ++d0;
        while (d1 >= d0 && weight(d1, i) == 0.0f) --d1;
      };
      const int padv = pad >= 0 ? pad : 0;
      int max_taps = 1, max_span = 1;   // longest exact source range; longest candidate range (resize_row16: <= 16)
      for (int i = 0; i < n_in; ++i) {
        int d0, d1;
        exact(i, d0, d1);
        max_taps = std::max(max_taps, d1 - d0 + 1);
        const int s0 = std::max((int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1, 0);
        const int s1 = std::min((int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1, n_out - 1);
        max_span = std::max(max_span, s1 - s0 + 1);
      }
      // tile edge and workgroup size: the pair that stages the fewest fine rows over the launch, among those whose
      // block fits the registers of the prefetch (kBtLoads vectors per thread) and leaves two workgroups on a CU
      struct Pick { int tc = 0, threads = 0, max_f = 0, wgs = 0; size_t lds = 0; double cost = 0; } best;
      const bool aligned = (n_out & 3) == 0 && ((uintptr_t)g & 15) == 0 && max_taps <= kBtTaps && max_span <= 16;
      const void* fn = nullptr;   // the instantiation this call takes
      {
#define SDFR_BT_FN(CO) (max_taps <= 6 ? reinterpret_cast<const void*>(&resize3_backward_tiled_kernel<CO, 6>)    \
                        : max_taps <= 8 ? reinterpret_cast<const void*>(&resize3_backward_tiled_kernel<CO, 8>)  \
                                        : reinterpret_cast<const void*>(&resize3_backward_tiled_kernel<CO, 12>))
        fn = !mix_w ? SDFR_BT_FN(0) : mix_cout == 1 ? SDFR_BT_FN(1) : mix_cout == 2 ? SDFR_BT_FN(2) : mix_cout == 3 ? SDFR_BT_FN(3) : SDFR_BT_FN(4);
#undef SDFR_BT_FN
      }
      static std::map<const void*, int> vgpr_waves;   // waves per SIMD the instantiation's registers allow
      static std::mutex vgpr_mutex;
      int waves_simd = 0;
      if (aligned) {
        std::lock_guard<std::mutex> lock(vgpr_mutex);
        auto it = vgpr_waves.find(fn);
        if (it == vgpr_waves.end()) {
          // (a block above 64 KiB of dynamic LDS needs the kernel's limit raised: once per instantiation)
          hipFuncAttributes fa;
          int waves = 0;   // (0: the runtime refused; the three-launch form below is taken)
          if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
              hipFuncGetAttributes(&fa, fn) == hipSuccess)
            waves = std::max(1, std::min(8, 512 / ((std::max(fa.numRegs, 1) + 7) / 8 * 8)));
          it = vgpr_waves.emplace(fn, waves).first;
        }
        waves_simd = it->second;
      }
      for (int tc = 2; aligned && tc <= std::min(kBtMaxTile, n_in); ++tc) {
        if (SDFR_BT_TILE > 0 && tc != std::min(SDFR_BT_TILE, n_in)) continue;
        int max_f = 1;
        for (int c0 = 0; c0 < n_in; c0 += tc) {
          int a0, a1, t;
          exact(c0, a0, t);
          exact(std::min(c0 + tc, n_in) - 1, t, a1);
          max_f = std::max(max_f, a1 - a0 + 1);
        }
        const int tiles = (n_in + tc - 1) / tc;
        const size_t lds = ((size_t)max_f * max_f * n_out + (size_t)max_f * tc * n_in) * sizeof(float);
        if (lds > 150 * 1024) continue;
        for (int threads = 256; threads <= kBtMaxThreads; threads *= 2) {
          if (SDFR_BT_THREADS > 0 && threads != SDFR_BT_THREADS) continue;
          if ((size_t)max_f * max_f * (n_out >> 2) > (size_t)kBtLoads * threads) continue;
          const size_t oc = tiles == 1 ? n_in + 2 * padv : tc + padv;   // columns of the padded tensor a tile writes, per axis
          if (oc * oc * pz > (size_t)kBtOut * threads) continue;
          // workgroups a CU holds (LDS, registers), and the cost: rows staged over the launch, with a charge for a
          // thin CU (few waves to hide the chain of a workgroup behind)
          const int wgs = (int)std::min<size_t>((160 * 1024) / (lds + 6 * 1024), (size_t)(4 * waves_simd) / (threads / 64));
          if (wgs < 1) continue;
          const double waves = (double)wgs * threads / 64;
          const double cost = (double)tiles * tiles * max_f * max_f * (1.0 + 4.0 / waves);
          if (!best.tc || cost < best.cost) { best.tc = tc; best.threads = threads; best.max_f = max_f; best.lds = lds; best.wgs = wgs; best.cost = cost; }
        }
      }
      if (best.tc) {
        const int tiles = (n_in + best.tc - 1) / best.tc, tt = tiles * tiles;
        // channel slots: each workgroup walks over `per` channels (tables and addresses once, the next channel's
        // rows prefetched); `per` so that the workgroups come in whole rounds of what the chip holds
        const long long cap = 256LL * best.wgs;
        int slots = 8;
        double best_t = 0;
        for (int per = 1; per <= 32; ++per) {
          const int sl = (int)(((nc + per - 1) / per + 7) / 8 * 8);
          const long long rounds = ((long long)tt * sl + cap - 1) / cap;
          const double t = (double)rounds * (per + 0.7);   // (+ the set-up of a workgroup, in channels)
          if (per == 1 || t < best_t) { best_t = t; slots = sl; }
        }
        const float* wm = mix_w ? mix_w : nullptr;
        const float* zb = d->d_params + d->zero_bias_off;
#define SDFR_BT(CO, TAPS)                                                                                             \
  hipLaunchKernelGGL((resize3_backward_tiled_kernel<CO, TAPS>), dim3((unsigned)(tt * slots)), dim3(best.threads),     \
                     best.lds, st, g, n_in, n_out, pad >= 0 ? act : nullptr, padv, pz, best.tc, best.max_f, (int)nc,  \
                     slots, wm, zb, buf[cur]);
#define SDFR_BT_TAPS(CO) { if (max_taps <= 6) SDFR_BT(CO, 6) else if (max_taps <= 8) SDFR_BT(CO, 8) else SDFR_BT(CO, 12) }
        if (!mix_w) SDFR_BT_TAPS(0)
        else if (mix_cout == 1) SDFR_BT_TAPS(1)
        else if (mix_cout == 2) SDFR_BT_TAPS(2)
        else if (mix_cout == 3) SDFR_BT_TAPS(3)
        else SDFR_BT_TAPS(4)
#undef SDFR_BT_TAPS
#undef SDFR_BT
        g = buf[cur];
        cur ^= 1;
        return;
      }
    }
    struct Pass { size_t outer; size_t inner; } passes[3] = {
        {nc * n_out * n_out, 1},                  // z:  [nc][no][no][no] -> [nc][no][no][ni]
        {nc * n_out, (size_t)n_in},               // y:  [nc][no][no][ni] -> [nc][no][ni][ni]
        {nc, (size_t)n_in * n_in}};               // x:  [nc][no][ni][ni] -> [nc][ni][ni][ni]
    for (int a = 0; a < 3; ++a) {
      if (a == 0 && few) {
        const size_t cnt = nc * n_out * n_in * n_in;
        hipLaunchKernelGGL(resize_zy_backward_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, g,
                           nc * n_out, n_in, n_out, buf[cur]);
        a = 1;
      } else if (a == 2 && mix_w) {
        const size_t np = (size_t)n_in + 2 * pad;
        const dim3 gm((unsigned)((np * np * pz + 255) / 256), N);
        const float* zb = d->d_params + d->zero_bias_off;
#define SDFR_MIX(CO) hipLaunchKernelGGL((resize_x_backward_mix_pad_kernel<CO>), gm, dim3(256), 0, st, g, C, n_in, n_out, mix_w, zb, act, pad, pz, buf[cur])
        if (mix_cout == 1) SDFR_MIX(1); else if (mix_cout == 2) SDFR_MIX(2); else if (mix_cout == 3) SDFR_MIX(3); else SDFR_MIX(4);
#undef SDFR_MIX
      } else if (a == 2 && pad >= 0) {
        const size_t np = (size_t)n_in + 2 * pad, cnt = nc * np * np * pz;
        hipLaunchKernelGGL(resize_x_backward_pad_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, g, nc,
                           n_in, n_out, act, pad, pz, buf[cur]);
      } else {
        const size_t cnt = passes[a].outer * n_in * passes[a].inner;
        hipLaunchKernelGGL(resize_axis_backward_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st,
                           g, passes[a].outer, n_in, n_out, passes[a].inner, buf[cur]);
      }
      g = buf[cur];
      cur ^= 1;
    }
  };
  // z-row pitch of the padded tensor layer l's transposed convolution reads: rows 16 bytes apart when the direct
  // convolution takes it (its 16-byte loads and register prefetch then apply whatever the size; 34 -> 36 floats)
  const bool ws_aligned = ((uintptr_t)wsp & 15) == 0;
  auto pitch_of = [&](int l) {
    const int k = d->conv_k[l], np = out_n[l] + 2 * (k - 1);
    const bool direct = ws_aligned && !d->conv_swap[l] && !(k == 1 && d->conv_cin[l] <= 4) &&
                        direct_ok(d->bwd_direct_off[l], np, d->conv_in_size[l], N);
    return direct ? (np + 3) & ~3 : np;
  };
  bool padded = false;  // g already is layer l's padded, masked output gradient (written by the resize above)
  // few latents: a transposed resize whose output only the transposed convolution of the layer below reads is not
  // launched -- it is left pending and runs inside that convolution's launch (conv3d_mfma_tresize_kernel)
  struct {
    bool on = false;
    VjpPlan plan;
    const float *g = nullptr, *act = nullptr, *mix_w = nullptr;
    int C = 0, n_in = 0, n_out = 0, mix_cout = 0;
  } pend;
  // the scaled form: the fused first stage adds the two volumes on load (it is the launch that reads grad_out when the
  // last layer is a swapped, ReLU-free 1x1x1 one: the `mix` step of the loop below); any other decoder gets them summed
  // by a launch of its own first
  bool sg_in_stage = false;
  if (sg) {
    const int l = d->n_conv - 1;
    if (sg->n == 1 && l > 0 && out_n[l] == d->volume && d->conv_swap[l] && !d->conv_relu[l] && d->conv_k[l] == 1 && d->conv_cin[l] <= 4 &&
        !d->conv_swap[l - 1] && out_n[l - 1] == d->conv_prev[l] && d->conv_cout[l - 1] == d->conv_cin[l] &&
        d->conv_prev[l] != d->conv_in_size[l] && ((uintptr_t)sg->g2 & 15) == 0)
      sg_in_stage = vjp_stage_plan(d, l - 1, d->conv_cout[l], d->conv_prev[l], d->conv_in_size[l], d->conv_cin[l], N, 1,
                                   grad_out, true).ok;
    if (!sg_in_stage)
      hipLaunchKernelGGL(scaled_combine_kernel, dim3((unsigned)((vox_in + 255) / 256)), dim3(256), 0, st,
                         const_cast<float*>(grad_out), sg->g2, sg->n, sg->cnt, sg->weight, vox_in);
  }
  int n = d->volume;
  if (out_n[d->n_conv - 1] != d->volume) {  // final resize
    const int ni = out_n[d->n_conv - 1];
    resize_backward(1, ni, d->volume);
    n = ni;
  }
  for (int l = d->n_conv - 1; l >= 0; --l) {
    const int k = d->conv_k[l], ci_n = d->conv_cin[l], co_n = d->conv_cout[l];
    const int nin = d->conv_in_size[l], m = out_n[l];
    const bool swap = d->conv_swap[l] != 0;
    const int prev = d->conv_prev[l];
    const int np = swap ? prev : m + 2 * (k - 1);   // size of the tensor the transposed conv reads
    const int nconv = swap ? prev : nin;            // ... and of the one it produces
    const int pz = swap ? np : pitch_of(l);         // ... and the floats between its z-rows
    // 1. ReLU' and zero padding of the output gradient
    const float* act = d->conv_relu[l] ? tape + (size_t)N * d->tape_conv_off[l] : nullptr;
    bool conv_done = false;
    // (does the resize in front of THIS layer fold into the stage of the layer below?  Then this stage's epilogue runs
    // that resize's z pass on its output rows, and the stage below starts from them.)
    const bool fuse_below = !swap && prev != nin && l > 0 && !d->conv_swap[l - 1] && out_n[l - 1] == prev &&
                            d->conv_cout[l - 1] == ci_n;
    VjpPlan below;
    if (pend.on) {   // steps 1 and 2 in one launch with the transposed resize above them
      pend.on = false;
      if (fuse_below) below = vjp_stage_plan(d, l - 1, ci_n, prev, nin, 0, N, 0, nullptr, false);
      if (!launch_vjp_stage(d, pend.plan, l, pend.g, pend.C, pend.n_in, pend.n_out, pend.act, pend.mix_w, pend.mix_cout,
                            below.ok ? prev : 0, buf[cur], N, st, (sg_in_stage && pend.g == grad_out) ? sg : nullptr))
        return fail(SDFR_E_INVALID, "sdfr_decoder_backward_latent: the fused stage could not be launched");
      g = buf[cur];
      cur ^= 1;
      n = nconv;
      conv_done = true;
    } else if (padded) {
      padded = false;
    } else if (!swap || act) {
      const size_t cntp = (size_t)co_n * (swap ? m : np) * (swap ? m : np) * (swap ? m : pz);
      hipLaunchKernelGGL(pad_mask_kernel, dim3((unsigned)((cntp + 255) / 256), N), dim3(256), 0, st, g, act,
                         co_n, m, swap ? 0 : k - 1, swap ? m : pz, buf[cur]);
      g = buf[cur];
      cur ^= 1;
    }
    // (swapped 1x1 layer: forward was conv -> resize, so the resize is transposed first)
    // single latents: its last pass also runs the transposed 1x1 layer and writes the padded, masked input of
    // the transposed convolution below (conv1x1 + pad_mask launches saved)
    const bool mix = swap && !act && k == 1 && ci_n <= 4 && l > 0 && !d->conv_swap[l - 1] && out_n[l - 1] == prev &&
                     d->conv_cout[l - 1] == ci_n && prev != nin &&
                     ((size_t)N * co_n * nin * nin * nin <= kFewElements || co_n == 1);
    if (mix) {
      const float* act_below = d->conv_relu[l - 1] ? tape + (size_t)N * d->tape_conv_off[l - 1] : nullptr;
      pend.plan = vjp_stage_plan(d, l - 1, co_n, prev, nin, ci_n, N, 1, g, g == grad_out);
      if (sg && sg->n == 1 && g == grad_out && pend.plan.ok != sg_in_stage)
        return fail(SDFR_E_INVALID, "sdfr_decoder_backward_latent: the scaled form's first stage changed its mind (internal)");
      if (pend.plan.ok) {
        pend.on = true;
        pend.g = g; pend.act = act_below; pend.mix_w = d->d_params + d->bwd_w_off[l];
        pend.C = co_n; pend.n_in = prev; pend.n_out = nin; pend.mix_cout = ci_n;
        n = prev;
        continue;
      }
      resize_backward(co_n, prev, nin, d->conv_k[l - 1] - 1,
                      d->conv_relu[l - 1] ? tape + (size_t)N * d->tape_conv_off[l - 1] : nullptr,
                      d->d_params + d->bwd_w_off[l], ci_n, pitch_of(l - 1));
      padded = true;
      n = prev;
      continue;
    }
    if (swap) resize_backward(co_n, prev, nin);
    // 2. data gradient = valid conv (kernel k) of the padded tensor with the flipped weights
    const int kpad = d->bwd_kpad[l];
    if (conv_done) {
      // (launched above)
    } else if (k == 1 && ci_n <= 4) {
      // transposed 1x1 layer: element-wise, like its forward (the MFMA form: 207 us per 256 latents, this: ~35)
      const int voxn = np * np * np;
      const dim3 g1((voxn + 255) / 256, N);
      const float* wb = d->d_params + d->bwd_w_off[l];
      const float* zb = d->d_params + d->zero_bias_off;
#define SDFR_CONV1(CO) hipLaunchKernelGGL((conv1x1_kernel<CO>), g1, dim3(256), 0, st, g, wb, zb, co_n, voxn, 0, buf[cur])
      if (ci_n == 1) SDFR_CONV1(1); else if (ci_n == 2) SDFR_CONV1(2); else if (ci_n == 3) SDFR_CONV1(3); else SDFR_CONV1(4);
#undef SDFR_CONV1
    } else if (!swap && launch_direct(d, d->bwd_direct_off[l], g, d->d_params + d->zero_bias_off, buf[cur], co_n, ci_n, np,
                               pz, nconv, 0, N, st)) {
      // (batched: direct VALU convolution)
    } else if (pz != np) {
      return fail(SDFR_E_INVALID, "sdfr_decoder_backward_latent: padded rows without the direct convolution (internal)");
    } else {
      const sdfr_decoder::ZPlan& zp = d->bwd_z[l];
      const int split = use_split_k(false, (nconv * nconv * nconv + 15) / 16, (ci_n + 15) / 16, N, kpad);
      const bool zgrp = !split && zp.zg > 1 && !swap &&
                        (long long)nconv * nconv * (nconv / zp.zg) * N >= kZGroupMinRows;
      const int kp = zgrp ? zp.kpad : kpad, rows = nconv * nconv * (nconv / (zgrp ? zp.zg : 1));
      const int nt = (rows + 15) / 16;
      launch_mfma(g, d->d_params + (zgrp ? zp.w_off : d->bwd_w_off[l]),
                  reinterpret_cast<const int*>(d->d_params + (zgrp ? zp.tab_off : d->bwd_tab_off[l])),
                  d->d_params + d->zero_bias_off, buf[cur], co_n, ci_n, np, nconv, kp, 0, nt,
                  zgrp ? 1 : (ci_n + 15) / 16, zgrp ? zp.zg : 1, split, N, st);
    }
    if (!conv_done) {
      g = buf[cur];
      cur ^= 1;
    }
    n = nconv;
    // 3. the resize in front of this layer, if any
    if (!swap && prev != nin) {
      // the layer below (l - 1) produced this tensor: if its transposed convolution is of the padded kind,
      // the last resize pass writes its input directly
      const bool fuse = l > 0 && !d->conv_swap[l - 1] && out_n[l - 1] == prev && d->conv_cout[l - 1] == ci_n;
      if (fuse) pend.plan = below.ok ? below : vjp_stage_plan(d, l - 1, ci_n, prev, nin, 0, N, 1, g, g == grad_out);
      if (fuse && pend.plan.ok) {
        pend.on = true;
        pend.g = g; pend.act = d->conv_relu[l - 1] ? tape + (size_t)N * d->tape_conv_off[l - 1] : nullptr;
        pend.mix_w = nullptr;
        pend.C = ci_n; pend.n_in = prev; pend.n_out = nin; pend.mix_cout = 0;
      } else if (fuse) {
        resize_backward(ci_n, prev, nin, d->conv_k[l - 1] - 1,
                        d->conv_relu[l - 1] ? tape + (size_t)N * d->tape_conv_off[l - 1] : nullptr, nullptr, 0,
                        pitch_of(l - 1));
        padded = true;
      } else {
        resize_backward(ci_n, prev, nin);
      }
      n = prev;
    }
  }
  // g is now the gradient w.r.t. the (ReLU'd) output of the Linear stack
  FcDesc fd;
  fd.n_fc = d->n_fc;
  fd.width[0] = d->latent;
  for (int l = 0; l < d->n_fc; ++l) {
    fd.width[l + 1] = d->fc_out[l];
    fd.w_off[l] = (long long)d->fc_w_off[l];
    fd.b_off[l] = (long long)d->fc_b_off[l];
  }
  // g == buf[cur ^ 1] here (every step above ends with g = buf[cur]; cur ^= 1): the transposed wide layer
  // must write the FREE buffer -- other workgroups still read g while this one stores.
  float* t_mid = buf[cur];
  {
    const int win = fd.width[d->n_fc - 1], wout = fd.width[d->n_fc];
    const float* act_fc = tape + (size_t)N * d->tape_fc_off;
    const float* wl = d->d_params + fd.w_off[d->n_fc - 1];
    constexpr int kIb = 5, kSb = 4;
    if (N >= 32 && (wout & 3) == 0 && (((uintptr_t)wl | (uintptr_t)g | (uintptr_t)act_fc) & 15) == 0)   // batches
      hipLaunchKernelGGL((fc_last_backward_batch_kernel<kIb, kSb>), dim3((win + kIb - 1) / kIb, (N + kSb - 1) / kSb),
                         dim3(kFcBlock), 0, st, d->d_params, fd, g, act_fc, N, t_mid);
    else
      hipLaunchKernelGGL(fc_last_backward_kernel, dim3(win, N), dim3(kFcBlock), 0, st, d->d_params, fd, g, act_fc,
                         t_mid, sg ? const_cast<float*>(grad_out) : (float*)nullptr, sg ? sg->g2 : (float*)nullptr,
                         (int)(vox_in / 4), sg ? sg->n : 0);
  }
  if (t_mid_out) *t_mid_out = t_mid;
  else if (decoder_fc_one_wave(d, fd))
    hipLaunchKernelGGL(fc_stack_backward_wave_kernel, dim3(N), dim3(64), 0, st, d->d_params, fd, z, t_mid, g_z);
  else hipLaunchKernelGGL(fc_stack_backward_kernel, dim3(N), dim3(kFcBlock), 0, st, d->d_params, fd, z, t_mid, g_z);
  SDFR_HIP_TRY(hipGetLastError());
  (void)n;
  return 0;
}
}  // namespace

extern "C" int sdfr_decoder_backward_latent(const sdfr_decoder* d, const float* z, const float* tape,
                                            const float* grad_out, int N, float* g_z,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  return decoder_backward_impl(d, z, tape, grad_out, N, g_z, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int sdfr_decoder_backward_latent_deferred(const sdfr_decoder* d, const float* z, const float* tape,
                                                     const float* grad_out, void* workspace, size_t workspace_bytes,
                                                     void* stream, const float** t_mid) {
  if (!t_mid) return fail(SDFR_E_NULL, "sdfr_decoder_backward_latent_deferred: NULL pointer argument");
  return decoder_backward_impl(d, z, tape, grad_out, 1, nullptr, workspace, workspace_bytes, stream, t_mid);
}

extern "C" int sdfr_decoder_backward_latent_deferred_batch(const sdfr_decoder* d, const float* z, const float* tape,
                                                           const float* grad_out, int N, void* workspace,
                                                           size_t workspace_bytes, void* stream, const float** t_mid) {
  if (!t_mid) return fail(SDFR_E_NULL, "sdfr_decoder_backward_latent_deferred_batch: NULL pointer argument");
  return decoder_backward_impl(d, z, tape, grad_out, N, nullptr, workspace, workspace_bytes, stream, t_mid);
}

extern "C" int sdfr_decoder_backward_latent_deferred_scaled(const sdfr_decoder* d, const float* z, const float* tape,
                                                            float* grad_out, float* grad_scaled, int n_scaled,
                                                            const float* count, float weight, void* workspace,
                                                            size_t workspace_bytes, void* stream, const float** t_mid) {
  if (!t_mid || !grad_scaled || !count)
    return fail(SDFR_E_NULL, "sdfr_decoder_backward_latent_deferred_scaled: NULL pointer argument");
  const ScaledGrad sg{grad_scaled, count, weight, n_scaled};
  return decoder_backward_impl(d, z, tape, grad_out, 1, nullptr, workspace, workspace_bytes, stream, t_mid, &sg);
}
