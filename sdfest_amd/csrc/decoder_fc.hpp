// decoder_fc.hpp -- the Linear stack's description and the backward of its small leading layers as a device
// function: decoder.hip runs it as its own launch (one workgroup per sample), loop.hip inside the one-workgroup tail of
// the captured render-and-compare iteration (one launch less per iteration).
#pragma once
#include <hip/hip_runtime.h>

struct sdfr_decoder;

namespace sdfr {

constexpr int kFcBlock = 256;
constexpr int kMaxHidden = 2048;  // widest Linear layer other than the last

struct FcDesc {
  int n_fc;
  int width[9];         // width[0] = latent, width[l+1] = out of layer l
  long long w_off[8];   // float offsets into params
  long long b_off[8];
};

// the decoder's Linear stack and its parameter block (device pointer); defined in decoder.hip
void decoder_fc_desc(const sdfr_decoder* d, FcDesc* out, const float** d_params, size_t* tape_fc_off);
// whether the backward of the stack's leading layers runs as one wave (fc_one_wave_ok, and not switched off for this
// handle: sdfr_decoder_set_option, SDFR_DECODER_OPT_FC_ONE_WAVE); defined in decoder.hip
bool decoder_fc_one_wave(const sdfr_decoder* dec, const FcDesc& d);

// Backward of the small leading layers for ONE sample, by one workgroup of kFcBlock threads: t_in (gradient w.r.t.
// the input of the last layer, not yet ReLU-masked) -> g_z.  Hidden activations are recomputed in LDS (80 KB static).
// Ends with the results stored but no barrier after them.  Narrow stacks take fc_stack_backward_one_wave below.
__device__ __forceinline__ void fc_stack_backward_sample(const float* __restrict__ params, const FcDesc& d,
                                                         const float* __restrict__ z, const float* __restrict__ t_in,
                                                         float* __restrict__ g_z) {
  __shared__ float act[8][kMaxHidden];   // act[l] = input of layer l (act[0] = z)
  __shared__ float gbuf[2][kMaxHidden];
  const int tid = threadIdx.x;
  for (int i = tid; i < d.width[0]; i += kFcBlock) act[0][i] = z[i];
  __syncthreads();
  for (int l = 0; l < d.n_fc - 1; ++l) {
    const int win = d.width[l], wout = d.width[l + 1];
    const float* w = params + d.w_off[l];
    const float* b = params + d.b_off[l];
    for (int o = tid; o < wout; o += kFcBlock) {
      float acc = b[o];
      for (int i = 0; i < win; ++i) acc = fmaf(w[(size_t)o * win + i], act[l][i], acc);
      act[l + 1][o] = fmaxf(acc, 0.0f);
    }
    __syncthreads();
  }
  int cur = 0;
  {
    const int l = d.n_fc - 1, win = d.width[l];
    // ReLU' of the layer that produced act[l] (l >= 1); the latent itself has no ReLU
    for (int i = tid; i < win; i += kFcBlock) {
      const float t = t_in[i];
      gbuf[cur][i] = (l == 0 || act[l][i] > 0.0f) ? t : 0.0f;
    }
    __syncthreads();
  }
  for (int l = d.n_fc - 2; l >= 0; --l) {
    const int win = d.width[l], wout = d.width[l + 1];
    const float* w = params + d.w_off[l];  // [out][in]
    for (int i = tid; i < win; i += kFcBlock) {
      float t = 0.0f;
      for (int o = 0; o < wout; ++o) t = fmaf(w[(size_t)o * win + i], gbuf[cur][o], t);
      gbuf[cur ^ 1][i] = (l == 0 || act[l][i] > 0.0f) ? t : 0.0f;
    }
    __syncthreads();
    cur ^= 1;
  }
  for (int i = tid; i < d.width[0]; i += kFcBlock) g_z[i] = gbuf[cur][i];
}

// The same by ONE wave, for stacks whose leading layers are narrow (the mug decoder: 8 -> 20 -> 50 -> 8192).  The
// function above is a chain of dependent global loads there -- a runtime-length loop of weight loads per layer and
// direction, 20 of 256 threads at work: 8.0 us of the captured loop's 14.7 us tail (tools/microbench/tail_stamps.py:
// 1.0 + 1.8 us for the two recomputed layers, 3.0 + 1.3 us for their transposes).  Here the wave first copies all
// weights and biases of these layers to LDS with independent loads (one round trip), then runs the same fmaf chains --
// same operands, same order, bit-identical results -- out of LDS, with no workgroup barrier: the other waves of the
// workgroup are free to do something else meanwhile (loop.hip's tail: the per-view reductions).  In the tail: 8.0 ->
// 4.7 us (1.7 us the staging round trip, 0.8 us the two recomputed layers, 1.5 us their transposes: LDS-latency chains;
// a lane's weights preloaded into a 64-register array, fully unrolled and predicated, made them 4.7 + 6.4 us).
constexpr int kFcWaveWidth = 64;       // widest layer input the one-wave form takes
constexpr int kFcWaveSpan = 6144;      // floats of the parameter block it stages (weights, biases, alignment gaps)
// The leading layers' weights and biases lie one after the other in the parameter block (sdfr_decoder_create); the
// wave copies that whole span.  span_of: its length, or -1 if some layer's parameters lie outside of it.
__host__ __device__ inline long long fc_wave_span(const FcDesc& d) {
  const int last = d.n_fc - 1;
  if (last < 1) return 0;
  const long long base = d.w_off[0], span = d.b_off[last - 1] + d.width[last] - base;
  for (int l = 0; l < last; ++l) {
    const long long w0 = d.w_off[l] - base, b0 = d.b_off[l] - base;
    if (w0 < 0 || w0 + (long long)d.width[l] * d.width[l + 1] > span || b0 < 0 || b0 + d.width[l + 1] > span) return -1;
  }
  return span;
}
__host__ __device__ inline bool fc_one_wave_ok(const FcDesc& d) {
  if (d.n_fc < 1 || d.n_fc > 8) return false;
  for (int l = 0; l < d.n_fc; ++l)
    if (d.width[l] > kFcWaveWidth) return false;
  const long long span = fc_wave_span(d);
  return span >= 0 && span <= kFcWaveSpan;
}

// n floats global -> LDS by one wave: all of a lane's loads (up to 24) are in flight before the first is stored
__device__ __forceinline__ void wave_stage(float* dst, const float* __restrict__ src, int n, int lane) {
  constexpr int U = 24;
  for (int e0 = 0; e0 < n; e0 += 64 * U) {
    float r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * 64 + lane;
      r[u] = e < n ? src[e] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * 64 + lane;
      if (e < n) dst[e] = r[u];
    }
  }
}

// The one-wave forms' LDS: the span of the parameter block at its own offsets, the layers' inputs, the gradients.
struct FcWaveLds {
  float p[kFcWaveSpan];
  float a[8][kFcWaveWidth];   // a[l] = input of layer l (a[0] = the latent)
  float g[2][kFcWaveWidth];
};
// The leading layers' forward by one wave out of LDS: a[0] (the latent) -> a[1] .. a[last]; L.p staged (wave_stage).
// The fmaf chains of fc_stack_kernel / fc_stack_backward_sample: bias first, inputs in ascending order.
__device__ __forceinline__ void fc_narrow_forward_one_wave(FcWaveLds& L, const FcDesc& d, int lane) {
  const int last = d.n_fc - 1;
  const long long base = d.w_off[0];
  for (int l = 0; l < last; ++l) {
    const int win = d.width[l], wout = d.width[l + 1];
    if (lane < wout) {
      float acc = L.p[d.b_off[l] - base + lane];
      const float* w = L.p + (d.w_off[l] - base) + lane * win;
#pragma unroll 8
      for (int i = 0; i < win; ++i) acc = fmaf(w[i], L.a[l][i], acc);
      L.a[l + 1][lane] = fmaxf(acc, 0.0f);
    }
    __builtin_amdgcn_wave_barrier();
  }
}
// 256 outputs of the LAST (wide) layer by a workgroup of kFcBlock threads, from its input a_last in LDS (every thread
// calls it behind a barrier): fc_stack_kernel's expression
__device__ __forceinline__ void fc_wide_slice(const float* __restrict__ params, const FcDesc& d, const float* a_last,
                                              int slice, float* __restrict__ out) {
  const int l = d.n_fc - 1;
  const int win = d.width[l], wout = d.width[l + 1];
  const int o = slice * kFcBlock + (int)threadIdx.x;
  if (o < wout) {
    const float* wt = params + d.w_off[l];  // transposed: [in][out]
    float acc = params[d.b_off[l] + o];
#pragma unroll 16
    for (int i = 0; i < win; ++i) acc = fmaf(wt[(size_t)i * wout + o], a_last[i], acc);
    out[o] = fmaxf(acc, 0.0f);
  }
}

// Called by the 64 lanes of one wave (lane = 0 .. 63); requires fc_one_wave_ok(d).  Results stored, nothing after.
__device__ __forceinline__ void fc_stack_backward_one_wave(FcWaveLds& L, const float* __restrict__ params,
                                                           const FcDesc& d, const float* __restrict__ z,
                                                           const float* __restrict__ t_in, float* g_z, int lane) {
  const int last = d.n_fc - 1;
  const long long base = d.w_off[0];
#ifdef SDFR_TAIL_STAMPS
  unsigned long long fcs[6];
  fcs[0] = wall_clock64();
#endif
  const float t_last = lane < d.width[last] ? t_in[lane] : 0.0f;
  const float z_lane = lane < d.width[0] ? z[lane] : 0.0f;
  wave_stage(L.p, params + base, (int)fc_wave_span(d), lane);
  if (lane < d.width[0]) L.a[0][lane] = z_lane;
  __builtin_amdgcn_wave_barrier();
#ifdef SDFR_TAIL_STAMPS
  fcs[1] = wall_clock64();
#endif
  fc_narrow_forward_one_wave(L, d, lane);
  int cur = 0;
#ifdef SDFR_TAIL_STAMPS
  fcs[2] = wall_clock64();
#endif
  if (lane < d.width[last]) L.g[cur][lane] = (last == 0 || L.a[last][lane] > 0.0f) ? t_last : 0.0f;
  __builtin_amdgcn_wave_barrier();
  for (int l = last - 1; l >= 0; --l) {
    const int win = d.width[l], wout = d.width[l + 1];
    if (lane < win) {
      const float* w = L.p + (d.w_off[l] - base) + lane;
      float t = 0.0f;
#pragma unroll 8
      for (int o = 0; o < wout; ++o) t = fmaf(w[o * win], L.g[cur][o], t);
      L.g[cur ^ 1][lane] = (l == 0 || L.a[l][lane] > 0.0f) ? t : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    cur ^= 1;
  }
  if (lane < d.width[0]) g_z[lane] = L.g[cur][lane];
#ifdef SDFR_TAIL_STAMPS
  fcs[3] = wall_clock64();
  if (lane == 0) printf("fc one wave: stage %.2f  forward %.2f  backward %.2f us\n", (double)(fcs[1] - fcs[0]) * 0.01,
                        (double)(fcs[2] - fcs[1]) * 0.01, (double)(fcs[3] - fcs[2]) * 0.01);
#endif
}

}  // namespace sdfr
