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

// Backward of the small leading layers for ONE sample, by one workgroup of kFcBlock threads: t_in (gradient w.r.t.
// the input of the last layer, not yet ReLU-masked) -> g_z.  Hidden activations are recomputed in LDS (80 KB static).
// Ends with the results stored but no barrier after them.  (Staging these layers' weights in LDS up front -- one round
// trip instead of one per layer -- was measured in the captured loop: no gain, 13.9 against 14.2 us for the tail.)
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

}  // namespace sdfr
