#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// microbenchmark (round 3): what does a gather wave-instruction cost as a function of its ACTIVE LANES and of the
// bytes per lane?  (The forward march turned out to be bound by the number of gather instructions it issues --
// fewer VALU instructions, fewer distinct chunks and more loads in flight per wave all left its time unchanged --
// so what a partially filled wave costs decides whether ray compaction can pay.)
// Each lane walks its own dependent chain through an L2-resident table; lanes >= `active` do not load.
template <int W>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, const int* __restrict__ idx, int iters, float* out,
                                         int tabmask, int active) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  int i = idx[tid & 65535];
  float acc = 0.f;
  if ((threadIdx.x & 63) < active) {
    for (int it = 0; it < iters; ++it) {
      if (W == 16) { float4 v = tab[i]; acc += v.x + v.y + v.z + v.w; i = (i + (int)(v.x)) & tabmask; }
      else if (W == 8) { float2 v = *reinterpret_cast<const float2*>(&tab[i]); acc += v.x + v.y; i = (i + (int)(v.x)) & tabmask; }
      else { float v = *reinterpret_cast<const float*>(&tab[i]); acc += v; i = (i + (int)v) & tabmask; }
    }
  }
  out[tid] = acc;
}
int main() {
  const int tabn = 1 << 16;
  std::vector<float4> h(tabn);
  for (int i = 0; i < tabn; ++i) h[i] = make_float4(4096.f, 0, 0, 0);
  float4* tab; hipMalloc(&tab, tabn * 16); hipMemcpy(tab, h.data(), tabn * 16, hipMemcpyHostToDevice);
  int* idx; hipMalloc(&idx, 65536 * 4);
  float* out; hipMalloc(&out, 256 * 2048 * 4);
  for (int wbytes : {16, 8, 4}) for (int spread : {0, 1}) for (int active : {64, 48, 32, 16, 8, 4}) {
    // spread 0: the active lanes share 64-byte chunks four by four (march-like: ~16 chunks per full wave);
    // spread 1: every active lane in a chunk of its own
    std::vector<int> hi(65536);
    for (int t = 0; t < 65536; ++t) {
      int lane = t & 63, wave = t >> 6;
      int c = spread ? lane : lane / 4, within = spread ? 0 : lane & 3;
      int base = (wave * 977) & (tabn / 4 - 1);
      hi[t] = (((base + c * 37) & (tabn / 4 - 1)) * 4 + within) & (tabn - 1);
    }
    hipMemcpy(idx, hi.data(), 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 400, blocks = 2048;
    auto launch = [&](int it) {
      if (wbytes == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, tab, idx, it, out, tabn - 1, active);
      else if (wbytes == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, tab, idx, it, out, tabn - 1, active);
      else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, tab, idx, it, out, tabn - 1, active);
    };
    launch(10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_cu = (double)blocks * 4 * iters / 256.0;
    printf("%2d B/lane, %s, %2d active lanes: %7.3f ms -> %6.1f cycles per wave-load per CU (2.35 GHz)\n", wbytes,
           spread ? "one chunk per lane " : "4 lanes per chunk  ", active, ms, ms * 1e-3 * 2.35e9 / wave_instr_per_cu);
  }
  return 0;
}
