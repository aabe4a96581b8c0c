#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// microbenchmark: cost of a 16-byte-per-lane gather as a function of distinct 64B chunks per wave
template <int W>  // bytes per lane: 4, 8, 16
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, const int* __restrict__ idx, int iters, float* out, int tabmask) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  int i = idx[tid & 65535];
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (W == 16) { float4 v = tab[i]; acc += v.x + v.y + v.z + v.w; i = (i + (int)(v.x)) & tabmask; }
    else if (W == 8) { float2 v = *reinterpret_cast<const float2*>(&tab[i]); acc += v.x + v.y; i = (i + (int)(v.x)) & tabmask; }
    else { float v = *reinterpret_cast<const float*>(&tab[i]); acc += v; i = (i + (int)v) & tabmask; }
  }
  out[tid] = acc;
}
int main() {
  const int tabn = 1 << 16;  // 1 MiB of float4 (L2-resident)
  std::vector<float4> h(tabn);
  for (int i = 0; i < tabn; ++i) h[i] = make_float4(4096.f, 0, 0, 0);   // stride: keeps pattern while moving
  float4* tab; hipMalloc(&tab, tabn * 16); hipMemcpy(tab, h.data(), tabn * 16, hipMemcpyHostToDevice);
  int* idx; hipMalloc(&idx, 65536 * 4);
  float* out; hipMalloc(&out, 256 * 2048 * 4);
  // (b) same number of 64-byte chunks, but paired into 128-byte lines: is the unit of cost the
  //     64-byte chunk or the 128-byte line?
  for (int lines : {8, 16, 32}) {
    std::vector<int> hi(65536);
    for (int t = 0; t < 65536; ++t) {
      int lane = t & 63, wave = t >> 6;
      int l = lane % lines;                     // 128-byte line id within the wave
      int within = (lane / lines) & 7;          // float4 within the line (both 64-byte halves)
      int base = (wave * 977) & (tabn / 8 - 1);
      hi[t] = (((base + l * 37) & (tabn / 8 - 1)) * 8 + within) & (tabn - 1);
    }
    hipMemcpy(idx, hi.data(), 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 400, blocks = 2048;
    hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, tab, idx, 10, out, tabn - 1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, tab, idx, iters, out, tabn - 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_cu = (double)blocks * 4 * iters / 256.0;
    printf("16 B/lane, %2d 128-byte lines/wave (= up to %2d chunks): %7.3f ms -> %6.1f cycles per wave-load per CU\n",
           lines, 2 * lines, ms, ms * 1e-3 * 2.2e9 / wave_instr_per_cu);
  }
  const int patterns[] = {1, 4, 8, 16, 32, 64};
  for (int wbytes : {16, 8, 4}) for (int chunks : patterns) {
    // per wave: 64 lanes spread over `chunks` distinct 64-byte chunks (4 float4 each)
    std::vector<int> hi(65536);
    for (int t = 0; t < 65536; ++t) {
      int lane = t & 63, wave = t >> 6;
      int c = lane % chunks;                    // chunk id within wave
      int within = (lane / chunks) & 3;         // float4 within the chunk
      int base = (wave * 977) & (tabn / 4 - 1);
      hi[t] = (((base + c * 37) & (tabn / 4 - 1)) * 4 + within) & (tabn - 1);
    }
    hipMemcpy(idx, hi.data(), 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 400, blocks = 2048;
    auto launch = [&](int it) {
      if (wbytes == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, tab, idx, it, out, tabn - 1);
      else if (wbytes == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, tab, idx, it, out, tabn - 1);
      else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, tab, idx, it, out, tabn - 1);
    };
    launch(10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_cu = (double)blocks * 4 * iters / 256.0;
    printf("%2d B/lane, %2d chunks/wave: %7.3f ms -> %6.1f cycles per wave-load per CU (2.2 GHz)\n", wbytes, chunks, ms, ms * 1e-3 * 2.2e9 / wave_instr_per_cu);
  }
  return 0;
}
