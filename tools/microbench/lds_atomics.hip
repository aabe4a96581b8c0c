#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// microbenchmark: LDS atomic throughput by op type and address pattern
template <int OP, int PAT>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
  __shared__ float buf[8192];
  __shared__ unsigned long long buf64[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) buf[i] = 0.f;
  for (int i = tid; i < 4096; i += 256) buf64[i] = 0ull;
  __syncthreads();
  unsigned idx;
  if (PAT == 0) idx = tid;                              // unique, conflict-free
  else if (PAT == 1) idx = (tid * 2654435761u) >> 19;   // random in 8192
  else if (PAT == 2) idx = (tid >> 3) * 33;             // 8 lanes share an address
  else idx = (tid >> 6);                                // whole wave same address
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    unsigned a = (idx + it * 97) & 4095;
    if (OP == 0) atomicAdd(&buf[a], 1.0f + lane);
    else if (OP == 1) atomicAdd((unsigned*)&buf[a], 1u + lane);
    else if (OP == 2) atomicAdd(&buf64[a], (unsigned long long)(1 + lane));
    else if (OP == 3) acc += (float)atomicCAS((int*)&buf[a], 0, lane);
    else if (OP == 4) { buf[a] += 1.0f; }               // plain RMW (racy) for reference
    else if (OP == 5) acc += atomicAdd(&buf[a], 1.0f);  // returning float add
  }
  __syncthreads();
  out[blockIdx.x * 256 + tid] = buf[tid] + acc + (float)buf64[tid & 4095];
}
template <int OP, int PAT> void run(const char* name, float* out) {
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(256), 0, 0, 10, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(256), 0, 0, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // 4 blocks/CU-ish: total wave-instrs = blocks*4*iters; per CU = that/256
  double per_cu = (double)blocks * 4 * iters / 256.0;
  double cyc = ms * 1e-3 * 2.3e9 / per_cu;
  printf("%-28s pattern %d: %8.3f ms  -> %6.1f cycles per wave-instruction per CU\n", name, PAT, ms, cyc);
}
int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
#define ALLPAT(OP, NAME) run<OP,0>(NAME,out); run<OP,1>(NAME,out); run<OP,2>(NAME,out); run<OP,3>(NAME,out);
  ALLPAT(0, "ds_add_f32 (no return)")
  ALLPAT(1, "ds_add_u32 (no return)")
  ALLPAT(2, "ds_add_u64 (no return)")
  ALLPAT(3, "ds_cmpst_rtn_b32")
  ALLPAT(4, "plain read+add+write")
  ALLPAT(5, "ds_add_rtn_f32")
  return 0;
}
