// atomic_scope.hip -- float atomic adds into a 64^3 volume (the d/dSDF accumulation of the renderer's and the sampler's
// backward) on MI355X: where they execute and what that costs.
//
//   hipcc -O3 --offload-arch=gfx950 atomic_scope.hip -o atomic_scope && ./atomic_scope
//
//   agent     atomicAdd as HIP emits it (relaxed, agent scope: global_atomic_add_f32 ... sc1): the eight XCDs' L2s are
//             not coherent with one another, so the add is forwarded to the memory side (TCC_EA0_ATOMIC counts it)
//   xcd       one private copy of the volume per XCD (index = the hardware's XCC_ID register, not a guess from the
//             workgroup id), workgroup-scope adds (no sc1): every workgroup that adds to a copy runs on the XCD whose
//             L2 holds it, so the L2 is their point of coherence and the add completes there; a second launch sums the
//             eight copies (the kernel boundary writes the L2s back).
// Each thread adds `per_thread` values at pseudo-random voxels (uniform: no locality at all) or inside a small
// neighbourhood per workgroup (the sampler's pattern: a workgroup's points share cells).  The total of the volume is
// checked against the number of adds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kVox = 64 * 64 * 64;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}
__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// MODE 0: agent scope, one volume.  MODE 1: workgroup scope, volume[xcc_id].  LOCAL: addresses within 4096 voxels
// of a per-workgroup base.
template <int MODE, bool LOCAL>
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ vol, int per_thread, unsigned* __restrict__ xcc_seen) {
  float* v = vol;
  if (MODE == 1) {
    const unsigned x = xcc_id();
    v = vol + (size_t)x * kVox;
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1u << x);
  }
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  const unsigned base = LOCAL ? hash(blockIdx.x * 977u + 13u) % (kVox - 4096) : 0u;
  for (int k = 0; k < per_thread; ++k) {
    const unsigned h = hash(gid * 131u + (unsigned)k * 2654435761u);
    const unsigned a = LOCAL ? base + (h & 4095u) : h % kVox;
    if (MODE == 0) __hip_atomic_fetch_add(&v[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(&v[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

__global__ __launch_bounds__(256) void sum_copies_kernel(const float* __restrict__ copies, int n_copies, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kVox) return;
  float s = 0.0f;
  for (int c = 0; c < n_copies; ++c) s += copies[(size_t)c * kVox + i];
  out[i] = s;
}

template <int MODE, bool LOCAL>
static void run(const char* name, float* vol, float* out, unsigned* seen, int wgs, int per_thread) {
  const int copies = MODE == 1 ? 8 : 1;
  hipEvent_t e0, e1, e2;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  float best = 1e9f, best_sum = 0.0f;
  double total = 0.0;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipMemset(vol, 0, (size_t)copies * kVox * sizeof(float)));
    CHECK(hipMemset(seen, 0, sizeof(unsigned)));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((add_kernel<MODE, LOCAL>), dim3(wgs), dim3(256), 0, 0, vol, per_thread, seen);
    CHECK(hipEventRecord(e1));
    if (MODE == 1) hipLaunchKernelGGL(sum_copies_kernel, dim3(kVox / 256), dim3(256), 0, 0, vol, copies, out);
    CHECK(hipEventRecord(e2));
    CHECK(hipDeviceSynchronize());
    float ms = 0, ms2 = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventElapsedTime(&ms2, e1, e2));
    if (ms < best) { best = ms; best_sum = ms2; }
    std::vector<float> h(kVox);
    CHECK(hipMemcpy(h.data(), MODE == 1 ? out : vol, kVox * sizeof(float), hipMemcpyDeviceToHost));
    total = 0.0;
    for (float x : h) total += x;
  }
  unsigned hs = 0;
  CHECK(hipMemcpy(&hs, seen, sizeof(unsigned), hipMemcpyDeviceToHost));
  const double n = (double)wgs * 256 * per_thread;
  printf("%-28s %9.0f adds: %7.1f us (%5.2f adds/ns)%s  total %.0f (%s)", name, n, best * 1e3, n / (best * 1e6),
         MODE == 1 ? "" : "", total, total == n ? "exact" : "WRONG");
  if (MODE == 1) printf("  + sum of 8 copies %.1f us, XCC ids seen 0x%x", best_sum * 1e3, hs);
  printf("\n");
}

int main() {
  float* vol; float* out; unsigned* seen;
  CHECK(hipMalloc(&vol, (size_t)8 * kVox * sizeof(float)));
  CHECK(hipMalloc(&out, (size_t)kVox * sizeof(float)));
  CHECK(hipMalloc(&seen, sizeof(unsigned)));
  for (int wgs : {1024, 4096}) {
    for (int per : {1, 4}) {
      run<0, false>("agent scope, uniform", vol, out, seen, wgs, per);
      run<1, false>("per-XCD copies, uniform", vol, out, seen, wgs, per);
      run<0, true>("agent scope, local", vol, out, seen, wgs, per);
      run<1, true>("per-XCD copies, local", vol, out, seen, wgs, per);
    }
  }
  return 0;
}
