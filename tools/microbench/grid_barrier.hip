// grid_barrier.hip -- what a barrier across the workgroups of ONE launch costs on MI355X: the
// price a single-launch form of the N = 1 decoder (8 dependent launches, 42 us in the captured loop) would pay per
// layer boundary instead of a launch boundary (~5 us each).
//
//   hipcc -O3 --offload-arch=gfx950 grid_barrier.hip -o grid_barrier && ./grid_barrier
//
// A round = every workgroup writes `words` floats of a shared array (its slice), crosses the barrier, and reads a
// slice another workgroup wrote this round (checked: a stale read is counted) -- the hand-off a layer boundary
// needs.  Barrier = one atomic counter per round-parity + a spin on it; variants differ in the fences around it:
//   agent   release fence (agent scope: L2 write-back on gfx950, the L2s of the 8 XCDs are not coherent) before the
//           arrive, acquire fence (L1 + L2 invalidate) after the wait: correct for workgroups anywhere on the chip
//   xcd     for workgroups on ONE XCD (they share the L2): stores are waited for (vmcnt(0)) and reach the shared L2
//           -- the vector L1 is write-through --, the readers bypass their L1 with agent-scope relaxed loads (sc1):
//           no L2 write-back, no invalidate.  Workgroups are kept on one XCD by launching 8 x G of them and letting
//           only those with blockIdx.x % 8 == 0 work (round-robin placement of consecutive ids over the XCDs).
//   coh     anywhere on the chip, NO fences: the hand-off data is written with agent-scope relaxed atomic stores
//           (global_store ... sc1: written through the writer's L2) and read with agent-scope relaxed atomic loads
//           (sc1: not served from a possibly stale line of the reader's L1 / L2); the barrier waits for the stores
//           (vmcnt(0)) before it arrives.  Only the hand-off data pays; nothing is written back or invalidated.
// Every participating workgroup must be resident at the same time (G <= CUs, one workgroup per CU here).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kCounterWords = 128 + 32 * 17;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// bounded: a wave that gives up says so (g_gave_up) and goes on -- the grid always drains
__device__ unsigned g_gave_up;
__device__ __forceinline__ void spin(unsigned* counter, unsigned target) {
  for (int poll = 0; poll < (1 << 22); ++poll) {
    if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
    __builtin_amdgcn_s_sleep(1);
  }
  atomicAdd(&g_gave_up, 1u);
}

__device__ __forceinline__ void barrier_agent(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin(counter, target);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// one XCD: no cache maintenance at all; the data itself is read with L1-bypassing loads by the caller
__device__ __forceinline__ void barrier_xcd(unsigned* counter, unsigned target) {
  __builtin_amdgcn_s_waitcnt(0);   // this thread's stores have left for the L2 (vmcnt(0), and everything else)
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin(counter, target);
  }
  __syncthreads();
}

// Two-level arrival: 256 adds on ONE word serialise at the memory side (~13 ns each: the 4.3 us of the flat form at
// G = 256); here groups of 16 workgroups arrive on their own word (own 128-byte line) and the last of a group arrives
// on the top word, which everybody polls.  ACQ: an agent-scope acquire fence (buffer_inv sc1: L1 + stale L2 lines)
// after the wait, so that the phase after the barrier may use plain, cached loads.
template <bool ACQ>
__device__ __forceinline__ void barrier_tree(unsigned* counters, int wg, int G, unsigned gen) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int grp = wg >> 4, n_grp = (G + 15) >> 4, in_grp = min(16, G - (grp << 4));
    const unsigned old = __hip_atomic_fetch_add(&counters[32 * (1 + grp)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == (unsigned)in_grp * gen)
      __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin(&counters[0], (unsigned)n_grp * gen);
    if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// MODE 0: agent fences, plain loads after the barrier.  MODE 1: one-XCD form, sc1 loads.  MODE 2: no barrier at all
// (the work of a round alone: subtract).  MODE 3: coherent stores and loads, no fences, any placement.
// MODE 4: coherent stores, two-level barrier with an acquire fence, plain loads.  MODE 5: coherent stores and loads,
// two-level barrier, no fence.
template <int MODE>
__global__ __launch_bounds__(256) void rounds_kernel(float* __restrict__ data, unsigned* __restrict__ counters,
                                                     int G, int stride_ids, int rounds, int words,
                                                     unsigned* __restrict__ stale) {
  if ((int)blockIdx.x % stride_ids != 0) return;
  const int wg = (int)blockIdx.x / stride_ids;
  unsigned* tree = counters + 128;   // [0]: top word, [32 * (1 + g)]: group g
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    const float tag = (float)(r + 1);
    for (int i = threadIdx.x; i < words; i += 256) {
      if (MODE >= 3) __hip_atomic_store(&data[(size_t)wg * words + i], tag + (float)wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else data[(size_t)wg * words + i] = tag + (float)wg;
    }
    if (MODE == 0) barrier_agent(&counters[r & 1 ? 32 : 0], (unsigned)G * (unsigned)(r / 2 + 1));
    if (MODE == 1 || MODE == 3) barrier_xcd(&counters[r & 1 ? 32 : 0], (unsigned)G * (unsigned)(r / 2 + 1));
    if (MODE == 4) barrier_tree<true>(tree, wg, G, (unsigned)(2 * r + 1));
    if (MODE == 5) barrier_tree<false>(tree, wg, G, (unsigned)(2 * r + 1));
    const int other = (wg + 1 + r % (G > 1 ? G - 1 : 1)) % G;
    for (int i = threadIdx.x; i < words; i += 256) {
      const float* p = &data[(size_t)other * words + i];
      const float v = (MODE == 1 || MODE == 3 || MODE == 5) ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
      if (MODE != 2 ? v != tag + (float)other : v == -1.0f) ++bad;   // (MODE 2: keeps the loads alive, never true)
    }
    // a second barrier so that nobody overwrites a slice that is still being read (as a layer chain would need)
    if (MODE == 0) barrier_agent(&counters[r & 1 ? 96 : 64], (unsigned)G * (unsigned)(r / 2 + 1));
    if (MODE == 1 || MODE == 3) barrier_xcd(&counters[r & 1 ? 96 : 64], (unsigned)G * (unsigned)(r / 2 + 1));
    if (MODE == 4) barrier_tree<true>(tree, wg, G, (unsigned)(2 * r + 2));
    if (MODE == 5) barrier_tree<false>(tree, wg, G, (unsigned)(2 * r + 2));
  }
  if (bad) atomicAdd(stale, bad);
}

template <int MODE>
static double run(int G, int stride_ids, int rounds, int words, float* data, unsigned* counters, unsigned* stale,
                  unsigned* h_stale) {
  CHECK(hipMemset(counters, 0, kCounterWords * sizeof(unsigned)));
  CHECK(hipMemset(stale, 0, sizeof(unsigned)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(rounds_kernel<MODE>, dim3(G * stride_ids), dim3(256), 0, 0, data, counters, G, stride_ids, 2, words, stale);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(counters, 0, kCounterWords * sizeof(unsigned)));
  CHECK(hipMemset(stale, 0, sizeof(unsigned)));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(rounds_kernel<MODE>, dim3(G * stride_ids), dim3(256), 0, 0, data, counters, G, stride_ids, rounds, words, stale);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(h_stale, stale, sizeof(unsigned), hipMemcpyDeviceToHost));
  return ms * 1e3 / rounds;   // us per round (= two barriers + the slice write / read)
}

int main() {
  const int rounds = 2000;
  float* data; unsigned* counters; unsigned* stale;
  CHECK(hipMalloc(&data, (size_t)256 * 65536 * sizeof(float)));
  CHECK(hipMalloc(&counters, kCounterWords * sizeof(unsigned)));
  CHECK(hipMalloc(&stale, sizeof(unsigned)));
  unsigned h = 0;
  printf("us per ROUND (slice write, barrier, read of another workgroup's slice, barrier); stale = wrong values read\n");
  for (int words : {256, 16384}) {
    for (int G : {256, 128, 64, 32, 8}) {
      const double w = run<2>(G, 1, rounds, words, data, counters, stale, &h);
      const double a = run<0>(G, 1, rounds, words, data, counters, stale, &h);
      const unsigned ha = h;
      const double c = run<3>(G, 1, rounds, words, data, counters, stale, &h);
      const unsigned hc = h;
      const double t4 = run<4>(G, 1, rounds, words, data, counters, stale, &h);
      const unsigned h4 = h;
      const double t5 = run<5>(G, 1, rounds, words, data, counters, stale, &h);
      printf("anywhere on the chip, G=%3d, %5d floats/wg: agent fences %7.2f us/round (work alone %5.2f) -> %5.2f us per barrier, stale %u; "
             "coherent stores + loads, no fences %7.2f -> %5.2f, stale %u; two-level + coherent stores + acquire, plain loads %7.2f -> %5.2f, stale %u; "
             "two-level + coherent stores + loads %7.2f -> %5.2f, stale %u\n",
             G, words, a, w, (a - w) / 2, ha, c, (c - w) / 2, hc, t4, (t4 - w) / 2, h4, t5, (t5 - w) / 2, h);
    }
    for (int G : {32, 16, 8}) {
      const double w = run<2>(G, 8, rounds, words, data, counters, stale, &h);
      const double a = run<0>(G, 8, rounds, words, data, counters, stale, &h);
      const unsigned ha = h;
      const double x = run<1>(G, 8, rounds, words, data, counters, stale, &h);
      printf("one XCD (ids = 0 mod 8), G=%3d, %5d floats/wg: agent fences %7.2f (stale %u), no cache maintenance + sc1 loads %7.2f (stale %u), "
             "work alone %5.2f -> %5.2f / %5.2f us per barrier\n", G, words, a, ha, x, h, w, (a - w) / 2, (x - w) / 2);
    }
  }
  unsigned gave_up = 0;
  CHECK(hipMemcpyFromSymbol(&gave_up, HIP_SYMBOL(g_gave_up), sizeof(unsigned)));
  printf("waits given up: %u\n", gave_up);
  return 0;
}
