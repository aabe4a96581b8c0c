// lds_march.hip -- how fast is the sphere-tracing march when the WHOLE grid sits in LDS?  
//
// A 64^3 grid (1 MiB) cannot be staged in the 160 KiB LDS, and a frustum brick per tile costs more to stage
// than the march reads.  This measures the upper bound any such scheme could reach: a 16^3
// grid (16 KiB) held completely in LDS by every workgroup, against the same march reading the same grid through
// the L1 (plain z-pair loads, and the product's 16-byte face records).  Same rays, same arithmetic, same results.
//   hipcc -O3 --offload-arch=gfx950 lds_march.hip -o lds_march && ./lds_march
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int R = 16, W = 640, H = 480, B = 256;
struct View { float rot[9], e[3], og[3], scale, isc; };

__device__ __forceinline__ float sample_lds(const float* g, float gx, float gy, float gz) {
  const float top = (float)(R - 2);
  const float bx = fminf(fmaxf(floorf(gx), 0.f), top), by = fminf(fmaxf(floorf(gy), 0.f), top), bz = fminf(fmaxf(floorf(gz), 0.f), top);
  const float ox = gx - bx, oy = gy - by, oz = gz - bz, ax = 1.f - ox, ay = 1.f - oy, az = 1.f - oz;
  const float* p = g + (int)fmaf(fmaf(bx, (float)R, by), (float)R, bz);
  const float v000 = p[0], v001 = p[1], v010 = p[R], v011 = p[R + 1], v100 = p[R * R], v101 = p[R * R + 1], v110 = p[R * R + R], v111 = p[R * R + R + 1];
  const float c00 = fmaf(v100, ox, v000 * ax), c01 = fmaf(v101, ox, v001 * ax), c10 = fmaf(v110, ox, v010 * ax), c11 = fmaf(v111, ox, v011 * ax);
  const float c0 = fmaf(c10, oy, c00 * ay), c1 = fmaf(c11, oy, c01 * ay);
  return fmaf(c1, oz, c0 * az);
}
__device__ __forceinline__ float sample_rec(const float4* rec, float gx, float gy, float gz) {  // face records, grid order
  const float top = (float)(R - 2);
  const float bx = fminf(fmaxf(floorf(gx), 0.f), top), by = fminf(fmaxf(floorf(gy), 0.f), top), bz = fminf(fmaxf(floorf(gz), 0.f), top);
  const float ox = gx - bx, oy = gy - by, oz = gz - bz, ax = 1.f - ox, ay = 1.f - oy, az = 1.f - oz;
  const int lin = (int)fmaf(fmaf(bx, (float)R, by), (float)R, bz);
  const float4 a = rec[lin], b = rec[lin + R * R];
  const float c00 = fmaf(b.x, ox, a.x * ax), c01 = fmaf(b.y, ox, a.y * ax), c10 = fmaf(b.z, ox, a.z * ax), c11 = fmaf(b.w, ox, a.w * ax);
  const float c0 = fmaf(c10, oy, c00 * ay), c1 = fmaf(c11, oy, c01 * ay);
  return fmaf(c1, oz, c0 * az);
}

// MODE 0: grid in LDS; 1: plain grid through the L1 (same addressing); 2: face records through the L1
template <int MODE>
__global__ __launch_bounds__(256) void march_kernel(const float* __restrict__ sdf, const float4* __restrict__ rec,
                                                    const View* __restrict__ views, float thr, float* __restrict__ depth,
                                                    unsigned long long* __restrict__ steps) {
  __shared__ float g[MODE == 0 ? R * R * R : 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (MODE == 0) {
    for (int i = tid; i < R * R * R / 4; i += 256) reinterpret_cast<float4*>(g)[i] = reinterpret_cast<const float4*>(sdf)[i];
    __syncthreads();
  }
  const View& s = views[blockIdx.z];
  float* img = depth + (size_t)blockIdx.z * W * H;
  unsigned long long nsteps = 0;
  for (int sub = 0; sub < 2; ++sub) {  // a 64 x 8 tile as two 32 x 8 sub-tiles, a wave = an 8 x 8 patch
    const int col = blockIdx.x * 64 + sub * 32 + (wave & 3) * 8 + (lane & 7), row = blockIdx.y * 8 + (lane >> 3);
    const float dx = ((float)col + 0.5f - 0.5f * W) * (2.0f / W), dy = -((float)row + 0.5f - 0.5f * H) * (2.0f / W);
    const float il = rsqrtf(dx * dx + dy * dy + 1.0f);
    const float d[3] = {(s.rot[0] * dx + s.rot[3] * dy - s.rot[6]) * il, (s.rot[1] * dx + s.rot[4] * dy - s.rot[7]) * il,
                        (s.rot[2] * dx + s.rot[5] * dy - s.rot[8]) * il};
    float tn = -1e-10f, tf = 1e10f;
    for (int a = 0; a < 3; ++a) {
      const float inv = 1.0f / d[a], ta = (s.e[a] + s.scale) * inv, tb = (s.e[a] - s.scale) * inv;
      tn = fmaxf(tn, fminf(ta, tb)); tf = fminf(tf, fmaxf(ta, tb));
    }
    float result = 0.f, t = fmaxf(tn, 0.f);
    if (!(tn > tf) && !(tf < 0.f) && t < tf) {
      const float k = s.isc * 0.5f * (R - 1);
      for (int n = 0; n < 4096; ++n) {
        ++nsteps;
        const float gx = fmaf(t, d[0] * k, s.og[0]), gy = fmaf(t, d[1] * k, s.og[1]), gz = fmaf(t, d[2] * k, s.og[2]);
        const float v = MODE == 0 ? sample_lds(g, gx, gy, gz) : (MODE == 1 ? sample_lds(sdf, gx, gy, gz) : sample_rec(rec, gx, gy, gz));
        const float dist = v * s.scale;
        if (dist < thr * t) { result = t * il; break; }
        t += dist;
        if (!(t < tf)) break;
      }
    }
    img[row * W + col] = result;
  }
  for (int off = 32; off; off >>= 1) nsteps += __shfl_xor(nsteps, off, 64);
  if (lane == 0 && steps) atomicAdd(steps, nsteps);
}

int main() {
  // 16^3 union of spheres, random poses like the benchmark's (scale 0.4-0.6, z 1.2-2)
  std::vector<float> sdf(R * R * R);
  srand(1);
  auto rnd = [] { return rand() / (float)RAND_MAX; };
  float c[8][4];
  for (auto& q : c) { q[0] = rnd() * 0.9f - 0.45f; q[1] = rnd() * 0.9f - 0.45f; q[2] = rnd() * 0.9f - 0.45f; q[3] = 0.15f + 0.2f * rnd(); }
  for (int x = 0; x < R; ++x) for (int y = 0; y < R; ++y) for (int z = 0; z < R; ++z) {
    const float X = -1 + 2.f * x / (R - 1), Y = -1 + 2.f * y / (R - 1), Z = -1 + 2.f * z / (R - 1);
    float m = 1e9f;
    for (auto& q : c) m = fminf(m, sqrtf((X - q[0]) * (X - q[0]) + (Y - q[1]) * (Y - q[1]) + (Z - q[2]) * (Z - q[2])) - q[3]);
    sdf[(x * R + y) * R + z] = m;
  }
  std::vector<float4> rec(R * R * R + R * R, make_float4(0, 0, 0, 0));
  for (int x = 0; x < R; ++x) for (int y = 0; y + 1 < R; ++y) for (int z = 0; z + 1 < R; ++z) {
    const float* p = &sdf[(x * R + y) * R + z];
    rec[(x * R + y) * R + z] = make_float4(p[0], p[1], p[R], p[R + 1]);
  }
  std::vector<View> views(B);
  for (auto& v : views) {
    float u1 = rnd(), u2 = rnd(), u3 = rnd();
    const float x = sqrtf(1 - u1) * sinf(6.2832f * u2), y = sqrtf(1 - u1) * cosf(6.2832f * u2), z = sqrtf(u1) * sinf(6.2832f * u3), w = sqrtf(u1) * cosf(6.2832f * u3);
    const float r[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    const float zz = 1.2f + 0.8f * rnd(), uu = (0.25f + 0.5f * rnd()) * W, vv = (0.25f + 0.5f * rnd()) * H;
    const float p[3] = {(uu - W / 2) * zz / (W / 2), -(vv - H / 2) * zz / (W / 2), -zz};
    v.scale = 0.4f + 0.2f * rnd(); v.isc = 1.0f / v.scale;
    for (int k = 0; k < 9; ++k) v.rot[k] = r[k];
    for (int k = 0; k < 3; ++k) { v.e[k] = r[k] * p[0] + r[3 + k] * p[1] + r[6 + k] * p[2]; v.og[k] = (-v.e[k] * v.isc + 1.0f) * 0.5f * (R - 1); }
  }
  float *d_sdf, *d_depth[3]; float4* d_rec; View* d_views; unsigned long long* d_steps;
  hipMalloc(&d_sdf, sdf.size() * 4); hipMemcpy(d_sdf, sdf.data(), sdf.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_rec, rec.size() * 16); hipMemcpy(d_rec, rec.data(), rec.size() * 16, hipMemcpyHostToDevice);
  hipMalloc(&d_views, B * sizeof(View)); hipMemcpy(d_views, views.data(), B * sizeof(View), hipMemcpyHostToDevice);
  hipMalloc(&d_steps, 8); hipMemset(d_steps, 0, 8);
  for (auto& p : d_depth) hipMalloc(&p, (size_t)B * W * H * 4);
  const dim3 grid(W / 64, H / 8, B);
  const char* names[3] = {"grid in LDS (16 KiB per workgroup)", "plain grid through the L1", "16-byte face records through the L1"};
  hipLaunchKernelGGL(march_kernel<1>, grid, dim3(256), 0, 0, d_sdf, d_rec, d_views, 0.005f, d_depth[1], d_steps);
  hipDeviceSynchronize();
  unsigned long long steps = 0; hipMemcpy(&steps, d_steps, 8, hipMemcpyDeviceToHost);
  printf("%d views of %dx%d, %d^3 grid: %.1f M march steps per launch\n", B, W, H, R, steps / 1e6);
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 3; ++m) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) {
        if (m == 0) hipLaunchKernelGGL(march_kernel<0>, grid, dim3(256), 0, 0, d_sdf, d_rec, d_views, 0.005f, d_depth[0], (unsigned long long*)nullptr);
        if (m == 1) hipLaunchKernelGGL(march_kernel<1>, grid, dim3(256), 0, 0, d_sdf, d_rec, d_views, 0.005f, d_depth[1], (unsigned long long*)nullptr);
        if (m == 2) hipLaunchKernelGGL(march_kernel<2>, grid, dim3(256), 0, 0, d_sdf, d_rec, d_views, 0.005f, d_depth[2], (unsigned long long*)nullptr);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("  %-40s %7.1f us per launch\n", names[m], ms * 100.f);
    }
  std::vector<float> h0((size_t)W * H), h1((size_t)W * H), h2((size_t)W * H);
  hipMemcpy(h0.data(), d_depth[0], h0.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h1.data(), d_depth[1], h1.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h2.data(), d_depth[2], h2.size() * 4, hipMemcpyDeviceToHost);
  size_t diff1 = 0, diff2 = 0, hits = 0;
  for (size_t i = 0; i < h0.size(); ++i) { diff1 += h0[i] != h1[i]; diff2 += h0[i] != h2[i]; hits += h0[i] > 0; }
  printf("view 0: %zu hit pixels; LDS vs plain: %zu differing pixels, LDS vs records: %zu\n", hits, diff1, diff2);
  return 0;
}
