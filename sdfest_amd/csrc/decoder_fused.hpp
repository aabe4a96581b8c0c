// decoder_fused.hpp -- the single decode's layer PAIRS as one launch each (round 6).
//
// The captured render-and-compare iteration (C5) is a chain of dependent launches; a single decode's layers are far too
// small to fill the chip, so what an iteration costs is the number of launches and the memory round trips inside each.
// Handing a tensor from one group of workgroups to another inside a launch was measured and lost
// (profiles/r05_chained_stages.md), so the pairs are fused WITHOUT a hand-off: a workgroup owns one column of the
// convolution's output (16 * ZT consecutive z at one (x, y)) and forms the operand patch under it -- [channel][3][3][z]
// -- in LDS from the PRODUCER's input, with the producer's own expression tree:
//   conv3d_mfma_up_kernel       resize3_kernel's trilinear tree on a coarse sub-box  (sdf_vae.py:235-246: F.interpolate
//                               in front of every Conv3d), then conv3d_mfma_kernel's contraction;
//   conv3d_mfma_tresize_kernel  the transposed resize of the VJP (z, y, x passes of resize3_backward_tiled_kernel on the
//                               fine block above the column, ReLU mask, the swapped 1x1x1 layer, zero padding), then the
//                               flipped-weight convolution;
//   fc_conv_kernel              the wide Linear layer's rows under the column (+ ReLU), then the first convolution.
// Every output value is the unfused kernels' chain of operations in the same order (the MFMA rows are independent, a
// tile's K order and split-K partition are kept), so results are bit for bit the unfused launches'
// (tests/test_decoder_gpu.py::test_single_latent_fused_pairs_are_bitwise_the_unfused_launches).  The recomputation a
// column's halo costs is free here: a single decode leaves most of the chip idle.
//
// Included by decoder.hip inside namespace sdfr { namespace {  (uses resize_axis, blend, resize_row16, stage_to_lds).
#pragma once

#ifdef SDFR_VS_STAMPS   // timing experiment: the stages of the first workgroup, 10 ns ticks (device printf)
#define VSS_DECL() unsigned long long vss[10]; int vsk = 0
#define VSS() do { if (vsk < 10) vss[vsk++] = wall_clock64(); } while (0)
#else
#define VSS_DECL() do { } while (0)
#define VSS() do { } while (0)
#endif

// stage_to_lds for a workgroup of nthr threads (n % 4 == 0, both 16-byte aligned; eight loads in flight per thread: a
// 432-tap layer's 27 KB in ONE round trip for 256 threads)
__device__ __forceinline__ void stage_to_lds_n(float* __restrict__ dst, const float* __restrict__ src, int n, int tid,
                                               int nthr) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  const int n4 = n >> 2;
  constexpr int kIn = 8;
  for (int i = tid; i < n4; i += kIn * nthr) {
    f32x4 v[kIn];
#pragma unroll
    for (int u = 0; u < kIn; ++u) v[u] = s4[min(i + u * nthr, n4 - 1)];
#pragma unroll
    for (int u = 0; u < kIn; ++u)
      if (i + u * nthr < n4) d4[i + u * nthr] = v[u];
  }
}

// floats behind a patch that the zero-weight padding taps (kk >= K) of the last tile may read: zeros
__host__ __device__ __forceinline__ int patch_slack(int ZT) { return 16 * ZT + 16; }

// A[row][tap kk] of a tile comes from patch[(kk / 3) * PZ + kk % 3 + row]: kk = channel * 27 + (a * 3 + b) * 3 + c
__device__ __forceinline__ int patch_tap(int kk, int PZ) {
  const int t = kk / 3;
  return t * PZ + (kk - 3 * t);
}

// The contraction of the column's tiles from the LDS patch, in the two orders conv3d_mfma_kernel takes for few latents:
//   SPLIT  its split-K form: the four waves 4t .. 4t+3 share tile t, wave w taking every fourth 32-tap chunk, the wave
//          (kpad / 32) & 3 the tail; the first of them adds the partial accumulators (w0 + w1) + (w2 + w3);
//   plain  wave t runs tile t's whole chain (32-tap groups, then 4-tap steps).
// Returns true in the waves that hold a finished tile (acc: rows kq * 4 + r of column lane & 15, bias not added);
// `tile` is that tile's index.  Every thread of the workgroup must call it (barrier inside).
template <bool SPLIT>
__device__ __forceinline__ bool column_contract(const float* __restrict__ patch, const float* __restrict__ w_l,
                                                float* __restrict__ red, int kpad, int PZ, int ZT, f32x4& acc, int& tile) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = lane & 15, kq = lane >> 4;
  acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if (SPLIT) {
    tile = wave >> 2;
    const int sub = wave & 3;
    if (tile < ZT) {
      const float* base = patch + tile * 16 + row;
      const int full = kpad / 32;
      for (int c = sub; c < full; c += 4) {
        const int kk0 = c * 32;
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int kk = kk0 + 4 * u + kq;
          a[u] = base[patch_tap(kk, PZ)];
          b[u] = w_l[kk * 16 + row];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
      }
      if (sub == (full & 3)) {
        for (int kk0 = full * 32; kk0 < kpad; kk0 += 4) {
          const int kk = kk0 + kq;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[patch_tap(kk, PZ)], w_l[kk * 16 + row], acc, 0, 0, 0);
        }
      }
      *reinterpret_cast<f32x4*>(red + (size_t)tid * 4) = acc;
    }
    __syncthreads();
    if (tile >= ZT || sub != 0) return false;
    const float* r0 = red + (size_t)(tile * 256 + lane) * 4;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0), a1 = *reinterpret_cast<const f32x4*>(r0 + 256);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(r0 + 512), a3 = *reinterpret_cast<const f32x4*>(r0 + 768);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (a0[r] + a1[r]) + (a2[r] + a3[r]);
    return true;
  } else {
    tile = wave;
    if (tile >= ZT) return false;
    const float* base = patch + tile * 16 + row;
    int kk0 = 0;
    for (; kk0 + 32 <= kpad; kk0 += 32) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = kk0 + 4 * u + kq;
        a[u] = base[patch_tap(kk, PZ)];
        b[u] = w_l[kk * 16 + row];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    for (; kk0 < kpad; kk0 += 4) {
      const int kk = kk0 + kq;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[patch_tap(kk, PZ)], w_l[kk * 16 + row], acc, 0, 0, 0);
    }
    return true;
  }
}

// bias, ReLU and store of a finished tile: out [Cout][m][m][m] of sample nb, column (x, y), rows z = tile * 16 + ...
__device__ __forceinline__ void column_store(const f32x4& acc, int tile, const float* __restrict__ bias, int co_tile,
                                             int Cout, int relu, float* __restrict__ out, int nb, int m, int x, int y) {
  const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
  const int co = co_tile * 16 + row;
  if (co >= Cout) return;
  const float bv = bias[co];
  float* dst = out + ((((size_t)nb * Cout + co) * m + x) * m + y) * m;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int z = tile * 16 + kq * 4 + r;
    if (z >= m) continue;
    float v = acc[r] + bv;
    if (relu) v = fmaxf(v, 0.0f);
    dst[z] = v;
  }
}

// Tap table of a patch [channel][IX][IY][PZ]: offset of tap kk = channel * 27 + (a * 3 + b) * 3 + c; 0 for the
// zero-weight padding taps kk >= K.
__device__ __forceinline__ void patch_taps(int* __restrict__ tap_l, int kpad, int K, int IX, int IY, int PZ, int tid,
                                           int nthr) {
  for (int kk = tid; kk < kpad; kk += nthr) {
    int off = 0;
    if (kk < K) {
      const int ch = kk / 27, r = kk - 27 * ch, a = r / 9, b = (r - 9 * a) / 3, c = r - 9 * a - 3 * b;
      off = ((ch * IX + a) * IY + b) * PZ + c;
    }
    tap_l[kk] = off;
  }
}

// One tile's contraction in conv3d_mfma_kernel's split-K ORDER -- four partial accumulators, partial w taking the
// 32-tap chunks c = w, w + 4, ... in ascending order and partial (kpad / 32) & 3 the tail, combined (p0 + p1) + (p2 + p3)
// -- by four waves (WPT == 4: waves 4t .. 4t+3 of the workgroup share tile t, as that kernel does; barrier inside, every
// thread must call) or by ONE wave holding all four partials (WPT == 1: workgroups of many tiles; the matrix cores are
// not what binds a single decode).  `base`: the patch at this lane's row of the tile; A[row][kk] = base[tap_l[kk]].
// Returns true where `out` holds the finished rows kq * 4 + r of column lane & 15 (bias not added).
template <int WPT>
__device__ __forceinline__ bool tile_contract(const float* __restrict__ base, bool has_tile, const int* __restrict__ tap_l,
                                              const float* __restrict__ w_l, float* __restrict__ red, int kpad,
                                              f32x4& out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = lane & 15, kq = lane >> 4;
  const int full = kpad / 32;
  auto chunk = [&](int kk0, f32x4 acc) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = kk0 + 4 * u + kq;
      a[u] = base[tap_l[kk]];
      b[u] = w_l[kk * 16 + row];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    return acc;
  };
  auto tail = [&](f32x4 acc) {
    for (int kk0 = full * 32; kk0 < kpad; kk0 += 4) {
      const int kk = kk0 + kq;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[tap_l[kk]], w_l[kk * 16 + row], acc, 0, 0, 0);
    }
    return acc;
  };
  if (WPT == 4) {
    const int sub = wave & 3;
    if (has_tile) {
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      for (int c = sub; c < full; c += 4) acc = chunk(c * 32, acc);
      if (sub == (full & 3)) acc = tail(acc);
      *reinterpret_cast<f32x4*>(red + (size_t)tid * 4) = acc;
    }
    __syncthreads();
    if (!has_tile || sub != 0) return false;
    const float* r0 = red + (size_t)tid * 4;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0), a1 = *reinterpret_cast<const f32x4*>(r0 + 256);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(r0 + 512), a3 = *reinterpret_cast<const f32x4*>(r0 + 768);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = (a0[r] + a1[r]) + (a2[r] + a3[r]);
    return true;
  } else {
    if (!has_tile) return false;
    f32x4 p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int c0 = 0; c0 < full; c0 += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < full) p[j] = chunk((c0 + j) * 32, p[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j == (full & 3)) p[j] = tail(p[j]);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = (p[0][r] + p[1][r]) + (p[2][r] + p[3][r]);
    return true;
  }
}

// resize ni -> n (trilinear, align_corners = False, up-sampling) + valid 3x3x3 convolution n -> m = n - 2, split-K order.
//   in [N][Cin][ni^3];  out [N][Cout][m^3];  wmat [co_tiles][kpad][16]
// A workgroup owns TX x TY output columns, all of z (ZT = ceil(m / 16) tiles each): few enough workgroups that every
// one has a CU to itself, and a patch whose halo is shared by the columns (one column per workgroup interpolates every
// fine element nine times over: measured, 900 workgroups of the 8 -> 4 layer 16.6 us against 5.4 + 8.6 as two launches).
// The patch [Cin][TX + 2][TY + 2][PZ] is formed in two steps with resize3_kernel's expression tree, z innermost:
//   A  Zc[ci][coarse column][fine z] = blend along z of the coarse columns under the patch, straight from memory (a
//      thread keeps its z: the terms once);
//   B  patch = blend along y, then x of four Zc columns (a thread keeps its (x, y): the terms once).
//   grid (tiles_x * tiles_y, co_tiles, N), block 64 * WPT * TX * TY * ZT (>= 256)
//   LDS: w_l [kpad * 16] | tap_l [kpad] | red [WPT == 4 ? 4 * nthr : 0] | patch | Zc [Cin][CX][CX][PZ]
// CX: the most coarse columns under TX + 2 (TY + 2) fine ones (host, the kernel's arithmetic).
// mix_out != NULL: the 1x1x1 layer that FOLLOWS (mix_co <= 4 output channels, its [Kpad][16] matrix and bias; a layer
// swapped with its resize, so no ReLU here) is applied to the finished rows -- conv1x1_kernel's chain over the channels
// in ascending order, "+ bias" last, exactly what resize3_mix_kernel forms at each corner -- and written to mix_out
// [N][mix_co][m^3]: the resize behind it then gathers ONE channel instead of Cout (`out` may be NULL without a tape).
template <int WPT>
__global__ __launch_bounds__(1024) void conv3d_mfma_up_kernel(
    const float* __restrict__ in, int ni, const float* __restrict__ wmat, const float* __restrict__ bias,
    float* __restrict__ out, int Cin, int Cout, int n, int m, int kpad, int relu, int CX, int ZT, int TX, int TY,
    const float* __restrict__ mix_w, const float* __restrict__ mix_b, int mix_co, float* __restrict__ mix_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  VSS_DECL();
  VSS();
  const int IX = TX + 2, IY = TY + 2, PZ = 16 * ZT + 2, patch_n = Cin * IX * IY * PZ;
  float* w_l = lds;
  int* tap_l = reinterpret_cast<int*>(w_l + (size_t)kpad * 16);
  float* red = reinterpret_cast<float*>(tap_l + kpad);
  float* patch = red + (WPT == 4 ? 4 * nthr : 0);
  float* Zc = patch + ((patch_n + 3) & ~3);
  const int co_tile = blockIdx.y, nb = blockIdx.z;
  const int tiles_y = (m + TY - 1) / TY;
  const int tx0 = ((int)blockIdx.x / tiles_y) * TX, ty0 = ((int)blockIdx.x % tiles_y) * TY;
  const float ratio = (float)ni / (float)n;
  int cx0, cx1, cy0, cy1, t;
  float fl;
  resize_axis(tx0, ratio, ni, cx0, t, fl);
  resize_axis(min(tx0 + IX, n) - 1, ratio, ni, t, cx1, fl);
  resize_axis(ty0, ratio, ni, cy0, t, fl);
  resize_axis(min(ty0 + IY, n) - 1, ratio, ni, t, cy1, fl);
  const int cxn = cx1 - cx0 + 1, cyn = cy1 - cy0 + 1;
  const size_t vi = (size_t)ni * ni * ni;
  const float* src = in + (size_t)nb * Cin * vi;
  {   // A: thread <-> (fine z, coarse columns g, g + gpz, ...), four columns' loads in flight
    const int gpz = nthr / PZ, fz = tid % PZ, g = tid / PZ;
    int z0, z1;
    float lz;
    resize_axis(min(fz, n - 1), ratio, ni, z0, z1, lz);
    const float wz0 = 1.0f - lz;
    const int columns = Cin * cxn * cyn;
    const unsigned m_cy = magic_of(cyn), m_cx = magic_of(cxn);
    bool first = true;
    constexpr int kA = 4;    // columns per pass (twelve -- a thread's ~10 columns in ONE pass -- measured: no faster, 3.3 us either way)
    for (int col0 = g; col0 < columns && g < gpz; col0 += kA * gpz) {
      float v0[kA], v1[kA];
      int at[kA];
#pragma unroll
      for (int u = 0; u < kA; ++u) {
        const int col = min(col0 + u * gpz, columns - 1);
        const int r1 = div_by(col, m_cy), jy = col - r1 * cyn;
        const int ci = div_by(r1, m_cx), jx = r1 - ci * cxn;
        const float* p = src + (size_t)ci * vi + ((size_t)(cx0 + jx) * ni + (cy0 + jy)) * ni;
        v0[u] = p[z0];
        v1[u] = p[z1];
        at[u] = col0 + u * gpz < columns ? ((ci * CX + jx) * CX + jy) * PZ + fz : -1;
      }
      if (first) {   // (the weights' loads behind the first columns': one wait covers both)
        stage_to_lds_n(w_l, wmat + (size_t)co_tile * kpad * 16, kpad * 16, tid, nthr);
        first = false;
      }
#pragma unroll
      for (int u = 0; u < kA; ++u)
        if (at[u] >= 0) Zc[at[u]] = blend(wz0, v0[u], lz, v1[u]);
    }
    if (first) stage_to_lds_n(w_l, wmat + (size_t)co_tile * kpad * 16, kpad * 16, tid, nthr);
  }
  patch_taps(tap_l, kpad, Cin * 27, IX, IY, PZ, tid, nthr);
  __syncthreads();
  VSS();   // 1: A (+ weights)
  {   // B: thread <-> (patch column (a, b), elements j, j + lpp, ... of its Cin * PZ)
    const int pairs = IX * IY, lpp = nthr / pairs, pair = tid / lpp, j0 = tid - pair * lpp;
    if (pair < pairs) {
      const int a = pair / IY, b = pair - a * IY;
      int x0, x1, y0, y1;
      float lx, ly;
      resize_axis(min(tx0 + a, n - 1), ratio, ni, x0, x1, lx);
      resize_axis(min(ty0 + b, n - 1), ratio, ni, y0, y1, ly);
      const float wx0 = 1.0f - lx, wy0 = 1.0f - ly;
      const int o00 = ((x0 - cx0) * CX + (y0 - cy0)) * PZ, o01 = ((x0 - cx0) * CX + (y1 - cy0)) * PZ;
      const int o10 = ((x1 - cx0) * CX + (y0 - cy0)) * PZ, o11 = ((x1 - cx0) * CX + (y1 - cy0)) * PZ;
      const unsigned m_pz = magic_of(PZ);
      const int per = Cin * PZ, zc_ch = CX * CX * PZ, p_ch = IX * IY * PZ;
      float* dst = patch + (a * IY + b) * PZ;
      // (four elements' reads in flight at a time was measured: no faster, 1.48 -> 1.60 us)
      for (int j = j0; j < per; j += lpp) {
        const int ci = div_by(j, m_pz), fz = j - ci * PZ;
        const float* zc = Zc + ci * zc_ch + fz;
        dst[ci * p_ch + fz] = blend(wx0, blend(wy0, zc[o00], ly, zc[o01]), lx, blend(wy0, zc[o10], ly, zc[o11]));
      }
    }
  }
  __syncthreads();
  VSS();   // 2: B
  const int tw = WPT == 4 ? wave >> 2 : wave;           // this wave's tile: (column lx, ly; z tile zt)
  const int zt = tw % ZT, cl = tw / ZT, ly_ = cl % TY, lx_ = cl / TY;
  const int x = tx0 + lx_, y = ty0 + ly_;
  const bool has = tw < TX * TY * ZT && x < m && y < m;
  f32x4 acc;
  const bool fin = tile_contract<WPT>(patch + (lx_ * IY + ly_) * PZ + zt * 16 + (lane & 15), has, tap_l, w_l, red, kpad, acc);
  VSS();   // 3: contraction
#ifdef SDFR_VS_STAMPS
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    printf("up-conv n %d m %d threads %d: A %.2f B %.2f contract %.2f us\n", n, m, nthr, (double)(vss[1] - vss[0]) * 0.01,
           (double)(vss[2] - vss[1]) * 0.01, (double)(vss[3] - vss[2]) * 0.01);
#endif
  if (fin) {
    if (out) column_store(acc, zt, bias, co_tile, Cout, relu, out, nb, m, x, y);
    if (mix_out) {   // (one column tile of channels: host)
      const int row = lane & 15, kq = lane >> 4;
      const float bv = row < Cout ? bias[row] : 0.0f;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[r] + bv;
        if (relu) v[r] = fmaxf(v[r], 0.0f);
      }
      float a2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      const int c2 = row < mix_co ? row : 0;
      for (int co = 0; co < Cout; ++co) {
        const float w = mix_w[co * 16 + c2];
#pragma unroll
        for (int r = 0; r < 4; ++r) a2[r] = fmaf(__shfl(v[r], kq * 16 + co, 64), w, a2[r]);
      }
      if (row < mix_co) {
        const float b2 = mix_b[row];
        float* dst = mix_out + ((((size_t)nb * mix_co + row) * m + x) * m + y) * m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int z = zt * 16 + kq * 4 + r;
          if (z < m) dst[z] = a2[r] + b2;
        }
      }
    }
  }
}

// One tile's contraction in conv3d_mfma_kernel's PLAIN order (what it takes where K = Cout * 27 is short): a single
// chain, 32 taps at a time, then 4.
__device__ __forceinline__ void tile_contract_plain(const float* __restrict__ base, const int* __restrict__ tap_l,
                                                    const float* __restrict__ w_l, int kpad, f32x4& out) {
  const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
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
  out = acc;
}

// One stage of the VJP for few latents: the transposed trilinear resize n_out -> n_in of C channels, the ReLU mask of the
// tensor it lands on, [the transposed 1x1x1 layer that was swapped with the resize: COUT > 0, one source channel -> COUT
// channels,] zero padding by `pad` -- none of it stored -- and the flipped-weight 3x3x3 convolution that reads the padded
// tensor P [CP][np^3] (np = n_in + 2 pad, CP = COUT > 0 ? COUT : C), giving [Cc][nc^3], nc = np - 2.
//   * a workgroup owns TX x TY output columns, all of z; it stages the block of fine rows above its (TX + 2) x (TY + 2)
//     columns of P and runs resize3_backward_tiled_kernel's passes on it -- z (in place), y, x; per pass the non-zero taps
//     in ascending source order, fmaf(w, v, acc) from 0 -- then mask, mix (`fmaf(acc, w_mix, 0) + bias`), into the LDS
//     patch; conv3d_mfma_kernel's contraction from there (MODE 0: its plain order, 4 / 1: its split-K order by four waves
//     / by one wave with four partial accumulators);
//   * the z pass is the heavy one (every fine row above the columns' halo), and a convolution's workgroup holds whole
//     z-rows of its OUTPUT: with e_nin > 0 it applies the z pass of the NEXT stage's transposed resize (nc -> e_nin) to
//     them before they are stored -- out [Cc][nc][nc][e_nin] -- and that stage (zin == 0) starts from rows that are
//     coarse along z already: it stages 2 - 3 x fewer floats and runs the two light passes only.  Same chains in the same
//     order either way: bit for bit the unfused launches.
//   g: zin ? [N][C][n_out^3] (n_out % 4 == 0, 16-byte aligned) : [N][C][n_out][n_out][n_in]
//   grid (tiles_x * tiles_y, co_tiles, N), block: MODE 4: 256 * tiles, else >= 64 * tiles (tiles = TX * TY * ZT <= 16)
//   LDS: w_l [kpad * 16] | tap_l [kpad] | red [MODE == 4 ? 4 * nthr : 0] | patch [CP][IX][IY][PZ] | Y [CK][FX][IY][n_in] |
//        F [CK][FX][FX][RL] (RL = zin ? n_out : n_in; reused as R [TX * TY][16][nc] by the epilogue)
// FX: the longest run of fine indices under TX + 2 (TY + 2) neighbouring coarse ones (host).  n_in, e_nin <= 64; exact
// source ranges <= TAPS (epilogue: kBtTaps), candidate ranges <= 16 (host).
struct VjpStage {
  const float* g;
  const float* act;
  const float* mix_w;
  const float* mix_b;
  const float* wmat;
  const float* bias;
  float* out;
  const float* tab;     // this stage's transposed resize: [n_in][16] = first source, taps, 12 weights (sdfr_decoder_create)
  const float* e_tab;   // the epilogue's ([e_nin][16]), or NULL
  int C, n_in, n_out, pad, Cc, kpad, CK, FX, ZT, TX, TY, zin, e_nin;
  // sdfr_decoder_backward_latent_deferred_scaled (ZIN only): the stage reads  g + k g2,  k = weight / cnt[0] (0 if the
  // count is 0) -- a second upstream volume whose normalisation its producer could not know (render.hip,
  // render_fused_l1_pc_kernel)
  const float* g2 = nullptr;
  const float* cnt = nullptr;
  float weight = 0.0f;
};
constexpr int kVjpXY = 6;   // TX + 2, TY + 2 <= 6
// ZIN: the stage runs the z pass itself (its registers: the next round's block and the z taps -- workgroups of <= 512
// threads); else its producer's epilogue has (s.zin == ZIN: host).
template <int COUT, int TAPS, int MODE, bool ZIN>
__global__ __launch_bounds__(ZIN ? 512 : 1024) void vjp_stage_kernel(VjpStage s) {
  constexpr int CO = COUT > 0 ? COUT : 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float w_tab[2 * kVjpXY + 64][kBtTaps];   // rows: a (x), kVjpXY + b (y), 2 kVjpXY + iz (z)
  __shared__ int d_tab[2 * kVjpXY + 64], n_tab[2 * kVjpXY];
  __shared__ float e_w[64][kBtTaps];                  // the epilogue's z pass
  __shared__ int e_d[64], e_n[64];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int co_tile = blockIdx.y, nb = blockIdx.z;
  VSS_DECL();
  VSS();
  const int C = s.C, n_in = s.n_in, n_out = s.n_out, pad = s.pad, TX = s.TX, TY = s.TY, ZT = s.ZT, FX = s.FX, CK = s.CK;
  const int np = n_in + 2 * pad, nc = np - 2;
  const int tiles_y = (nc + TY - 1) / TY;
  const int tx0 = ((int)blockIdx.x / tiles_y) * TX, ty0 = ((int)blockIdx.x % tiles_y) * TY;
  const int CP = COUT > 0 ? COUT : C;
  const int IX = TX + 2, IY = TY + 2, PZ = 16 * ZT + 2, patch_n = CP * IX * IY * PZ, ab_n = IX * IY;
  const int RL = ZIN ? n_out : n_in;
  float* w_l = lds;
  int* tap_l = reinterpret_cast<int*>(w_l + (size_t)s.kpad * 16);
  float* red = reinterpret_cast<float*>(tap_l + s.kpad);
  float* patch = red + (MODE == 4 ? 4 * nthr : 0);
  float* Y = patch + ((patch_n + 3) & ~3);
  float* F = Y + (((size_t)CK * FX * IY * n_in + 3) & ~(size_t)3);
  const size_t coarse_vol = (size_t)n_in * n_in * n_in;
  const unsigned m_ni = magic_of(n_in), m_ab = magic_of(ab_n), m_iy = magic_of(IY);
  // an item of the x pass: e -> (ck, a, b, jz); its place in the coarse tensor (-1: outside, a zero of the padding)
  auto x_item = [&](int e, int& ck, int& ab, int& a, int& b, int& jz) {
    const int r0 = div_by(e, m_ni);
    jz = e - r0 * n_in;
    ck = div_by(r0, m_ab);
    ab = r0 - ck * ab_n;
    a = div_by(ab, m_iy);
    b = ab - a * IY;
    const int ix = tx0 + a - pad, iy = ty0 + b - pad;
    return (ix < 0 || ix >= n_in || iy < 0 || iy >= n_in) ? -1 : (ix * n_in + iy) * n_in + jz;
  };
  // the ReLU mask of this thread's first item of the first round: on its way before anything else
  float mask0[CO];
  if (s.act) {
    int ck, ab, a, b, jz;
    const int at = tid < min(CK, C) * ab_n * n_in ? x_item(tid, ck, ab, a, b, jz) : -1;
#pragma unroll
    for (int co = 0; co < CO; ++co)
      mask0[co] = at >= 0 ? s.act[((size_t)nb * CP + (COUT > 0 ? co : ck)) * coarse_vol + at] : 0.0f;
  }
  // the block of fine rows above the workgroup's columns of P: x in [fx0, fx0 + fnx), y in [fy0, fy0 + fny), whole rows
  // (first source and taps of the first / last coarse index inside the tensor: four rows of the table, scalar loads)
  const int a_lo = max(pad - tx0, 0), a_hi = min(n_in - 1 - tx0 + pad, IX - 1);
  const int b_lo = max(pad - ty0, 0), b_hi = min(n_in - 1 - ty0 + pad, IY - 1);
  const int* itab = reinterpret_cast<const int*>(s.tab);
  const int fx0 = itab[(tx0 + a_lo - pad) * 16], fx1 = itab[(tx0 + a_hi - pad) * 16] + itab[(tx0 + a_hi - pad) * 16 + 1];
  const int fy0 = itab[(ty0 + b_lo - pad) * 16], fy1 = itab[(ty0 + b_hi - pad) * 16] + itab[(ty0 + b_hi - pad) * 16 + 1];
  const int fnx = fx1 - fx0, fny = fy1 - fy0;
  const size_t src_vol = (size_t)n_out * n_out * RL;
  const int slab = fny * RL, slab_u = ZIN ? slab >> 2 : slab;   // units: 16-byte vectors (ZIN: RL % 4 == 0) / floats
  const unsigned m_fy = magic_of(fny), m_fx = magic_of(fnx), m_su = magic_of(slab_u);
  auto unit_at = [&](int e, int& lds_off) {   // unit e of a round's block -> float offsets in g's channels / in F
    const int sl = div_by(e, m_su), off = (e - sl * slab_u) * (ZIN ? 4 : 1);
    const int ck = div_by(sl, m_fx), fx = sl - ck * fnx;
    lds_off = (ck * FX + fx) * FX * RL + off;
    return (size_t)ck * src_vol + ((size_t)(fx0 + fx) * n_out + fy0) * RL + off;
  };
  constexpr int kFl = ZIN ? 8 : 1;
  f32x4 pre[kFl];
  // (with a second volume the registers hold half as many units, each from both volumes: pre[j], pre[j + kFl / 2])
  const bool two = ZIN && s.g2 != nullptr;
  const int kfl = two ? kFl / 2 : kFl;
  float k2 = 0.0f;
  if (two) {
    const float cnt = s.cnt[0];
    k2 = cnt > 0.0f ? s.weight / cnt : 0.0f;
  }
  auto prefetch = [&](int c0) {
    const int total = min(CK, C - c0) * fnx * slab_u;
    const float* src = s.g + ((size_t)nb * C + c0) * src_vol;
    if (two) {
      const float* src2 = s.g2 + ((size_t)nb * C + c0) * src_vol;
#pragma unroll
      for (int j = 0; j < kFl / 2; ++j) {
        int lo;
        const size_t g = unit_at(min(tid + nthr * j, total - 1), lo);
        pre[j] = *reinterpret_cast<const f32x4*>(src + g);
        pre[(j + kFl / 2) % kFl] = *reinterpret_cast<const f32x4*>(src2 + g);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < kFl; ++j) {
      int lo;
      const size_t g = unit_at(min(tid + nthr * j, total - 1), lo);
      pre[j] = *reinterpret_cast<const f32x4*>(src + g);
    }
  };
  // the tables of this column block -- rows a (x), kVjpXY + b (y), 2 kVjpXY + iz (z, ZIN), then the epilogue's --: loaded
  // first, stored behind everything else that is issued (the block's loads do not wait for them)
  constexpr int kTv = 4;
  float tv[kTv];
  const int t_rows = 2 * kVjpXY + (ZIN ? n_in : 0), t_total = (t_rows + s.e_nin) * 16;
  auto table_row = [&](int row, bool& epi) {
    epi = row >= t_rows;
    int i;
    if (epi) i = row - t_rows;
    else if (row < kVjpXY) { i = tx0 + row - pad; if (row >= IX || i < 0 || i >= n_in) i = -1; }
    else if (row < 2 * kVjpXY) { i = ty0 + row - kVjpXY - pad; if (row - kVjpXY >= IY || i < 0 || i >= n_in) i = -1; }
    else i = row - 2 * kVjpXY;
    return i;
  };
#pragma unroll
  for (int u = 0; u < kTv; ++u) {
    const int e = tid + nthr * u;
    bool epi;
    const int i = e < t_total ? table_row(e >> 4, epi) : -1;
    tv[u] = i >= 0 ? (epi ? s.e_tab : s.tab)[i * 16 + (e & 15)] : 0.0f;
  }
  if (ZIN) prefetch(0);   // (in flight while the tables and the weights are staged)
  auto table_store = [&](int e, float v) {
    const int row = e >> 4, k = e & 15;
    bool epi;
    const int i = table_row(row, epi);
    const int vi = __builtin_bit_cast(int, v);
    if (epi) {
      if (k == 0) e_d[i] = vi; else if (k == 1) e_n[i] = vi; else if (k - 2 < kBtTaps) e_w[i][k - 2] = v;
    } else {
      if (k == 0) d_tab[row] = vi; else if (k == 1) { if (row < 2 * kVjpXY) n_tab[row] = vi; } else if (k - 2 < kBtTaps) w_tab[row][k - 2] = v;
    }
  };
  for (int e = tid; e < patch_n; e += nthr) patch[e] = 0.0f;   // the padding
  patch_taps(tap_l, s.kpad, CP * 27, IX, IY, PZ, tid, nthr);
  VSS();   // 1: set-up issued
  auto tables_to_lds = [&]() {
#pragma unroll
    for (int u = 0; u < kTv; ++u)
      if (tid + nthr * u < t_total) table_store(tid + nthr * u, tv[u]);
    for (int e = tid + nthr * kTv; e < t_total; e += nthr) {   // (more rows than the registers hold: rare)
      bool epi;
      const int i = table_row(e >> 4, epi);
      table_store(e, i >= 0 ? (epi ? s.e_tab : s.tab)[i * 16 + (e & 15)] : 0.0f);
    }
  };
  // lane -> (slot, iz): 64 / nl rows per wave step, nl = the power of two that holds a coarse row
  const int nl = n_in <= 8 ? 8 : (n_in <= 16 ? 16 : (n_in <= 32 ? 32 : 64));
  const int spw = 64 / nl, slot = lane / nl, iz = lane - slot * nl;
  const bool zok = iz < n_in;
  for (int c0 = 0; c0 < C; c0 += CK) {
    const int ck_n = min(CK, C - c0);
    const int total = ck_n * fnx * slab_u;
    if (c0 > 0) __syncthreads();   // the previous round's passes have left F and Y
    if (ZIN) {
      if (c0 == 0) {
        tables_to_lds();
        stage_to_lds_n(w_l, s.wmat + (size_t)co_tile * s.kpad * 16, s.kpad * 16, tid, nthr);
      }
#pragma unroll
      for (int j = 0; j < kFl; ++j) {
        const int e = tid + nthr * j;
        if (j < kfl && e < total) {
          int lo;
          (void)unit_at(e, lo);
          f32x4 v = pre[j];
          if (two) {
            const f32x4 v2 = pre[(j + kFl / 2) % kFl];
            v = f32x4{fmaf(k2, v2.x, v.x), fmaf(k2, v2.y, v.y), fmaf(k2, v2.z, v.z), fmaf(k2, v2.w, v.w)};
          }
          *reinterpret_cast<f32x4*>(F + lo) = v;
        }
      }
      if (total > kfl * nthr) {   // (blocks larger than the prefetch: the rest straight from memory)
        const float* src = s.g + ((size_t)nb * C + c0) * src_vol;
        const float* src2 = two ? s.g2 + ((size_t)nb * C + c0) * src_vol : nullptr;
        for (int e = tid + nthr * kfl; e < total; e += nthr) {
          int lo;
          const size_t g = unit_at(e, lo);
          f32x4 v = *reinterpret_cast<const f32x4*>(src + g);
          if (two) {
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(src2 + g);
            v = f32x4{fmaf(k2, v2.x, v.x), fmaf(k2, v2.y, v.y), fmaf(k2, v2.z, v.z), fmaf(k2, v2.w, v.w)};
          }
          *reinterpret_cast<f32x4*>(F + lo) = v;
        }
      }
      __syncthreads();
      if (c0 + CK < C) prefetch(c0 + CK);
      // z pass, in place: row (ck, fx, fy) -> its head [0, n_in)
      float wz[TAPS];
      int oz[TAPS];
      {
        const int dz = zok ? d_tab[2 * kVjpXY + iz] : 0;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
          wz[k] = zok ? w_tab[2 * kVjpXY + iz][k] : 0.0f;
          oz[k] = min(dz + k, n_out - 1);
        }
      }
      const int rows = ck_n * fnx * fny;
      for (int r = wave * spw + slot; r < rows; r += nw * spw) {
        const int r1 = div_by(r, m_fy), fy = r - r1 * fny;
        const int ck = div_by(r1, m_fx), fx = r1 - ck * fnx;
        float* f = F + (size_t)((ck * FX + fx) * FX + fy) * RL;
        float v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) v[k] = f[oz[k]];
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (wz[k] != 0.0f) ? fmaf(wz[k], v[k], acc) : acc;
        if (zok) f[iz] = acc;
      }
    } else {
      // rows that the producer's epilogue left coarse along z: eight loads in flight per thread
      const float* src = s.g + ((size_t)nb * C + c0) * src_vol;
      constexpr int kB = 8;
      bool staged = c0 > 0;
      for (int e0 = tid; e0 < total; e0 += kB * nthr) {
        float v[kB];
        int lo[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
          const size_t g = unit_at(min(e0 + u * nthr, total - 1), lo[u]);
          v[u] = src[g];
        }
        if (!staged) {
          tables_to_lds();
          stage_to_lds_n(w_l, s.wmat + (size_t)co_tile * s.kpad * 16, s.kpad * 16, tid, nthr);
          staged = true;
        }
#pragma unroll
        for (int u = 0; u < kB; ++u)
          if (e0 + u * nthr < total) F[lo[u]] = v[u];
      }
      if (!staged) {
        tables_to_lds();
        stage_to_lds_n(w_l, s.wmat + (size_t)co_tile * s.kpad * 16, s.kpad * 16, tid, nthr);
      }
    }
    __syncthreads();
    VSS();   // 2: block staged (+ z pass)
    {   // y pass: (ck, fx, b) -> Y[ck][fx][b][iz]
      const int trip = ck_n * fnx * IY;
      for (int r = wave * spw + slot; r < trip; r += nw * spw) {
        const int r1 = div_by(r, m_iy), b = r - r1 * IY;
        const int ck = div_by(r1, m_fx), fx = r1 - ck * fnx;
        const int d0 = d_tab[kVjpXY + b] - fy0;
        const float* zc = F + (size_t)((ck * FX + fx) * FX) * RL + (zok ? iz : 0);
        float w[TAPS], v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
          w[k] = w_tab[kVjpXY + b][k];
          v[k] = zc[min(max(d0 + k, 0), fny - 1) * RL];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (w[k] != 0.0f) ? fmaf(w[k], v[k], acc) : acc;
        if (zok) Y[((ck * FX + fx) * IY + b) * n_in + iz] = acc;
      }
    }
    __syncthreads();
    VSS();   // 3: y pass
    {   // x pass, mask, (mix,) into the patch: items (ck, a, b, iz), inside the tensor only
      const int items = ck_n * ab_n * n_in;
      for (int e = tid; e < items; e += nthr) {
        int ck, ab, a, b, jz;
        const int at = x_item(e, ck, ab, a, b, jz);
        if (at < 0) continue;
        float mask[CO];
        if (s.act) {
#pragma unroll
          for (int co = 0; co < CO; ++co)
            mask[co] = (e == tid && c0 == 0) ? mask0[co]
                                             : s.act[((size_t)nb * CP + (COUT > 0 ? co : c0 + ck)) * coarse_vol + at];
        }
        const int d0 = d_tab[a] - fx0;
        const float* yc = Y + (size_t)((ck * FX) * IY + b) * n_in + jz;
        float w[TAPS], v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
          w[k] = w_tab[a][k];
          v[k] = yc[min(max(d0 + k, 0), fnx - 1) * IY * n_in];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (w[k] != 0.0f) ? fmaf(w[k], v[k], acc) : acc;
        if (COUT == 0) {
          if (s.act && !(mask[0] > 0.0f)) acc = 0.0f;
          patch[((c0 + ck) * ab_n + ab) * PZ + jz + pad] = acc;
        } else {
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            const float v1 = fmaf(acc, s.mix_w[co], 0.0f) + s.mix_b[co];
            patch[(co * ab_n + ab) * PZ + jz + pad] = (!s.act || mask[co] > 0.0f) ? v1 : 0.0f;
          }
        }
      }
    }
  }
  __syncthreads();
  VSS();   // 4: x pass
  // contraction: this wave's tile (column lx, ly; z tile zt)
  const int tw = MODE == 4 ? wave >> 2 : wave;
  const int zt = tw % ZT, cl = tw / ZT, ly_ = cl % TY, lx_ = cl / TY;
  const int x = tx0 + lx_, y = ty0 + ly_;
  const bool has = tw < TX * TY * ZT && x < nc && y < nc;
  const float* base = patch + (lx_ * IY + ly_) * PZ + zt * 16 + (lane & 15);
  f32x4 acc;
  bool done;
  if (MODE == 0) {
    done = has;
    if (has) tile_contract_plain(base, tap_l, w_l, s.kpad, acc);
  } else {
    done = tile_contract<MODE == 4 ? 4 : 1>(base, has, tap_l, w_l, red, s.kpad, acc);
  }
  VSS();   // 5: contraction
#ifdef SDFR_VS_STAMPS
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    printf("vjp stage n_in %d n_out %d zin %d threads %d: tables %.2f stage%s %.2f y %.2f x %.2f contract %.2f us (first round)\n", n_in,
           n_out, (int)ZIN, nthr, (double)(vss[1] - vss[0]) * 0.01, ZIN ? "+z" : "", (double)(vss[2] - vss[1]) * 0.01,
           (double)(vss[3] - vss[2]) * 0.01, (double)(vss[4] - vss[3]) * 0.01, (double)(vss[5] - vss[4]) * 0.01);
#endif
  if (s.e_nin == 0) {
    if (done) column_store(acc, zt, s.bias, co_tile, s.Cc, 0, s.out, nb, nc, x, y);
    return;
  }
  // epilogue: the z pass of the next stage's transposed resize (nc -> e_nin) on the finished z-rows
  float* R = F;   // [TX * TY][16][nc]  (every pass has left F: the barrier in front of the contraction)
  if (done) {
    const int row = lane & 15, kq = lane >> 4, co = co_tile * 16 + row;
    const float bv = co < s.Cc ? s.bias[co] : 0.0f;
    float* rr = R + (size_t)((lx_ * TY + ly_) * 16 + row) * nc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int zz = zt * 16 + kq * 4 + r;
      if (zz < nc) rr[zz] = acc[r] + bv;
    }
  }
  __syncthreads();
  {
    const int e_nin = s.e_nin, items = TX * TY * 16 * e_nin;
    const unsigned m_e = magic_of(e_nin);
    for (int e = tid; e < items; e += nthr) {
      const int r0 = div_by(e, m_e), jz = e - r0 * e_nin;
      const int col = r0 >> 4, row = r0 & 15, co = co_tile * 16 + row;
      const int lx = col / TY, ly = col - lx * TY, xx = tx0 + lx, yy = ty0 + ly;
      if (co >= s.Cc || xx >= nc || yy >= nc) continue;
      const float* rr = R + (size_t)r0 * nc + e_d[jz];
      const int nt = e_n[jz];
      float a2 = 0.0f;
      for (int k = 0; k < nt; ++k) {
        const float w = e_w[jz][k];
        a2 = (w != 0.0f) ? fmaf(w, rr[k], a2) : a2;
      }
      s.out[((((size_t)nb * s.Cc + co) * nc + xx) * nc + yy) * e_nin + jz] = a2;
    }
  }
}

// The Linear stack (narrow leading layers: fc_one_wave_ok) and the FIRST convolution (3x3x3, no resize in front of it:
// sdf_vae.py:223-238) in one launch: every workgroup runs the narrow layers as fc_stack_kernel<true> does, forms the
// rows of the wide layer under its column -- bias first, inputs in ascending order, ReLU: fc_stack_kernel's chain --
// straight into the operand patch, and contracts it (split-K order).  A row's weights (one per input of the wide layer,
// <= kFcWaveWidth) are ALL in flight before the narrow layers start: one memory round trip for the whole stack.
// The wide layer's output is needed again by the VJP (its ReLU mask): with fc_out != NULL the column that owns a fine
// (x, y) -- (min(x, m - 1), min(y, m - 1)) -- stores it.
//   grid (m * m, co_tiles, N), block >= 256 * ZT;  LDS (dynamic): w_l [kpad * 16] | tap_l [kpad] | red [4 * nthr] | patch [Cin * 9 * PZ]
__global__ __launch_bounds__(1024) void fc_conv_kernel(const float* __restrict__ params, FcDesc d,
                                                       const float* __restrict__ z, float* __restrict__ fc_out,
                                                       const float* __restrict__ wmat, const float* __restrict__ bias,
                                                       float* __restrict__ out, int Cin, int Cout, int n, int m, int kpad,
                                                       int relu, int ZT) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float act[2][kFcWaveWidth];
  __shared__ float p_lds[kFcWaveSpan];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  VSS_DECL();
  VSS();
  const int PZ = 16 * ZT + 2, patch_n = Cin * 9 * PZ;
  float* w_l = lds;
  int* tap_l = reinterpret_cast<int*>(w_l + (size_t)kpad * 16);
  float* red = reinterpret_cast<float*>(tap_l + kpad);
  float* patch = red + 4 * nthr;
  const int co_tile = blockIdx.y, nb = blockIdx.z;
  const int x = (int)blockIdx.x / m, y = (int)blockIdx.x - x * m;
  const int lw = d.n_fc - 1, win = d.width[lw], wout = d.width[lw + 1];
  const float* wt = params + d.w_off[lw];   // transposed: [in][out]
  const float* bl = params + d.b_off[lw];
  const int n3 = n * n * n, items = Cin * 9 * n;
  const unsigned m_n = magic_of(n);
  // row e of the patch: (channel, a, b, z) -> its output index o of the wide layer, its place in the patch
  auto row_of = [&](int e, int& o, int& at, bool& own) {
    const int r0 = div_by(e, m_n), zz = e - r0 * n;
    const int ci = r0 / 9, ab = r0 - 9 * ci, a = ab / 3, b = ab - 3 * a;
    o = ci * n3 + ((x + a) * n + (y + b)) * n + zz;
    at = r0 * PZ + zz;
    own = min(x + a, m - 1) == x && min(y + b, m - 1) == y;
  };
  // this thread's first row: its weights on their way before anything else
  float wr[kFcWaveWidth];
  float b0 = 0.0f;
  int o0 = 0, at0 = 0;
  bool own0 = false;
  if (tid < items) {
    row_of(tid, o0, at0, own0);
    b0 = bl[o0];
#pragma unroll
    for (int i = 0; i < kFcWaveWidth; ++i) wr[i] = i < win ? wt[(size_t)i * wout + o0] : 0.0f;
  }
  // the narrow layers (fc_stack_kernel<true>): their parameters to LDS with all loads in flight, then wave 0
  int cur = 0;
  {
    const long long base = d.w_off[0];
    const int span = (int)fc_wave_span(d);
    const float z_t = tid < d.width[0] ? z[(size_t)nb * d.width[0] + tid] : 0.0f;
    constexpr int U = 6;
    for (int e0 = 0; e0 < span; e0 += U * nthr) {
      float r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * nthr + tid;
        r[u] = e < span ? params[base + e] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * nthr + tid;
        if (e < span) p_lds[e] = r[u];
      }
    }
    if (tid < d.width[0]) act[0][tid] = z_t;
    __syncthreads();
    VSS();   // 1: parameters staged
    if (tid < 64) {
      for (int l = 0; l < d.n_fc - 1; ++l) {
        const int wi = d.width[l], wo = d.width[l + 1];
        if (tid < wo) {
          float acc = p_lds[d.b_off[l] - base + tid];
          const float* w = p_lds + (d.w_off[l] - base) + tid * wi;
#pragma unroll 8
          for (int i = 0; i < wi; ++i) acc = fmaf(w[i], act[cur][i], acc);
          act[cur ^ 1][tid] = fmaxf(acc, 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
        cur ^= 1;
      }
    } else {
      // (meanwhile: the convolution's weights, the tap table, the rows of the patch no output reads)
      cur = (d.n_fc - 1) & 1;
    }
    stage_to_lds_n(w_l, wmat + (size_t)co_tile * kpad * 16, kpad * 16, tid, nthr);
    patch_taps(tap_l, kpad, Cin * 27, 3, 3, PZ, tid, nthr);
    for (int e = tid; e < patch_n; e += nthr) patch[e] = 0.0f;
    __syncthreads();
    VSS();   // 2: narrow layers (+ conv weights)
  }
  // the wide layer's rows under the column
  float* fo = (fc_out && co_tile == 0) ? fc_out + (size_t)nb * wout : nullptr;
  if (tid < items) {
    float acc = b0;
#pragma unroll
    for (int i = 0; i < kFcWaveWidth; ++i)
      if (i < win) acc = fmaf(wr[i], act[cur][i], acc);
    acc = fmaxf(acc, 0.0f);
    patch[at0] = acc;
    if (fo && own0) fo[o0] = acc;
  }
  for (int e = tid + nthr; e < items; e += nthr) {
    int o, at;
    bool own;
    row_of(e, o, at, own);
    float acc = bl[o];
#pragma unroll 16
    for (int i = 0; i < win; ++i) acc = fmaf(wt[(size_t)i * wout + o], act[cur][i], acc);
    acc = fmaxf(acc, 0.0f);
    patch[at] = acc;
    if (fo && own) fo[o] = acc;
  }
  __syncthreads();
  VSS();   // 3: wide rows
  const int tw = wave >> 2;
  f32x4 acc;
  const bool fin = tile_contract<4>(patch + tw * 16 + (lane & 15), tw < ZT, tap_l, w_l, red, kpad, acc);
  VSS();   // 4: contraction
#ifdef SDFR_VS_STAMPS
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    printf("fc+conv threads %d: params %.2f narrow %.2f wide %.2f contract %.2f us\n", nthr, (double)(vss[1] - vss[0]) * 0.01,
           (double)(vss[2] - vss[1]) * 0.01, (double)(vss[3] - vss[2]) * 0.01, (double)(vss[4] - vss[3]) * 0.01);
#endif
  if (fin) column_store(acc, tw, bias, co_tile, Cout, relu, out, nb, m, x, y);
}
