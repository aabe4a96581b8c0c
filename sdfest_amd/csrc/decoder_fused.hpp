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

// stage_to_lds for a workgroup of nthr threads (n % 4 == 0, both 16-byte aligned; four loads in flight per thread)
__device__ __forceinline__ void stage_to_lds_n(float* __restrict__ dst, const float* __restrict__ src, int n, int tid,
                                               int nthr) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  const int n4 = n >> 2;
  for (int i = tid; i < n4; i += 4 * nthr) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = s4[min(i + u * nthr, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
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

// resize ni -> n (trilinear, align_corners = False, up-sampling) + valid 3x3x3 convolution n -> m = n - 2, split-K order.
//   in [N][Cin][ni^3];  out [N][Cout][m^3];  wmat [co_tiles][kpad][16]
//   grid (m * m, co_tiles, N), block 256 * ZT, ZT = ceil(m / 16)
//   LDS: w_l [kpad * 16] | red [1024 * ZT] | patch [Cin * 9 * PZ + patch_slack] | coarse [Cin][CX][CX][ni]
// CX: the most coarse columns under three fine ones along an axis (host, the kernel's arithmetic).
__global__ __launch_bounds__(1024) void conv3d_mfma_up_kernel(
    const float* __restrict__ in, int ni, const float* __restrict__ wmat, const float* __restrict__ bias,
    float* __restrict__ out, int Cin, int Cout, int n, int m, int kpad, int relu, int CX, int ZT) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PZ = 16 * ZT + 2, patch_n = Cin * 9 * PZ;
  float* w_l = lds;
  float* red = w_l + (size_t)kpad * 16;
  float* patch = red + 1024 * ZT;
  float* coarse = patch + ((patch_n + patch_slack(ZT) + 3) & ~3);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int co_tile = blockIdx.y, nb = blockIdx.z;
  const int x = (int)blockIdx.x / m, y = (int)blockIdx.x - x * m;
  const float ratio = (float)ni / (float)n;
  int cx0, cx1, cy0, cy1, t;
  float fl;
  resize_axis(x, ratio, ni, cx0, t, fl);
  resize_axis(x + 2, ratio, ni, t, cx1, fl);
  resize_axis(y, ratio, ni, cy0, t, fl);
  resize_axis(y + 2, ratio, ni, t, cy1, fl);
  const int cxn = cx1 - cx0 + 1, cyn = cy1 - cy0 + 1;
  const size_t vi = (size_t)ni * ni * ni;
  const float* src = in + (size_t)nb * Cin * vi;
  // the coarse columns under the patch (all of z): loads in flight before the weights are staged
  const int c_total = Cin * cxn * cyn * ni;
  const unsigned m_ni = magic_of(ni), m_cy = magic_of(cyn), m_cx = magic_of(cxn);
  auto coarse_at = [&](int e, int& lds_off) {
    const int r0 = div_by(e, m_ni), jz = e - r0 * ni;
    const int r1 = div_by(r0, m_cy), jy = r0 - r1 * cyn;
    const int ci = div_by(r1, m_cx), jx = r1 - ci * cxn;
    lds_off = ((ci * CX + jx) * CX + jy) * ni + jz;
    return (size_t)ci * vi + ((size_t)(cx0 + jx) * ni + (cy0 + jy)) * ni + jz;
  };
  constexpr int kCo = 4;
  float cv[kCo];
  int ca[kCo];
#pragma unroll
  for (int u = 0; u < kCo; ++u) {
    const int e = tid + nthr * u;
    int lo;
    const size_t g = coarse_at(min(e, c_total - 1), lo);
    cv[u] = src[g];
    ca[u] = e < c_total ? lo : -1;
  }
  stage_to_lds_n(w_l, wmat + (size_t)co_tile * kpad * 16, kpad * 16, tid, nthr);
#pragma unroll
  for (int u = 0; u < kCo; ++u)
    if (ca[u] >= 0) coarse[ca[u]] = cv[u];
  for (int e = tid + nthr * kCo; e < c_total; e += nthr) {
    int lo;
    const size_t g = coarse_at(e, lo);
    coarse[lo] = src[g];
  }
  if (tid < patch_slack(ZT)) patch[patch_n + tid] = 0.0f;   // (read by the zero-weight padding taps)
  __syncthreads();
  // the fine patch: resize3_kernel's expression tree, z innermost
  const unsigned m_pz = magic_of(PZ);
  for (int e = tid; e < patch_n; e += nthr) {
    const int r0 = div_by(e, m_pz), zz = e - r0 * PZ;
    const int r1 = r0 / 3, b = r0 - 3 * r1, ci = r1 / 3, a = r1 - 3 * ci;
    int x0, x1, y0, y1, z0, z1;
    float lx, ly, lz;
    resize_axis(x + a, ratio, ni, x0, x1, lx);
    resize_axis(y + b, ratio, ni, y0, y1, ly);
    resize_axis(min(zz, n - 1), ratio, ni, z0, z1, lz);
    const float* p = coarse + (size_t)ci * CX * CX * ni;
#define AT(ix, iy, iz) p[(((ix) - cx0) * CX + ((iy) - cy0)) * ni + (iz)]
    const float wx0 = 1.0f - lx, wy0 = 1.0f - ly, wz0 = 1.0f - lz;
    patch[e] = blend(wx0, blend(wy0, blend(wz0, AT(x0, y0, z0), lz, AT(x0, y0, z1)),
                                ly, blend(wz0, AT(x0, y1, z0), lz, AT(x0, y1, z1))),
                     lx, blend(wy0, blend(wz0, AT(x1, y0, z0), lz, AT(x1, y0, z1)),
                               ly, blend(wz0, AT(x1, y1, z0), lz, AT(x1, y1, z1))));
#undef AT
  }
  __syncthreads();
  f32x4 acc;
  int tile;
  if (column_contract<true>(patch, w_l, red, kpad, PZ, ZT, acc, tile))
    column_store(acc, tile, bias, co_tile, Cout, relu, out, nb, m, x, y);
}

// Transposed trilinear resize n_out -> n_in (the VJP of an up-sampling resize), ReLU mask, [the transposed 1x1x1 layer
// that was swapped with the resize: COUT > 0, one source channel -> COUT channels,] zero padding by `pad` -- all of it on
// the fine block above one output column -- and the flipped-weight 3x3x3 convolution that reads the padded tensor:
//   g_out [N][C][n_out^3]  ->  (never stored) P [N][CP][np^3], np = n_in + 2 pad, CP = COUT > 0 ? COUT : C
//                          ->  out [N][Cc][nc^3], nc = np - 2
// Same chains as resize3_backward_tiled_kernel (per pass the non-zero taps in ascending source order, fmaf(w, v, acc)
// from 0; z, then y, then x; `fmaf(acc, w_mix, 0) + bias` for the 1x1 layer) and as conv3d_mfma_kernel (SPLIT or plain).
//   grid (nc * nc, co_tiles, N), block nthr (SPLIT: 256 * ZT .. 1024, plain: 256 .. 512), ZT = ceil(nc / 16)
//   the channels of g_out are taken CK at a time (C % CK may be anything)
//   LDS: w_l [kpad * 16] | red [SPLIT ? 1024 * ZT : 0] | patch [CP * 9 * PZ + patch_slack] | Y [CK][FX][3][n_in] | F [CK][FX][FX][n_out]
// FX: the longest run of fine indices that feed three neighbouring coarse ones (host).  n_out % 4 == 0, g_out 16-byte
// aligned, n_in <= 64, the exact source range of a coarse index <= TAPS long and its candidate range <= 16 (host).
template <int COUT, int TAPS, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 1024 : 512) void conv3d_mfma_tresize_kernel(
    const float* __restrict__ g_out, int C, int n_in, int n_out, const float* __restrict__ act, int pad,
    const float* __restrict__ mix_w, const float* __restrict__ mix_b, const float* __restrict__ wmat,
    const float* __restrict__ bias, float* __restrict__ out, int Cc, int kpad, int CK, int FX, int ZT) {
  constexpr int CO = COUT > 0 ? COUT : 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float w_tab[6 + 64][kBtTaps];   // rows: a = 0..2 (x), 3 + b (y), 6 + iz (z)
  __shared__ int d_tab[6 + 64], n_tab[6];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int co_tile = blockIdx.y, nb = blockIdx.z;
  const int np = n_in + 2 * pad, nc = np - 2;
  const int x = (int)blockIdx.x / nc, y = (int)blockIdx.x - x * nc;
  const int CP = COUT > 0 ? COUT : C;
  const int PZ = 16 * ZT + 2, patch_n = CP * 9 * PZ;
  float* w_l = lds;
  float* red = w_l + (size_t)kpad * 16;
  float* patch = red + (SPLIT ? 1024 * ZT : 0);
  float* Y = patch + ((patch_n + patch_slack(ZT) + 3) & ~3);
  float* F = Y + (((size_t)CK * FX * 3 * n_in + 3) & ~(size_t)3);
  const float ratio = (float)n_in / (float)n_out;
  // weight tables: a row per 16-lane group (resize_row16); rows of coarse indices outside the tensor are empty
  {
    const int k16 = tid & 15, g16 = lane & 48;
    for (int row = tid >> 4; row < 6 + n_in; row += nthr >> 4) {
      int i;
      if (row < 3) { i = x + row - pad; if (i < 0 || i >= n_in) i = -1; }
      else if (row < 6) { i = y + row - 3 - pad; if (i < 0 || i >= n_in) i = -1; }
      else i = row - 6;
      int d0, nt;
      float w;
      resize_row16(i, k16, g16, ratio, n_in, n_out, d0, nt, w);
      if (k16 == 0) {
        d_tab[row] = d0;
        if (row < 6) n_tab[row] = nt;
      }
      if (k16 < kBtTaps) w_tab[row][k16] = w;
    }
  }
  for (int e = tid; e < patch_n + patch_slack(ZT); e += nthr) patch[e] = 0.0f;   // the padding, and what the zero-weight taps read
  __syncthreads();
  // the fine block above the column: x in [fx0, fx0 + fnx), y in [fy0, fy0 + fny), all of z
  const int a_lo = max(pad - x, 0), a_hi = min(n_in - 1 - x + pad, 2);
  const int b_lo = max(pad - y, 0), b_hi = min(n_in - 1 - y + pad, 2);
  const int fx0 = d_tab[a_lo], fnx = d_tab[a_hi] + n_tab[a_hi] - fx0;
  const int fy0 = d_tab[3 + b_lo], fny = d_tab[3 + b_hi] + n_tab[3 + b_hi] - fy0;
  const int q4 = n_out >> 2;
  const size_t fine_vol = (size_t)n_out * n_out * n_out, coarse_vol = (size_t)n_in * n_in * n_in;
  const unsigned m_q = magic_of(q4), m_fy = magic_of(fny), m_fx = magic_of(fnx), m_ni = magic_of(n_in);
  constexpr int kFl = 8;
  f32x4 pre[kFl];
  auto fine_at = [&](int e, int& lds_off) {   // vector e of a round's block -> offsets (floats) in g_out's channel / in F
    const int r0 = div_by(e, m_q), q = e - r0 * q4;
    const int r1 = div_by(r0, m_fy), fy = r0 - r1 * fny;
    const int ck = div_by(r1, m_fx), fx = r1 - ck * fnx;
    lds_off = ((ck * FX + fx) * FX + fy) * n_out + 4 * q;
    return (size_t)ck * fine_vol + ((size_t)(fx0 + fx) * n_out + (fy0 + fy)) * n_out + 4 * q;
  };
  auto prefetch = [&](int c0) {
    const int total4 = min(CK, C - c0) * fnx * fny * q4;
    const float* src = g_out + ((size_t)nb * C + c0) * fine_vol;
#pragma unroll
    for (int j = 0; j < kFl; ++j) {
      int lo;
      const size_t g = fine_at(min(tid + nthr * j, total4 - 1), lo);
      pre[j] = *reinterpret_cast<const f32x4*>(src + g);
    }
  };
  prefetch(0);   // (in flight while the weights are staged: their wait covers both)
  stage_to_lds_n(w_l, wmat + (size_t)co_tile * kpad * 16, kpad * 16, tid, nthr);
  // lane -> (slot, iz): 64 / nl rows per wave step, nl = the power of two that holds a coarse row
  const int nl = n_in <= 8 ? 8 : (n_in <= 16 ? 16 : (n_in <= 32 ? 32 : 64));
  const int spw = 64 / nl, slot = lane / nl, iz = lane - slot * nl;
  const bool zok = iz < n_in;
  const int dz = zok ? d_tab[6 + iz] : 0;
  float wz[TAPS];
  int oz[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; ++k) {
    wz[k] = zok ? w_tab[6 + iz][k] : 0.0f;
    oz[k] = min(dz + k, n_out - 1);
  }
  for (int c0 = 0; c0 < C; c0 += CK) {
    const int ck_n = min(CK, C - c0);
    const int total4 = ck_n * fnx * fny * q4;
    if (c0 > 0) __syncthreads();   // the previous round's passes have left F and Y
#pragma unroll
    for (int j = 0; j < kFl; ++j) {
      const int e = tid + nthr * j;
      if (e < total4) {
        int lo;
        (void)fine_at(e, lo);
        *reinterpret_cast<f32x4*>(F + lo) = pre[j];
      }
    }
    if (total4 > kFl * nthr) {   // (blocks larger than the prefetch: the rest straight from memory)
      const float* src = g_out + ((size_t)nb * C + c0) * fine_vol;
      for (int e = tid + nthr * kFl; e < total4; e += nthr) {
        int lo;
        const size_t g = fine_at(e, lo);
        *reinterpret_cast<f32x4*>(F + lo) = *reinterpret_cast<const f32x4*>(src + g);
      }
    }
    __syncthreads();
    if (c0 + CK < C) prefetch(c0 + CK);
    {   // z pass, in place: row (ck, fx, fy) -> its head [0, n_in)
      const int rows = ck_n * fnx * fny;
      for (int r = wave * spw + slot; r < rows; r += nw * spw) {
        const int r1 = div_by(r, m_fy), fy = r - r1 * fny;
        const int ck = div_by(r1, m_fx), fx = r1 - ck * fnx;
        float* f = F + (size_t)((ck * FX + fx) * FX + fy) * n_out;
        float v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) v[k] = f[oz[k]];
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (wz[k] != 0.0f) ? fmaf(wz[k], v[k], acc) : acc;
        if (zok) f[iz] = acc;
      }
    }
    __syncthreads();
    {   // y pass: (ck, fx, b) -> Y[ck][fx][b][iz]
      const int trip = ck_n * fnx * 3;
      for (int r = wave * spw + slot; r < trip; r += nw * spw) {
        const int r1 = r / 3, b = r - 3 * r1;
        const int ck = div_by(r1, m_fx), fx = r1 - ck * fnx;
        const int d0 = d_tab[3 + b] - fy0;
        const float* zc = F + (size_t)((ck * FX + fx) * FX) * n_out + (zok ? iz : 0);
        float w[TAPS], v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
          w[k] = w_tab[3 + b][k];
          v[k] = zc[min(max(d0 + k, 0), fny - 1) * n_out];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (w[k] != 0.0f) ? fmaf(w[k], v[k], acc) : acc;
        if (zok) Y[((ck * FX + fx) * 3 + b) * n_in + iz] = acc;
      }
    }
    __syncthreads();
    {   // x pass, mask, (mix,) into the patch: items (ck, a, b, iz), inside the tensor only
      const int items = ck_n * 9 * n_in;
      for (int e = tid; e < items; e += nthr) {
        const int r0 = div_by(e, m_ni), jz = e - r0 * n_in;
        const int ck = r0 / 9, ab = r0 - 9 * ck, a = ab / 3, b = ab - 3 * a;
        const int ix = x + a - pad, iy = y + b - pad;
        if (ix < 0 || ix >= n_in || iy < 0 || iy >= n_in) continue;
        const size_t at = ((size_t)ix * n_in + iy) * n_in + jz;
        float mask[CO];
        if (act) {
#pragma unroll
          for (int co = 0; co < CO; ++co)
            mask[co] = act[((size_t)nb * CP + (COUT > 0 ? co : c0 + ck)) * coarse_vol + at];
        }
        const int d0 = d_tab[a] - fx0;
        const float* yc = Y + (size_t)((ck * FX) * 3 + b) * n_in + jz;
        float w[TAPS], v[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
          w[k] = w_tab[a][k];
          v[k] = yc[min(max(d0 + k, 0), fnx - 1) * 3 * n_in];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc = (w[k] != 0.0f) ? fmaf(w[k], v[k], acc) : acc;
        if (COUT == 0) {
          if (act && !(mask[0] > 0.0f)) acc = 0.0f;
          patch[((c0 + ck) * 9 + ab) * PZ + jz + pad] = acc;
        } else {
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            const float v1 = fmaf(acc, mix_w[co], 0.0f) + mix_b[co];
            patch[(co * 9 + ab) * PZ + jz + pad] = (!act || mask[co] > 0.0f) ? v1 : 0.0f;
          }
        }
      }
    }
  }
  __syncthreads();
  f32x4 acc;
  int tile;
  if (column_contract<SPLIT>(patch, w_l, red, kpad, PZ, ZT, acc, tile))
    column_store(acc, tile, bias, co_tile, Cc, 0, out, nb, nc, x, y);
}

// The Linear stack (narrow leading layers: fc_one_wave_ok) and the FIRST convolution (3x3x3, no resize in front of it:
// sdf_vae.py:223-238) in one launch: every workgroup runs the narrow layers as fc_stack_kernel<true> does, forms the
// rows of the wide layer under its column -- bias first, inputs in ascending order, ReLU: fc_stack_kernel's chain --
// straight into the operand patch, and contracts it (split-K order).  The wide layer's output is needed again by the VJP
// (its ReLU mask): with fc_out != NULL the column that owns a fine (x, y) -- (min(x, m - 1), min(y, m - 1)) -- stores it.
//   grid (m * m, co_tiles, N), block >= 256 * ZT;  LDS (dynamic): w_l [kpad * 16] | red [1024 * ZT] | patch [Cin * 9 * PZ + slack]
// VEC4: a thread forms four consecutive z of a row with 16-byte weight loads (n % 4 == 0, wide weights 16-byte aligned).
template <bool VEC4>
__global__ __launch_bounds__(1024) void fc_conv_kernel(const float* __restrict__ params, FcDesc d,
                                                       const float* __restrict__ z, float* __restrict__ fc_out,
                                                       const float* __restrict__ wmat, const float* __restrict__ bias,
                                                       float* __restrict__ out, int Cin, int Cout, int n, int m, int kpad,
                                                       int relu, int ZT) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float act[2][kFcWaveWidth];
  __shared__ float p_lds[kFcWaveSpan];
  const int PZ = 16 * ZT + 2, patch_n = Cin * 9 * PZ;
  float* w_l = lds;
  float* red = w_l + (size_t)kpad * 16;
  float* patch = red + 1024 * ZT;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int co_tile = blockIdx.y, nb = blockIdx.z;
  const int x = (int)blockIdx.x / m, y = (int)blockIdx.x - x * m;
  // the narrow layers (fc_stack_kernel<true>): their parameters to LDS with all loads in flight, then wave 0
  int cur = 0;
  {
    const long long base = d.w_off[0];
    const int span = (int)fc_wave_span(d);
    const float z_t = tid < d.width[0] ? z[(size_t)nb * d.width[0] + tid] : 0.0f;
    constexpr int U = kFcWaveSpan / kFcBlock;   // (enough for 256 threads; larger workgroups leave the tail idle)
    float r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = u * nthr + tid;
      r[u] = e < span ? params[base + e] : 0.0f;
    }
    stage_to_lds_n(w_l, wmat + (size_t)co_tile * kpad * 16, kpad * 16, tid, nthr);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = u * nthr + tid;
      if (e < span) p_lds[e] = r[u];
    }
    for (int e = tid; e < patch_n + patch_slack(ZT); e += nthr) patch[e] = 0.0f;
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
  }
  // the wide layer's rows under the column
  {
    const int l = d.n_fc - 1;
    const int win = d.width[l], wout = d.width[l + 1];
    const float* wt = params + d.w_off[l];   // transposed: [in][out]
    const float* bl = params + d.b_off[l];
    const int n3 = n * n * n;
    const int own_x = min(x, m - 1), own_y = min(y, m - 1);   // (this column's own coordinates: x < m, y < m)
    float* fo = (fc_out && co_tile == 0) ? fc_out + (size_t)nb * wout : nullptr;
    if (VEC4) {
      const int q4 = n >> 2, items = Cin * 9 * q4;
      const unsigned m_q = magic_of(q4);
      for (int e = tid; e < items; e += nthr) {
        const int r0 = div_by(e, m_q), zz = (e - r0 * q4) << 2;
        const int ci = r0 / 9, ab = r0 - 9 * ci, a = ab / 3, b = ab - 3 * a;
        const int o = ci * n3 + ((x + a) * n + (y + b)) * n + zz;
        f32x4 acc = *reinterpret_cast<const f32x4*>(bl + o);
#pragma unroll 16
        for (int i = 0; i < win; ++i) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(wt + (size_t)i * wout + o);
          const float h = act[cur][i];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[j], h, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.0f);
        float* pr = patch + r0 * PZ + zz;
#pragma unroll
        for (int j = 0; j < 4; ++j) pr[j] = acc[j];
        if (fo && min(x + a, m - 1) == own_x && min(y + b, m - 1) == own_y) *reinterpret_cast<f32x4*>(fo + o) = acc;
      }
    } else {
      const int items = Cin * 9 * n;
      const unsigned m_n = magic_of(n);
      for (int e = tid; e < items; e += nthr) {
        const int r0 = div_by(e, m_n), zz = e - r0 * n;
        const int ci = r0 / 9, ab = r0 - 9 * ci, a = ab / 3, b = ab - 3 * a;
        const int o = ci * n3 + ((x + a) * n + (y + b)) * n + zz;
        float acc = bl[o];
#pragma unroll 16
        for (int i = 0; i < win; ++i) acc = fmaf(wt[(size_t)i * wout + o], act[cur][i], acc);
        acc = fmaxf(acc, 0.0f);
        patch[r0 * PZ + zz] = acc;
        if (fo && min(x + a, m - 1) == own_x && min(y + b, m - 1) == own_y) fo[o] = acc;
      }
    }
  }
  __syncthreads();
  f32x4 acc;
  int tile;
  if (column_contract<true>(patch, w_l, red, kpad, PZ, ZT, acc, tile))
    column_store(acc, tile, bias, co_tile, Cout, relu, out, nb, m, x, y);
}
