// sampler_device.hpp -- the device side of the sampler's backward as a workgroup-level function, shared by
// pc_loss_backward_kernel (sampler.hip) and by the renderer's backward when the two run in ONE launch
// (render.hip, render_backward_pc_kernel: the captured loop's two independent backward passes side by side).
#pragma once
#include "common.hpp"
#include "device.hpp"

namespace sdfr {
namespace {

constexpr int kPts = kSamplerPts;  // points per workgroup
using SamplerHash = PairRunHash<512>;   // z-pair runs: one LDS add per column of a cell (device.hpp)

struct PointFrame {
  float qn[4];
  float inv_norm;
  float rot[9];  // R(q^) row-major
  V3 p;
  float scale;
};

__device__ __forceinline__ PointFrame load_frame(const float* __restrict__ pos,
                                                 const float* __restrict__ quat,
                                                 const float* __restrict__ scale, int v) {
  PointFrame f;
  const float x0 = quat[4 * v], y0 = quat[4 * v + 1], z0 = quat[4 * v + 2], w0 = quat[4 * v + 3];
  const float n2 = x0 * x0 + y0 * y0 + z0 * z0 + w0 * w0;
  f.inv_norm = 1.0f / sqrtf(n2);
  const float x = x0 * f.inv_norm, y = y0 * f.inv_norm, z = z0 * f.inv_norm, w = w0 * f.inv_norm;
  f.qn[0] = x; f.qn[1] = y; f.qn[2] = z; f.qn[3] = w;
  f.rot[0] = 1 - 2 * (y * y + z * z); f.rot[1] = 2 * (x * y - w * z);     f.rot[2] = 2 * (x * z + w * y);
  f.rot[3] = 2 * (x * y + w * z);     f.rot[4] = 1 - 2 * (x * x + z * z); f.rot[5] = 2 * (y * z - w * x);
  f.rot[6] = 2 * (x * z - w * y);     f.rot[7] = 2 * (y * z + w * x);     f.rot[8] = 1 - 2 * (x * x + y * y);
  f.p = mk(pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]);
  f.scale = scale[v];
  return f;
}

// object-frame point o = R^T (P - p) and the cell; returns false for a masked (outside) point
template <int RT>
__device__ __forceinline__ bool sample_cell(const PointFrame& f, const float* __restrict__ vol,
                                            int R, V3 P, V3& vrel, V3& o, Cell& c) {
  const int Rr = RT > 0 ? RT : R;
  vrel = P - f.p;
  o = mk(fmaf(f.rot[0], vrel.x, fmaf(f.rot[3], vrel.y, f.rot[6] * vrel.z)),
         fmaf(f.rot[1], vrel.x, fmaf(f.rot[4], vrel.y, f.rot[7] * vrel.z)),
         fmaf(f.rot[2], vrel.x, fmaf(f.rot[5], vrel.y, f.rot[8] * vrel.z)));
  const float h = 0.5f * (float)(Rr - 1);
  const float gx = (o.x / f.scale + 1.0f) * h, gy = (o.y / f.scale + 1.0f) * h,
              gz = (o.z / f.scale + 1.0f) * h;
  const float top = (float)(Rr - 2);
  const float cxf = floorf(gx), cyf = floorf(gy), czf = floorf(gz);
  const bool inside = !(cxf < 0.0f) && !(cyf < 0.0f) && !(czf < 0.0f) && !(cxf > top) && !(cyf > top) &&
                      !(czf > top) && (gx == gx) && (gy == gy) && (gz == gz);
  gather_cell<RT>(vol, R, gx, gy, gz, c);  // clamps the cell, so the loads are always safe
  return inside;
}


struct PcBackwardArgs {
  const float* grad_out;   // plain form: upstream gradient per point
  const float* points;
  const int* offsets;
  int n_single;
  const float* pos;
  const float* quat;
  const float* scale;
  const float* sdf;
  int R;
  long long sdf_view_stride;
  float* g_sdf;
  long long g_sdf_view_stride;
  float* partials;
  int nblk;
  float l1_weight;
  float* loss_part;
  int groups = 0;   // workgroups per view, striding over the view's blocks (0: one workgroup per block, groups = nblk)
};

struct PcBackwardLds {
  // 4-voxel runs x 512 slots: back-projected depth images are coherent (measured on 64 rendered
  // views, 1.13 M points: 122 -> 99 us against 2 x 1024; uniformly random points 80 -> 83 us)
  SamplerHash hash;
  float wave_part[kPts / 64][8];
  float wave_abs[kPts / 64];
  int blk_max_bits;
};

// L1: the upstream gradient is not read but formed here, for the loss  weight * mean |value|  over
// the view's points (simple_setup.py:144): go = +-weight / M_v by the sign of the point's value, and
// the block's sum of |value| goes to `loss_part` -- the loop then needs neither the sampler's
// forward launch nor the loss launch.
// DET (SDFR_SDF_GRAD_DETERMINISTIC, render.hip): `g_sdf` is the 64-bit fixed-point volume; every point's eight
// contributions are rounded once to the quantum 2^-SDFR_FIXED_QUANTUM_BITS and added as integers, straight to the
// volume (no LDS table: the mode is for reproducible runs, not for speed).
// GROUPS (a.groups): the grid has `groups` workgroups per view and workgroup bx takes the blocks bx, bx + groups, ...
// of the view, one after the other, each through a table of its own (clear .. flush) -- so the grid need not grow
// with the CAPACITY of the point buffers (sdfr_depth_to_points_resident: room for every pixel, 1 200 blocks per
// 640x480 view, of which a mug fills ~60).  (Several blocks through ONE table, flushed once, was measured and
// dropped: the block is a dependent chain and every block added to it costs its full length, profiles/r05_pc_rounds.md.)
// DIRECT: every point's eight contributions go straight to the volume's float atomics -- no table, no block maximum,
// two barriers less per block: for launches of a few dozen blocks (a small object in one or two views of the captured
// loop), where what a block costs is the depth of its dependent chain and not the number of atomics it sends.
// SDFG = false: nobody wants d/dSDF (a loop that does not optimise the shape) -- pose sums and the loss only.
template <int RT, bool L1, bool DET = false, bool DIRECT = false, bool SDFG = true>
__device__ __forceinline__ void pc_backward_block(PcBackwardLds& lds, const PcBackwardArgs& a, int bx, int v) {
  SamplerHash& hash = lds.hash;
  float (&wave_part)[kPts / 64][8] = lds.wave_part;
  float (&wave_abs)[kPts / 64] = lds.wave_abs;
  int& blk_max_bits = lds.blk_max_bits;
  const float* __restrict__ grad_out = a.grad_out;
  const float* __restrict__ points = a.points;
  const int* __restrict__ offsets = a.offsets;
  const float* __restrict__ pos = a.pos;
  const float* __restrict__ quat = a.quat;
  const float* __restrict__ scale = a.scale;
  const float* __restrict__ sdf = a.sdf;
  float* __restrict__ g_sdf = a.g_sdf;
  float* __restrict__ partials = a.partials;
  float* __restrict__ loss_part = a.loss_part;
  const int n_single = a.n_single, R = a.R, nblk = a.nblk;
  const long long sdf_view_stride = a.sdf_view_stride, g_sdf_view_stride = a.g_sdf_view_stride;
  const float l1_weight = a.l1_weight;
  const int groups = a.groups > 0 ? a.groups : nblk;

  const int Rr = RT > 0 ? RT : R;
  const int begin = offsets ? offsets[v] : 0;
  const int end = offsets ? offsets[v + 1] : n_single;
  if (bx * kPts >= end - begin) return;  // the reducer never reads these blocks' slots
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const PointFrame f = load_frame(pos, quat, scale, v);
  const float* vol = sdf + (size_t)v * sdf_view_stride;
  float* gvol = g_sdf + (size_t)v * g_sdf_view_stride;

#pragma unroll 1
  for (int blk = bx; blk * kPts < end - begin; blk += groups) {   // (workgroup-uniform)
    const int i = begin + blk * kPts + tid;
    if (!DET && !DIRECT) hash.clear(tid, kPts);   // (after the previous block's flush: its reads end at the barrier below)
    if (!DIRECT && tid == 0) blk_max_bits = 0;

    bool live = false;
    float go = 0.0f, l1_abs = 0.0f;
    V3 vrel = mk(0, 0, 0), o = mk(0, 0, 0);
    Cell c;
    c.lin = 0; c.ox = c.oy = c.oz = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.v[k] = 0.0f;
    if (i < end) {
      const V3 P = mk(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
      live = sample_cell<RT>(f, vol, R, P, vrel, o, c);
      if (L1) {
        const float val = live ? trilerp(c) * f.scale : 0.0f;  // losses.py:133-135: masked values are 0
        const float k = l1_weight / (float)(end - begin);       // as pc_l1_kernel (loop.hip)
        go = val > 0.0f ? k : (val < 0.0f ? -k : 0.0f);
        l1_abs = fabsf(val);
      } else {
        go = live ? grad_out[i] : 0.0f;
      }
    }
    if (!DIRECT) {
      __syncthreads();   // (table cleared, blk_max_bits reset; the previous block's wave_part reads are done)
      const float gmax = wave_max(fabsf(go));
      if (lane == 0) atomicMax(&blk_max_bits, __float_as_int(gmax));
      __syncthreads();
    }

    // the fixed-point scale: from the block's largest |go|
    const float bound = DIRECT ? 0.0f : 2.0f * __int_as_float(blk_max_bits) * fabsf(f.scale);
    int e2;
    (void)frexpf(bound, &e2);
    const bool fixed_ok = !DIRECT && (bound > 0.0f) && (bound < 1e30f) && (e2 > -80);
    const float to_fixed = fixed_ok ? ldexpf(1.0f, SamplerHash::kBits - e2) : 0.0f;
    const float from_fixed = ldexpf(1.0f, e2 - SamplerHash::kBits);
    const float weight_limit = SamplerHash::kWeightLimit;

    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
      const float gsz = 2.0f / (float)(Rr - 1);  // grid size
      const float tri = trilerp(c);
      const float ax = 1.0f - c.ox, ay = 1.0f - c.oy, az = 1.0f - c.oz;
      const float c00 = fmaf(c.v[4], c.ox, c.v[0] * ax), c01 = fmaf(c.v[5], c.ox, c.v[1] * ax);
      const float c10 = fmaf(c.v[6], c.ox, c.v[2] * ax), c11 = fmaf(c.v[7], c.ox, c.v[3] * ax);
      V3 G;  // d tri / d (cell coordinate)
      G.x = ((c.v[4] - c.v[0]) * ay + (c.v[6] - c.v[2]) * c.oy) * az +
            ((c.v[5] - c.v[1]) * ay + (c.v[7] - c.v[3]) * c.oy) * c.oz;
      G.y = (c10 - c00) * az + (c11 - c01) * c.oz;
      G.z = fmaf(c11, c.oy, c01 * ay) - fmaf(c10, c.oy, c00 * ay);
      // value = tri(off) * scale, off = (o/scale - cellpos)/g  =>  d value / d o = G / g
      const V3 dvo = mk(G.x / gsz, G.y / gsz, G.z / gsz);
      // d value / d scale = tri - (dvo . o) / scale
      acc[7] = go * (tri - dot(dvo, o) / f.scale);
      // o = R^T (P - p): d/dp = -R dvo
      acc[0] = -go * fmaf(f.rot[0], dvo.x, fmaf(f.rot[1], dvo.y, f.rot[2] * dvo.z));
      acc[1] = -go * fmaf(f.rot[3], dvo.x, fmaf(f.rot[4], dvo.y, f.rot[5] * dvo.z));
      acc[2] = -go * fmaf(f.rot[6], dvo.x, fmaf(f.rot[7], dvo.y, f.rot[8] * dvo.z));
      // R^T v = (1 - 2|u|^2) v + 2 u (u.v) - 2 w (u x v)  (the matrix form of losses.py:65-77):
      //   d/du_k = -4 u_k v + 2 e_k (u.v) + 2 u v_k - 2 w (e_k x v),   d/dw = -2 (u x v)
      const V3 u = mk(f.qn[0], f.qn[1], f.qn[2]);
      const float w = f.qn[3];
      const float udv = dot(u, vrel), Dv = dot(dvo, vrel), Du = dot(dvo, u);
      const V3 vxD = cross(vrel, dvo);  // dvo . (e_k x v) = (v x dvo)_k
      acc[3] = go * (-4.0f * u.x * Dv + 2.0f * udv * dvo.x + 2.0f * vrel.x * Du - 2.0f * w * vxD.x);
      acc[4] = go * (-4.0f * u.y * Dv + 2.0f * udv * dvo.y + 2.0f * vrel.y * Du - 2.0f * w * vxD.y);
      acc[5] = go * (-4.0f * u.z * Dv + 2.0f * udv * dvo.z + 2.0f * vrel.z * Du - 2.0f * w * vxD.z);
      acc[6] = go * (-2.0f * dot(dvo, cross(u, vrel)));

      // d/dsdf: go * scale * trilinear weight, through the fixed-point run-hash
      const float gs = go * f.scale;
      const float x0w = ax * gs, x1w = c.ox * gs;
      const float w0 = x0w * ay * az, w1 = x0w * ay * c.oz, w2 = x0w * c.oy * az, w3 = x0w * c.oy * c.oz;
      const float w4 = x1w * ay * az, w5 = x1w * ay * c.oz, w6 = x1w * c.oy * az, w7 = x1w * c.oy * c.oz;
      // (a NaN upstream gradient can be dropped by the block's fmaxf-based maximum: such a lane, like
      // any lane beyond the fixed-point range, adds in float, so NaN/Inf reach g_sdf as in autograd)
      if (!SDFG) {
      } else if (DET) {
        // (not finite: the conversion saturates, NaN counts as 0 -- include/sdfr.h)
        if (go != 0.0f) {
          unsigned long long* g0 = reinterpret_cast<unsigned long long*>(gvol) + c.lin;
          const float wk[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
          const float q = (float)(1ll << SDFR_FIXED_QUANTUM_BITS);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            atomicAdd(g0 + ((j & 4) ? Rr * Rr : 0) + ((j & 2) ? Rr : 0) + (j & 1),
                      (unsigned long long)__float2ll_rn(wk[j] * q));
        }
      } else if (fixed_ok && fabsf(gs) * to_fixed < weight_limit) {
        const float wk[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
        hash.add_cell(gvol, c.lin, Rr, wk, to_fixed);
      } else if (go != 0.0f) {
        float* g0 = gvol + c.lin;
        atomicAdd(g0, w0);                atomicAdd(g0 + 1, w1);
        atomicAdd(g0 + Rr, w2);           atomicAdd(g0 + Rr + 1, w3);
        atomicAdd(g0 + Rr * Rr, w4);      atomicAdd(g0 + Rr * Rr + 1, w5);
        atomicAdd(g0 + Rr * Rr + Rr, w6); atomicAdd(g0 + Rr * Rr + Rr + 1, w7);
      }
    }
    {
      const float sk = wave_sum8(acc, lane);
      if ((lane & 7) == 0) wave_part[wave][lane >> 3] = sk;
    }
    if (L1) {
      const float sa = wave_sum(l1_abs);
      if (lane == 0) wave_abs[wave] = sa;
    }
    __syncthreads();
    if (tid < 8) {
      float t = 0.0f;
#pragma unroll
      for (int wv = 0; wv < kPts / 64; ++wv) t += wave_part[wv][tid];
      partials[((size_t)v * nblk + blk) * 8 + tid] = t;
    }
    if (L1 && tid == 0) {
      float t = 0.0f;
#pragma unroll
      for (int wv = 0; wv < kPts / 64; ++wv) t += wave_abs[wv];
      loss_part[(size_t)v * nblk + blk] = t;
    }
    // (the barrier above orders every table add before the flush's reads)
    if (!DET && !DIRECT) hash.flush(gvol, Rr * Rr * Rr, from_fixed, tid, kPts);
    __syncthreads();   // the flush's reads before the next block's clear (DIRECT: wave_part's before its next writes)
  }
}


}  // namespace
}  // namespace sdfr
