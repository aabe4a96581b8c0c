// device.hpp -- device-side pieces shared by the render and sampler kernels (gfx950 only).
#pragma once

#include "common.hpp"

namespace sdfr {
namespace {

// ---------------------------------------------------------------------------------------------
// shared device pieces
// ---------------------------------------------------------------------------------------------
// unit ray through a pixel centre (camera frame, OpenGL).  sdf_renderer_cuda.cu:137-154.
__device__ __forceinline__ V3 pixel_ray(int row, int col, float cx, float cy, float rfx, float rfy) {
  const float dx = ((float)col + 0.5f - cx) * rfx;
  const float dy = -((float)row + 0.5f - cy) * rfy;
  const float inv_len = __builtin_amdgcn_rsqf(fmaf(dx, dx, fmaf(dy, dy, 1.0f)));
  return mk(dx * inv_len, dy * inv_len, -inv_len);
}

__device__ __forceinline__ V3 rot_t(const ViewSetup& s, V3 v) {  // R^T v
  return mk(fmaf(s.rot[0], v.x, fmaf(s.rot[3], v.y, s.rot[6] * v.z)),
            fmaf(s.rot[1], v.x, fmaf(s.rot[4], v.y, s.rot[7] * v.z)),
            fmaf(s.rot[2], v.x, fmaf(s.rot[5], v.y, s.rot[8] * v.z)));
}
__device__ __forceinline__ V3 rot_f(const ViewSetup& s, V3 v) {  // R v
  return mk(fmaf(s.rot[0], v.x, fmaf(s.rot[1], v.y, s.rot[2] * v.z)),
            fmaf(s.rot[3], v.x, fmaf(s.rot[4], v.y, s.rot[5] * v.z)),
            fmaf(s.rot[6], v.x, fmaf(s.rot[7], v.y, s.rot[8] * v.z)));
}

// v_min_f32 / v_max_f32 as they are (IEEE mode: the result is the non-NaN operand, a signalling NaN comes out
// quiet): fminf / fmaxf compile to the same instruction PLUS a canonicalising v_max(x, x) per operand the compiler
// cannot prove quiet.
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// z-adjacent corner pair: 8-byte load at 4-byte alignment (global memory allows it)
struct __attribute__((packed, aligned(4))) ZPair {
  float lo, hi;
};

struct Cell {
  float v[8];  // corner values, index 4*ix + 2*iy + iz
  float ox, oy, oz;
  float bx, by, bz;  // the cell's corner 000 (integers in float)
  int lin;  // linear index of corner 000
};

// Locate the cell of a grid-space point and fetch its 8 corners from the plain grid (four z-pair
// loads).  Same cell choice and the same un-clamped (extrapolating) cell coordinate as
// sdf_renderer_cuda.cu:196-239.  Used once per hit pixel by the backward and once per point by the
// sampler; the forward's march has its own sample routine (march_sample).
template <int RT>
__device__ __forceinline__ void gather_cell(const float* __restrict__ src, int R, float gx, float gy,
                                            float gz, Cell& c) {
  const int Rr = RT > 0 ? RT : R;
  const float top = (float)(Rr - 2);
  const float bx = fminf(fmaxf(floorf(gx), 0.0f), top);
  const float by = fminf(fmaxf(floorf(gy), 0.0f), top);
  const float bz = fminf(fmaxf(floorf(gz), 0.0f), top);
  c.ox = gx - bx; c.oy = gy - by; c.oz = gz - bz;
  c.bx = bx; c.by = by; c.bz = bz;
  // linear index formed in float (exact: < 2^24 for R <= 256) -- one conversion instead of three
  const int lin = (Rr <= 256) ? (int)fmaf(fmaf(bx, (float)Rr, by), (float)Rr, bz)
                              : ((int)bx * Rr + (int)by) * Rr + (int)bz;
  c.lin = lin;
  const float* base = src + lin;
  const ZPair p00 = *reinterpret_cast<const ZPair*>(base);
  const ZPair p01 = *reinterpret_cast<const ZPair*>(base + Rr);
  const ZPair p10 = *reinterpret_cast<const ZPair*>(base + Rr * Rr);
  const ZPair p11 = *reinterpret_cast<const ZPair*>(base + Rr * Rr + Rr);
  c.v[0] = p00.lo; c.v[1] = p00.hi; c.v[2] = p01.lo; c.v[3] = p01.hi;
  c.v[4] = p10.lo; c.v[5] = p10.hi; c.v[6] = p11.lo; c.v[7] = p11.hi;
}

// Face records are stored x-major with 2 x 2 blocks in (y, z): a 64-byte chunk of the record array
// holds the faces of a 2 x 2 neighbourhood, so the cells a wave touches in one step fall into fewer
// chunks -- the gather unit costs ~2 cycles per distinct 64-byte chunk per load
// (tools/microbench/gather_bench.hip).  Measured: forward 247.5 -> 232.3 us per 256 random views.
// (Three arrays -- faces perpendicular to x, y, z, each view marching the one for its dominant
// viewing axis so that the 2 x 2 blocks lie in the plane its rays sweep -- were built and measured:
// 261 us.  12 MiB of records no longer fit the 4 MiB per-XCD L2.)
// Hb = ceil(R/2) blocks per axis; one x-slab = 4 Hb^2 records.
constexpr bool kRecordsBlocked = true;
__host__ __device__ __forceinline__ int record_slab(int R) {
  return kRecordsBlocked ? 4 * ((R + 1) >> 1) * ((R + 1) >> 1) : R * R;
}
__device__ __forceinline__ int record_index(int x, int y, int z, int Hb) {
  if (!kRecordsBlocked) return (x * (2 * Hb) + y) * (2 * Hb) + z;  // timing only: even R
  return (((x * Hb + (y >> 1)) * Hb + (z >> 1)) << 2) | ((y & 1) << 1) | (z & 1);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// The march's sample: trilinear SDF value at a grid-space point, written for the VALU.
//
// In round 1 rocprofv3 showed the forward kernel issuing VALU instructions in ~97 % of its SIMD cycles
// (SQ_ACTIVE_INST_VALU x 4 vs. busy cycles) at 49 VALU per step; with ~30 per step the VALU is now second to the
// two gathers.  This version does the same
// arithmetic, bit for bit (same floor/clamp, same (1-f)*a + f*b lerps in the order x, y, z), in
// 2-wide packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 run two fp32 lanes per issue
// slot on gfx950) and forms the record index in float (exact below 2^24) with a bit permute
// instead of three conversions and ten integer ops.
// `src`: buffer descriptor of the face records (PACKED) or of the plain grid.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ constexpr bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

template <int RT, bool PACKED>
__device__ __forceinline__ float march_sample(__amdgpu_buffer_rsrc_t src, int R, f32x2 gxy, float gz) {
  const int Rr = RT > 0 ? RT : R;
  const float top = (float)(Rr - 2), fR = (float)Rr;
  const f32x2 bxy = {fminf(fmaxf(floorf(gxy.x), 0.0f), top), fminf(fmaxf(floorf(gxy.y), 0.0f), top)};
  const float bz = fminf(fmaxf(floorf(gz), 0.0f), top);
  const f32x2 oxy = gxy - bxy, axy = 1.0f - oxy;
  const float oz = gz - bz, az = 1.0f - oz;
  // linear index of corner 000, formed in float where that is exact (R^3 <= 2^24)
  const int lin = (Rr <= 256) ? (int)fmaf(fmaf(bxy.x, fR, bxy.y), fR, bz)
                              : ((int)bxy.x * Rr + (int)bxy.y) * Rr + (int)bz;
  f32x2 a01, a23, b01, b23;  // corner pairs (z, z+1) at (x,y), (x,y+1), (x+1,y), (x+1,y+1)
  if (PACKED) {
    int rix;
    if (!kRecordsBlocked) {
      rix = lin;
    } else if (RT > 0 && is_pow2(RT) && RT >= 4) {
      // lin = [x | y5..y1 y0 | z5..z1 z0]  ->  record_index = [x | y5..y1 | z5..z1 | y0 | z0]
      // as two bit-field inserts (the compiler expands the C form into and/and/and/or3)
      constexpr int L = __builtin_ctz(RT > 0 ? RT : 2);
      constexpr int zhi = ((1 << L) - 2) << 1;  // where z5..z1 land
      int t;
      asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(t) : "s"(zhi), "v"(lin << 1), "v"(lin));  // VOP3: no literals
      asm("v_bfi_b32 %0, 2, %1, %2" : "=v"(rix) : "v"(lin >> (L - 1)), "v"(t));
    } else {
      rix = record_index((int)bxy.x, (int)bxy.y, (int)bz, (Rr + 1) >> 1);
    }
    const i32x4 a = __builtin_amdgcn_raw_buffer_load_b128(src, rix * 16, 0, 0);
    const i32x4 b = __builtin_amdgcn_raw_buffer_load_b128(src, rix * 16, record_slab(Rr) * 16, 0);
    a01 = f32x2{__int_as_float(a.x), __int_as_float(a.y)};
    a23 = f32x2{__int_as_float(a.z), __int_as_float(a.w)};
    b01 = f32x2{__int_as_float(b.x), __int_as_float(b.y)};
    b23 = f32x2{__int_as_float(b.z), __int_as_float(b.w)};
  } else {
    // plain grid: four z-pairs, 8-byte loads at 4-byte alignment
    const int off = lin * 4;
    const i32x2 p00 = __builtin_amdgcn_raw_buffer_load_b64(src, off, 0, 0);
    const i32x2 p01 = __builtin_amdgcn_raw_buffer_load_b64(src, off, Rr * 4, 0);
    const i32x2 p10 = __builtin_amdgcn_raw_buffer_load_b64(src, off, Rr * Rr * 4, 0);
    const i32x2 p11 = __builtin_amdgcn_raw_buffer_load_b64(src, off, (Rr * Rr + Rr) * 4, 0);
    a01 = f32x2{__int_as_float(p00.x), __int_as_float(p00.y)};
    a23 = f32x2{__int_as_float(p01.x), __int_as_float(p01.y)};
    b01 = f32x2{__int_as_float(p10.x), __int_as_float(p10.y)};
    b23 = f32x2{__int_as_float(p11.x), __int_as_float(p11.y)};
  }
  // lerp x (both z of the y and the y+1 edge), then y, then z: sdf_renderer_cuda.cu:231-238
  const f32x2 ox2 = {oxy.x, oxy.x}, ax2 = {axy.x, axy.x}, oy2 = {oxy.y, oxy.y}, ay2 = {axy.y, axy.y};
  const f32x2 c0 = __builtin_elementwise_fma(b01, ox2, a01 * ax2);
  const f32x2 c1 = __builtin_elementwise_fma(b23, ox2, a23 * ax2);
  const f32x2 c = __builtin_elementwise_fma(c1, oy2, c0 * ay2);
  return fmaf(c.y, oz, c.x * az);
}

// trilinear value, lerp order x, y, z (sdf_renderer_cuda.cu:231-238)
__device__ __forceinline__ float trilerp(const Cell& c) {
  const float ax = 1.0f - c.ox, ay = 1.0f - c.oy, az = 1.0f - c.oz;
  const float c00 = fmaf(c.v[4], c.ox, c.v[0] * ax);
  const float c01 = fmaf(c.v[5], c.ox, c.v[1] * ax);
  const float c10 = fmaf(c.v[6], c.ox, c.v[2] * ax);
  const float c11 = fmaf(c.v[7], c.ox, c.v[3] * ax);
  const float c0 = fmaf(c10, c.oy, c00 * ay);
  const float c1 = fmaf(c11, c.oy, c01 * ay);
  return fmaf(c1, c.oz, c0 * az);
}

// lane -> pixel of the wave's 8x8 patch, row-major: a row of 8 pixels is one 32-byte segment for
// the image loads and stores.  (A Morton order -- 2x2 pixel blocks per lane quad, hoping that quad
// lanes share a record -- was measured: no gain for the gathers, whose cost follows the distinct
// 64-byte chunks of the whole wave, and 2-4 % slower because the image accesses split into 8-byte
// pieces.)
// A wave's pixel patch is PW x (64/PW): 8x8 in the forward (most coherent ray bundle: 227 vs
// 229 us for 16x4, 250 for 32x2), 16x4 in the backward (64-byte row segments for the two image
// reads: 186 vs 195 us).  4 waves tile a 32 x 8 sub-tile.
constexpr int kPatchWFwd = 8, kPatchWBwd = 16;
template <int PW> struct Patch {
  static constexpr int W = PW, H = 64 / PW, waves_x = kSubW / PW;
  static __device__ __forceinline__ int ox(int wave) { return (wave % waves_x) * W; }
  static __device__ __forceinline__ int oy(int wave) { return (wave / waves_x) * H; }
  static __device__ __forceinline__ int x(int lane) { return lane % W; }
  static __device__ __forceinline__ int y(int lane) { return lane / W; }
};
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sums of 8 per-lane values over the wave with 10 cross-lane exchanges instead of 48: a butterfly
// that halves the payload at each of the first three steps (lanes 32 apart split the 8 values 4/4,
// lanes 16 apart 2/2, lanes 8 apart 1/1), then three plain steps on the one value left.  Lane 8k
// (and the 7 lanes after it) ends up with the total of value k.  Fixed order: reproducible.
__device__ __forceinline__ float wave_sum8(const float (&v)[8], int lane) {
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
  float a[4], b[2], c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float keep = b5 ? v[j + 4] : v[j], send = b5 ? v[j] : v[j + 4];
    a[j] = keep + __shfl_xor(send, 32, 64);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float keep = b4 ? a[j + 2] : a[j], send = b4 ? a[j] : a[j + 2];
    b[j] = keep + __shfl_xor(send, 16, 64);
  }
  {
    const float keep = b3 ? b[1] : b[0], send = b3 ? b[0] : b[1];
    c = keep + __shfl_xor(send, 8, 64);
  }
  c += __shfl_xor(c, 4, 64);
  c += __shfl_xor(c, 2, 64);
  c += __shfl_xor(c, 1, 64);
  return c;
}

// LDS hash of short z-runs of the gradient volume: key = linear voxel index >> SHIFT.
//
// Sums are kept in 64-bit FIXED POINT, not float: on gfx950 ds_add_f32 executes one lane at a
// time (~185 cycles per wave-instruction even without conflicts, measured), ds_add_u64 takes ~8.
// Every contribution of a tile is bounded by M = max|grad_depth| * scale over its hit pixels
// (trilinear weights <= 1, |d.z| <= 1), a voxel receives at most 512 pixels x 8 corners of them,
// so with 2^-e >= M the integers round(c * 2^(44-e)) sum to less than 2^58 in magnitude.
// The scale is a power of two (exact in float), every float contribution is represented exactly
// (24 significant bits), and integer addition is associative: a tile's sums are exact and
// independent of the order in which its lanes arrive.
//
// Geometry (measured, backward of 256 views / of one 160x120 view):  8-voxel runs x 256 slots
// 188 us / 47 us;  4 x 512: 180 / 39;  2 x 1024: 227 / 20;  4 x 1024: 278 / 20.  A table that
// overflows falls back to global float atomics (slow: low-resolution views touch a new cell with
// every pixel), a big table costs occupancy.  Batches use 4 x 512, small calls 2 x 1024.
constexpr int kFixedBits = 44;

// float -> 64-bit integer (round to nearest) for |x| < 2^51.  gfx950 has no f32 -> i64 conversion;
// the compiler's expansion is ~12 VALU instructions and a tile performs 8 per hit pixel (a sixth of
// the backward's VALU work).  Adding 1.5 * 2^52 in double leaves the integer in the mantissa:
// convert, add, and one 32-bit subtract on the high word.
__device__ __forceinline__ long long fixed_from_float(float x) {
  const double d = (double)x + 6755399441055744.0;
  return __double_as_longlong(d) - 0x4338000000000000LL;
}

template <int SHIFT, int SLOTS>
struct RunHash {
  static constexpr int kShift = SHIFT, kSlots = SLOTS, kLen = 1 << SHIFT;
  static constexpr int kBits = kFixedBits;                 // contributions are scaled to < 2^kBits
  static constexpr float kWeightLimit = 3.5e13f;           // 2^45: larger scaled weights bypass the table
  unsigned long long vals[SLOTS << SHIFT];
  int keys[SLOTS];

  __device__ __forceinline__ void clear(int tid, int nthreads) {
    for (int i = tid; i < SLOTS; i += nthreads) keys[i] = -1;
    for (int i = tid; i < (SLOTS << SHIFT); i += nthreads) vals[i] = 0ull;
  }

  // Slot of run `key`: a plain LDS read first -- once a run has been installed (by an earlier lane
  // or sub-tile; several pixels share a cell) the look-up needs no atomic at all -- and a compare-
  // and-swap only on an empty slot.  Keys never change once written, so a stale read can only be
  // "empty".  (Measured: CAS on every look-up cost 96 of the backward's 267 us.)
  __device__ __forceinline__ int slot_of(int key) {
    unsigned h = (((unsigned)key * 2654435761u) >> 16) & (SLOTS - 1);
#pragma unroll 1
    for (int probe = 0; probe < 32; ++probe) {
      int cur = __builtin_nontemporal_load(&keys[h]);  // not cached in a register across probes
      if (cur == -1) cur = atomicCAS(&keys[h], -1, key);
      if (cur == -1 || cur == key) return (int)h;
      h = (h + 1) & (SLOTS - 1);
    }
    return -1;
  }

  // one fixed-point add into a resolved slot, or a global atomic when the table is crowded.
  // DET (deterministic d/dSDF, render.hip): `gvol` is the 64-bit fixed-point volume and everything that reaches it
  // is an integer -- sums that do not depend on the order of arrival, across tiles, views and ranks.
  template <bool DET = false>
  __device__ __forceinline__ void add_one(float* __restrict__ gvol, int slot, int lin, float w,
                                          float to_fixed) {
    if (slot >= 0)
      atomicAdd(&vals[(slot << SHIFT) + (lin & (kLen - 1))], (unsigned long long)fixed_from_float(w * to_fixed));
    else if (DET)
      atomicAdd(reinterpret_cast<unsigned long long*>(gvol) + lin, (unsigned long long)fixed_from_float(w * to_fixed));
    else
      atomicAdd(gvol + lin, w);
  }

  // Add the 8 corner contributions of the cell at `lin` (corner order 000,001,010,011,100,...).
  // The four z-pairs live in four runs (more when a pair straddles two runs): their first-probe key
  // reads are issued together and waited for once; only a miss takes the probing loop.  The
  // look-ups are a latency chain per lane (LDS round trip each), so overlapping them matters more
  // than the atomic's own cost.
  template <bool DET = false>
  __device__ __forceinline__ void add_cell(float* __restrict__ gvol, int lin, int Rr,
                                           const float (&w)[8], float to_fixed) {
    const int col[4] = {lin, lin + Rr, lin + Rr * Rr, lin + Rr * Rr + Rr};
    int key[4], cur[4];
    unsigned h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      key[j] = col[j] >> SHIFT;
      h[j] = (((unsigned)key[j] * 2654435761u) >> 16) & (SLOTS - 1);
      cur[j] = __builtin_nontemporal_load(&keys[h[j]]);
    }
    int slot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[j] = (cur[j] == key[j]) ? (int)h[j] : slot_of(key[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      add_one<DET>(gvol, slot[j], col[j], w[2 * j], to_fixed);
      const bool split = (col[j] & (kLen - 1)) == kLen - 1;  // the +z corner starts the next run
      const int s_hi = split ? slot_of(key[j] + 1) : slot[j];
      add_one<DET>(gvol, s_hi, col[j] + 1, w[2 * j + 1], to_fixed);
    }
  }

  // flush: consecutive lanes write consecutive voxels of a run (contiguous global float atomics; DET: 64-bit
  // integer atomics of the sums themselves)
  template <bool DET = false>
  __device__ __forceinline__ void flush(float* __restrict__ gvol, int nvox, float from_fixed, int tid,
                                        int nthreads) {
    // (A list of the slots in use, so that the flush need not scan the table, was measured:
    // +12 us per backward launch -- the extra LDS atomics and bytes cost more than the scan.)
    for (int i = tid; i < (SLOTS << SHIFT); i += nthreads) {
      const int key = keys[i >> SHIFT];
      if (key < 0) continue;  // most slots of a tile stay empty
      const long long q = (long long)vals[i];
      const int lin = (key << SHIFT) + (i & (kLen - 1));
      if (q != 0 && lin < nvox) {
        if (DET) atomicAdd(reinterpret_cast<unsigned long long*>(gvol) + lin, (unsigned long long)q);
        else atomicAdd(gvol + lin, (float)q * from_fixed);
      }
    }
  }
};
// Runs of 4 voxels whose entries are z-PAIRS: entry e of a run is one 64-bit word holding two signed 32-bit
// fixed-point sums, low half = voxel 4 key + e, high half = voxel 4 key + e + 1 (so a run reaches one voxel
// into the next one and a cell's z-pair never straddles two runs).  A column of a cell -- two corners -- is
// ONE LDS atomic and one look-up: 4 adds and 4 look-ups per hit pixel instead of 8 adds and 4-8 look-ups.
// The backward's cost follows the number of LDS instructions a tile issues, not their
// width.  Adding (hi << 32) + lo as one 64-bit integer keeps both halves exact as long as the low sum stays
// inside 32 bits: a voxel receives at most one contribution per pixel, so the caller scales contributions to
// < 2^(31 - log2(pixels) - 1) (render.hip, tile_fixed_bits: 2^21 for the 512-pixel tiles, 2^20 for the 1024-pixel
// ones, 2^22 for the sampler's 256 points) and sends anything at or above twice that to the global-atomic bypass.
// The price is resolution: a contribution is rounded to 2^-bits of the tile's bound (2 max|grad| scale) instead
// of being represented exactly; sums stay independent of the order of arrival.
template <int SLOTS>
struct PairRunHash {
  static constexpr int kSlots = SLOTS, kLen = 4;
  static constexpr int kBits = 22;
  static constexpr float kWeightLimit = 4194304.0f;  // 2^22: 256 such contributions (the sampler's block) stay below 2^31
  unsigned long long vals[SLOTS * 4];
  int keys[SLOTS];

  __device__ __forceinline__ void clear(int tid, int nthreads) {
    // 16-byte stores: the cost of an LDS phase follows its instruction count
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    i32x4v* k4 = reinterpret_cast<i32x4v*>(keys);
    i32x4v* v4 = reinterpret_cast<i32x4v*>(vals);
    for (int i = tid; i < SLOTS / 4; i += nthreads) k4[i] = i32x4v{-1, -1, -1, -1};
    for (int i = tid; i < SLOTS * 2; i += nthreads) v4[i] = i32x4v{0, 0, 0, 0};
  }

  // as RunHash::slot_of, continuing from a first probe that has been read already (`cur` = keys[h]): no
  // second read of a slot that was seen empty or taken
  __device__ __forceinline__ int slot_after(int key, unsigned h, int cur) {
#pragma unroll 1
    for (int probe = 0; probe < 32; ++probe) {
      if (cur == -1) cur = atomicCAS(&keys[h], -1, key);
      if (cur == -1 || cur == key) return (int)h;
      h = (h + 1) & (SLOTS - 1);
      cur = __builtin_nontemporal_load(&keys[h]);
    }
    return -1;
  }

  template <bool DET = false>
  __device__ __forceinline__ void add_cell(float* __restrict__ gvol, int lin, int Rr, const float (&w)[8],
                                           float to_fixed) {
    static_assert(!DET, "32-bit pair sums cannot hold the deterministic mode's quantum: use RunHash");
    const int col[4] = {lin, lin + Rr, lin + Rr * Rr, lin + Rr * Rr + Rr};
    int key[4], cur[4];
    unsigned h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      key[j] = col[j] >> 2;
      h[j] = (((unsigned)key[j] * 2654435761u) >> 16) & (SLOTS - 1);
      cur[j] = __builtin_nontemporal_load(&keys[h[j]]);
    }
    int slot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) slot[j] = (cur[j] == key[j]) ? (int)h[j] : slot_after(key[j], h[j], cur[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (slot[j] >= 0) {
        const long long lo = fixed_from_float(w[2 * j] * to_fixed), hi = fixed_from_float(w[2 * j + 1] * to_fixed);
        atomicAdd(&vals[slot[j] * 4 + (col[j] & 3)], (unsigned long long)(lo + (hi << 32)));
      } else {  // crowded table
        atomicAdd(gvol + col[j], w[2 * j]);
        atomicAdd(gvol + col[j] + 1, w[2 * j + 1]);
      }
    }
  }

  // thread -> (slot, voxel k = 0..4 of the run): low half of entry k plus high half of entry k - 1.
  // (One lane per ENTRY with the neighbour's high half fetched by DPP -- 2 LDS reads per step instead of 3 --
  // needs a second, quarter-filled global atomic for voxel 4: backward 158.7 -> 209 us.  The flush's global
  // atomics, not its LDS reads, are the expensive part.)
  template <bool DET = false>
  __device__ __forceinline__ void flush(float* __restrict__ gvol, int nvox, float from_fixed, int tid, int nthreads) {
    static_assert(!DET, "use RunHash");
    for (int i = tid; i < SLOTS * 5; i += nthreads) {
      const int slot = i / 5, k = i - slot * 5;
      const int key = keys[slot];
      if (key < 0) continue;
      long long sum = 0;
      if (k < 4) sum += (long long)(int)(unsigned)(vals[slot * 4 + k] & 0xffffffffull);
      if (k > 0) {
        const long long q = (long long)vals[slot * 4 + k - 1];
        sum += (q - (long long)(int)(unsigned)((unsigned long long)q & 0xffffffffull)) >> 32;
      }
      const int lin = key * 4 + k;
      if (sum != 0 && lin < nvox) atomicAdd(gvol + lin, (float)sum * from_fixed);
    }
  }
};
using BatchHash = RunHash<2, 512>;    // 64 x 8-pixel tiles of a batch; blocks of 256 points of the sampler
using SmallHash = RunHash<1, 1024>;   // 32 x 8-pixel tiles of small calls, any resolution

// Minimum / maximum over the wave by DPP (row shifts, then the row broadcasts of gfx9): 6 VALU instructions, no LDS
// traffic; lane 63 ends up with the result.  min / max are idempotent, so a lane without a valid source simply
// combines with itself (`old` = its own value, bound_ctrl off) and every row may take part in the broadcasts.
template <int CTRL> __device__ __forceinline__ int dpp_self(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_min_to63(int v) {
  v = min(v, dpp_self<0x111>(v));  // row_shr:1
  v = min(v, dpp_self<0x112>(v));  // row_shr:2
  v = min(v, dpp_self<0x114>(v));  // row_shr:4
  v = min(v, dpp_self<0x118>(v));  // row_shr:8
  v = min(v, dpp_self<0x142>(v));  // row_bcast:15
  v = min(v, dpp_self<0x143>(v));  // row_bcast:31
  return v;
}
__device__ __forceinline__ int wave_max_to63(int v) {
  v = max(v, dpp_self<0x111>(v));
  v = max(v, dpp_self<0x112>(v));
  v = max(v, dpp_self<0x114>(v));
  v = max(v, dpp_self<0x118>(v));
  v = max(v, dpp_self<0x142>(v));
  v = max(v, dpp_self<0x143>(v));
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}


}  // namespace
}  // namespace sdfr
