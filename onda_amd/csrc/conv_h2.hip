// fp32-accurate convolution on the f16 matrix pipe with TWO limbs ("f16x2"): half the MFMA work of the
// three-limb bf16 split (round 1, retired: `git show c558d15:onda_amd/csrc/conv_bf3.hip`), which matters because that kernel is bound by a power-managed
// clock, not by issue slots or operand delivery.
//
//   x * s = h1 + h2 + e,   h1 = f16(x * s), h2 = f16(x * s - h1),   |e| <= 2^-22 |x * s|  (or <= 2^-25 absolute)
//
// with s a power of two per TENSOR chosen so that the largest |x * s| lies in [2^14, 2^15) (f16 tops out at
// 65504): f16 has 11
// significant bits but only 5 exponent bits, so unlike bf16 it needs the scale; a power of two keeps the
// scaling exact, and a per-tensor factor comes out of the whole contraction (a per-pixel one would not:
// an im2col row mixes pixels).  The second limb is stored times 2^11 (the residual of an f16 rounding is
// <= 2^-11 of the first limb), so both limbs keep 11 significant bits for every element down to 2^-28 of the
// tensor maximum; below that the limbs slide into f16 subnormals (nothing is flushed) and precision falls off
// gradually.  Measured on outputs that depend ONLY on small elements (a pixel far below the largest one,
// through a 1x1 conv; tests/test_hip_kernels.py::test_f16x2_interpixel_range): 1.3e-7 down to 2^-24 of the
// maximum, 6e-7 at 2^-28, 1e-5 at 2^-32 -- 7-8 decades of per-element range at full accuracy, against fp32's
// own 2^-126; the retired "bf16x3" mode had no such dependence.
// The product is evaluated as a1*b1 + (a1*b2 + a2*b1), each exact in fp32 (11 x 11 bits),
// accumulated in fp32 by v_mfma_f32_16x16x32_f16 -- a1*b1 in one accumulator set, the two cross products
// (2^11 too large) in a second one that is folded in with 2^-11 at the end; what is dropped (a2*b2 and the
// representation error) is ~2^-22 |a||b| per product.  Measured on the GPU: 1.3-3e-7 relative L2 against
// fp64, where fp32 FMA chains give 3e-7 and the bf16x3 kernels 0.7-2.4e-7.
//
// Same geometry, LDS images (two limb planes instead of three: 48 KB), weight DMA, hybrid stream-K
// schedule and epilogue as conv_fwd_bf3_kernel; the accumulators are multiplied by 1 / (sa * sb)
// (exact) before the epilogue or the partial-tile store.  The scale of a tensor lives in device memory as
// its max|x| (written by the kernel that produced the tensor, or by absmax_kernel) -- no host round trip.
#include "conv_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// two floats -> two f16 (round to nearest even) in one dword
__device__ __forceinline__ unsigned cvt2h(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
__device__ __forceinline__ f32x2 unpack2h(unsigned p) {
  return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2);
}

// The second limb is stored times 2^11: the residual of an f16 rounding is <= 2^-11 of the first limb, so
// h2 * 2^11 lives in the first limb's own exponent range instead of sliding into f16 subnormals for small
// elements.  The cross products a1*b2' + a2'*b1 go to a second accumulator set that is folded in with 2^-11
// at the end (exact scaling): full two-limb precision for every element down to 2^-28 of the tensor maximum.
constexpr float LIMB2_SCALE = ONDA_LIMB2_SCALE, LIMB2_UNSCALE = 1.f / ONDA_LIMB2_SCALE;  // common.h

// float4 (already scaled) -> two limbs, each 4 f16 packed in 8 bytes
__device__ __forceinline__ void split2(const f32x4 v, u32x2& l1, u32x2& l2) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float x0 = v[2 * h], x1 = v[2 * h + 1];
    const unsigned p = cvt2h(x0, x1);
    const f32x2 f = unpack2h(p);
    l1[h] = p;
    l2[h] = cvt2h((x0 - f[0]) * LIMB2_SCALE, (x1 - f[1]) * LIMB2_SCALE);
  }
}

constexpr unsigned OOB = 0x80000000u;     // every operand is < 2 GiB - 4 KiB (checked on the host)
constexpr unsigned CH_OOB = 0x7FFFF000u;  // second addend: row + channel never wraps, stays out of range
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// chunk swizzle of the 64-byte-row LDS images (as in the retired conv_bf3.hip)
__device__ __forceinline__ int swz_row(int row) {
  const int q = (row >> 2) & 3;
  return q ^ ((q & 1) << 1) ^ ((row >> 1) & 1);
}

// ---- per-tensor scale --------------------------------------------------------------------------
// A tensor's scale travels as its max|x| in ONDA_AMAX_FLOATS device floats (written by absmax_kernel below or,
// fused, by the kernel that produced the tensor: BatchNorm apply / backward, the conv epilogue; common.h
// amax_update / amax_read).  Every consumer derives s = 2^e, 1/s = 2^-e with amax * 2^e in [2^14, 2^15) from
// it -- a few loads and scalar operations per wave, no finalisation kernel, no host round trip.  An all-zero
// (or non-finite) tensor gets e = 0.
struct Scale2 {
  float s, inv;
};
__device__ __forceinline__ Scale2 scale_of(const float* __restrict__ amax) {
  const float m = amax_read(amax);
  int e = 0;
  if (m > 0.f && m < 3.0e38f) {
    int ex;
    frexpf(m, &ex);  // m = f * 2^ex, f in [0.5, 1)
    e = 15 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  return Scale2{ldexpf(1.f, e), ldexpf(1.f, -e)};
}

// amax[ONDA_AMAX_FLOATS] (zero on entry): slot-wise max|x| over x[rows][ld] (C valid channels)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long rows, int C, int ld,
                                                     float* __restrict__ amax) {
  const int c4 = C >> 2;
  const long long n = rows * c4;
  float m = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const long long row = e / c4;
    const int ch = (int)(e - row * c4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ld + ch);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  // a flat tensor (rows == 1) whose length is not a multiple of 4: the last 1-3 values
  if (rows == 1 && blockIdx.x == 0 && threadIdx.x < (C & 3)) m = fmaxf(m, fabsf(x[(C & ~3) + threadIdx.x]));
  __shared__ float red[4];
  amax_update_block(amax, m, red);
}

// OIHW fp32 -> limb planes dst[2][rows_pad][Kp] f16 of w * 2^e (e from *amax).  dgrad = 0: row n, k = tap*Cin + c.
// dgrad = 1: row c, k = tap'*Cout_pad + n with the taps flipped (data-gradient operand).
__global__ void pack_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int Cout, int Cin, int taps,
                               int rows_pad, int Kp, int dgrad, int Cout_pad, const float* __restrict__ amax) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t plane = (size_t)rows_pad * Kp;
  if (e >= plane) return;
  const int k = (int)(e % Kp), row = (int)(e / Kp);
  float v = 0.f;
  if (!dgrad) {
    if (row < Cout && k < taps * Cin) {
      const int tap = k / Cin, cc = k - tap * Cin;
      v = w[((size_t)row * Cin + cc) * taps + tap];
    }
  } else {
    const int tap = k / Cout_pad, n = k - tap * Cout_pad;
    if (row < Cin && tap < taps && n < Cout) v = w[((size_t)n * Cin + row) * taps + (taps - 1 - tap)];
  }
  v *= scale_of(amax).s;
  const _Float16 a = (_Float16)v;
  dst[e] = a;
  dst[plane + e] = (_Float16)((v - (float)a) * LIMB2_SCALE);
}

// ---- all conv weights of a model in two launches --------------------------------------------------
// The student's weights change every step (SGD) and so do the teacher's (EMA): ~100 weight tensors to re-scale and
// re-split, forward and data-gradient form.  One absmax + one pack launch per tensor cost 3 ms of launch overhead per
// step; a table of tensors in device memory makes it one launch each.
// The launches are FLAT: an entry owns the blocks [first_block, first_block + onda_pack_blocks(entry)) of one 1-D grid
// (a (largest tensor) x (entries) grid dispatched 106 000 workgroups for a ResNet-50, nine in ten of them empty: the
// dispatch alone took longer than the copy).  A workgroup finds its entry by bisection over the table.
constexpr int PK_BLOCK_ELEMS = 9216;   // weights per block: one 32 x 32 x 9 unit, or nine 32 x 32 x 1 units (loaded together)
constexpr int PK_T = 32;               // channels per side of a unit
constexpr int PK_MAXTAPS = 9;          // 3 x 3 (other filters take the element-per-thread path)
__host__ __device__ inline bool pack_fast(int Cout, int Cin, int taps) {
  return (taps == 1 || taps == 9) && Cin % PK_T == 0 && Cout % PK_T == 0;
}
__host__ __device__ inline int pack_blocks_of(int Cout, int Cin, int taps) {
  if (pack_fast(Cout, Cin, taps)) {
    const int units = (Cout / PK_T) * (Cin / PK_T), upb = PK_BLOCK_ELEMS / (PK_T * PK_T * taps);
    return (units + upb - 1) / upb;
  }
  const long long elems = (long long)Cout * Cin * taps;
  return (int)((elems + PK_BLOCK_ELEMS - 1) / PK_BLOCK_ELEMS);
}
__device__ __forceinline__ int pack_entry_of(const OndaPackEntry* __restrict__ table, int n, int block) {
  __shared__ int starts[1024];  // block starts in LDS: the bisection's dependent loads stay off the memory system
  const bool in_lds = n <= 1024;
  if (in_lds) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) starts[i] = table[i].first_block;
    __syncthreads();
  }
  int lo = 0, hi = n - 1;  // largest i with table[i].first_block <= block
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((in_lds ? starts[mid] : table[mid].first_block) <= block) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void absmax_multi_kernel(const OndaPackEntry* __restrict__ table, int n_entries) {
  const OndaPackEntry e = table[pack_entry_of(table, n_entries, blockIdx.x)];
  const int lb = blockIdx.x - e.first_block, nb = pack_blocks_of(e.Cout, e.Cin, e.taps);
  const long long n = (long long)e.Cout * e.Cin * e.taps, per = (n + nb - 1) / nb;
  const long long i0 = lb * per, i1 = i0 + per < n ? i0 + per : n;
  float m = 0.f;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) m = fmaxf(m, fabsf(e.w[i]));
  __shared__ float red[4];
  amax_update_block(e.amax, m, red);
}

// Both packed forms of a weight in one pass over it, through LDS: a workgroup takes 32 output x 32 input channels x all
// taps (reads: runs of 32*taps contiguous floats), keeps them as [n][c][tap] (rows padded by one float: the transposed
// read below is conflict-free) and writes the forward rows [n][tap][c] and the data-gradient rows [c][tap'][n] (taps
// flipped) in runs of 32 f16 per limb plane.  (The element-per-thread version gathered with a stride of `taps` floats
// and scattered 2-byte stores: 280 us for a ResNet-50's weights, 5 x the time this traffic takes at HBM rate.)
template <int TAPS>
__device__ __forceinline__ void pack_blocks(const OndaPackEntry& e, float s, float* tile, int lb) {
  constexpr int ROW = PK_T * TAPS + 1, PER_N = PK_T * TAPS, TOTAL = PK_T * PER_N;
  constexpr int UPB = PK_BLOCK_ELEMS / (PK_T * PK_T * TAPS), SLOT = PK_T * ROW;  // units per block, floats per unit in LDS
  const size_t plane = (size_t)e.Cout * e.Cin * TAPS;
  const int K = TAPS * e.Cin, Kd = TAPS * e.Cout;
  _Float16* fwd = static_cast<_Float16*>(e.fwd);
  _Float16* dg = static_cast<_Float16*>(e.dgrad);
  const int cblocks = e.Cin / PK_T, units = (e.Cout / PK_T) * cblocks;
  const int u0 = lb * UPB, nu = units - u0 < UPB ? units - u0 : UPB;
  // all of the block's units are loaded before the one barrier (a 1 x 1 unit alone is 4 KB: nine load / barrier / store
  // round trips in a row were latency, not bandwidth)
  for (int i = threadIdx.x; i < nu * TOTAL; i += 256) {
    const int us = i / TOTAL, k = i - us * TOTAL, nl = k / PER_N, r = k - nl * PER_N;  // r = c_l * TAPS + tap: contiguous
    const int u = u0 + us, n0 = (u / cblocks) * PK_T, c0 = (u % cblocks) * PK_T;
    tile[us * SLOT + nl * ROW + r] = e.w[((size_t)(n0 + nl) * e.Cin + c0) * TAPS + r] * s;
  }
  __syncthreads();
  // two neighbouring elements per thread: 4-byte stores, runs of 32 f16 per limb plane
  for (int i = threadIdx.x; i < nu * (TOTAL / 2); i += 256) {  // forward form: (n_l, tap, c_l), c_l fastest
    const int us = i / (TOTAL / 2), j = i - us * (TOTAL / 2);
    const int u = u0 + us, n0 = (u / cblocks) * PK_T, c0 = (u % cblocks) * PK_T;
    const int cl = (j % (PK_T / 2)) * 2, q = j / (PK_T / 2), tap = q % TAPS, nl = q / TAPS;
    const float* tl = tile + us * SLOT;
    const float v0 = tl[nl * ROW + cl * TAPS + tap], v1 = tl[nl * ROW + (cl + 1) * TAPS + tap];
    const unsigned p1 = cvt2h(v0, v1);
    const f32x2 f = unpack2h(p1);
    const size_t o = (size_t)(n0 + nl) * K + (size_t)tap * e.Cin + c0 + cl;
    *reinterpret_cast<unsigned*>(fwd + o) = p1;
    *reinterpret_cast<unsigned*>(fwd + plane + o) = cvt2h((v0 - f[0]) * LIMB2_SCALE, (v1 - f[1]) * LIMB2_SCALE);
  }
  if (dg != nullptr) {
    for (int i = threadIdx.x; i < nu * (TOTAL / 2); i += 256) {  // data-gradient form: (c_l, tap', n_l), n_l fastest
      const int us = i / (TOTAL / 2), j = i - us * (TOTAL / 2);
      const int u = u0 + us, n0 = (u / cblocks) * PK_T, c0 = (u % cblocks) * PK_T;
      const int nl = (j % (PK_T / 2)) * 2, q = j / (PK_T / 2), tapd = q % TAPS, cl = q / TAPS;
      const float* tl = tile + us * SLOT;
      const float v0 = tl[nl * ROW + cl * TAPS + (TAPS - 1 - tapd)], v1 = tl[(nl + 1) * ROW + cl * TAPS + (TAPS - 1 - tapd)];
      const unsigned p1 = cvt2h(v0, v1);
      const f32x2 f = unpack2h(p1);
      const size_t o = (size_t)(c0 + cl) * Kd + (size_t)tapd * e.Cout + n0 + nl;
      *reinterpret_cast<unsigned*>(dg + o) = p1;
      *reinterpret_cast<unsigned*>(dg + plane + o) = cvt2h((v0 - f[0]) * LIMB2_SCALE, (v1 - f[1]) * LIMB2_SCALE);
    }
  }
}

__global__ __launch_bounds__(256) void pack_h2_multi_kernel(const OndaPackEntry* __restrict__ table, int n_entries) {
  const OndaPackEntry e = table[pack_entry_of(table, n_entries, blockIdx.x)];
  const int lb = blockIdx.x - e.first_block;
  const size_t plane = (size_t)e.Cout * e.Cin * e.taps;  // both forms have Cout*taps*Cin elements per limb plane
  const float s = scale_of(e.amax).s;
  const int K = e.taps * e.Cin, Kd = e.taps * e.Cout;
  _Float16* fwd = static_cast<_Float16*>(e.fwd);
  _Float16* dg = static_cast<_Float16*>(e.dgrad);
  if (pack_fast(e.Cout, e.Cin, e.taps)) {
    __shared__ float tile[9 * PK_T * (PK_T + 1) > PK_T * (PK_T * PK_MAXTAPS + 1) ? 9 * PK_T * (PK_T + 1) : PK_T * (PK_T * PK_MAXTAPS + 1)];
    if (e.taps == 1) pack_blocks<1>(e, s, tile, lb);
    else pack_blocks<9>(e, s, tile, lb);
    return;
  }
  const size_t i_end = (size_t)(lb + 1) * PK_BLOCK_ELEMS < plane ? (size_t)(lb + 1) * PK_BLOCK_ELEMS : plane;
  for (size_t i = (size_t)lb * PK_BLOCK_ELEMS + threadIdx.x; i < i_end; i += 256) {
    {  // forward form: row n, k = tap*Cin + c
      const int k = (int)(i % K), n = (int)(i / K);
      const int tap = k / e.Cin, cc = k - tap * e.Cin;
      const float v = e.w[((size_t)n * e.Cin + cc) * e.taps + tap] * s;
      const _Float16 a = (_Float16)v;
      fwd[i] = a;
      fwd[plane + i] = (_Float16)((v - (float)a) * LIMB2_SCALE);
    }
    if (dg != nullptr) {  // data-gradient form: row c, k = tap'*Cout + n, taps flipped
      const int k = (int)(i % Kd), c = (int)(i / Kd);
      const int tap = k / e.Cout, n = k - tap * e.Cout;
      const float v = e.w[((size_t)n * e.Cin + c) * e.taps + (e.taps - 1 - tap)] * s;
      const _Float16 a = (_Float16)v;
      dg[i] = a;
      dg[plane + i] = (_Float16)((v - (float)a) * LIMB2_SCALE);
    }
  }
}

// ---- forward / data gradient ----------------------------------------------------------------------
template <int BM, int BN, bool SK>
__global__ __launch_bounds__(256, 2) void conv_fwd_h2_kernel(const ConvK a, unsigned limb_stride, unsigned x_bytes, unsigned w_bytes,
                                                             const float* __restrict__ xamax, const float* __restrict__ wamax) {
  constexpr int WAVES_M = 2, WAVES_N = 2;
  constexpr int MF = 16;
  constexpr int TM = BM / (MF * WAVES_M), TN = BN / (MF * WAVES_N);
  constexpr int AL = BM / 32;
  constexpr int PLANE_A = BM * 64;         // activation limb plane, 64-byte rows, swizzled
  constexpr int PLANE_B = BN * 64;         // weight limb plane, 64-byte rows, source-swizzled
  constexpr int A_BYTES = 2 * PLANE_A, B_STAGE = 2 * PLANE_B;
  constexpr int CHUNKS = BN / 16;          // 1-KiB DMA pieces per limb plane
  constexpr int DPW = 2 * CHUNKS / 4;      // DMA instructions per wave per K-step
  __shared__ __attribute__((aligned(16))) unsigned char lds[A_BYTES + 2 * B_STAGE];

  const OndaConv& c = a.c;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int KT = a.taps * a.kcper;
  const int tiles_all = a.tilesM * a.tilesN;
  const int tiles_dp = SK ? a.tiles_dp : tiles_all;
  const long long U = (long long)(tiles_all - tiles_dp) * KT;
  long long u = SK ? swz * U / nblk : 0;
  const long long u_begin = u;
  const long long u_end = SK ? (swz + 1) * U / nblk : 0;
  int dp_tile = swz;
  const int ccol = (t & 7) * 4, rbase = t >> 3;
  const int wstride = a.taps * c.Cin;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, x_bytes), rw = make_rsrc(a.w, w_bytes);
  const Scale2 sx = scale_of(xamax), sw = scale_of(wamax);
  const float sa = sx.s;                                 // power of two: x * sa has its largest magnitude in [2^13, 2^14)
  const float unscale_a = sx.inv, unscale_b = sw.inv;  // applied one after the other: their product may underflow

  while (dp_tile < tiles_dp || u < u_end) {
    const bool dp = dp_tile < tiles_dp;
    const int tile = dp ? dp_tile : tiles_dp + (int)(u / KT);
    const int k_begin = dp ? 0 : (int)(u - (long long)(tile - tiles_dp) * KT);
    const int k_end = dp ? KT : (int)min((long long)KT, k_begin + (u_end - u));
    const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    int hi0[AL], wi0[AL], bH[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int m = m0 + rbase + 32 * i;
      const bool vm = m < a.M;
      const int mm = vm ? m : 0;
      const int wo = mm % c.Wo, tq = mm / c.Wo;
      const int ho = tq % c.Ho, b = tq / c.Ho;
      hi0[i] = vm ? ho * c.stride - c.pad : -(1 << 28);
      wi0[i] = wo * c.stride - c.pad;
      bH[i] = b * c.Hi;
    }
    // this wave's DMA pieces: piece p = wave*DPW + d -> limb p / CHUNKS, 1-KiB chunk p % CHUNKS;
    // lane -> LDS slot (row = chunk*16 + lane/4, c' = lane & 3) <- data chunk c' ^ swz_row(row)
    unsigned dofs[DPW];
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
      const int p = wave * DPW + d;
      const int l = p / CHUNKS, j = p % CHUNKS;
      const int row = j * 16 + (lane >> 2), cq = (lane & 3) ^ swz_row(row);
      const int n = n0 + row;
      dofs[d] = n < c.Cout ? (l * limb_stride + (unsigned)n * wstride) * 2u + cq * 16u : OOB;
    }

    unsigned aofs[AL];
    f32x4 ar[AL];
    int tap = k_begin / a.kcper, c0 = (k_begin - tap * a.kcper) * BK;
    auto set_tap = [&](int tp) {
      const int rr = tp / c.kw, ss = tp - rr * c.kw;
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        const int hi = hi0[i] + rr * c.dil, wi = wi0[i] + ss * c.dil;
        const bool ok = (unsigned)hi < (unsigned)c.Hi && (unsigned)wi < (unsigned)c.Wi;
        aofs[i] = ok ? (unsigned)(((bH[i] + hi) * c.Wi + wi) * c.ldx + ccol) * 4u : OOB;
      }
    };
    auto gload_a = [&]() {
      const int sa = c0 * 4;
#pragma unroll
      for (int i = 0; i < AL; ++i)
        ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, aofs[i], sa, 0));
    };
    auto dma_b = [&](int stage) {
      const int sw = (tap * c.Cin + c0) * 2;
#pragma unroll
      for (int d = 0; d < DPW; ++d) {
        const int p = wave * DPW + d;
        const int l = p / CHUNKS, j = p % CHUNKS;
        unsigned char* dst = lds + A_BYTES + stage * B_STAGE + l * PLANE_B + j * 1024;
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass drops the whole kernel stub if it sees this cast in a lambda)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, dofs[d], sw, 0, 0);
#else
        (void)dst;
        (void)sw;
#endif
      }
    };
    auto sstore_a = [&]() {
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        u32x2 l1, l2;
        split2(ar[i] * sa, l1, l2);
        const int row = rbase + 32 * i;
        const int off = row * 64 + ((((t & 7) >> 1) ^ swz_row(row)) << 4) + (t & 1) * 8;
        *reinterpret_cast<u32x2*>(lds + 0 * PLANE_A + off) = l1;
        *reinterpret_cast<u32x2*>(lds + 1 * PLANE_A + off) = l2;
      }
    };

    f32x4 acc[TM][TN], accx[TM][TN];  // a1*b1, and the cross products (2^11 too large)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = accx[i][j][e] = 0.f;

    __syncthreads();  // the previous segment's readers are done with every LDS region
    set_tap(tap);
    gload_a();
    dma_b(0);
    int cur = 0;
    for (int kt = k_begin; kt < k_end; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's A rows and weight DMA have landed
      __syncthreads();                                   // ... and everybody else's; A image is free
      sstore_a();
      __syncthreads();
      if (kt + 1 < k_end) {
        c0 += BK;
        if (c0 == c.Cin) {
          c0 = 0;
          ++tap;
          set_tap(tap);
        }
        gload_a();
        dma_b(cur ^ 1);  // the stage read one step ago; all waves are past that compute
      }
      // lane l: row l & 15 of each 16-row block, data chunk l >> 4 (swizzle is the same for every block)
      const int frag = (lane & 15) * 64 + (((lane >> 4) ^ swz_row(lane & 15)) << 4);
      const unsigned char* Ab = lds + wm * TM * MF * 64 + frag;
      const unsigned char* Bb = lds + A_BYTES + cur * B_STAGE + wn * TN * MF * 64 + frag;
      // A limbs stay in registers; B limbs stream 2 -> 1 (smaller products first): a1*b2, a2*b1, a1*b1
      f16x8 af[TM][2];
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][l] = *reinterpret_cast<const f16x8*>(Ab + l * PLANE_A + i * MF * 64);
#pragma unroll
      for (int l = 1; l >= 0; --l) {
        f16x8 bf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f16x8*>(Bb + l * PLANE_B + j * MF * 64);
#pragma unroll
        for (int la = 1 - l; la >= 0; --la)  // a_{la+1} * b_{l+1} with la + l <= 1
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (la + l == 0)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][la], bf[j], acc[i][j], 0, 0, 0);
              else
                accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][la], bf[j], accx[i][j], 0, 0, 0);
            }
      }
      cur ^= 1;
    }

    // back to the operands' own units (exact: powers of two) before anything reads the accumulators
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = ((acc[i][j] + accx[i][j] * LIMB2_UNSCALE) * unscale_a) * unscale_b;
    if (dp) dp_tile += nblk; else u += k_end - k_begin;
    if (SK && (k_begin != 0 || k_end != KT)) {
      float* slot = a.ws + ((size_t)swz * 2 + (u - (k_end - k_begin) == u_begin ? 0 : 1)) * (BM * BN);
      conv_store_partial<BN, TM, TN, MF>(slot, acc, wm, wn, lane);
      continue;
    }
    __syncthreads();
    conv_epilogue<BM, BN, TM, TN, WAVES_M, MF>(a, acc, reinterpret_cast<float*>(lds), tile_m, m0, n0, wm, wn, lane);
  }
}


// ---- weight gradient ------------------------------------------------------------------------------
// conv_wgrad_bf3_kernel (round 1, retired: `git show c558d15:onda_amd/csrc/conv_bf3.hip`) with two f16 limbs per operand: the staging thread multiplies its
// operand by that tensor's power-of-two scale before the split, the slabs are written in the operands' own
// units (acc * 1 / (sx * sdy), exact).  LDS image: 64-byte rows; inside each 16-row block the
// row index is transposed as a 4 x 4 matrix and the 16-byte chunk index is XOR-ed with row bits
// {0,1} and {3,4}.  Conflict-free for the ds_read_b128 fragment read (lane l: row l & 15, chunk
// l >> 4) AND for both ds_write_b128 staging patterns (8 lanes on rows 4l + j, or on 8
// consecutive rows): SQ_LDS_BANK_CONFLICT 0.32 -> 0 of the LDS-active cycles.
__device__ __forceinline__ int wg_slot(int row, int chunk) {
  const int phys = (row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3);
  return phys * 64 + ((chunk ^ (row & 3) ^ ((row >> 3) & 3)) << 4);
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_h2_kernel(const WgradK a, unsigned x_bytes, unsigned dy_bytes, const float* __restrict__ xamax,
                                                              const float* __restrict__ dyamax) {
  constexpr int WAVES_N = 2;
  constexpr int MF = 16;
  constexpr int TM = BM / (2 * MF), TN = BN / (2 * MF);
  constexpr int ROWS = BM + BN;
  constexpr int PLANE = ROWS * 64;  // 64-byte rows, placed by wg_slot()
  constexpr int CPT = (BM > BN ? BM : BN) / 32;  // channel columns per staging thread
  static_assert(BM == BN, "one staging half per operand");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PLANE];
  __shared__ unsigned pofs[33];  // [32]: does any of the 32 pixels of the K-step see a real input pixel for this tap?

  const OndaConv& c = a.c;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  int bid = blockIdx.x;
  const int tile_c = bid % a.tilesC;
  bid /= a.tilesC;
  const int tap = bid % a.taps;
  bid /= a.taps;
  const int tile_n = bid % a.tilesN;
  const int ks = bid / a.tilesN;
  const int n0 = tile_n * BM, c0 = tile_c * BN;
  const int mbeg = ks * a.mchunk;
  const int mend = min(a.M, mbeg + a.mchunk);
  const int KT = mend > mbeg ? (mend - mbeg + BK - 1) / BK : 0;
  const int rr = tap / c.kw, ss = tap - rr * c.kw;
  const int dh = rr * c.dil - c.pad, dw = ss * c.dil - c.pad;

  // staging role of this thread
  // the role is wave-uniform; readfirstlane tells the compiler so (descriptor and scalar offset stay
  // in SGPRs instead of a per-lane "waterfall" loop around every buffer load)
  const bool is_x = __builtin_amdgcn_readfirstlane(t >> 7) != 0;
  const Scale2 sx = scale_of(xamax), sd = scale_of(dyamax);
  const float sop = is_x ? sx.s : sd.s;               // per-tensor power of two of this thread's operand
  const float unscale_a = sx.inv, unscale_b = sd.inv;  // applied one after the other (no underflow of the product)
  const int kgroup = (t >> 5) & 3;  // 8 pixels kgroup*8 .. +7
  const __amdgpu_buffer_rsrc_t rs = is_x ? make_rsrc(a.x, x_bytes) : make_rsrc(a.dy, dy_bytes);
  const int chmax = is_x ? c.Cin : c.Cout;

  // byte offset (OOB = zero row) of pixel m of the X operand for this tap
  auto pixel_offset = [&](int m) -> unsigned {
    if (m >= mend) return OOB;
    const int wo = m % c.Wo, tq = m / c.Wo;
    const int ho = tq % c.Ho, b = tq / c.Ho;
    const int hi = ho * c.stride + dh, wi = wo * c.stride + dw;
    if ((unsigned)hi >= (unsigned)c.Hi || (unsigned)wi >= (unsigned)c.Wi) return OOB;
    return (unsigned)(((b * c.Hi + hi) * c.Wi + wi) * c.ldx) * 4u;
  };
  // row (bytes, OOB past the chunk) of pixel slot q = kgroup*8 + p of K-step mb
  auto row_offset = [&](int mb, int q) -> unsigned {
    return is_x ? pofs[q] : (mb + q < mend ? (unsigned)(q * a.lddy) * 4u : OOB);
  };

  // WIDE: a thread owns 4 consecutive channels x 8 pixels, fetched as one 16-byte load per pixel
  // (32 lanes = 512 contiguous bytes of a pixel row); the 4 x 8 register block is read out
  // column-wise, so the transposition is free.  Otherwise (64-wide tiles): 2 channel columns of
  // scalar loads as in conv_wgrad_bf3_kernel.
  constexpr bool WIDE = BM == 128;
  constexpr int NV = WIDE ? 8 : CPT * 2;
  f32x4 v[NV];
  const int cl = t & 31;
  const int chb = (is_x ? c0 : n0) + (WIDE ? 4 * cl : cl);
  unsigned chofs[WIDE ? 1 : CPT];
  if constexpr (WIDE) {
    chofs[0] = chb < chmax ? (unsigned)chb * 4u : CH_OOB;
  } else {
#pragma unroll
    for (int j = 0; j < CPT; ++j) chofs[j] = chb + 32 * j < chmax ? (unsigned)(chb + 32 * j) * 4u : CH_OOB;
  }
  auto gload = [&](int mb) {
    const int so = is_x ? 0 : mb * a.lddy * 4;  // dY: the scalar part mb*lddy rides in the soffset
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const unsigned row = row_offset(mb, kgroup * 8 + p);
      if constexpr (WIDE) {
        v[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, row + chofs[0], so, 0));
      } else {
#pragma unroll
        for (int j = 0; j < CPT; ++j)
          v[2 * j + (p >> 2)][p & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, row + chofs[j], so, 0));
      }
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < (WIDE ? 4 : CPT); ++j) {
      u32x2 a1, a2, b1, b2;
      if constexpr (WIDE) {
        split2(f32x4{v[0][j], v[1][j], v[2][j], v[3][j]} * sop, a1, a2);
        split2(f32x4{v[4][j], v[5][j], v[6][j], v[7][j]} * sop, b1, b2);
      } else {
        split2(v[2 * j] * sop, a1, a2);
        split2(v[2 * j + 1] * sop, b1, b2);
      }
      const int row = (is_x ? BM : 0) + (WIDE ? 4 * cl + j : cl + 32 * j);
      unsigned char* dst = lds + wg_slot(row, kgroup);
      *reinterpret_cast<u32x4*>(dst + 0 * PLANE) = u32x4{a1[0], a1[1], b1[0], b1[1]};
      *reinterpret_cast<u32x4*>(dst + 1 * PLANE) = u32x4{a2[0], a2[1], b2[0], b2[1]};
    }
  };

  f32x4 acc[TM][TN], accx[TM][TN];  // a1*b1, and the cross products (2^11 too large)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = accx[i][j][e] = 0.f;

  // K-steps whose 32 pixels all fall into the padding for this tap (a dilated tap near the image
  // border: 9-34 % of the ASPP weight-gradient work) contribute exact zeros and are skipped: no loads,
  // no split, no MFMAs.  The flag rides with the offset table, so the test is uniform.
  auto fill_offsets = [&](int mb) {
    if (t < 32) {
      const unsigned o = pixel_offset(mb + t);
      pofs[t] = o;
      const unsigned long long any = __ballot(o != OOB);
      if (t == 0) pofs[32] = (any & 0xFFFFFFFFull) != 0;
    }
  };
  bool live = false;  // the K-step held in registers has work
  if (KT > 0) {
    fill_offsets(mbeg);
    __syncthreads();
    live = pofs[32] != 0;
    if (live) gload(mbeg);
  }
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();  // LDS image and pofs are free
    const bool cur = live;
    if (cur) sstore();
    if (kt + 1 < KT) fill_offsets(mbeg + (kt + 1) * BK);
    __syncthreads();
    live = kt + 1 < KT && pofs[32] != 0;
    if (live) gload(mbeg + (kt + 1) * BK);
    if (!cur) continue;
    // odd 16-row blocks (row bit 4) flip chunk bit 1: byte offset ^ 32
    const int frag = wg_slot(lane & 15, lane >> 4);
    const unsigned char* Ab = lds + wm * TM * MF * 64;
    const unsigned char* Bb = lds + (BM + wn * TN * MF) * 64;
    f16x8 af[TM][2];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i][l] = *reinterpret_cast<const f16x8*>(Ab + l * PLANE + i * MF * 64 + (frag ^ ((i & 1) << 5)));
#pragma unroll
    for (int l = 1; l >= 0; --l) {
      f16x8 bf[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f16x8*>(Bb + l * PLANE + j * MF * 64 + (frag ^ ((j & 1) << 5)));
#pragma unroll
      for (int la = 1 - l; la >= 0; --la)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (la + l == 0)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][la], bf[j], acc[i][j], 0, 0, 0);
            else
              accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][la], bf[j], accx[i][j], 0, 0, 0);
          }
    }
  }

#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int cc = c0 + (wn * TN + jn) * MF + (lane & 15);
    if (cc >= c.Cin) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + (wm * TM + i) * MF + 4 * (lane >> 4) + e;
        if (n >= c.Cout) continue;
        a.slabs[(((size_t)ks * c.Cout + n) * a.taps + tap) * c.Cin + cc] = ((acc[i][jn][e] + accx[i][jn][e] * LIMB2_UNSCALE) * unscale_a) * unscale_b;
      }
  }
}


}  // namespace

extern "C" {

int onda_absmax(const float* x, int64_t rows, int C, int ld, float* amax, onda_stream_t s) {
  ONDA_REQUIRE(x && amax && rows > 0 && C > 0 && ld >= C && (rows == 1 || (C % 4 == 0 && ld % 4 == 0)));
  if (!ONDA_ALIGNED16(x)) return ONDA_EALIGN;
  const long long n = rows * (C / 4);
  const int blocks = (int)(n / 256 / 8 + 1 > 1024 ? 1024 : n / 256 / 8 + 1);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, ONDA_STREAM(s), x, (long long)rows, C, ld, amax);
  return ONDA_LAUNCH_RESULT();
}

int onda_pack_weight_h2(const float* w_oihw, void* dst, int Cout, int Cin, int taps, int rows_pad, int Kp, int dgrad,
                        int Cout_pad, const float* amax, onda_stream_t s) {
  ONDA_REQUIRE(w_oihw && dst && amax && Cout > 0 && Cin > 0 && taps > 0 && rows_pad > 0 && Kp > 0);
  const size_t plane = (size_t)rows_pad * Kp;
  hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, ONDA_STREAM(s), w_oihw,
                     static_cast<_Float16*>(dst), Cout, Cin, taps, rows_pad, Kp, dgrad, Cout_pad, amax);
  return ONDA_LAUNCH_RESULT();
}

int onda_pack_blocks(int Cout, int Cin, int taps) { return Cout > 0 && Cin > 0 && taps > 0 ? pack_blocks_of(Cout, Cin, taps) : 0; }

int onda_pack_weights_h2_multi(const OndaPackEntry* table, int n, int64_t total_blocks, onda_stream_t s) {
  ONDA_REQUIRE(table && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31));
  hipLaunchKernelGGL(absmax_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, ONDA_STREAM(s), table, n);
  hipLaunchKernelGGL(pack_h2_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, ONDA_STREAM(s), table, n);
  return ONDA_LAUNCH_RESULT();
}

int onda_conv2d_fwd_h2(const float* x, const float* xamax, const void* w2, const float* wamax, float* y,
                       const float* scale, const float* shift, const float* residual, float* stats, float* ws,
                       float* yamax, const OndaConv* c, onda_stream_t s) {
  ONDA_REQUIRE(x && xamax && w2 && wamax && y && c);
  ONDA_REQUIRE(c->run_if == nullptr);  // device predicates: pre-split kernels only (conv_l2.hip)
  ONDA_REQUIRE(c->Cin > 0 && c->Cin % 32 == 0 && c->Cout > 0 && c->Cout % 4 == 0 && c->ldx % 4 == 0 && c->ldx >= c->Cin);
  ONDA_REQUIRE(c->kh >= 1 && c->kw >= 1 && c->stride >= 1 && c->dil >= 1 && c->out_os >= 1);
  if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(w2)) return ONDA_EALIGN;
  if (ws && (c->ldy % 4 != 0 || !ONDA_ALIGNED16(y) || (residual && (c->ldr % 4 != 0 || !ONDA_ALIGNED16(residual)))))
    ws = nullptr;
  ConvK k;
  k.x = x; k.w = w2; k.y = y; k.scale = scale; k.shift = shift; k.res = residual; k.stats = stats; k.ws = ws;
  k.amax = yamax;
  k.c = *c;
  const long long M = (long long)c->B * c->Ho * c->Wo;
  ONDA_REQUIRE(M > 0 && M < (1ll << 31));
  ONDA_REQUIRE((long long)c->B * c->Hi * c->Wi * c->ldx * 4 < 0x7FFFF000ll);  // 32-bit byte offsets
  k.M = (int)M;
  k.taps = c->kh * c->kw;
  k.kcper = c->Cin / 32;
  k.tilesM = (k.M + 127) / 128;
  const bool wide = c->Cout > 64;
  k.tilesN = wide ? (c->Cout + 127) / 128 : (c->Cout + 63) / 64;
  const size_t limb_elems = (size_t)c->Cout * k.taps * c->Cin;  // planes are [Cout][taps*Cin]
  ONDA_REQUIRE(limb_elems * 4 < (1ull << 31));
  const unsigned limb_stride = (unsigned)limb_elems;
  const unsigned x_bytes = (unsigned)((size_t)c->B * c->Hi * c->Wi * c->ldx * 4), w_bytes = (unsigned)(limb_elems * 4);
  const int tiles = k.tilesM * k.tilesN, KT = k.taps * k.kcper, G = conv_resident_workgroups();
  const int rem = tiles % G;
  k.tiles_dp = tiles - rem;
  const double t_tile_us = 2.0 * 128.0 * (wide ? 128.0 : 64.0) * k.taps * c->Cin / 0.5e6;  // one tile, half a CU, ~250 TF/s chip
  const double fix_us = 8.0 + (G + 2.0 * rem) * (wide ? 0.03 : 0.015);  // partial tiles written + read
  bool balanced = ws != nullptr && rem != 0 && KT >= 4 && t_tile_us * (1.0 - (double)rem / G) > fix_us;
  if (const int force = conv_sched_override()) {  // ONDA_CONV_SCHED: 1 tile-per-workgroup, 2 hybrid, 3 pure stream-K
    if (force == 1 || ws == nullptr) {
      balanced = false;
    } else {
      balanced = true;
      if (force == 3) k.tiles_dp = 0;
    }
  }
  hipStream_t st = ONDA_STREAM(s);
  if (balanced) {
    if (wide)
      hipLaunchKernelGGL((conv_fwd_h2_kernel<128, 128, true>), dim3(G), dim3(256), 0, st, k, limb_stride, x_bytes, w_bytes, xamax, wamax);
    else
      hipLaunchKernelGGL((conv_fwd_h2_kernel<128, 64, true>), dim3(G), dim3(256), 0, st, k, limb_stride, x_bytes, w_bytes, xamax, wamax);
    return conv_launch_fixup(k, G, wide, st);
  }
  if (wide)
    hipLaunchKernelGGL((conv_fwd_h2_kernel<128, 128, false>), dim3(tiles), dim3(256), 0, st, k, limb_stride, x_bytes, w_bytes, xamax, wamax);
  else
    hipLaunchKernelGGL((conv_fwd_h2_kernel<128, 64, false>), dim3(tiles), dim3(256), 0, st, k, limb_stride, x_bytes, w_bytes, xamax, wamax);
  return ONDA_LAUNCH_RESULT();
}

int onda_conv2d_wgrad_h2(const float* x, const float* xamax, const float* dy, const float* dyamax, float* slabs, int lddy,
                         int splitk, const OndaConv* c, onda_stream_t s) {
  ONDA_REQUIRE(x && dy && xamax && dyamax && slabs && c && splitk >= 1);
  const long long M = (long long)c->B * c->Ho * c->Wo;
  ONDA_REQUIRE(M > 0 && M < (1ll << 31));
  ONDA_REQUIRE((long long)c->B * c->Hi * c->Wi * c->ldx * 4 < 0x7FFFF000ll && M * lddy * 4 < 0x7FFFF000ll);
  const unsigned x_bytes = (unsigned)((size_t)c->B * c->Hi * c->Wi * c->ldx * 4), dy_bytes = (unsigned)(M * lddy * 4);
  WgradK k;
  k.x = x; k.dy = dy; k.slabs = slabs; k.c = *c;
  k.M = (int)M;
  k.lddy = lddy;
  k.splitk = splitk;
  k.mchunk = (int)(((M + splitk - 1) / splitk + 31) / 32 * 32);
  k.taps = c->kh * c->kw;
  if (c->Cout > 64 && c->Cin > 64) {
    // 16-byte loads along the channel axis of both operands
    if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(dy) || (c->ldx & 3) || (lddy & 3)) return ONDA_EALIGN;
    k.tilesN = (c->Cout + 127) / 128;
    k.tilesC = (c->Cin + 127) / 128;
    hipLaunchKernelGGL((conv_wgrad_h2_kernel<128, 128>), dim3(k.tilesN * k.tilesC * k.taps * splitk), dim3(256), 0,
                       ONDA_STREAM(s), k, x_bytes, dy_bytes, xamax, dyamax);
  } else {
    k.tilesN = (c->Cout + 63) / 64;
    k.tilesC = (c->Cin + 63) / 64;
    hipLaunchKernelGGL((conv_wgrad_h2_kernel<64, 64>), dim3(k.tilesN * k.tilesC * k.taps * splitk), dim3(256), 0,
                       ONDA_STREAM(s), k, x_bytes, dy_bytes, xamax, dyamax);
  }
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"
