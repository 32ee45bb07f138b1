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
// The convolution kernels themselves are in conv_l2.hip (both operands pre-split in memory, LDS-DMA only); this file keeps the
// arithmetic's description, max|x| (absmax) and the weight packers.  (The round-1 kernels that split the activations in
// registers between two barriers were removed in round 5: `git show d53d4f5:onda_amd/csrc/conv_h2.hip`.)
// The scale of a tensor lives in device memory as its max|x| (written by the kernel that produced the tensor, or by
// absmax_kernel) -- no host round trip.
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

// OIHW fp32 -> limb rows dst[rows_pad][Kp / 32][2][32] f16 of w * 2^e (e from *amax; common.h: limb_at).  dgrad = 0: row n,
// k = tap*Cin + c.  dgrad = 1: row c, k = tap'*Cout_pad + n with the taps flipped (data-gradient operand).
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
  const size_t o = limb_at((size_t)row, k, Kp);
  dst[o] = a;
  dst[o + LIMB2_OFS] = (_Float16)((v - (float)a) * LIMB2_SCALE);
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
    const size_t o = limb_at((size_t)(n0 + nl), tap * e.Cin + c0 + cl, K);
    *reinterpret_cast<unsigned*>(fwd + o) = p1;
    *reinterpret_cast<unsigned*>(fwd + o + LIMB2_OFS) = cvt2h((v0 - f[0]) * LIMB2_SCALE, (v1 - f[1]) * LIMB2_SCALE);
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
      const size_t o = limb_at((size_t)(c0 + cl), tapd * e.Cout + n0 + nl, Kd);
      *reinterpret_cast<unsigned*>(dg + o) = p1;
      *reinterpret_cast<unsigned*>(dg + o + LIMB2_OFS) = cvt2h((v0 - f[0]) * LIMB2_SCALE, (v1 - f[1]) * LIMB2_SCALE);
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
      const size_t o = limb_at((size_t)n, k, K);
      fwd[o] = a;
      fwd[o + LIMB2_OFS] = (_Float16)((v - (float)a) * LIMB2_SCALE);
    }
    if (dg != nullptr) {  // data-gradient form: row c, k = tap'*Cout + n, taps flipped
      const int k = (int)(i % Kd), c = (int)(i / Kd);
      const int tap = k / e.Cout, n = k - tap * e.Cout;
      const float v = e.w[((size_t)n * e.Cin + c) * e.taps + (e.taps - 1 - tap)] * s;
      const _Float16 a = (_Float16)v;
      const size_t o = limb_at((size_t)c, k, Kd);
      dg[o] = a;
      dg[o + LIMB2_OFS] = (_Float16)((v - (float)a) * LIMB2_SCALE);
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
  ONDA_REQUIRE(w_oihw && dst && amax && Cout > 0 && Cin > 0 && taps > 0 && rows_pad > 0 && Kp > 0 && Kp % 32 == 0);
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

}  // extern "C"
