// "f16x2" convolution with BOTH operands pre-split: forward and data gradient as an implicit GEMM whose
// activation rows and weight rows arrive in LDS by LDS-DMA only (buffer_load ... lds) -- no VALU work and
// no ds_write in the K loop (the register-split kernel of conv_h2.hip spends its K-step between two
// barriers on exactly those, beside the other wave's MFMA stream that starves them).
//
// Operand format ("limb rows", the same for activations and weights; common.h limb_at):
//   row r, block of 32 channels b:  [32 x l1][32 x l2] f16,  l1 = f16(x * s),  l2 = f16((x * s - l1) * 2^11),  s = 2^e from max|x|
// (conv_h2.hip explains the arithmetic): the 128 bytes a K-step needs of a row are ONE cache line (until round 4 the limbs
// were separate planes, two half lines per row and K-step: the kernels' K loops were bound by line requests per CU,
// tools/micro/dma_rate.hip).  Activation rows are written by the kernel that produces the tensor (BatchNorm apply /
// backward, eval-mode conv epilogue, SE gate, the stem's patch kernel) or by split_h2_kernel below from an fp32 tensor
// whose max|x| is known; 4 bytes per element, like fp32.
//
// Geometry: WM x WN waves, each a 64 x 64 output (4 x 4 MFMA tiles of 16 x 16, two accumulator sets), K-step 32
// channels of one filter tap, a ring of STAGES LDS stages filled two K-steps ahead:
//   s_waitcnt vmcnt(own DMAs of the next step)  ->  s_barrier  ->  issue the DMAs of step k+2  ->  fragments + MFMAs
// one barrier per K-step, never vmcnt(0) inside the loop.  256 x 128 tiles (8 waves, 144 KB, one workgroup per
// CU) halve the L2 -> LDS bytes per MFMA of the 128 x 128 kernel.  LDS image: 128-byte rows (both limbs of the K-step's
// 32 channels, as they lie in memory: one LDS-DMA instruction = 8 rows x 128 B), the 16-byte chunk index XOR-ed with a
// function of the row (swz8) on the DMA's source side and in the fragment read: conflict-free ds_read_b128.
#include "conv_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Order of a 4 x 4 grid of MFMAs: a snake -- row i runs left to right, row i + 1 right to left -- so that consecutive
// instructions differ in ONE operand fragment only.  These loops run against the chip's power management:
// fewer operand switches = a slightly higher clock (measured: -1.2 % on the 512 -> 512 3x3 conv, -0.6 % on an ASPP branch,
// bit-identical results -- every accumulator still sees its K-steps in the same order).
#define ZZ(i, jj) (((i) & 1) ? 3 - (jj) : (jj))

constexpr float LIMB2_SCALE = ONDA_LIMB2_SCALE, LIMB2_UNSCALE = 1.f / ONDA_LIMB2_SCALE;  // common.h

// Output tiles leave as NON-TEMPORAL (streaming) stores: a convolution never reads its output again, and every round of
// tiles writes as many bytes as the 4 MB L2 of an XCD holds -- allocated normally they push the weight panel and the
// activation rows the XCD's workgroups share out of it (measured on the short-K 1x1 convolutions: -7..8 % with `nt`).
#ifndef ONDA_NT_OUT
#define ONDA_NT_OUT 1
#endif
constexpr int NT_AUX = ONDA_NT_OUT ? 2 : 0;  // buffer instruction cache policy: bit 1 = nt
template <class T>
__device__ __forceinline__ void store_out(T* p, T v) {
  if constexpr (ONDA_NT_OUT) __builtin_nontemporal_store(v, p); else *p = v;
}
constexpr unsigned OOB = 0x80000000u;  // every operand is < 2 GiB - 4 KiB (checked on the host)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// LDS image of an operand tile: 128-byte rows = 8 chunks of 16 bytes (source chunks 0-3: first limb's channels 0-31, 4-7: second
// limb's).  Chunk c of row r sits at position c ^ swz8(r).  A fragment read takes rows 0-15 of a 16-row block at one chunk
// per 16 lanes: rows r and r + 2 start 256 bytes = one pass over the 64 banks apart, so the eight rows of one parity need
// eight different positions: swz8(r) = (r & 15) >> 1.
__device__ __forceinline__ int swz8(int row) { return (row & 15) >> 1; }
// source side of an LDS-DMA piece (8 rows x 128 B; lane L lands at byte 16 L of the piece = row L >> 3, position L & 7):
// the byte offset inside the row's line this lane has to fetch; `odd` = the piece holds rows 8..15 of a 16-row block
__device__ __forceinline__ unsigned dma_chunk16(int lane, int odd) { return (unsigned)(((lane & 7) ^ (4 * odd + (lane >> 4))) << 4); }
// fragment read: lane l takes row l & 15 of a 16-row block, K chunk l >> 4 (8 channels) of limb `limb`
__device__ __forceinline__ int frag_ofs(int lane, int limb) {
  return (lane & 15) * 128 + ((((limb << 2) | (lane >> 4)) ^ swz8(lane & 15)) << 4);
}
// a tile row's (input row, input column, image row base) for tap (0, 0), packed: set_tap unpacks it once per filter tap
struct RowPos {
  int hw;  // (hi0 & 0xFFFF) | wi0 << 16;  hi0 = -32768: the row does not exist (past M)
  int bH;
};
// (b0: the image the offsets are relative to -- the first one the tile touches, see x_window)
__device__ __forceinline__ RowPos row_pos(int m, int M, const OndaConv& c, int b0) {
  const bool vm = m < M;
  const int mm = vm ? m : 0;
  const int wo = mm % c.Wo, tq = mm / c.Wo;
  const int ho = tq % c.Ho, b = tq / c.Ho;
  const int hi0 = vm ? ho * c.stride - c.pad : -32768, wi0 = wo * c.stride - c.pad;
  return RowPos{(int)((unsigned)(hi0 & 0xFFFF) | ((unsigned)wi0 << 16)), (vm ? b - b0 : 0) * c.Hi};
}
// The input operand as a tile sees it: a buffer that starts at the first image its rows touch.  32-bit byte offsets then only
// have to span the images of ONE tile (two, for images of at least a tile's rows), not the tensor: 4 + 4 images of
// 1024 x 2048 put layer4's 2048-channel activations at 2.17 GB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x_window(const ConvK& a, int m0, int& b0) {
  const OndaConv& c = a.c;
  b0 = m0 / (c.Ho * c.Wo);
  const long long at = (long long)b0 * c.Hi * c.Wi * c.ldx * 4, left = a.x_total - at;
  return make_rsrc(reinterpret_cast<const char*>(a.x) + at, (unsigned)(left < 0x7FFFF000ll ? left : 0x7FFFF000ll));
}
// byte offset of the row's 128-byte block 0 for filter tap (rr, ss), or OOB (padding / no such row)
__device__ __forceinline__ unsigned row_tap_ofs(const RowPos& p, int rr, int ss, const OndaConv& c, unsigned chunk16) {
  const int hi = (int)(short)(p.hw & 0xFFFF) + rr * c.dil, wi = (p.hw >> 16) + ss * c.dil;
  const bool ok = (unsigned)hi < (unsigned)c.Hi && (unsigned)wi < (unsigned)c.Wi;
  return ok ? (unsigned)(((p.bH + hi) * c.Wi + wi) * c.ldx) * 4u + chunk16 : 0x80000000u;
}

struct Scale2 {
  float s, inv;
};
__device__ __forceinline__ Scale2 scale_from(float m);
__device__ __forceinline__ Scale2 scale_of(const float* __restrict__ amax) { return scale_from(amax_read(amax)); }
__device__ __forceinline__ Scale2 scale_from(const float m) {
  int e = 0;
  if (m > 0.f && m < 3.0e38f) {
    int ex;
    frexpf(m, &ex);  // m = f * 2^ex, f in [0.5, 1)
    e = 15 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  return Scale2{ldexpf(1.f, e), ldexpf(1.f, -e)};
}

__device__ __forceinline__ unsigned cvt2h(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
__device__ __forceinline__ f32x2 unpack2h(unsigned p) {
  return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2);
}

// x[rows][ldx] fp32 (C valid channels) -> limb rows dst[rows][ldo / 32][2][32] f16 with the scale of *amax
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ x, long long rows, int C, int ldx,
                                                       _Float16* __restrict__ dst, int ldo, long long plane,
                                                       const float* __restrict__ amax) {
  const float s = scale_of(amax).s;
  const int c8 = C >> 3;
  const long long n = rows * c8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const long long row = e / c8;
    const int ch = (int)(e - row * c8) * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + row * ldx + ch) * s;
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + row * ldx + ch + 4) * s;
    u32x4 l1, l2;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float a = h < 2 ? v0[2 * h] : v1[2 * h - 4], b = h < 2 ? v0[2 * h + 1] : v1[2 * h - 3];
      const unsigned p = cvt2h(a, b);
      const f32x2 f = unpack2h(p);
      l1[h] = p;
      l2[h] = cvt2h((a - f[0]) * LIMB2_SCALE, (b - f[1]) * LIMB2_SCALE);
    }
    _Float16* o = dst + limb_at((size_t)row, ch, ldo);
    *reinterpret_cast<u32x4*>(o) = l1;
    *reinterpret_cast<u32x4*>(o + LIMB2_OFS) = l2;
  }
}

// Stem patches (7 x 7, stride 2, pad 3 on the NCHW image) written as limb planes: col[2][M][Kp], row m = output pixel,
// k = (r*7 + s)*3 + c, zero padded to Kp.  A patch matrix holds copies of image values and zeros, so max|col| is the
// image's max|x| (known before this kernel runs): no fp32 patch matrix, no max pass, no split pass.
__global__ __launch_bounds__(256) void stem_im2col_l2_kernel(const float* __restrict__ x, _Float16* __restrict__ dst,
                                                             long long plane, int B, int H, int W, int Ho, int Wo, int Kp,
                                                             const float* __restrict__ amax) {
  const float sc = scale_of(amax).s;
  const int k8 = Kp >> 3;
  const long long n = (long long)B * Ho * Wo * k8;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const long long m = e / k8;
    const int k0 = (int)(e - m * k8) * 8;
    const int wo = (int)(m % Wo);
    const long long tq = m / Wo;
    const int ho = (int)(tq % Ho), b = (int)(tq / Ho);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float val = 0.f;
      if (k < 147) {
        const int tap = k / 3, cc = k - tap * 3;
        const int r = tap / 7, t7 = tap - r * 7;
        const int hi = ho * 2 - 3 + r, wi = wo * 2 - 3 + t7;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) val = x[(((size_t)b * 3 + cc) * H + hi) * W + wi];
      }
      v[j] = val * sc;
    }
    u32x4 l1, l2;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const unsigned pk = cvt2h(v[2 * h], v[2 * h + 1]);
      const f32x2 f = unpack2h(pk);
      l1[h] = pk;
      l2[h] = cvt2h((v[2 * h] - f[0]) * LIMB2_SCALE, (v[2 * h + 1] - f[1]) * LIMB2_SCALE);
    }
    _Float16* o = dst + limb_at((size_t)m, k0, Kp);
    *reinterpret_cast<u32x4*>(o) = l1;
    *reinterpret_cast<u32x4*>(o + LIMB2_OFS) = l2;
  }
}

// ---- epilogue of a (64*WM) x (64*WN) tile held as 4 x 4 MFMA tiles of 16 x 16 per wave ----------------------------
// The accumulator layout has a lane's 16 values of one MFMA column 4 rows apart: stored as they stand that is 64
// four-byte stores per lane in 64-byte runs, and the store ISSUE (not bandwidth) is what a tile then waits for
// (measured: 26 000 cycles per 256 x 128 tile, more than the K loop of a 1 x 1 convolution with 256 input
// channels).  So each wave transposes its 64 x 64 sub-tile through LDS, 16 rows at a time, and stores whole
// 256-byte row segments as 16-byte vectors: 16 stores per lane, scale / shift loaded once per lane.
// `scratch`: LDS nobody else touches during the epilogue: 4 KiB per wave, then WM*BN*4 + WM*WN floats of statistics.
// COUNTED: every wave issues exactly 16 output stores (buffer stores; rows past M / channels past Cout get an
// out-of-range offset and are dropped by the hardware) -- conv_l2x_kernel counts them in its vmcnt waits.
// x[lane] (+, min, max) x[lane ^ 16] and x[lane ^ 32] on the VALU (v_permlane16/32_swap: gfx950), four values per call.
// (Inline assembly: this compiler folds the builtin's two results into one when both inputs are the same value.)
#define ONDA_SWAP4(INSN, A, B)                                                                                          \
  asm volatile("s_nop 1\n\t" INSN " %0, %4\n\t" INSN " %1, %5\n\t" INSN " %2, %6\n\t" INSN " %3, %7\n\ts_nop 0"         \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]))
__device__ __forceinline__ void rows_reduce4(float& s1, float& s2, float& mn, float& mx) {
#if defined(__HIP_DEVICE_COMPILE__)
  float A[4] = {s1, s2, mn, mx}, B[4] = {s1, s2, mn, mx};
  ONDA_SWAP4("v_permlane16_swap_b32", A, B);
  float C[4] = {A[0] + B[0], A[1] + B[1], fminf(A[2], B[2]), fmaxf(A[3], B[3])};
  float D[4] = {C[0], C[1], C[2], C[3]};
  ONDA_SWAP4("v_permlane32_swap_b32", C, D);
  s1 = C[0] + D[0];
  s2 = C[1] + D[1];
  mn = fminf(C[2], D[2]);
  mx = fmaxf(C[3], D[3]);
#endif
}

// AFFINE = false: no per-channel scale / shift, residual or ReLU (train-mode convolutions, plain data gradients): the
// epilogue then issues no global LOAD at all.  That matters in the continuous stream (COUNTED): vmcnt counts in issue
// order, so waiting for any load issued here means waiting for the next tile's DMAs that are already in flight, and
// the compiler has to place such a wait (vmcnt(0)) as soon as a load MAY have been issued.  For the same reason the
// barriers of the COUNTED path are bare s_barrier + lgkmcnt waits, not __syncthreads() (whose release fence is a
// vmcnt(0)): measured 4 000 of the epilogue's 13 000 cycles.
template <int WM, int WN, bool COUNTED = false, bool AFFINE = true>
__device__ __forceinline__ void l2_epilogue(const ConvK& a, const f32x4 (&acc)[4][4], unsigned char* scratch, int tile_m, int m0,
                                            int n0, int wm, int wn, int lane, float ua, float ub, unsigned y_bytes = 0) {
  // `acc` holds RAW sums (operand units: value * 2^ea * 2^eb); ua = 2^-ea, ub = 2^-eb (exact) are applied to the four
  // statistics of a column and folded into the per-channel scale of the output instead of to all 64 accumulators
  constexpr int BN = 64 * WN, NW = WM * WN, NT = NW * 64;
  constexpr int TRS = 68;  // floats per row of the transposition buffer: rows 4 apart land 16 banks apart (ds_write_b32
                           // banks are (a/4) % 32 per 32-lane half; 64 would put lanes l and l+16 on one bank)
  const OndaConv& c = a.c;
  const int t = threadIdx.x, wave = t >> 6;
  // diagnostics: the phases of the workgroup's LAST tile (overwritten per tile: no load here, see AFFINE)
  unsigned long long* est = a.stamps != nullptr && t == 0 ? a.stamps + (size_t)blockIdx.x * 32 + 24 : nullptr;
  if (est) est[0] = __builtin_amdgcn_s_memtime();
  float* red = reinterpret_cast<float*>(scratch + NW * (16 * TRS * 4));
  const int SR = a.stats_rows;  // 2, or 4 with the per-channel min / max of the raw tile (rows past M count as zeros:
                                // the extrema only have to BOUND the tensor's, norm_l2.hip)
  if (a.stats != nullptr) {
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      float s1 = 0.f, s2 = 0.f, mn = 3.0e38f, mxv = -3.0e38f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[i][jn][e];
          s1 += v;
          s2 += v * v;
          mn = fminf(mn, v);
          mxv = fmaxf(mxv, v);
        }
      rows_reduce4(s1, s2, mn, mxv);
      if (lane < 16) {
        const int col = (wn * 4 + jn) * 16 + lane;
        red[(wm * BN + col) * 4 + 0] = (s1 * ua) * ub;
        red[(wm * BN + col) * 4 + 1] = (((s2 * ua) * ub) * ua) * ub;
        red[(wm * BN + col) * 4 + 2] = (mn * ua) * ub;
        red[(wm * BN + col) * 4 + 3] = (mxv * ua) * ub;
      }
    }
  }

  if (est) est[1] = __builtin_amdgcn_s_memtime();
  const bool plain = (c.out_os == 1 && c.Hf == c.Ho && c.Wf == c.Wo);
  // after the transposition: lane -> row 4*r + (lane >> 4) of a 16-row chunk (r = 0..3), columns 4*(lane & 15) .. +3
  const int cl = (lane & 15) * 4, rl = lane >> 4;
  const int n = n0 + wn * 64 + cl;
  const bool vn = n < c.Cout;  // Cout is a multiple of 4
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if constexpr (AFFINE) {
    if (vn && a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
    if (vn && a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
  }
  sc = (sc * ua) * ub;
  float* tr = reinterpret_cast<float*>(scratch + wave * (16 * TRS * 4));
  float mx = 0.f;
  // the residual (a shortcut in eval mode; the running gradient sum of a shared activation, ops.GradSink) is fetched
  // up front -- 16 independent 16-byte loads per lane into the registers the second accumulator set just vacated --
  // instead of one round trip per 16-row chunk in the store loop
  f32x4 rv[4][4];
  if (AFFINE && a.res) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (wm * 4 + i) * 16 + 4 * r + rl;
        const bool live = m < a.M && vn;
        rv[i][r] = live ? *reinterpret_cast<const f32x4*>(a.res + (size_t)m * c.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  // Dense output (row m of the GEMM is row m of y: every convolution but the strided data gradient): the address of
  // (i, r) is a per-lane base plus a workgroup-uniform term -- one VALU add per store instead of the 64-bit index
  // arithmetic; with buffer stores (COUNTED) the rows past M lie past the end of the buffer (y_bytes covers M dense
  // rows) and the columns past Cout carry an out-of-range base: the hardware drops both, no per-row test.
  const bool track = a.amax != nullptr;
  const bool full_rows = m0 + 64 * WM <= a.M;
  const int mw = m0 + wm * 64 + rl;  // this lane's output row for (i, r) = (0, 0)
  // (buffer stores: the output as a buffer that starts at the tile's first row and ends behind row M - 1 -- 32-bit offsets
  //  span one tile, the tensor may pass 2 GiB)
  const unsigned vbase = vn ? (unsigned)(((size_t)(mw - m0) * c.ldy + n) * 4) : OOB;
  const long long y_left = (long long)(a.M - m0) * c.ldy * 4;
  const unsigned y_win = (unsigned)(y_left < 0x7FFFF000ll ? y_left : 0x7FFFF000ll);
  float* const ytile = a.y + (size_t)m0 * c.ldy;
  float* const ybase = a.y + (size_t)mw * c.ldy + n;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // this wave's own rows of the buffer: only its own earlier reads have to be out of the way
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int e = 0; e < 4; ++e) tr[(4 * (lane >> 4) + e) * TRS + jn * 16 + (lane & 15)] = acc[i][jn][e];
    __builtin_amdgcn_wave_barrier();  // one wave, and the LDS executes a wave's instructions in order: no wait, no s_barrier
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * r + rl;
      f32x4 v = *reinterpret_cast<const f32x4*>(tr + row * TRS + cl);
      const int m = mw + i * 16 + 4 * r;
      const bool live = (full_rows || m < a.M) && vn;
      if constexpr (AFFINE) {
        v = v * sc + sh;
        if (a.res) v += rv[i][r];
        if (c.relu) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
      } else {
        v = v * sc;
      }
      if (plain) {
        if constexpr (COUNTED) {
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), make_rsrc(ytile, y_win),
                                                 vbase + (unsigned)((i * 16 + 4 * r) * c.ldy * 4), 0, NT_AUX);
#endif
        } else if (live) {
          store_out(reinterpret_cast<f32x4*>(ybase + (size_t)((i * 16 + 4 * r) * c.ldy)), v);
        }
      } else {  // scattered rows (stride-2 data gradient)
        const int mm = live ? m : 0;
        const int wo = mm % c.Wo, tq = mm / c.Wo;
        const int ho = tq % c.Ho, b = tq / c.Ho;
        const size_t orow = ((size_t)b * c.Hf + (size_t)ho * c.out_os) * c.Wf + (size_t)wo * c.out_os;
        if constexpr (COUNTED) {
#if defined(__HIP_DEVICE_COMPILE__)
          const unsigned off = live ? (unsigned)((orow * c.ldy + n) * 4) : OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), make_rsrc(a.y, y_bytes), off, 0, NT_AUX);
#endif
        } else if (live) {
          store_out(reinterpret_cast<f32x4*>(a.y + orow * c.ldy + n), v);
        }
      }
      if (track && live) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (est) est[2] = __builtin_amdgcn_s_memtime();
  float* ar = red + WM * BN * 4;
  if (a.amax != nullptr) {
    mx = wave_max(mx);
    if (lane == 0) ar[wave] = mx;
  }
  if (a.stats != nullptr || a.amax != nullptr) {
    if constexpr (COUNTED) {  // the partial statistics in LDS are all the barrier has to publish
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
  }
  if (est) est[3] = __builtin_amdgcn_s_memtime();
  if (a.stats != nullptr) {
    for (int col = t; col < BN; col += NT) {
      if (n0 + col >= c.Cout) continue;
      float s1 = 0.f, s2 = 0.f, mn = 3.0e38f, mxv = -3.0e38f;
#pragma unroll
      for (int w_ = 0; w_ < WM; ++w_) {
        s1 += red[(w_ * BN + col) * 4 + 0];
        s2 += red[(w_ * BN + col) * 4 + 1];
        mn = fminf(mn, red[(w_ * BN + col) * 4 + 2]);
        mxv = fmaxf(mxv, red[(w_ * BN + col) * 4 + 3]);
      }
      float* dst = a.stats + (size_t)tile_m * SR * c.Cout + n0 + col;
      dst[0] = s1;
      dst[c.Cout] = s2;
      if (SR == 4) {
        dst[2 * c.Cout] = mn;
        dst[3 * c.Cout] = mxv;
      }
    }
  }
  if (a.amax != nullptr && t == 0) {  // one atomic per tile
    float m = ar[0];
#pragma unroll
    for (int w_ = 1; w_ < NW; ++w_) m = fmaxf(m, ar[w_]);
    if (m > 0.f)
      atomicMax(reinterpret_cast<unsigned*>(a.amax) + (blockIdx.x & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE, __float_as_uint(m));
  }
  if (est) est[4] = __builtin_amdgcn_s_memtime();
}

// ---- epilogue that writes LIMB PLANES (eval mode: conv + folded BatchNorm [+ residual limbs] [+ ReLU] -> the next conv's
// operand format, no fp32 tensor and no split pass in between).  The planes' scale has to exist before the first element
// is stored: it comes from an a-priori bound
//     max|y| <= max|x| * max_c(|scale_c| * sum_k |w_ck|) + max_c |shift_c| + max|residual|
// (every workgroup computes it from three device scalars and writes it to its slot of `ybound`: same value everywhere).
// The bound is 2^7..2^10 above the true maximum for these layers; a scale 2^k too small costs k of the 28 bits of range
// the format has below the maximum, nothing else.  The TRUE maximum of the stored values still goes to `amax` (one atomic
// per tile), because it is what bounds the NEXT layer: bounds do not compound.
__device__ __forceinline__ float limb_out_bound(const ConvK& a) {
  float b = amax_read(a.xtrue) * a.kb[0] + a.kb[1];
  if (a.resl != nullptr) b += amax_read(a.res_true);
  return b * 1.0001f;  // (fp32 rounding of the sums and of the bound itself)
}

template <int WM, int WN, bool COUNTED>
__device__ __forceinline__ void l2_epilogue_limbs(const ConvK& a, const f32x4 (&acc)[4][4], unsigned char* scratch, int m0, int n0,
                                                  int wm, int wn, int lane, float ua, float ub) {
  constexpr int BN = 64 * WN, NW = WM * WN;
  constexpr int TRS = 68;
  const OndaConv& c = a.c;
  if constexpr (COUNTED) asm volatile("" : "+v"(lane));
  int t = threadIdx.x;
  if constexpr (COUNTED) asm volatile("" : "+v"(t));
  const int wave = t >> 6;
  const float bound = limb_out_bound(a);
  const float so = scale_from(bound).s;
  if (t == 0) a.ybound[(blockIdx.x & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE] = bound;
  const float ri = a.resl != nullptr ? scale_of(a.res_amax).inv : 0.f;
  const int cl = (lane & 15) * 4, rl = lane >> 4;
  const int n = n0 + wn * 64 + cl;
  const bool vn = n < c.Cout;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (vn && a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
  if (vn && a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
  sc = (sc * ua) * ub;
  float* tr = reinterpret_cast<float*>(scratch + wave * (16 * TRS * 4));
  float* ar = reinterpret_cast<float*>(scratch + NW * (16 * TRS * 4));
  const int mw = m0 + wm * 64 + rl;
  // residual limbs, all 16 row positions of this lane up front (8 bytes per plane each)
  u32x2 r1[4][4], r2[4][4];
  if (a.resl != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mw + i * 16 + 4 * r;
        const bool live = m < a.M && vn;
        const _Float16* p = a.resl + limb_at((size_t)(live ? m : 0), live ? n : 0, c.ldr);
        r1[i][r] = live ? *reinterpret_cast<const u32x2*>(p) : u32x2{0u, 0u};
        r2[i][r] = live ? *reinterpret_cast<const u32x2*>(p + LIMB2_OFS) : u32x2{0u, 0u};
      }
  }
  float mx = 0.f;
  // the output's limb rows as a buffer from the tile's first row on: rows past M fall outside, offsets span one tile
  const long long out_left = (long long)(a.M - m0) * c.ldy * 4;
  const unsigned out_bytes = (unsigned)(out_left < 0x7FFFF000ll ? out_left : 0x7FFFF000ll);
  _Float16* const ytile = a.yl + (size_t)m0 * 2 * c.ldy;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int e = 0; e < 4; ++e) tr[(4 * (lane >> 4) + e) * TRS + jn * 16 + (lane & 15)] = acc[i][jn][e];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * r + rl;
      f32x4 v = *reinterpret_cast<const f32x4*>(tr + row * TRS + cl);
      const int m = mw + i * 16 + 4 * r;
      const bool live = m < a.M && vn;
      v = v * sc + sh;
      if (a.resl != nullptr) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 p1 = unpack2h(r1[i][r][h]), p2 = unpack2h(r2[i][r][h]);
          v[2 * h] += (p1[0] + p2[0] * LIMB2_UNSCALE) * ri;
          v[2 * h + 1] += (p1[1] + p2[1] * LIMB2_UNSCALE) * ri;
        }
      }
      if (c.relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
      }
      if (live) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      const f32x4 w = v * so;
      u32x2 l1, l2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned pk = cvt2h(w[2 * h], w[2 * h + 1]);
        const f32x2 f = unpack2h(pk);
        l1[h] = pk;
        l2[h] = cvt2h((w[2 * h] - f[0]) * LIMB2_SCALE, (w[2 * h + 1] - f[1]) * LIMB2_SCALE);
      }
      if constexpr (COUNTED) {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned off = vn ? (unsigned)(limb_at((size_t)(m - m0), n, c.ldy) * 2) : OOB;  // rows past M: past the buffer's end
        __builtin_amdgcn_raw_buffer_store_b64(l1, make_rsrc(ytile, out_bytes), off, 0, NT_AUX);
        __builtin_amdgcn_raw_buffer_store_b64(l2, make_rsrc(ytile, out_bytes), off, 2 * LIMB2_OFS, NT_AUX);
#endif
      } else if (live) {
        _Float16* dst = a.yl + limb_at((size_t)m, n, c.ldy);
        store_out(reinterpret_cast<u32x2*>(dst), l1);
        store_out(reinterpret_cast<u32x2*>(dst + LIMB2_OFS), l2);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (a.amax != nullptr) {  // the true maximum of what was stored (bounds the next layer)
    mx = wave_max(mx);
    if (lane == 0) ar[wave] = mx;
    if constexpr (COUNTED) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
    if (t == 0) {
      float mm = ar[0];
#pragma unroll
      for (int w_ = 1; w_ < NW; ++w_) mm = fmaxf(mm, ar[w_]);
      if (mm > 0.f)
        atomicMax(reinterpret_cast<unsigned*>(a.amax) + (blockIdx.x & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE, __float_as_uint(mm));
    }
  }
}

// ---- forward / data gradient ------------------------------------------------------------------------------------------
template <int WM, int WN, int STAGES, int OCC, bool SK, int DBG = 0>
__global__ __launch_bounds__(WM * WN * 64, OCC) void conv_l2_kernel(const ConvK a, unsigned xplane, unsigned wplane, unsigned x_bytes,
                                                                   unsigned w_bytes, const float* __restrict__ xamax,
                                                                   const float* __restrict__ wamax) {
  if (a.c.run_if != nullptr && *a.c.run_if == 0) return;  // predicated launch (onda_switch_step decided on the device)
  constexpr int NW = WM * WN;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128;  // 128-byte rows: both limbs of the K-step's 32 channels
  constexpr int APW = (BM / 8) / NW, BPW = (BN / 8) / NW;        // 8-row pieces (1 KiB: one LDS-DMA instruction) per wave and stage
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0 && APW % 2 == 0 && BPW % 2 == 0, "whole 16-row blocks per wave");
  constexpr int DPW = APW + BPW;  // LDS-DMA instructions per wave per K-step
  static_assert(STAGES >= 2, "ring: one stage being read, STAGES - 1 in flight");
  constexpr int AHEAD = STAGES >= 3 ? 2 : 1;     // K-steps in flight beyond the one being read
  constexpr bool STAGGER = NW == 8 && DBG != 6;  // two waves per SIMD: the second half of the workgroup runs half a step late
  static_assert(!STAGGER || STAGES >= 3, "the staggered halves need two steps in flight");
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * STAGE];

  const OndaConv& c = a.c;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool late = wave >= NW / 2;

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
  const int wstride = a.taps * c.Cin;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, w_bytes);
  const Scale2 sx = scale_of(xamax), sw = scale_of(wamax);
  const float unscale_a = sx.inv, unscale_b = sw.inv;  // applied one after the other: their product may underflow

  const int lrow = lane >> 3;                                                // row inside an 8-row piece
  const unsigned cq[2] = {dma_chunk16(lane, 0), dma_chunk16(lane, 1)};       // source chunk of this lane's LDS slot (even / odd piece)
  const int fr0 = frag_ofs(lane, 0), fr1 = frag_ofs(lane, 1);                // fragment reads: first / second limb

  unsigned long long tk_setup = 0, tk_loop = 0, tk_epi = 0, tk_mark = DBG == 5 ? __builtin_amdgcn_s_memtime() : 0;
  const unsigned long long tk_start = tk_mark;
  auto stamp = [&](unsigned long long& acc_) {
    if (DBG == 5) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      acc_ += now - tk_mark;
      tk_mark = now;
    }
  };
  while (dp_tile < tiles_dp || u < u_end) {
    const bool dp = dp_tile < tiles_dp;
    const int tile = dp ? dp_tile : tiles_dp + (int)(u / KT);
    const int k_begin = dp ? 0 : (int)(u - (long long)(tile - tiles_dp) * KT);
    const int k_end = dp ? KT : (int)min((long long)KT, k_begin + (u_end - u));
    const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    int b0;
    const __amdgpu_buffer_rsrc_t rx = x_window(a, m0, b0);
    RowPos rp[APW];
#pragma unroll
    for (int d = 0; d < APW; ++d) rp[d] = row_pos(m0 + (wave * APW + d) * 8 + lrow, a.M, c, b0);
    unsigned bofs[BPW];
#pragma unroll
    for (int d = 0; d < BPW; ++d) {
      const int n = n0 + (wave * BPW + d) * 8 + lrow;
      bofs[d] = n < c.Cout ? (unsigned)n * wstride * 4u + cq[d & 1] : OOB;  // a weight row: 2 * K f16
    }

    // Filter taps whose input rows all lie outside the image for EVERY output row of this tile (whole image rows of a
    // dilated tap: a quarter of the K-steps of the dilation-24 ASPP branch) contribute exact zeros: a whole tile skips
    // them -- no DMA, no MFMA.  (Stream-K pieces keep the full K range their unit arithmetic is written in.)
    unsigned live = 0xFFFFFFFFu;
    if (dp && a.taps > 1 && a.skip_dead_taps) {
      live = 0;
      const int m_last = min(a.M, m0 + BM) - 1;
      const int r0 = m0 / c.Wo, r1 = m_last / c.Wo;  // global output rows (b*Ho + ho) the tile touches
      const int ho0 = r0 % c.Ho;
      for (int tp = 0; tp < a.taps; ++tp) {
        const int dh = (tp / c.kw) * c.dil - c.pad;
        bool alive = false;
        for (int r = r0, ho = ho0; r <= r1; ++r) {
          alive |= (unsigned)(ho * c.stride + dh) < (unsigned)c.Hi;
          if (++ho == c.Ho) ho = 0;
        }
        live |= (alive ? 1u : 0u) << tp;
      }
      if (live == 0) live = 1;  // (cannot happen for a tile with a valid row and pad < kernel reach; keeps the loop non-empty)
    }
    auto next_live_tap = [&](int tp) {
      while (tp < a.taps && !((live >> tp) & 1u)) ++tp;
      return tp;
    };
    unsigned aofs[APW];
    int tap_i = dp ? next_live_tap(0) : k_begin / a.kcper;
    int c0_i = dp ? 0 : (k_begin - tap_i * a.kcper) * BK;  // the K-step the next issue() fetches
    auto set_tap = [&](int tp) {
      const int rr = tp / c.kw, ss = tp - rr * c.kw;
#pragma unroll
      for (int d = 0; d < APW; ++d) aofs[d] = row_tap_ofs(rp[d], rr, ss, c, cq[d & 1]);
    };
    auto issue = [&](int stage_off) {
#if defined(__HIP_DEVICE_COMPILE__)
      const int sa = c0_i * 4;                       // the row's 128-byte block of channels c0 .. c0 + 31
      const int sb = (tap_i * c.Cin + c0_i) * 4;
#pragma unroll
      for (int d = 0; d < (DBG == 9 ? APW / 2 : APW); ++d) {  // (DBG 9, ablation: half of the DMA instructions)
        unsigned char* dst = lds + stage_off + (wave * APW + d) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, aofs[d], sa, 0, 0);
      }
#pragma unroll
      for (int d = 0; d < (DBG == 9 ? BPW / 2 : BPW); ++d) {
        unsigned char* dst = lds + stage_off + A_BYTES + (wave * BPW + d) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, bofs[d], sb, 0, 0);
      }
#else
      (void)stage_off;
#endif
      c0_i += BK;
      if (c0_i == c.Cin) {
        c0_i = 0;
        tap_i = next_live_tap(tap_i + 1);
        if (tap_i < a.taps) set_tap(tap_i);
      }
    };

    f32x4 acc[4][4], accx[4][4];  // a1*b1, and the cross products (2^11 too large)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = accx[i][j][e] = 0.f;

    __syncthreads();  // the previous tile's readers (fragments, epilogue scratch) are done with every LDS region
    set_tap(tap_i);
    int st_issue = 0, st_read = 0;  // byte offsets of the ring stages
    const int nsteps = dp && a.taps > 1 && a.skip_dead_taps ? __builtin_popcount(live & ((1u << a.taps) - 1u)) * a.kcper : k_end - k_begin;
    issue(st_issue);
    st_issue += STAGE;
    if (AHEAD == 2 && nsteps > 1) {
      issue(st_issue);
      st_issue += STAGE;
    }
    if (st_issue == STAGES * STAGE) st_issue = 0;
    stamp(tk_setup);
    auto wait_landed = [&](bool more_in_flight) {  // this wave's DMAs of a step have landed; those of the step after may still fly
      if (DBG == 1) return;
      if (more_in_flight)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto issue_next = [&]() {
      issue(st_issue);
      st_issue = st_issue + STAGE == STAGES * STAGE ? 0 : st_issue + STAGE;
    };
    // fragments: A limbs stay in registers; B limbs stream 2 -> 1 (smaller products first): a1*b2, a2*b1, a1*b1
    f16x8 af[4][2], bf[4], b1[4];
    if constexpr (DBG == 8 || DBG >= 10) {  // (ablation: some value in every fragment register)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i][0] = af[i][1] = f16x8{(_Float16)1.f, (_Float16)0.5f, (_Float16)lane, 0, 0, 0, 0, 0};
        bf[i] = b1[i] = f16x8{(_Float16)0.25f, (_Float16)2.f, (_Float16)lane, 0, 0, 0, 0, 0};
      }
    }
    const unsigned char *Ab, *Ab2, *Bb, *Bb2;  // this wave's 16-row blocks of the stage being read: first / second limb
    auto prepare = [&]() {  // "P": first fragments of the stage at st_read (ALL of them when the halves are staggered)
      Ab = lds + st_read + wm * 64 * 128 + fr0;
      Ab2 = lds + st_read + wm * 64 * 128 + fr1;
      Bb = lds + st_read + A_BYTES + wn * 64 * 128 + fr0;
      Bb2 = lds + st_read + A_BYTES + wn * 64 * 128 + fr1;
      st_read = st_read + STAGE == STAGES * STAGE ? 0 : st_read + STAGE;
      if constexpr (DBG == 8 || DBG >= 10) return;  // ablation: no fragment reads at all (the MFMAs run on whatever the registers hold)
      if constexpr (DBG == 7) {        // ablation: half of the fragment reads (first limbs only, used for every product)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][1] = af[i][0] = *reinterpret_cast<const f16x8*>(Ab + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = bf[j] = *reinterpret_cast<const f16x8*>(Bb + j * 2048);
        return;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i][0] = *reinterpret_cast<const f16x8*>(Ab + i * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i][1] = *reinterpret_cast<const f16x8*>(Ab2 + i * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(Bb2 + j * 2048);
      if constexpr (STAGGER) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const f16x8*>(Bb + j * 2048);
      }
    };
    auto compute = [&]() {  // "C": 48 MFMAs (one wave per SIMD: the b1 fragments are fetched behind the first 16)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
      if constexpr (!STAGGER && DBG != 7 && DBG != 8 && DBG < 10) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const f16x8*>(Bb + j * 2048);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], b1[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], b1[ZZ(i, jj)], acc[i][ZZ(i, jj)], 0, 0, 0);
    };
    if constexpr (!STAGGER) {
      // (a two-stage ring -- half the LDS, two workgroups per CU hide each other's waits -- has one step in flight)
      for (int kt = 0; kt < nsteps; ++kt) {
        wait_landed(AHEAD == 2 && kt + 1 < nsteps);
        __builtin_amdgcn_s_barrier();  // everybody's DMAs of step kt have landed; the stage read in step kt-1 is free
        if (DBG != 2 && kt + AHEAD < nsteps) issue_next();
        prepare();
        compute();
      }
    } else {
      // Two waves per SIMD.  Left alone they leave every barrier at the same point of the same program: both issue their
      // DMAs, both wait for their fragments, then their MFMAs queue behind one another (measured: 2 450 cycles per K-step
      // against 1 536 of MFMA).  So a K-step is cut into two slots, "P" (DMA issue, fragment reads) and "C" (MFMAs),
      // with a barrier between slots, and the second half of the workgroup runs ONE SLOT LATE: in every slot one wave
      // of each SIMD computes while the other prepares.  Same code for both halves; only the places of the DMA issue
      // and of the vmcnt wait differ (each sits where that half's stage is free / has to be published).
      // The compute slot reads nothing from LDS (all fragments are fetched in the prepare slot), so the stage of step
      // k-1 is last read in slot 2k-1 (late P) and BOTH halves issue the DMAs of step k+2 in their own PREPARE slot --
      // early in slot 2k, late in slot 2k+1: the ~60 cycles an LDS-DMA instruction costs its wave (6 per step) fall
      // beside the other half's MFMAs instead of in front of the wave's own (measured before: slot pairs of ~1 170 + 800
      // cycles against 2 x 768 of MFMA).
      //   barrier #2k   : early half has waited for its DMAs of step k;  late half for its DMAs of step k (before #2k)
      if (late) {
        wait_landed(nsteps > 1);
        __builtin_amdgcn_s_barrier();
      }
      for (int kt = 0; kt < nsteps; ++kt) {
        // (DBG 12, measurement: s_memtime of one K-step in the middle of the loop, lane 0 of the first wave of each half)
        const bool stamp_now = DBG == 12 && a.stamps != nullptr && kt == nsteps / 2 && lane == 0 && (wave == 0 || wave == NW / 2) && dp && dp_tile == swz;
        unsigned long long* sp = a.stamps + ((size_t)bid * 2 + (late ? 1 : 0)) * 8;  // (ONDA_L2X_STAMP=1: the tail of the workspace)
        if (stamp_now) sp[0] = __builtin_amdgcn_s_memtime();
        if (!late) wait_landed(kt + 1 < nsteps);
        if (stamp_now) sp[1] = __builtin_amdgcn_s_memtime();
        if (DBG != 11) __builtin_amdgcn_s_barrier();  // (DBG 10: neither DMA nor fragment reads; 11: nor the slot barriers)
        if (stamp_now) sp[2] = __builtin_amdgcn_s_memtime();
        if (DBG != 2 && DBG < 10 && kt + 2 < nsteps) issue_next();
        if (stamp_now) sp[3] = __builtin_amdgcn_s_memtime();
        prepare();
        if (DBG == 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (stamp_now) sp[4] = __builtin_amdgcn_s_memtime();
        if (late && kt + 1 < nsteps) wait_landed(kt + 2 < nsteps);
        if (stamp_now) sp[5] = __builtin_amdgcn_s_memtime();
        if (DBG != 11) __builtin_amdgcn_s_barrier();
        if (stamp_now) sp[6] = __builtin_amdgcn_s_memtime();
        compute();
        if (stamp_now) sp[7] = __builtin_amdgcn_s_memtime();
      }
      if (!late) __builtin_amdgcn_s_barrier();
    }
    stamp(tk_loop);
    // back to the operands' own units (exact: powers of two) before anything reads the accumulators
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = acc[i][j] + accx[i][j] * LIMB2_UNSCALE;
    if (dp) dp_tile += nblk; else u += k_end - k_begin;
    if (SK && (k_begin != 0 || k_end != KT)) {
      // a piece goes to the workspace in the operands' own units (exact: powers of two, one after the other)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (acc[i][j] * unscale_a) * unscale_b;
      float* slot = a.ws + ((size_t)swz * 2 + (u - (k_end - k_begin) == u_begin ? 0 : 1)) * (BM * BN);
      conv_store_partial<BN, 4, 4, 16>(slot, acc, wm, wn, lane);
      continue;
    }
    // (the epilogue folds the two exact unscale factors into its per-column constants; if their PRODUCT left the normal
    //  range -- tensors with max|x| around 2^-50 and less -- apply them here, one after the other, as the pieces do)
    float ua = unscale_a, ub = unscale_b;
    if (const float u = ua * ub; !(u >= 0x1p-100f && u <= 0x1p100f)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (acc[i][j] * ua) * ub;
      ua = ub = 1.f;
    }
    __syncthreads();
    if (a.yl != nullptr) l2_epilogue_limbs<WM, WN, false>(a, acc, lds, m0, n0, wm, wn, lane, ua, ub);
    else l2_epilogue<WM, WN>(a, acc, lds, tile_m, m0, n0, wm, wn, lane, ua, ub);
    if (DBG == 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(tk_epi);
    }
  }
  if (DBG == 5 && t == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.ws) + (size_t)bid * 4;
    o[0] = tk_setup; o[1] = tk_loop; o[2] = tk_epi; o[3] = __builtin_amdgcn_s_memtime() - tk_start;
  }
}


// ---- the same kernel as ONE continuous K-step stream over all of a workgroup's tiles ------------------------------------
// conv_l2_kernel starts every tile cold: row decomposition, two DMA round trips before the first MFMA, and it ends it with
// an epilogue during which nothing is in flight -- 3 500-7 700 + ~6 000 cycles per tile, as much as the K loop itself for
// a 1 x 1 convolution with 256 input channels (8 K-steps of ~2 300 cycles).  Here the DMA issue runs ahead ACROSS tile
// boundaries: while the last two K-steps of a tile are being multiplied the first two stages of the workgroup's next tile
// are already on their way, the epilogue's stores are issued behind them, and the next tile's first MFMA waits for
// neither (s_waitcnt vmcnt counts in issue order: the two waits after an epilogue leave its 16 stores per wave -- buffer
// stores, out-of-range lanes dropped by the hardware, so that the count is exact -- and the younger DMAs in flight).
// The epilogue transposes through the ring stage that was read last (free until the next DMA issue, one barrier later).
// Work items of a workgroup: its whole tiles, then its one or two stream-K segments of remainder tiles.
template <int WM, int WN, int STAGES, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void conv_l2x_kernel(const ConvK a, unsigned xplane, unsigned wplane, unsigned x_bytes,
                                                                    unsigned w_bytes, unsigned y_bytes, const float* __restrict__ xamax,
                                                                    const float* __restrict__ wamax) {
  if (a.c.run_if != nullptr && *a.c.run_if == 0) return;  // predicated launch (onda_switch_step decided on the device)
  constexpr int NW = WM * WN;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128;  // 128-byte rows: both limbs (see conv_l2_kernel)
  constexpr int APW = (BM / 8) / NW, BPW = (BN / 8) / NW;        // 8-row pieces per wave and stage
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0 && APW % 2 == 0 && BPW % 2 == 0, "whole 16-row blocks per wave");
  constexpr int DPW = APW + BPW;
  constexpr int EST = 16;  // buffer stores every wave issues per epilogue (l2_epilogue<.., true>)
  static_assert(STAGES == 3 && DPW + EST <= 63, "ring of three; vmcnt holds 6 bits");
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * STAGE];

  const OndaConv& c = a.c;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int KT = a.taps * a.kcper;
  const int tiles_all = a.tilesM * a.tilesN, tiles_dp = a.tiles_dp;
  const long long U = (long long)(tiles_all - tiles_dp) * KT;
  const long long u_begin = swz * U / nblk, u_end = (swz + 1) * U / nblk;
  const int wstride = a.taps * c.Cin;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, w_bytes);
  __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, 0);  // (the window of the item being issued: open_issue_item)
  const Scale2 sx = scale_of(xamax), sw = scale_of(wamax);
  const float unscale_a = sx.inv, unscale_b = sw.inv;
  const int lrow = lane >> 3;
  const unsigned cq[2] = {dma_chunk16(lane, 0), dma_chunk16(lane, 1)};
  const int fr0 = frag_ofs(lane, 0), fr1 = frag_ofs(lane, 1);

  // ---- work items: (tile, k_begin, k_end); a cursor is (whole tile index, stream-K unit) ------------------------------
  struct Cursor {
    int dp_tile;
    long long u;
  };
  auto item_valid = [&](const Cursor& cu) { return cu.dp_tile < tiles_dp || cu.u < u_end; };
  auto item_of = [&](const Cursor& cu, int& tile, int& k_begin, int& k_end) {
    if (cu.dp_tile < tiles_dp) {
      tile = cu.dp_tile;
      k_begin = 0;
      k_end = KT;
    } else {
      tile = tiles_dp + (int)(cu.u / KT);
      k_begin = (int)(cu.u - (long long)(tile - tiles_dp) * KT);
      k_end = (int)min((long long)KT, k_begin + (u_end - cu.u));
    }
  };
  auto item_next = [&](Cursor& cu, int k_begin, int k_end) {
    if (cu.dp_tile < tiles_dp) cu.dp_tile += nblk; else cu.u += k_end - k_begin;
  };

  // ---- issue side ------------------------------------------------------------------------------------------------------
  Cursor ci{swz, u_begin};
  int i_left = 0;  // K-steps of the issue item still to be issued
  RowPos rp[APW];
  int tap_i = 0, c0_i = 0;
  unsigned bofs[BPW], aofs[APW];
  int st_issue = 0, st_read = 0, in_flight = 0;  // in_flight: issued steps whose DMAs have not been waited for
  auto set_tap = [&](int tp) {
    const int rr = tp / c.kw, ss = tp - rr * c.kw;
#pragma unroll
    for (int d = 0; d < APW; ++d) aofs[d] = row_tap_ofs(rp[d], rr, ss, c, cq[d & 1]);
  };
  auto open_issue_item = [&]() {  // row decomposition of the item at `ci` (the VALU work that used to start every tile)
    int tile, k_begin, k_end;
    item_of(ci, tile, k_begin, k_end);
    i_left = k_end - k_begin;
    const int m0 = (tile / a.tilesN) * BM, n0 = (tile % a.tilesN) * BN;
    int b0;
    rx = x_window(a, m0, b0);
#pragma unroll
    for (int d = 0; d < APW; ++d) rp[d] = row_pos(m0 + (wave * APW + d) * 8 + lrow, a.M, c, b0);
#pragma unroll
    for (int d = 0; d < BPW; ++d) {
      const int n = n0 + (wave * BPW + d) * 8 + lrow;
      bofs[d] = n < c.Cout ? (unsigned)n * wstride * 4u + cq[d & 1] : OOB;
    }
    tap_i = k_begin / a.kcper;
    c0_i = (k_begin - tap_i * a.kcper) * BK;
    set_tap(tap_i);
    item_next(ci, k_begin, k_end);
  };
  auto issue_step = [&]() {  // the next K-step of the stream, if there is one
    if (i_left == 0) {
      if (!item_valid(ci)) return;
      open_issue_item();
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const int sa = c0_i * 4, sb = (tap_i * c.Cin + c0_i) * 4;
#pragma unroll
    for (int d = 0; d < APW; ++d) {
      unsigned char* dst = lds + st_issue + (wave * APW + d) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, aofs[d], sa, 0, 0);
    }
#pragma unroll
    for (int d = 0; d < BPW; ++d) {
      unsigned char* dst = lds + st_issue + A_BYTES + (wave * BPW + d) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, bofs[d], sb, 0, 0);
    }
#endif
    st_issue = st_issue + STAGE == STAGES * STAGE ? 0 : st_issue + STAGE;
    ++in_flight;
    --i_left;
    c0_i += BK;
    if (c0_i == c.Cin) {
      c0_i = 0;
      ++tap_i;
      if (i_left > 0) set_tap(tap_i);
    }
  };

  // ---- compute side ----------------------------------------------------------------------------------------------------
  f16x8 af[4][2], bf[4];
  const unsigned char *Ab, *Ab2, *Bb, *Bb2;
  auto prepare = [&]() {
    Ab = lds + st_read + wm * 64 * 128 + fr0;
    Ab2 = lds + st_read + wm * 64 * 128 + fr1;
    Bb = lds + st_read + A_BYTES + wn * 64 * 128 + fr0;
    Bb2 = lds + st_read + A_BYTES + wn * 64 * 128 + fr1;
    st_read = st_read + STAGE == STAGES * STAGE ? 0 : st_read + STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i][0] = *reinterpret_cast<const f16x8*>(Ab + i * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i][1] = *reinterpret_cast<const f16x8*>(Ab2 + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(Bb2 + j * 2048);
  };
  int stores_young = 0;  // waits during which an epilogue's stores are still younger than the DMAs waited for
  const bool limb_out = a.yl != nullptr;  // limb-plane output: two 8-byte stores where the fp32 output has one of 16 bytes
  static_assert(DPW + 2 * EST <= 63, "vmcnt holds 6 bits");
  auto wait_step = [&]() {  // the DMAs of the oldest step in flight have landed
    // outstanding, oldest first: [that step] [the step after it, if issued] [an epilogue's stores, for two waits]
    if (in_flight > 1) {
      if (stores_young && limb_out) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW + 2 * EST) : "memory");
      else if (stores_young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW + EST) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
    } else {
      if (stores_young && limb_out) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * EST) : "memory");
      else if (stores_young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (stores_young) --stores_young;
    --in_flight;
  };

  const bool late = a.late_issue && wave >= NW / 2;
  Cursor cc{swz, u_begin};
  int n_stamp = 0;
  auto stamp = [&]() {
    if (a.stamps != nullptr && t == 0 && n_stamp < 32) a.stamps[(size_t)bid * 32 + n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  issue_step();
  issue_step();
  while (item_valid(cc)) {
    int tile, k_begin, k_end;
    item_of(cc, tile, k_begin, k_end);
    const bool whole = k_begin == 0 && k_end == KT;
    const bool first_piece = cc.dp_tile >= tiles_dp && cc.u == u_begin;
    item_next(cc, k_begin, k_end);
    f32x4 acc[4][4], accx[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = accx[i][j][e] = 0.f;
    for (int kt = k_begin; kt < k_end; ++kt) {
      wait_step();
      __builtin_amdgcn_s_barrier();  // everybody's DMAs of this step have landed; the stage read one step ago is free
      // two steps ahead in the stream, whatever tile that is.  The two waves of a SIMD leave the barrier together: one
      // (first half of the workgroup) issues its DMAs now, the other goes straight to its fragments and MFMAs and issues
      // behind them -- the matrix pipe starts ~360 cycles earlier and the second wave's issue overlaps the first one's
      // MFMAs (the stage it fills was read one step ago either way; same order of DMAs and epilogue stores per wave, so
      // the vmcnt arithmetic of wait_step is unchanged)
      if (!late) issue_step();
      prepare();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(Bb + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], bf[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[ZZ(i, jj)], acc[i][ZZ(i, jj)], 0, 0, 0);
      if (late) issue_step();
    }
    stamp();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = acc[i][j] + accx[i][j] * LIMB2_UNSCALE;
    if (!whole) {  // stream-K piece: raw accumulators to this workgroup's slot (64 unconditional stores per lane)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (acc[i][j] * unscale_a) * unscale_b;
      float* slot = a.ws + ((size_t)swz * 2 + (first_piece ? 0 : 1)) * (BM * BN);
      conv_store_partial<BN, 4, 4, 16>(slot, acc, wm, wn, lane);
      stores_young = 2;
      stamp();
      continue;
    }
    // (the epilogue folds the two exact unscale factors into its per-column constants; if their PRODUCT left the normal
    //  range -- tensors with max|x| around 2^-50 and less -- apply them here, one after the other, as the pieces do)
    float ua = unscale_a, ub = unscale_b;
    if (const float u = ua * ub; !(u >= 0x1p-100f && u <= 0x1p100f)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (acc[i][j] * ua) * ub;
      ua = ub = 1.f;
    }
    // every wave has its fragments of the last step (they fed its MFMAs): that stage is scratch now.  A bare barrier:
    // the DMAs of the next two steps stay in flight (see l2_epilogue)
    __builtin_amdgcn_s_barrier();
    const int scratch = st_read == 0 ? (STAGES - 1) * STAGE : st_read - STAGE;  // the stage read last
    if (limb_out)
      l2_epilogue_limbs<WM, WN, true>(a, acc, lds + scratch, (tile / a.tilesN) * BM, (tile % a.tilesN) * BN, wm, wn, lane, ua, ub);
    else if (a.scale != nullptr || a.shift != nullptr || a.res != nullptr || c.relu)
      l2_epilogue<WM, WN, true, true>(a, acc, lds + scratch, tile / a.tilesN, (tile / a.tilesN) * BM, (tile % a.tilesN) * BN, wm, wn,
                                      lane, ua, ub, y_bytes);
    else
      l2_epilogue<WM, WN, true, false>(a, acc, lds + scratch, tile / a.tilesN, (tile / a.tilesN) * BM, (tile % a.tilesN) * BN, wm, wn,
                                       lane, ua, ub, y_bytes);
    stores_young = 2;
    stamp();
  }
}

// ---- activation-stationary 1 x 1 convolution: the activation rows of a workgroup live in REGISTERS, only weight rows stream ----
// The short-K 1 x 1 convolutions (Cin <= 256: conv3 / downsample of layer1-3, the data gradients of their conv1) have as
// few as 2-8 K-steps per tile: in the tile kernels above every column tile re-delivers its 256 (128) activation rows into
// LDS -- 8-16 times per row tile, two thirds of the bytes a K-step waits for -- and pays a cold start and an epilogue of
// ~12 000 cycles per 8 K-steps during which the matrix pipe idles (round 3-5 stamps).
// Here a workgroup (4 waves, one per SIMD, 128 rows) loads its rows' WHOLE K extent once, straight from memory into the
// MFMA fragment layout (a lane's 16 bytes of a fragment are contiguous in a limb row: no LDS, no transposition) -- the
// 512-register budget of one wave per SIMD is what makes that possible -- and then sweeps the column tiles of Cout against
// them: per K-step only 16 KB of weight rows (hot in every L2) arrive by LDS-DMA, through a ring that runs ahead ACROSS
// column tiles and row panels.  With 256 input channels both limbs of the rows would take 256 registers per lane, which the
// compiler cannot hold beside the accumulators (it spilled the fragments as it loaded them): A2L keeps the SECOND limbs
// (used by one of the three products) in LDS instead -- 64 KB per panel, filled once per panel by LDS-DMA, 64-byte rows
// with the 16-byte chunk XOR-ed by (row >> 2) & 3 (conflict-free ds_read_b128) -- and the first limbs in registers.
// Work items (row panel, column tile) in row-major order are dealt to the workgroups in equal contiguous runs.
//
// THE EPILOGUE RUNS UNDER THE NEXT TILE'S K LOOP.  A finished tile (64 registers: acc + accx * 2^-11) is kept in `old` while
// the next one accumulates, and its epilogue -- the statistics of l2_epilogue, the transposition through LDS, the stores; or
// scale / shift / residual / ReLU; or limb rows -- is cut into phases that are spread over the K-steps of the next item
// (ep_step): with one wave per SIMD nothing else would fill the matrix pipe during those ~8 000 cycles, and nothing else
// would fill the issue slots between the MFMAs.  What makes that delicate is the ring's counted wait: s_waitcnt vmcnt(N)
// counts EVERY vector-memory operation in issue order, so the stores / residual loads of the interleaved epilogue must be
// counted exactly -- too few in N and a K-step waits for a store's round trip, too many and it reads a stage that has not
// landed.  So every K-step issues exactly EPS epilogue operations behind its DMA issue: the phases' real ones (buffer
// operations, out-of-range lanes dropped by the hardware: the count does not depend on the data) padded with stores to an
// empty buffer; the prologue pads as well.  N = (NST - 2) * BPW + (NST - 1) * EPS, one constant per instantiation.
// One epilogue (EPI 0): plain output + BatchNorm statistics (train-mode forward, plain data gradients).  Everything with a load
// in it stays on the tile kernels: a form that fetched the residual one K-step ahead (counted buffer loads) spilled under the
// 256-register cap and ran 2.2-2.4x slower than the tile kernel, "add into y" by no-return buffer_atomic_add_f32 (no load at all)
// 3-4x slower (34 M four-byte atomics per launch): profiles/r06_l2a_activation_stationary_kernel.txt, `git log` for the code.
// Same products in the same order per accumulator as the tile kernels, same epilogue arithmetic: bit-identical results.
template <int KB, int EPI>
struct L2aPlan {  // epilogue operations per wave and K-step
  static constexpr int CPS = 8 / KB;               // 4-row chunks of the pending tile stored per K-step (8 per wave)
  static constexpr int T = 1;                      // output operations per chunk: one 16-byte store
  static constexpr int F = 1;                      // the statistics store (last step)
  static constexpr int EPS = CPS * T + F;          // every step is padded to the last step's count
};

template <int KB, int NST, bool A2L, int EPI>
__global__ __launch_bounds__(512, 2) void conv_l2a_kernel(const ConvK a, unsigned w_bytes, unsigned y_bytes, const float* __restrict__ xamax,
                                                          const float* __restrict__ wamax) {
  if (a.c.run_if != nullptr && *a.c.run_if == 0) return;  // predicated launch (onda_switch_step decided on the device)
  constexpr int WM = 4, WN = 2, NW = 8, MI = 2, BM = 128, BN = 128;  // 8 waves (two per SIMD), each 32 rows x 64 columns
  constexpr int STAGE = BN * 128;          // 128 weight rows x (32 channels x 2 limbs)
  constexpr int BPW = (BN / 8) / NW;       // 8-row pieces (one LDS-DMA instruction) per wave and K-step
  constexpr int AHEAD = NST - 1;           // K-steps in flight, the one being waited for included
  constexpr int EPS = L2aPlan<KB, EPI>::EPS;
#ifndef ONDA_L2A_HINTS
#define ONDA_L2A_HINTS 0
#endif
#ifndef ONDA_L2A_PADS
#define ONDA_L2A_PADS 1
#endif
  // (measurement builds: ONDA_L2A_PADS=0 drops the padding and waits for the younger DMAs only -- conservative, still correct)
  constexpr int NWAIT = (AHEAD - 1) * BPW + (ONDA_L2A_PADS ? AHEAD * EPS : 0);  // operations younger than the DMAs a K-step waits for
  static_assert(NWAIT <= 63, "vmcnt holds 6 bits");
  static_assert((KB == 8 || KB == 4 || KB == 2) && EPI == 0, "schedules of ep_step");
  constexpr int TRS = 68;                  // floats per row of a wave's transposition buffer (l2_epilogue)
  constexpr int SCRATCH = NW * (16 * TRS * 4) + WM * BN * 4 * 4 + 64;  // transposition buffers, statistics partials
  constexpr int A2_BYTES = A2L ? KB * (BM / 16) * 1024 : 0;           // second limbs: [kb][16-row block][16 rows x 64 B]
  constexpr int NA = A2L ? 1 : 2;                                     // limbs of the rows held in registers
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE + SCRATCH + A2_BYTES];
  unsigned char* const a2lds = lds + NST * STAGE + SCRATCH;
  unsigned char* const scratch = lds + NST * STAGE;

  const OndaConv& c = a.c;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
#ifndef ONDA_L2A_STAGGER
#define ONDA_L2A_STAGGER 0
#endif
  const bool late = ONDA_L2A_STAGGER && wave >= NW / 2;  // the second wave of its SIMD takes its epilogue slice first (measurement switch)
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const long long items = (long long)a.tilesM * a.tilesN;
  const int it_begin = (int)(swz * items / nblk), it_end = (int)((swz + 1) * items / nblk);
  if (it_begin >= it_end) return;
  const int wstride = c.Cin;  // (taps == 1)
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, w_bytes);
  const __amdgpu_buffer_rsrc_t rnone = make_rsrc(a.w, 0);  // an empty buffer: every access is out of range (counted, dropped)
  const Scale2 sx = scale_of(xamax), sw = scale_of(wamax);
  // the two exact unscale factors are folded into the epilogue's per-column constants -- unless their PRODUCT leaves the normal
  // range (tensors with max|x| around 2^-50): then they are applied to the sums, one after the other (pre_a, pre_b)
  const bool unscale_first = !(sx.inv * sw.inv >= 0x1p-100f && sx.inv * sw.inv <= 0x1p100f);
  const float ua = unscale_first ? 1.f : sx.inv, ub = unscale_first ? 1.f : sw.inv;
  const float pre_a = unscale_first ? sx.inv : 1.f, pre_b = unscale_first ? sw.inv : 1.f;
  const int lrow = lane >> 3;
  const unsigned cq[2] = {dma_chunk16(lane, 0), dma_chunk16(lane, 1)};
  const int fr0 = frag_ofs(lane, 0), fr1 = frag_ofs(lane, 1);
  const int fa2o = (lane & 15) * 64 + (((lane >> 4) ^ (((lane & 15) >> 2) & 3)) << 4);  // this lane's 16 bytes of a 1 KB A2 unit

  // Column tile of an item: row panel p sweeps the column tiles starting at tile p % tilesN, so that neighbouring
  // workgroups (neighbouring panels, in step) do not all ask the L2 for the same weight lines in the same few hundred cycles
  auto col_tile_of = [&](int it) {
    const int pp = it / a.tilesN, j = it - pp * a.tilesN + pp % a.tilesN;
    return j >= a.tilesN ? j - a.tilesN : j;
  };

  // ---- issue side: the weight stream, one step = (item, kb); behind the last item the offsets are out of range (the DMA
  // writes zeros into a stage nobody reads): every step issues, so the number of DMAs in flight is a constant
  int i_item = it_begin, i_kb = 0, st_issue = 0;
  unsigned bofs[BPW];
  auto open_issue_item = [&]() {
    const int n0 = col_tile_of(i_item) * BN;
#pragma unroll
    for (int d = 0; d < BPW; ++d) {
      const int n = n0 + (wave * BPW + d) * 8 + lrow;
      bofs[d] = i_item < it_end && n < c.Cout ? (unsigned)n * wstride * 4u + cq[d & 1] : OOB;
    }
  };
  auto issue_step = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int d = 0; d < BPW; ++d) {
      unsigned char* dst = lds + st_issue + (wave * BPW + d) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, bofs[d], i_kb * 128, 0, 0);
    }
#endif
    st_issue = st_issue + STAGE == NST * STAGE ? 0 : st_issue + STAGE;
    if (++i_kb == KB) {
      i_kb = 0;
      ++i_item;
      open_issue_item();
    }
  };
  auto pad = [&](int n) {  // n counted operations that do nothing (stores to the empty buffer)
#if defined(__HIP_DEVICE_COMPILE__)
    for (int k = 0; k < (ONDA_L2A_PADS ? n : 0); ++k) __builtin_amdgcn_raw_buffer_store_b32(0u, rnone, 0, 0, 0);
#endif
  };

  open_issue_item();
#pragma unroll
  for (int sI = 0; sI < AHEAD; ++sI) {
    issue_step();
    pad(EPS);
  }

  // ---- the finished tile whose epilogue is under way -----------------------------------------------------------------------
  // Everything below is BRANCH-FREE (a K-step must stay one basic block for the scheduler to interleave these phases with the
  // MFMAs): absent operands are read from the empty buffer (zeros), absent outputs go to it, "no tile pending yet" (the first
  // item) is a tile whose every offset is out of range.
  f32x4 old[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) old[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool o_valid = false;
  int o_p = 0, o_m0 = 0, o_n0 = 0;
  const int cl = (lane & 15) * 4, rl = lane >> 4;  // after the transposition: row 4*r + rl of a 16-row chunk, columns cl .. cl + 3
  float* const tr = reinterpret_cast<float*>(scratch + wave * (16 * TRS * 4));
  float* const red = reinterpret_cast<float*>(scratch + NW * (16 * TRS * 4));
  const int SR = a.stats_rows;
  auto ep_n = [&]() { return o_n0 + wn * 64 + cl; };
  // statistics of column block jn: sum, sum of squares, min, max over this wave's 64 rows -> red (l2_epilogue's arithmetic)
  auto ep_S = [&](int jn) {
    float s1 = 0.f, s2 = 0.f, mn = 3.0e38f, mxv = -3.0e38f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = old[i][jn][e];
        s1 += v;
        s2 += v * v;
        mn = fminf(mn, v);
        mxv = fmaxf(mxv, v);
      }
    rows_reduce4(s1, s2, mn, mxv);
    if (lane < 16) {
      const int col = (wn * 4 + jn) * 16 + lane;
      *reinterpret_cast<f32x4*>(red + (wm * BN + col) * 4) = f32x4{(s1 * ua) * ub, (((s2 * ua) * ub) * ua) * ub, (mn * ua) * ub, (mxv * ua) * ub};
    }
  };
  // The pending tile leaves in 8 chunks per wave (row block i = g / 4 of the wave's two, rows 4*rq + rl of it, rq = g % 4), CPS
  // chunks per K-step, spread evenly over the K loop.  (All of a row block's stores in one step -- the first form of this
  // kernel -- made every CU of the chip store in the same two of eight steps: those steps took twice as long, the stores
  // queueing at HBM's write rate.)
  // row block i of the pending tile -> the wave's transposition buffer
  auto ep_Tw = [&](int i) {
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int e = 0; e < 4; ++e) tr[(4 * (lane >> 4) + e) * TRS + jn * 16 + (lane & 15)] = old[i][jn][e];
    __builtin_amdgcn_wave_barrier();
  };
  // chunk g: out of the buffer, unscaled, T counted operations (rows past M / columns past Cout: out of range, dropped)
  auto ep_chunk = [&](int g) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int i = g >> 2, rq = g & 3;
    const int n = ep_n();
    const bool vn = o_valid && n < c.Cout;
    const int mw = o_m0 + wm * (16 * MI) + rl;
    const long long y_left = (long long)(a.M - o_m0) * c.ldy * 4;
    const unsigned y_win = (unsigned)(y_left < 0x7FFFF000ll ? (y_left > 0 ? y_left : 0) : 0x7FFFF000ll);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y + (size_t)o_m0 * c.ldy, y_win);
    const unsigned off = vn ? (unsigned)(((size_t)(mw - o_m0 + i * 16 + 4 * rq) * c.ldy + n) * 4) : OOB;
    f32x4 v = *reinterpret_cast<const f32x4*>(tr + (4 * rq + rl) * TRS + cl);
    v = v * f32x4{ua * ub, ua * ub, ua * ub, ua * ub};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, off, 0, NT_AUX);
    if (rq == 3) __builtin_amdgcn_wave_barrier();
#else
    (void)g;
#endif
  };
  // the tile's statistics row: one counted store per wave (no statistics wanted: to the empty buffer).  A barrier lies
  // between the last ep_S and this (the K-steps' own)
  auto ep_F = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    const int col = t & (BN - 1), h = t >> 7;  // 128 columns x the four statistics: one value per thread
    f32x4 v = *reinterpret_cast<const f32x4*>(red + col * 4);  // the four row groups in ascending order
#pragma unroll
    for (int w_ = 1; w_ < WM; ++w_) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(red + (w_ * BN + col) * 4);
      v = f32x4{v[0] + o[0], v[1] + o[1], fminf(v[2], o[2]), fmaxf(v[3], o[3])};
    }
    const __amdgpu_buffer_rsrc_t rs = a.stats != nullptr ? make_rsrc(a.stats + (size_t)o_p * SR * c.Cout, (unsigned)(SR * c.Cout) * 4u) : rnone;
    const bool vc = o_valid && o_n0 + col < c.Cout;
    const int kk = SR == 4 ? h : (h & 1);  // (two statistic rows only: the sums are stored twice)
    const float val = kk == 0 ? v[0] : (kk == 1 ? v[1] : (kk == 2 ? v[2] : v[3]));
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rs, vc ? (unsigned)((kk * c.Cout + o_n0 + col) * 4) : OOB, 0, 0);
#endif
  };
  // the slice of K-step kb: the statistics of two / four column blocks in the first steps, CPS chunks of the pending tile in
  // every step, the statistics row in the last.  Every step issues exactly EPS counted operations.
  using PL = L2aPlan<KB, EPI>;
  constexpr int CPS = PL::CPS;
  auto ep_step = [&](int kb) {
    if constexpr (EPI == 0) {
      if constexpr (KB == 8) {
        if (kb == 0) { ep_S(0); ep_S(1); }
        if (kb == 1) { ep_S(2); ep_S(3); }
      } else {
        if (kb == 0) { ep_S(0); ep_S(1); ep_S(2); ep_S(3); }
      }
    }
#pragma unroll
    for (int u = 0; u < CPS; ++u) {
      const int g = kb * CPS + u;
      if ((g & 3) == 0) ep_Tw(g >> 2);
      ep_chunk(g);
    }
    if (kb + 1 < KB) {
      pad(PL::F);
    } else {
      if constexpr (EPI == 0) ep_F(); else pad(PL::F);
    }
  };

  // ---- the row panel's activations: fragment (kb, i, limb) of this lane = 16 bytes of row m0 + wm*64 + i*16 + (lane & 15):
  // bytes [limb*64 + (lane >> 4)*16, +16) of the row's 128-byte block kb
  f16x8 af[KB][MI][NA];
  int cur_p = -1, st_read = 0;
  // input pixel of GEMM row m: the pixel itself, or (stride 2) the one it samples
  auto pixel_of = [&](int m) -> long long {
    if (c.stride == 1) return m;
    const int mm = m < a.M ? m : a.M - 1;
    const int wo = mm % c.Wo, tq = mm / c.Wo;
    const int ho = tq % c.Ho, b = tq / c.Ho;
    return ((long long)b * c.Hi + (long long)ho * c.stride) * c.Wi + (long long)wo * c.stride;
  };
  int n_stamp = 0;   // diagnostics (ONDA_L2X_STAMP=1, tools/l2a_stamps.py): s_memtime at the start, then per item: rows ready, end of
  auto stamp = [&]() {  // the K loop
    if (a.stamps != nullptr && t == 0 && n_stamp < 32) a.stamps[(size_t)bid * 32 + n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  stamp();
  for (int item = it_begin; item < it_end; ++item) {
    const int p = item / a.tilesN, tile_n = col_tile_of(item);
    const int m0 = p * BM, n0 = tile_n * BN;
    if (p != cur_p) {
      cur_p = p;
      // (the weight ring keeps its contents and its order; only the rows change)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(__HIP_DEVICE_COMPILE__)
      if constexpr (A2L) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everybody has read its last second-limb fragments of the old panel
        // second limbs -> LDS: unit (kb, rb) = 16 rows x 64 B; lane L lands at byte 16 L = row L >> 2, position L & 3, and
        // fetches chunk (L & 3) ^ ((row >> 2) & 3) of its row's second-limb half.  The buffer starts at the panel's first pixel.
        const long long pix0 = pixel_of(m0);
        const long long left = a.x_total - pix0 * c.ldx * 4;
        const __amdgpu_buffer_rsrc_t rxa = make_rsrc(reinterpret_cast<const char*>(a.x) + pix0 * c.ldx * 4,
                                                     (unsigned)(left < 0x7FFFF000ll ? left : 0x7FFFF000ll));
        const int ur = lane >> 2;
        const unsigned uch = (unsigned)((((lane & 3) ^ ((ur >> 2) & 3)) << 4) + 64);
#pragma unroll
        for (int u = 0; u < 1; ++u) {  // this wave's 16-row block
          const int rb = wave;
          const int m = m0 + rb * 16 + ur;
          const unsigned rofs = m < a.M ? (unsigned)((pixel_of(m) - pix0) * c.ldx * 4) + uch : OOB;
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            unsigned char* dst = a2lds + ((kb * (BM / 16) + rb) << 10);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxa, (__attribute__((address_space(3))) void*)dst, 16, rofs, kb * 128, 0, 0);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (16 * MI) + i * 16 + (lane & 15);
        const char* rowp = reinterpret_cast<const char*>(a.x) + pixel_of(m) * c.ldx * 4 + (lane >> 4) * 16;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int l = 0; l < NA; ++l) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < a.M) v = *reinterpret_cast<const u32x4*>(rowp + kb * 128 + l * 64);
            af[kb][i][l] = __builtin_bit_cast(f16x8, v);
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (A2L) __builtin_amdgcn_s_barrier();  // everybody's second-limb pieces have landed
#endif
    }
    stamp();
    f32x4 acc[MI][4], accx[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = accx[i][j][e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");  // this wave's DMAs of this step have landed (and its LDS writes)
      __builtin_amdgcn_s_barrier();  // everybody's have; the stage read one step ago is free; `red` of the last ep_S is complete
      issue_step();                  // AHEAD steps ahead in the stream, whatever column tile / row panel that is
      const unsigned char* Bb = lds + st_read + wn * 64 * 128 + fr0;
      const unsigned char* Bb2 = lds + st_read + wn * 64 * 128 + fr1;
      st_read = st_read + STAGE == NST * STAGE ? 0 : st_read + STAGE;
      f16x8 bf[4], a2[MI];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(Bb2 + j * 2048);
      if constexpr (A2L) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a2[i] = *reinterpret_cast<const f16x8*>(a2lds + ((kb * (BM / 16) + wm * MI + i) << 10) + fa2o);
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) a2[i] = af[kb][i][NA - 1];
      }
      // The two waves of a SIMD (w and w + 4) leave the barrier together; left in step they would both multiply and then both
      // run their slice of the pending epilogue.  The second one takes the slice FIRST: one wave's VALU / LDS / store
      // instructions beside the other's MFMAs, in every K-step (the slot stagger of conv_l2_kernel, with the epilogue as the
      // "prepare" slot).  Same instructions per wave either way: the counted wait above does not notice.
      if (late) ep_step(kb);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kb][i][0], bf[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(Bb + j * 2048);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[i], bf[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kb][i][0], bf[ZZ(i, jj)], acc[i][ZZ(i, jj)], 0, 0, 0);
#ifdef ONDA_L2A_STEP_STAMPS  // (measurement builds: a branch per K-step, which keeps the compiler from interleaving the slice with the MFMAs)
      if (a.stamps != nullptr && t == 0 && item == it_begin + 2) a.stamps[(size_t)bid * 32 + 16 + kb] = __builtin_amdgcn_s_memtime();
#endif
      if (!late) ep_step(kb);
    }
    stamp();
    // this tile becomes the pending one (raw sums; the unscale factors are folded into the epilogue's constants)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) old[i][j] = ((acc[i][j] + accx[i][j] * LIMB2_UNSCALE) * pre_a) * pre_b;
    o_valid = true;
    o_p = p;
    o_m0 = m0;
    o_n0 = n0;
  }
  // ---- the last tile's epilogue, chunk by chunk ----------------------------------------------------------------------------
  if constexpr (EPI == 0) {
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) ep_S(jn);
  }
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if ((g & 3) == 0) ep_Tw(g >> 2);
    ep_chunk(g);
  }
  if constexpr (EPI == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ep_F();
  }
}

// ---- stream-K remainder: partial tiles -> output, in ONE wide launch ---------------------------------------------------
// A remainder tile was cut into pieces by the workgroups of conv_l2_kernel<.., SK = true> (raw accumulators in `ws`).
// One workgroup per 8 rows of a remainder tile sums that tile's pieces in ascending-workgroup order (fixed order:
// deterministic) and runs the epilogue on them; its statistics partial goes to a row of its own (sub-block 0: the tile's
// own row, the others: extra rows behind the regular ones -- bn_finalize sums over all rows anyway), so no second stage
// and no cross-workgroup reduction is needed.  Replaces the two-launch piece_sum + fixup of the older kernels
// (measured there: 31 us per convolution, 5.4 ms per adaptation step).
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_l2_fixup_kernel(const ConvK a, int G, int rows_regular) {
  if (a.c.run_if != nullptr && *a.c.run_if == 0) return;  // predicated launch (onda_switch_step decided on the device)
  constexpr int C4 = BN / 4;      // float4 columns of a tile row
  constexpr int RG = 256 / C4;    // rows covered by the workgroup (one element group per thread)
  constexpr int SUB = BM / RG;    // sub-blocks per tile
  __shared__ f32x4 red[4][256];
  __shared__ int piece[1024];
  __shared__ int wave_count[4];
  const OndaConv& c = a.c;
  const int KT = a.taps * a.kcper;
  const long long U = (long long)(a.tilesM * a.tilesN - a.tiles_dp) * KT;
  // grid.x walks every tile of the tile ROWS that hold remainder tiles (a statistic row spans all channel tiles of a tile
  // row, so the extra rows must be complete): tiles of the first such row that ran one-per-workgroup only fill identities
  const int sub = blockIdx.y;
  const int first_m = a.tiles_dp / a.tilesN;
  const int tile = first_m * a.tilesN + blockIdx.x;
  const int lt = tile - a.tiles_dp;  // remainder-local tile (negative: not a remainder tile)
  const long long t0 = (long long)lt * KT, t1 = t0 + KT;
  const int vs = lt < 0 ? 0 : (int)(((t0 + 1) * G + U - 1) / U - 1), ve = lt < 0 ? 0 : (int)((t1 * G + U - 1) / U - 1);
  const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
  const int t = threadIdx.x, col = (t % C4) * 4, rg = t / C4;
  const int n = tile_n * BN + col;
  const bool vn = n < c.Cout;
  const int SR = a.stats_rows;
  float* srow = a.stats ? a.stats + (size_t)(sub == 0 ? tile_m : rows_regular + (tile_m - first_m) * (SUB - 1) + sub - 1) * SR * c.Cout + n
                        : nullptr;
  if (vs == ve) {  // computed whole by one workgroup: its own epilogue ran; the extra statistic rows are identities
    if (srow && sub != 0 && rg == 0 && vn) {
      *reinterpret_cast<f32x4*>(srow) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(srow + c.Cout) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (SR == 4) {
        *reinterpret_cast<f32x4*>(srow + 2 * c.Cout) = f32x4{3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
        *reinterpret_cast<f32x4*>(srow + 3 * c.Cout) = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
      }
    }
    return;
  }
  // the workgroups whose K range meets this tile, compacted in ascending order (with fewer remainder K-steps than
  // workgroups some ranges are empty); entry = workgroup * 2 + (0: its first piece, 1: its second)
  const int npieces = ve - vs + 1;
  int nvalid = 0;
  for (int p0 = 0; p0 < npieces; p0 += 256) {
    const int p = p0 + t;
    int val = -1;
    if (p < npieces) {
      const int vb = vs + p;
      const long long b0 = (long long)vb * U / G, b1 = (long long)(vb + 1) * U / G;
      const long long g0 = max(b0, t0), g1 = min(b1, t1);
      if (g1 > g0) val = vb * 2 + (g0 == b0 ? 0 : 1);
    }
    const unsigned long long mask = __ballot(val >= 0);
    if ((t & 63) == 0) wave_count[t >> 6] = __popcll(mask);
    __syncthreads();
    int off = nvalid;
    for (int w = 0; w < (t >> 6); ++w) off += wave_count[w];
    if (val >= 0) piece[off + __popcll(mask & ((1ull << (t & 63)) - 1ull))] = val;
    nvalid += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
    __syncthreads();
  }
  const int row = sub * RG + rg;
  const int eo = row * BN + col;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 v = zero;
  // eight pieces in flight (a chain of dependent loads was most of this kernel's 10-14 us); added in ascending order
  for (int p = 0; p < nvalid; p += 8) {
    f32x4 part[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pc = piece[min(p + j, nvalid - 1)];
      part[j] = *reinterpret_cast<const f32x4*>(a.ws + (size_t)pc * (BM * BN) + eo);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v += p + j < nvalid ? part[j] : zero;
  }
  const int m = tile_m * BM + row;
  float mx = 0.f;
  if (a.yl != nullptr) {  // limb-plane output (l2_epilogue_limbs): same bound, same scale, same arithmetic
    const float bound = limb_out_bound(a);
    const float so = scale_from(bound).s;
    if (t == 0) a.ybound[(blockIdx.x & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE] = bound;
    const float ri = a.resl != nullptr ? scale_of(a.res_amax).inv : 0.f;
    if (m < a.M && vn) {
      f32x4 o = v;
      if (a.scale) o *= *reinterpret_cast<const f32x4*>(a.scale + n);
      if (a.shift) o += *reinterpret_cast<const f32x4*>(a.shift + n);
      if (a.resl != nullptr) {
        const _Float16* p = a.resl + limb_at((size_t)m, n, c.ldr);
        const u32x2 q1 = *reinterpret_cast<const u32x2*>(p), q2 = *reinterpret_cast<const u32x2*>(p + LIMB2_OFS);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 p1 = unpack2h(q1[h]), p2 = unpack2h(q2[h]);
          o[2 * h] += (p1[0] + p2[0] * LIMB2_UNSCALE) * ri;
          o[2 * h + 1] += (p1[1] + p2[1] * LIMB2_UNSCALE) * ri;
        }
      }
      if (c.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
      }
      mx = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
      const f32x4 w = o * so;
      u32x2 l1, l2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned pk = cvt2h(w[2 * h], w[2 * h + 1]);
        const f32x2 f = unpack2h(pk);
        l1[h] = pk;
        l2[h] = cvt2h((w[2 * h] - f[0]) * LIMB2_SCALE, (w[2 * h + 1] - f[1]) * LIMB2_SCALE);
      }
      _Float16* dst = a.yl + limb_at((size_t)m, n, c.ldy);
      *reinterpret_cast<u32x2*>(dst) = l1;
      *reinterpret_cast<u32x2*>(dst + LIMB2_OFS) = l2;
    }
    if (a.amax != nullptr) amax_update_block(a.amax, mx, reinterpret_cast<float*>(&red[0][0]));
    return;
  }
  if (m < a.M && vn) {
    f32x4 o = v;
    if (a.scale) o *= *reinterpret_cast<const f32x4*>(a.scale + n);
    if (a.shift) o += *reinterpret_cast<const f32x4*>(a.shift + n);
    if (a.res) o += *reinterpret_cast<const f32x4*>(a.res + (size_t)m * c.ldr + n);
    if (c.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
    }
    size_t orow = m;
    if (!(c.out_os == 1 && c.Hf == c.Ho && c.Wf == c.Wo)) {
      const int wo = m % c.Wo, tq = m / c.Wo;
      const int ho = tq % c.Ho, b = tq / c.Ho;
      orow = ((size_t)b * c.Hf + (size_t)ho * c.out_os) * c.Wf + (size_t)wo * c.out_os;
    }
    *reinterpret_cast<f32x4*>(a.y + orow * c.ldy + n) = o;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  if (a.amax != nullptr) amax_update_block(a.amax, mx, reinterpret_cast<float*>(&red[0][0]));
  if (a.stats != nullptr) {
    __syncthreads();
    red[0][t] = v;
    red[1][t] = v * v;
    red[2][t] = v;
    red[3][t] = v;
    __syncthreads();
    if (rg == 0 && vn) {
      f32x4 s1 = red[0][t], s2 = red[1][t], mn = v, mxv = v;
      for (int g = 1; g < RG; ++g) {
        s1 += red[0][g * C4 + t];
        s2 += red[1][g * C4 + t];
        const f32x4 o = red[2][g * C4 + t];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mn[j] = fminf(mn[j], o[j]);
          mxv[j] = fmaxf(mxv[j], o[j]);
        }
      }
      *reinterpret_cast<f32x4*>(srow) = s1;
      *reinterpret_cast<f32x4*>(srow + c.Cout) = s2;
      if (SR == 4) {
        *reinterpret_cast<f32x4*>(srow + 2 * c.Cout) = mn;
        *reinterpret_cast<f32x4*>(srow + 3 * c.Cout) = mxv;
      }
    }
  }
}

// ---- weight gradient ----------------------------------------------------------------------------------------------------
// dw[n][tap][c] = sum over pixels m of dy[m][n] * x[pix(m, tap)][c]: the contraction index (pixels) is the SLOW memory axis
// of both operands.  Both arrive as they lie in HBM -- [pixel][channel] limb rows, by LDS-DMA, 2 pixels x (128 channels x 2
// limbs) = four whole cache lines per pixel and instruction -- and are read out of LDS TRANSPOSED by ds_read_b64_tr_b16 (4
// pixels x 16 channels per 16 lanes, each lane receives one channel's 4 pixels): two of them make the 8 consecutive k of a
// 16x16x32 MFMA operand.  No VALU split, no register transposition, no LDS stores.  LDS image per 128-channel half:
// [32 pixels][512 B], a row = the 128 first limbs (256 B) then the 128 second limbs, each half with its 16-byte chunk index
// XOR-ed by ((row & 3) << 2 | (row >> 2) & 3) (conflict-free for the transposed reads), applied on the DMA's source side.
// An instruction's lanes 0-31 / 32-63 fetch pixels 4g + 2h and 4g + 2h + 2 ... no: pixels 4g + 2 lambda + h (lambda = lane >> 5,
// h = the group's first / second instruction) into LDS rows 4g + 2h + lambda -- the middle two pixels of every four swap
// rows, for BOTH operands alike (the contraction does not care), so that a lane's two pixels are neighbours (m, m + 1).  Tile 64*WM output channels x 64*WN input channels per (tap, pixel range); ring and slot stagger as in
// conv_l2_kernel.  K-steps whose 32 pixels all fall into the padding for this tap are skipped (whole dead rows of a
// dilated tap: up to half of the ASPP weight-gradient work).
// The transposed reads are inline assembly: behind the builtin the compiler waits for vmcnt(0) -- every LDS-DMA in flight --
// before each group of reads (it cannot tell the stage being read from the stages being filled), which serialises the
// ring.  In assembly nothing is waited for automatically: lds_wait() below is the s_waitcnt lgkmcnt(0) for them, tied to
// the destination registers so that no MFMA that uses them can be scheduled above it.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int OFS = 0>  // (byte offset in the instruction's immediate field: the second limbs lie 256 bytes behind the first)
__device__ __forceinline__ f16x8 tr_read8(unsigned a0, unsigned a1) {
  u32x2_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(OFS));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(OFS));
  const u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(f16x8, r);
}
__device__ __forceinline__ void lds_wait(f16x8& a, f16x8& b, f16x8& c_, f16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c_), "+v"(d));
}

template <int WM, int WN, int STAGES, int OCC, int MODE = 0>
__global__ __launch_bounds__(WM * WN * 64, OCC) void conv_wgrad_l2_kernel(const WgradK a, unsigned xplane, unsigned dyplane, unsigned x_bytes,
                                                                         unsigned dy_bytes, const float* __restrict__ xamax,
                                                                         const float* __restrict__ dyamax) {
  constexpr int NW = WM * WN;
  constexpr int SA = (64 * WM) / 128, SB = (64 * WN) / 128;  // 128-channel sub-images of dy / x
  static_assert(SA >= 1 && SB >= 1, "tiles are multiples of 128 channels");
  constexpr int SUB = 32 * 512;                              // one sub-image (128 channels, both limbs): 32 pixels x 512 B
  constexpr int A_BYTES = SA * SUB, STAGE = A_BYTES + SB * SUB;
  constexpr int GPW = 8 / NW;                                // 4-pixel groups per wave
  static_assert(GPW >= 1 && 8 % NW == 0, "8 pixel groups per K-step");
  constexpr int DPW = GPW * (SA + SB) * 2;                   // LDS-DMA instructions per wave per K-step
  constexpr bool STAGGER = NW == 8;
  constexpr int MAX_KT = 2048;  // K-steps of one workgroup (pixel range / 32); the host splits longer ranges
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * STAGE + MAX_KT * 2];

  const OndaConv& c = a.c;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool late = wave >= NW / 2;

  // workgroups are dealt round-robin over the 8 XCDs: hand each XCD a contiguous band of the (pixel range, tap,
  // channel tile) order, so that the workgroups that stream the same dy rows (all input-channel tiles of one tap and
  // pixel range) and the same x rows (the taps of one pixel range) find them in one L2
  int bid;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int tile_c = bid % a.tilesC;
  bid /= a.tilesC;
  const int tap = bid % a.taps;
  bid /= a.taps;
  const int tile_n = bid % a.tilesN;
  const int ks = bid / a.tilesN;
  const int n0 = tile_n * (64 * WM), c0 = tile_c * (64 * WN);
  const int mbeg = ks * a.mchunk;
  const int mend = min(a.M, mbeg + a.mchunk);
  const int KT = mend > mbeg ? (mend - mbeg + 31) / 32 : 0;
  const int rr = tap / c.kw, ss = tap - rr * c.kw;
  const int dh = rr * c.dil - c.pad, dw = ss * c.dil - c.pad;
  unsigned long long* stp = a.stamps != nullptr && t == 0 && blockIdx.x < 4096 ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
  if (stp) stp[0] = __builtin_amdgcn_s_memtime();
  // Both operands as buffers that start where this workgroup's pixel range starts (dy: its first pixel; x: the first image that
  // pixel lies in): the 32-bit offsets span one pixel range (<= 65 536 pixels and the images they touch), the tensors may be
  // larger than 2 GiB (4 + 4 images of 1024 x 2048: 2.17 GB of 2048-channel limb rows)
  constexpr unsigned PAST = 0x7FFFF000u;  // an offset at or behind the end of any window, that OOB + PAST does not wrap
  const int b_first = mbeg / (c.Ho * c.Wo);
  const long long x_at = (long long)b_first * c.Hi * c.Wi * c.ldx * 4, dy_at = (long long)mbeg * a.lddy * 4;
  const long long x_left = a.x_total - x_at;
  // (MODE 1: the dy window ends with the pixel range -- rows past it read as zeros, no per-pixel test)
  const long long dy_left = MODE == 1 ? (long long)(mend - mbeg) * a.lddy * 4 : a.dy_total - dy_at;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const char*>(a.x) + x_at, (unsigned)(x_left < PAST ? (x_left > 0 ? x_left : 0) : PAST));
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(reinterpret_cast<const char*>(a.dy) + dy_at, (unsigned)(dy_left < PAST ? (dy_left > 0 ? dy_left : 0) : PAST));
  const Scale2 sx = scale_of(xamax), sd = scale_of(dyamax);
  const float unscale_a = sx.inv, unscale_b = sd.inv;

  // The live K-steps of this workgroup, listed once into LDS (a K-step -- 32 pixels -- is dead when every output row it
  // touches maps to an input row outside the image for this tap: whole rows of a dilated tap).  The integer divisions
  // happen here, a few per thread, instead of several per K-step on the scalar unit.
  unsigned short* live_list = reinterpret_cast<unsigned short*>(lds + STAGES * STAGE);
  int nlive = 0;
  {
    unsigned char* flags = lds;  // scratch: the ring is not in use yet
    for (int kt = t; kt < KT; kt += NW * 64) {
      const int m_first = mbeg + kt * 32, m_last = min(mend, m_first + 32) - 1;
      const int r_first = m_first / c.Wo, r_last = m_last / c.Wo;  // global output row (b*Ho + ho)
      bool live = false;
      for (int r = r_first; r <= r_last; ++r) live |= (unsigned)((r % c.Ho) * c.stride + dh) < (unsigned)c.Hi;
      flags[kt] = live;
    }
    __syncthreads();
    for (int base = 0; base < KT; base += 64) {  // every wave compacts the whole list (same result; no second barrier needed
      const bool f = base + lane < KT && flags[base + lane];  // before the list is read: each wave reads what it wrote)
      const unsigned long long mask = __ballot(f);
      if (f) live_list[nlive + __popcll(mask & ((1ull << lane) - 1))] = (unsigned short)(base + lane);
      nlive += __popcll(mask);
    }
    __syncthreads();  // (all waves wrote identical values; the barrier also frees `flags` for the ring)
  }
  auto live_at = [&](int i) -> int { return __builtin_amdgcn_readfirstlane((int)live_list[i]); };

  // DMA roles: this wave moves pixel groups grp = wave*GPW + d (4 pixels each) of every K-step, both operands: two
  // instructions h = 0, 1 per group and 128-channel half, each 2 pixels x 512 B (lanes 0-31: one pixel, its 16 first-limb chunks
  // then its 16 second-limb chunks)
  const int lam = lane >> 5, limb = (lane >> 4) & 1;
  unsigned ch_dy[GPW][2][SA], ch_x[GPW][2][SB];
#pragma unroll
  for (int d = 0; d < GPW; ++d)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = (wave * GPW + d) * 4 + 2 * h + lam;  // LDS row this lane fills
      const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
      const int ch = (lane & 15) ^ swz;  // data chunk (8 channels) that lands in this lane's LDS slot
#pragma unroll
      for (int sI = 0; sI < SA; ++sI) {
        const int n = n0 + sI * 128 + ch * 8;
        ch_dy[d][h][sI] = n < c.Cout ? (unsigned)(limb_at(0, n, 0) + limb * LIMB2_OFS) * 2u : PAST;  // past the buffer whatever pixel offset is added
      }
#pragma unroll
      for (int sI = 0; sI < SB; ++sI) {
        const int cc = c0 + sI * 128 + ch * 8;
        ch_x[d][h][sI] = cc < c.Cin ? (unsigned)(limb_at(0, cc, 0) + limb * LIMB2_OFS) * 2u : PAST;
      }
    }
  // this lane's FIRST pixel of each group (its second one is the next pixel): (image, output row, output column), advanced
  // incrementally (no division in the loop)
  int p_m[GPW], p_wo[GPW], p_ho[GPW], p_b[GPW];
#pragma unroll
  for (int d = 0; d < GPW; ++d) {
    const int m = mbeg + (wave * GPW + d) * 4 + 2 * lam;
    p_m[d] = m;
    p_wo[d] = m % c.Wo;
    const int tq = m / c.Wo;
    p_ho[d] = tq % c.Ho;
    p_b[d] = tq / c.Ho - b_first;  // (relative to the window of x)
  }
  int k_decoded = 0;  // the K-step p_* stand at
  // A step's DMAs in two parts -- first limb planes, second limb planes -- so that the staggered kernel can issue one
  // part in the prepare slot and the other behind the first MFMAs of the compute slot.  Stamps (tools/wgrad_stamps.py):
  // 2 650 cycles per K-step with every workgroup within 10 of that; the first half's prepare slot = 620 (live-list
  // lookup, pixel offsets, DMA issue) + 640 (32 transposed fragment reads: LDS bandwidth of four waves) + 140 at the
  // barrier, its compute slot 768 of MFMA.  With the split: 2 550.  Tried and slower: reads issued before the DMAs
  // (2 950: the read issue itself blocks for 780 cycles and the DMAs behind it for 1 000), next step's offsets worked
  // out in the compute slot (3 190: branchy VALU code and an LDS lookup break the MFMA stream).
  unsigned pdy_s[GPW][2], px_s[GPW][2];
  int stage_s = 0;
  auto issue_part = [&](int h) {  // the h-th instruction of every group: LDS rows 4 grp + 2 h, + 1
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int d = 0; d < GPW; ++d) {
      const int grp = wave * GPW + d;
#pragma unroll
      for (int sI = 0; sI < SA; ++sI) {
        unsigned char* dst = lds + stage_s + sI * SUB + (grp * 2 + h) * 1024;
        const unsigned vo = pdy_s[d][h] + ch_dy[d][h][sI];  // OOB (2^31) + anything below 2^31 stays out of range, no wrap
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (__attribute__((address_space(3))) void*)dst, 16, vo, 0, 0, 0);
      }
#pragma unroll
      for (int sI = 0; sI < SB; ++sI) {
        unsigned char* dst = lds + stage_s + A_BYTES + sI * SUB + (grp * 2 + h) * 1024;
        const unsigned vo = px_s[d][h] + ch_x[d][h][sI];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)dst, 16, vo, 0, 0, 0);
      }
    }
#else
    (void)h;
#endif
  };
  auto issue_addr = [&](int kt, int stage_off) {  // the step's pixel offsets (kept for both parts)
    stage_s = stage_off;
    const int adv = (kt - k_decoded) * 32;
    k_decoded = kt;
#pragma unroll
    for (int d = 0; d < GPW; ++d) {
      p_m[d] += adv;
      p_wo[d] += adv;
      while (p_wo[d] >= c.Wo) {
        p_wo[d] -= c.Wo;
        if (++p_ho[d] == c.Ho) {
          p_ho[d] = 0;
          ++p_b[d];
        }
      }
      int m = p_m[d], wo = p_wo[d], ho = p_ho[d], b = p_b[d];
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // this lane's two pixels: m and m + 1
        unsigned pdy = OOB, px = OOB;
        if (m < mend) {
          pdy = (unsigned)(m - mbeg) * (unsigned)a.lddy * 4u;  // a limb row: 4 bytes per channel
          const int hi = ho * c.stride + dh, wi = wo * c.stride + dw;
          if ((unsigned)hi < (unsigned)c.Hi && (unsigned)wi < (unsigned)c.Wi) px = (unsigned)(((b * c.Hi + hi) * c.Wi + wi) * c.ldx) * 4u;
        }
        pdy_s[d][h] = pdy;
        px_s[d][h] = px;
        ++m;
        if (++wo == c.Wo) {
          wo = 0;
          if (++ho == c.Ho) {
            ho = 0;
            ++b;
          }
        }
      }
    }
  };
  auto issue = [&](int kt, int stage_off) {
    issue_addr(kt, stage_off);
    issue_part(0);
    issue_part(1);
  };

  // transposed fragment reads: 16-lane group g = k-group (pixels 8g .. 8g+7), lane 4q+p of the group addresses row q,
  // 8 bytes p & 1 of chunk (p >> 1) of the tile's two chunks
  unsigned fa[4][2], fb[4][2];
  {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * g + 4 * h + q;
      const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cha = (wm & 1) * 8 + 2 * i + (pp >> 1), chb = (wn & 1) * 8 + 2 * i + (pp >> 1);
        fa[i][h] = (unsigned)((wm >> 1) * SUB + 512 * row + 16 * (cha ^ swz) + 8 * (pp & 1));  // (+ 256: the second limbs)
        fb[i][h] = (unsigned)(A_BYTES + (wn >> 1) * SUB + 512 * row + 16 * (chb ^ swz) + 8 * (pp & 1));
        // (opaque: left alone the compiler recomputes all sixteen from the lane id in every K-step -- 65 VALU instructions
        //  in the prepare slot -- rather than keep them in registers)
        asm volatile("" : "+v"(fa[i][h]), "+v"(fb[i][h]));
      }
    }
  }

  f32x4 acc[4][4], accx[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = accx[i][j][e] = 0.f;

  if (stp) {
    stp[1] = __builtin_amdgcn_s_memtime();
    stp[5] = (unsigned long long)nlive;
  }
  // the live K-steps, fetched two ahead
  int i_cur = 0, i_iss = 0;
  int st_issue = 0, st_read = 0;
  auto issue_next = [&]() {
    issue(live_at(i_iss), st_issue);
    st_issue = st_issue + STAGE == STAGES * STAGE ? 0 : st_issue + STAGE;
    ++i_iss;
  };
  bool second_part_due = false;
  auto issue_first_part = [&]() {   // prepare slot
    second_part_due = i_iss < nlive;
    if (second_part_due) {
      issue_addr(live_at(i_iss), st_issue);
      issue_part(0);
      st_issue = st_issue + STAGE == STAGES * STAGE ? 0 : st_issue + STAGE;
      ++i_iss;
    }
  };
  auto issue_second_part = [&]() {  // compute slot, behind the first 16 MFMAs
    if (second_part_due) issue_part(1);
    second_part_due = false;
  };
  auto wait_landed_half = [&](bool first_part_in_flight) {  // (second half of the workgroup: only the first part of the
    if (first_part_in_flight)                               //  younger step has been issued when it waits)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW / 2) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto wait_landed = [&](bool more_in_flight) {
    if (more_in_flight)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  constexpr int AHEAD = STAGES >= 3 ? 2 : 1;  // (two-stage ring: two workgroups per CU cover each other's waits)
  static_assert(!STAGGER || STAGES >= 3, "the staggered halves need two steps in flight");
  if constexpr (MODE == 0) {
    if (i_iss < nlive) issue_next();
    if (AHEAD == 2 && i_iss < nlive) issue_next();
  }
  f16x8 af[4][2], bf[4], b1[4];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  unsigned base;
  auto prepare = [&]() {  // "P": issue the transposed reads of the a1, a2, b2 fragments of the stage at st_read
    base = lds0 + st_read;
    st_read = st_read + STAGE == STAGES * STAGE ? 0 : st_read + STAGE;
    // (one address add per register pair -- the stage -- and the limb in the immediate: every VALU instruction of the prepare
    //  slot is paid by the OTHER wave of the SIMD, whose MFMAs it delays: 32 of them measured -16 % on the forward kernel)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned a0 = base + fa[i][0], a1 = base + fa[i][1];
      af[i][0] = tr_read8<0>(a0, a1);
      af[i][1] = tr_read8<256>(a0, a1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned a0 = base + fb[j][0], a1 = base + fb[j][1];
      bf[j] = tr_read8<256>(a0, a1);
      if constexpr (STAGGER) b1[j] = tr_read8<0>(a0, a1);  // staggered halves: the compute slot reads nothing from LDS (see conv_l2_kernel)
    }
  };
  auto prepared = [&]() {  // ... and wait for them (before the barrier that lets the stage be refilled / before the MFMAs)
    lds_wait(af[0][0], af[1][0], af[2][0], af[3][0]);
    lds_wait(af[0][1], af[1][1], af[2][1], af[3][1]);
    lds_wait(bf[0], bf[1], bf[2], bf[3]);
    if constexpr (STAGGER) lds_wait(b1[0], b1[1], b1[2], b1[3]);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&]() {  // "C" (one wave per SIMD: the b1 fragments arrive behind the first 16 MFMAs)
    if constexpr (!STAGGER) {
#pragma unroll
      for (int j = 0; j < 4; ++j) b1[j] = tr_read8(base + fb[j][0], base + fb[j][1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
    if constexpr (!STAGGER) lds_wait(b1[0], b1[1], b1[2], b1[3]);
    if constexpr (STAGGER) issue_second_part();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], b1[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], b1[ZZ(i, jj)], acc[i][ZZ(i, jj)], 0, 0, 0);
  };
  if constexpr (MODE >= 1) {
    // MODE 1 / 2 (round 5; MODE 0 above is the loop of rounds 2-4, kept for the four-wave tile and as the fallback).  The stamps
    // said the PREPARE slot was the long one (620 cycles of pixel arithmetic + DMA issue, 640 of fragment reads, against 768
    // of MFMA in the compute slot), so the pixel arithmetic is gone and the DMA issue is shared between the slots:
    //   * a step's pixel offsets cost a dozen VALU instructions, two per MFMA under the compute slot's first 16 MFMAs:
    //     MODE 1 is the 1 x 1 stride-1 convolution (x pixel = output pixel); MODE 2 reads the wave's four input pixels of a
    //     step from the geometry's table (a.pix) with ONE scalar load, issued a slot early (branchy index arithmetic among
    //     the MFMAs had lost, see above);
    //   * its first three DMAs follow there, one per two MFMAs, its other three at the start of the NEXT prepare slot;
    //   * the late half runs one step further ahead than the early one (in its compute slot the stage it refills has just
    //     been read for the last time -- by itself), so every DMA has two slots of flight or more.
    // Measured (tools/ab_conv_shapes.sh, one box): 313-376 -> 390-500 TFLOP/s per shape.
    static_assert(STAGGER && GPW == 1 && STAGES == 3, "the eight-wave tile");
    const unsigned lddy4 = (unsigned)a.lddy * 4u, ldx4 = (unsigned)c.ldx * 4u;
    unsigned dy_lane[2], x_lane[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned pix = (unsigned)(wave * 4 + 2 * lam + h);
      dy_lane[h] = pix * lddy4;
      x_lane[h] = (pix + (unsigned)(mbeg - b_first * c.Ho * c.Wo)) * ldx4;
    }
    const int base_px = b_first * c.Hi * c.Wi;  // first pixel of the x window
    const i32x4* tab = MODE == 2 ? reinterpret_cast<const i32x4*>(a.pix + (size_t)tap * a.pix_stride + mbeg + wave * 4) : nullptr;
    i32x4 t_next = {-1, -1, -1, -1};
    // Every slot issues: behind the last live step the offsets get `kill` added (out of range: the DMA writes zeros into a
    // stage nobody reads any more), so that the ring's waits are the same constant everywhere and the loop body is ONE basic
    // block -- which is what lets the offsets and the DMA issue be interleaved with the MFMAs below.
    auto offsets = [&](int kt, const i32x4& te, unsigned kill) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pdy_s[0][h] = (unsigned)kt * 32u * lddy4 + dy_lane[h] + kill;
        if constexpr (MODE == 1) {
          px_s[0][h] = (unsigned)kt * 32u * ldx4 + x_lane[h] + kill;
        } else {
          const int lo = te[h], hi = te[2 + h];
          const int e = lam ? hi : lo;
          px_s[0][h] = (e >= 0 ? __umul24((unsigned)(e - base_px), ldx4) : PAST) + kill;  // (a window is < 2^24 pixels, a row < 2^24 bytes)
        }
      }
    };
    auto issue_first = [&]() {
      stage_s = st_issue;
      issue_part(0);
    };
    auto issue_second = [&]() {
      issue_part(1);
      st_issue = st_issue + STAGE == STAGES * STAGE ? 0 : st_issue + STAGE;
      ++i_iss;
    };
    auto issue_dmas = [&]() {  // (the prologue's whole steps)
      issue_first();
      issue_second();
    };
    // (a SCALAR load, in assembly: the compiler makes the table read a vector load, which would sit in the vmcnt queue that
    //  the ring's waits count; scalar loads return through lgkmcnt, which every prepare slot drains anyway.  The value is
    //  valid behind the next entry_wait only.)
    const unsigned long long tab_u = reinterpret_cast<unsigned long long>(tab);
    const unsigned tab_lo = __builtin_amdgcn_readfirstlane((unsigned)tab_u), tab_hi = __builtin_amdgcn_readfirstlane((unsigned)(tab_u >> 32));
    auto entry = [&](int kt) -> i32x4 {
      i32x4 r = {0, 0, 0, 0};
      if constexpr (MODE == 2) {
        const unsigned long long p = (((unsigned long long)tab_hi << 32) | tab_lo) + (unsigned long long)(unsigned)kt * 128ull;
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(r) : "s"(p));
      }
      return r;
    };
    auto entry_wait = [&](i32x4& r) {
      if constexpr (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));
    };
    const int n_pre = late ? 3 : 2;
    if (nlive > 0) {  // (a workgroup whose pixel range is empty or all padding stores a slab of zeros)
      for (int j = 0; j < n_pre; ++j) {
        const int kt = live_at(j < nlive ? j : 0);
        i32x4 te = entry(kt);
        entry_wait(te);
        offsets(kt, te, j < nlive ? 0u : PAST);
        issue_dmas();
      }
      int kt_next = live_at(i_iss < nlive ? i_iss : 0);
      t_next = entry(kt_next);
      entry_wait(t_next);
      if (late) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPW) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      for (; i_cur < nlive; ++i_cur) {
        // the second three DMAs of a step go out at the start of the NEXT prepare slot (a DMA instruction holds its wave's
        // issue port for ~60 cycles wherever it stands: three beside the reads, three beside the MFMAs -- all six in the
        // compute slot: 4-12 % slower).  Early half: behind step i_cur only the first three of the next step are in flight.
        if (!late) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW / 2) : "memory");
        __builtin_amdgcn_s_barrier();
        if (i_cur > 0) issue_second();
        // the step this wave issues in its NEXT compute slot: its index from the live list now (an LDS read among the fragment
        // reads), its table entry by a scalar load that has a whole slot to arrive
        const unsigned short ktv = live_list[i_iss + 1 < nlive ? i_iss + 1 : 0];
        prepare();
        prepared();
        const int kt_after = __builtin_amdgcn_readfirstlane((int)ktv);
        i32x4 t_after = entry(kt_after);
        if (late) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
        __builtin_amdgcn_s_barrier();
        offsets(kt_next, t_next, i_iss < nlive ? 0u : PAST);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
        issue_first();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) accx[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], b1[ZZ(i, jj)], accx[i][ZZ(i, jj)], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[i][ZZ(i, jj)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], b1[ZZ(i, jj)], acc[i][ZZ(i, jj)], 0, 0, 0);
        // the order the scheduler is asked for: the offsets' VALU two per MFMA under the first 16, then one DMA (its m0, its
        // address add) per two MFMAs, then the rest of the MFMAs
        constexpr int NI = DPW / 2;
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // (an MFMA-only head: VALU in a compute segment's first ~250 cycles does not overlap)
#pragma unroll
        for (int q = 0; q < 14; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 48 - 16 - 14 - 2 * NI, 0);
        __builtin_amdgcn_sched_barrier(0);
        entry_wait(t_after);  // (issued a slot ago: nothing to wait for)
        kt_next = kt_after;
        t_next = t_after;
      }
      if (!late) __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the zero-fill DMAs behind the last step: the slab store reuses the ring)
    }
  } else if constexpr (!STAGGER) {
    for (; i_cur < nlive; ++i_cur) {
      wait_landed(AHEAD == 2 && i_cur + 1 < nlive);
      __builtin_amdgcn_s_barrier();
      if (i_iss < nlive) issue_next();
      prepare();
      prepared();
      compute();
    }
  } else {
    if (late && nlive > 0) {
      wait_landed(nlive > 1);
      __builtin_amdgcn_s_barrier();
    }
    for (; i_cur < nlive; ++i_cur) {
      if (!late) wait_landed(i_cur + 1 < nlive);
      __builtin_amdgcn_s_barrier();
      unsigned long long tk0 = 0, tk1 = 0, tk2 = 0;
      if (stp) tk0 = __builtin_amdgcn_s_memtime();
      issue_first_part();  // both halves: first limb planes here, second ones behind the first MFMAs of the compute slot
      if (stp) tk1 = __builtin_amdgcn_s_memtime();
      prepare();
      prepared();
      if (stp) tk2 = __builtin_amdgcn_s_memtime();
      if (late && i_cur + 1 < nlive) wait_landed_half(i_cur + 2 < nlive);
      __builtin_amdgcn_s_barrier();
      if (stp && i_cur == nlive / 2) {  // one step in the middle: DMA issue | fragment reads | wait at the barrier
        const unsigned long long tk3 = __builtin_amdgcn_s_memtime();
        stp[4] = ((tk1 - tk0) << 40) | ((tk2 - tk1) << 20) | (tk3 - tk2);
        stp[6] = tk3;
      }
      compute();
      if (stp && i_cur == nlive / 2) stp[7] = __builtin_amdgcn_s_memtime() - stp[6];
    }
    if (!late && nlive > 0) __builtin_amdgcn_s_barrier();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = ((acc[i][j] + accx[i][j] * LIMB2_UNSCALE) * unscale_a) * unscale_b;
  if (stp) stp[2] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  // slab store through the wave's own 4 KiB of LDS: 16 rows (output channels) x 64 input channels at a time, 16-byte stores
  float* tr = reinterpret_cast<float*>(lds + (t >> 6) * 4096);
  const int cl = (lane & 15) * 4, rl = lane >> 4;
  const int cc = c0 + wn * 64 + cl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int e = 0; e < 4; ++e) tr[(4 * (lane >> 4) + e) * 64 + jn * 16 + (lane & 15)] = acc[i][jn][e];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * r + rl;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tr + row * 64 + cl);
      const int n = n0 + (wm * 4 + i) * 16 + row;
      if (n < c.Cout && cc < c.Cin) *reinterpret_cast<f32x4*>(a.slabs + (((size_t)ks * c.Cout + n) * a.taps + tap) * c.Cin + cc) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (stp) stp[3] = __builtin_amdgcn_s_memtime();
}

}  // namespace

extern "C" {

int onda_split_h2(const float* x, int64_t rows, int C, int ldx, void* dst, int ldo, int64_t plane, const float* amax,
                  onda_stream_t s) {
  // limb rows (common.h limb_at): a row is ldo / 32 blocks of [32 x l1][32 x l2] -- a stride that is not a multiple of 32 would
  // put the channels past the last whole block into the next row (`plane` is ignored since round 5: the limbs share a row)
  (void)plane;
  ONDA_REQUIRE(x && dst && amax && rows > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldx % 4 == 0 && ldo >= C && ldo % 32 == 0);
  if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(dst)) return ONDA_EALIGN;
  const long long n = rows * (C / 8);
  const int blocks = (int)(n / 256 / 4 + 1 > 2048 ? 2048 : n / 256 / 4 + 1);
  hipLaunchKernelGGL(split_h2_kernel, dim3(blocks), dim3(256), 0, ONDA_STREAM(s), x, (long long)rows, C, ldx,
                     static_cast<_Float16*>(dst), ldo, (long long)plane, amax);
  return ONDA_LAUNCH_RESULT();
}

int onda_stem_im2col_l2(const float* x_nchw, const float* xamax, void* dst, int64_t plane, int B, int H, int W, int Ho, int Wo,
                        int Kp, onda_stream_t s) {
  (void)plane;
  ONDA_REQUIRE(x_nchw && xamax && dst && Kp % 32 == 0 && Kp >= 147 && B > 0 && Ho > 0 && Wo > 0);  // limb rows: whole 32-channel blocks
  if (!ONDA_ALIGNED16(dst)) return ONDA_EALIGN;
  const long long n = (long long)B * Ho * Wo * (Kp / 8);
  const int blocks = (int)(n / 256 / 2 + 1 > 4096 ? 4096 : n / 256 / 2 + 1);
  hipLaunchKernelGGL(stem_im2col_l2_kernel, dim3(blocks), dim3(256), 0, ONDA_STREAM(s), x_nchw, static_cast<_Float16*>(dst),
                     (long long)plane, B, H, W, Ho, Wo, Kp, xamax);
  return ONDA_LAUNCH_RESULT();
}

// tile variant of the pre-split conv for an (M, Cout) problem: 0 = 256 x 128 (8 waves), 1 = 128 x 128, 2 = 256 x 64
int onda_conv_l2_variant(int64_t M, int Cout) {
  if (const char* e = getenv("ONDA_L2_VARIANT")) return atoi(e);
  if (Cout <= 64) return 2;
  const long long t256 = ((M + 255) / 256) * ((Cout + 127) / 128);
  return t256 >= 200 ? 0 : 1;
}

// ... and for an (M, Cout, K) problem: short K loops whose epilogue outweighs them run faster as 128 x 128 tiles on TWO
// four-wave workgroups per CU (one's epilogue beside the other's K loop) than in the 256 x 128 stream kernel -- when there
// are many column tiles to share the activation rows in L2, or so few K-steps that the tile is all epilogue.  Measured per
// shape (tools/ab_conv_shapes.sh with ONDA_L2_VARIANT=1, profiles/r05_short_k_tiles_ab.txt): -4...-23 % where the rule
// says yes at 4 images, +2...+15 % for 256 output channels with 16-32 K-steps, where it says no; at 8 images the data
// gradients gain the same and the train-mode forwards are +-3 %; the step: 92.81 -> 92.35 ms (three alternating runs).
static int l2_variant_k(long long M, int Cout, int taps, int Cin) {
  const int v = onda_conv_l2_variant(M, Cout);
  static const int on = getenv("ONDA_L2_SHORTK128") ? atoi(getenv("ONDA_L2_SHORTK128")) : 1;
  if (v != 0 || !on || getenv("ONDA_L2_VARIANT")) return v;
  const int kts = taps * (Cin / 32);
  return kts <= 8 || (kts <= 16 && Cout >= 512) || (kts <= 32 && Cout >= 2048) ? 1 : 0;
}

// ... and 1 x 1 convolutions with 64 / 128 / 256 input channels and at least 128 output channels run ACTIVATION-STATIONARY
// (conv_l2a_kernel: a workgroup's 128 rows in registers, the weight rows streamed): the same 128-row statistic tiles as
// variant 1, no stream-K remainder.  ONDA_L2_STATIONARY=1 switches it on (default: the tile kernels of round 5).
static bool l2_stationary(long long M, int Cout, int taps, int Cin) {
  // OFF by default: measured alone the kernel is 7-18 % faster than the 128 x 128 tile kernel on the train-mode forward convolutions
  // it takes (256 -> 1024: 76-77 us against 82.7; 128 -> 512: 33.5 against 39.4; 64 -> 256 at 129 x 257: 52 against 64), the
  // adaptation step does not move (94.88 / 94.87 ms on, 94.77 / 94.72 off, alternating runs on one box): profiles/r06_l2a_*.txt
  const char* e = getenv("ONDA_L2_STATIONARY");  // (read per call: the kernel's own test switches it on for its launches)
  if (e == nullptr || atoi(e) == 0 || getenv("ONDA_L2_VARIANT")) return false;
  // (at least two column tiles: with one, nothing amortises a panel's rows)
  return taps == 1 && (Cin == 64 || Cin == 128 || Cin == 256) && Cout >= 256 && l2_variant_k(M, Cout, taps, Cin) == 1;
}

/* which device kernel onda_conv2d_fwd_l2 launches for a problem: the tile variant (0: 256 x 128, 1: 128 x 128, 2: 256 x 64
 * = conv_l2_kernel<4,2> / <2,2> / <4,1>), 3: conv_l2x_kernel<4,2>, the continuous K-step stream taken by 256 x 128
 * problems with at most 32 K-steps per tile, or 4: conv_l2a_kernel, the activation-stationary 1 x 1 kernel (bench.py names
 * its per-kernel figures after this) */
// K-steps per tile up to which a 256 x 128 problem runs as the continuous K-step stream (conv_l2x_kernel).  (Round 6 re-measured
// the threshold: 72-step tiles -- 256 -> 256 3 x 3 -- 130 us on the stream kernel against 120.5, 144-step tiles 391 against 363.)
static constexpr int l2x_max_ksteps() { return 32; }

int onda_conv_l2_kernel_id(int64_t M, int Cout, int taps, int Cin) {
  if (l2_stationary(M, Cout, taps, Cin)) return 4;
  const int variant = l2_variant_k(M, Cout, taps, Cin);
  const bool short_k = taps * (Cin / 32) <= l2x_max_ksteps();
  return variant == 0 && short_k ? 3 : variant;
}

}  // extern "C"

namespace {
// How an (M, Cout, K) problem is scheduled: whole rounds of one tile per workgroup; a remainder of tiles (tiles mod the
// resident workgroups) is cut into equal K ranges over all workgroups when that pays (hybrid stream-K).
struct L2Schedule {
  int variant, BM, BN, tilesM, tilesN, G, rem, sub;
  bool balanced, stationary;
  int rem_rows() const { return tilesM - (tilesM * tilesN - rem) / tilesN; }  // tile rows that hold remainder tiles
  int stats_rows_total() const { return tilesM + (balanced ? rem_rows() * (sub - 1) : 0); }
};
bool l2_small_ring2() { return true; }  // (the four-wave tiles on a three-stage ring, one workgroup per CU, lost: DESIGN.md section 3)

L2Schedule l2_schedule(long long M, int Cout, int taps, int Cin, bool have_ws, long long stat_split = 0, bool plain = false) {
  L2Schedule q;
  q.variant = l2_variant_k(M, Cout, taps, Cin);
  q.BM = q.variant == 1 ? 128 : 256;
  q.BN = q.variant == 2 ? 64 : 128;
  q.tilesM = (int)((M + q.BM - 1) / q.BM);
  q.tilesN = (Cout + q.BN - 1) / q.BN;
  // 256 x 128 tiles: one workgroup per CU (144 KB of LDS, 3-stage ring).  The four-wave tiles (128 x 128, 256 x 64) run a
  // 2-stage ring (64 / 80 KB): two workgroups per CU, each covering the other's DMA waits, prologue and epilogue
  q.G = q.variant == 0 || !l2_small_ring2() ? conv_resident_workgroups() / 2 : conv_resident_workgroups();
  q.sub = q.BM / (256 / (q.BN / 4));
  const int tiles = q.tilesM * q.tilesN, KT = taps * (Cin / 32);
  q.rem = tiles % q.G;
  const double t_tile_us = 2.0 * q.BM * q.BN * taps * Cin / 1.4e6;  // one tile on one CU at ~360 TF/s chip-wide
  const double fix_us = 6.0 + (q.G + 2.0 * q.rem) * (q.BM * q.BN / 16384.0) * 0.02;  // partial tiles written + read
  q.balanced = have_ws && q.rem != 0 && KT >= 4 && t_tile_us * (1.0 - (double)q.rem / q.G) > fix_us &&
               (size_t)q.G * 2 * q.BM * q.BN <= (size_t)onda_conv_ws_floats() && q.G <= 1024;
  if (plain) q.balanced = false;  // OndaConv.plain_schedule
  q.stationary = l2_stationary(M, Cout, taps, Cin);
  if (q.stationary) q.balanced = false;  // contiguous runs of (row panel, column tile) items: balanced to one item by construction
  if (const int force = conv_sched_override()) {  // ONDA_CONV_SCHED: 1 tile-per-workgroup, 2 hybrid
    if (force == 1 || !have_ws) q.balanced = false;
    else if (force == 2) q.balanced = q.rem != 0;
  }
  // Two row groups with separate BatchNorm statistics (OndaConv.stat_split): the statistics row of the tile that straddles
  // the boundary is split by onda_bn_finalize_l2, which needs that row to cover the WHOLE tile -- a stream-K remainder tile
  // spreads its statistics over sub-block rows.  Remainder tiles are the last ones; keep the straddler out of their tile rows
  // (only tiny problems -- fewer tiles than two rounds -- lose their balanced schedule to this).
  // (the same when the boundary falls BETWEEN two tile rows: the remainder's extra statistic rows sit behind all regular
  //  rows, where the BatchNorm kernels count them with group 1 -- every remainder tile must lie in group 1)
  if (q.balanced && stat_split > 0 && (stat_split + q.BM - 1) / q.BM > (tiles - q.rem) / q.tilesN) q.balanced = false;
  return q;
}
}  // namespace

extern "C" {

/* rows of the `stats` partials the conv will write for this problem (tile rows + the extra rows of a stream-K remainder) */
int onda_conv_l2_tiles_m(int64_t M, int Cout, int taps, int Cin) { return l2_schedule(M, Cout, taps, Cin, true).stats_rows_total(); }
int onda_conv_l2_tiles_m_split(int64_t M, int Cout, int taps, int Cin, int64_t stat_split, int plain_schedule, int* tile_rows) {
  const L2Schedule q = l2_schedule(M, Cout, taps, Cin, true, stat_split, plain_schedule != 0);
  if (tile_rows) *tile_rows = q.BM;
  return q.stats_rows_total();
}

/* Measurement helpers (bench.py's `pipe.executed_tflops`): the share of a problem's K-steps the kernel really issues.
 * conv_l2_kernel skips, per whole tile, the filter taps that only see padding; the continuous-stream kernel and stream-K
 * pieces run the full K range; the weight gradient skips dead 32-pixel steps per tap.  Same arithmetic as the kernels. */
double onda_conv_l2_live_fraction(const OndaConv* c, int with_stats) {
  if (!c) return 1.0;
  const long long M = (long long)c->B * c->Ho * c->Wo;
  const int taps = c->kh * c->kw, kcper = c->Cin / 32;
  if (M <= 0 || taps <= 1 || kcper <= 0) return 1.0;
  const L2Schedule q = l2_schedule(M, c->Cout, taps, c->Cin, true, with_stats ? (long long)c->stat_split : 0, c->plain_schedule != 0);
  if (onda_conv_l2_kernel_id(M, c->Cout, taps, c->Cin) == 3) return 1.0;
  const int tiles = q.tilesM * q.tilesN, tiles_dp = q.balanced ? tiles - q.rem : tiles;
  long long live_steps = (long long)(tiles - tiles_dp) * taps * kcper;
  for (int tile_m = 0; tile_m * q.tilesN < tiles_dp; ++tile_m) {
    const int n_dp = tiles_dp - tile_m * q.tilesN < q.tilesN ? tiles_dp - tile_m * q.tilesN : q.tilesN;
    const long long m0 = (long long)tile_m * q.BM, m_last = (M < m0 + q.BM ? M : m0 + q.BM) - 1;
    const long long r0 = m0 / c->Wo, r1 = m_last / c->Wo;
    int live = 0;
    for (int tp = 0; tp < taps; ++tp) {
      const int dh = (tp / c->kw) * c->dil - c->pad;
      bool alive = false;
      for (long long r = r0; r <= r1 && !alive; ++r) alive = (unsigned)((int)(r % c->Ho) * c->stride + dh) < (unsigned)c->Hi;
      live += alive;
    }
    if (live == 0) live = 1;
    live_steps += (long long)n_dp * live * kcper;
  }
  return (double)live_steps / ((double)tiles * taps * kcper);
}

double onda_conv_wgrad_l2_live_fraction(const OndaConv* c, int splitk) {
  if (!c || splitk < 1) return 1.0;
  const long long M = (long long)c->B * c->Ho * c->Wo;
  const int taps = c->kh * c->kw;
  if (M <= 0 || taps <= 1) return 1.0;
  const long long mchunk = ((M + splitk - 1) / splitk + 31) / 32 * 32;
  long long live = 0, all = 0;
  for (int ks = 0; ks < splitk; ++ks) {
    const long long mbeg = ks * mchunk, mend = M < mbeg + mchunk ? M : mbeg + mchunk;
    const long long KT = mend > mbeg ? (mend - mbeg + 31) / 32 : 0;
    all += KT * taps;
    for (int tp = 0; tp < taps; ++tp) {
      const int dh = (tp / c->kw) * c->dil - c->pad;
      for (long long kt = 0; kt < KT; ++kt) {
        const long long m_first = mbeg + kt * 32, m_last = (mend < m_first + 32 ? mend : m_first + 32) - 1;
        bool alive = false;
        for (long long r = m_first / c->Wo; r <= m_last / c->Wo && !alive; ++r)
          alive = (unsigned)((int)(r % c->Ho) * c->stride + dh) < (unsigned)c->Hi;
        live += alive;
      }
    }
  }
  return all > 0 ? (double)live / (double)all : 1.0;
}

static int l2_fwd_impl(const void* xl, int64_t xplane, const float* xamax, const void* w2, const float* wamax, float* y,
                       const float* scale, const float* shift, const float* residual, float* stats, int stats_rows, float* ws,
                       float* yamax, const OndaConv* c, onda_stream_t s, const OndaLimbOut* lo) {
  ONDA_REQUIRE(xl && xamax && w2 && wamax && (y || lo) && c && ws && (stats_rows == 2 || stats_rows == 4));
  if (lo != nullptr) {  // limb-plane output: dense [M][ldy] planes, no statistics, residual (if any) as limb planes
    ONDA_REQUIRE(lo->out && lo->out_bound && lo->kb && lo->xtrue && !stats && !residual);
    ONDA_REQUIRE(c->out_os == 1 && c->Hf == c->Ho && c->Wf == c->Wo && c->ldy % 32 == 0 && c->ldy >= c->Cout);
    ONDA_REQUIRE(!lo->res || (lo->res_amax && c->ldr % 32 == 0 && c->ldr >= c->Cout));
    if (!ONDA_ALIGNED16(lo->out) || (lo->res && !ONDA_ALIGNED16(lo->res))) return ONDA_EALIGN;
  }
  ONDA_REQUIRE(c->Cin > 0 && c->Cin % 32 == 0 && c->Cout > 0 && c->Cout % 4 == 0 && c->ldx % 32 == 0 && c->ldx >= c->Cin);
  ONDA_REQUIRE(c->kh >= 1 && c->kw >= 1 && c->stride >= 1 && c->dil >= 1 && c->out_os >= 1);
  (void)xplane;  // (limb rows: the two limbs of a 32-channel block are 64 bytes apart inside the row; kept in the signature)
  if (!ONDA_ALIGNED16(xl) || !ONDA_ALIGNED16(w2)) return ONDA_EALIGN;
  // the epilogue stores (and reads the residual in) 16-byte vectors
  if (c->ldy % 4 != 0 || (y && !ONDA_ALIGNED16(y)) || (residual && (c->ldr % 4 != 0 || !ONDA_ALIGNED16(residual)))) return ONDA_EALIGN;
  if (scale && !ONDA_ALIGNED16(scale)) return ONDA_EALIGN;
  if (shift && !ONDA_ALIGNED16(shift)) return ONDA_EALIGN;
  ConvK k;
  k.x = static_cast<const float*>(xl); k.w = w2; k.y = y; k.scale = scale; k.shift = shift; k.res = residual; k.stats = stats; k.ws = ws;
  k.amax = yamax;
  k.stats_rows = stats_rows;
  if (lo != nullptr) {
    k.yl = static_cast<_Float16*>(lo->out);
    k.yplane = lo->out_plane;
    k.ybound = lo->out_bound;
    k.amax = lo->out_amax;
    k.kb = lo->kb;
    k.xtrue = lo->xtrue;
    k.resl = static_cast<const _Float16*>(lo->res);
    k.resplane = lo->res_plane;
    k.res_amax = lo->res_amax;
    k.res_true = lo->res_true ? lo->res_true : lo->res_amax;
  }
  k.skip_dead_taps = getenv("ONDA_L2A_WKB") ? 78 : 1;
  k.late_issue = 1;
  static const int stamp_on = getenv("ONDA_L2X_STAMP") ? atoi(getenv("ONDA_L2X_STAMP")) : 0;
  if (stamp_on)  // the last 64 KiB of the workspace (beyond anything the schedules use: checked below)
    k.stamps = reinterpret_cast<unsigned long long*>(ws + onda_conv_ws_floats()) - 1024 * 32;
  k.c = *c;
  const long long M = (long long)c->B * c->Ho * c->Wo;
  ONDA_REQUIRE(M > 0 && M < (1ll << 31));
  // RowPos packs a tile row's input coordinates of tap (0, 0) into 16 bits each (-32768 = no such row)
  ONDA_REQUIRE((long long)c->Ho * c->stride + (long long)c->dil * (c->kh - 1) < 32768 &&
               (long long)c->Wo * c->stride + (long long)c->dil * (c->kw - 1) < 32768 && c->pad < 32768);
  const long long x_total = (long long)c->B * c->Hi * c->Wi * c->ldx * 4;  // limb rows: 4 bytes per element
  {  // 32-bit byte offsets inside a tile's WINDOW (x_window: the images its <= 256 rows touch), not inside the tensor
    const long long img = (long long)c->Hi * c->Wi * c->ldx * 4, per_tile = 256 / ((long long)c->Ho * c->Wo) + 2;
    ONDA_REQUIRE((per_tile < c->B ? per_tile : c->B) * img < 0x7FFFF000ll);
  }
  k.x_total = x_total;
  k.M = (int)M;
  k.taps = c->kh * c->kw;
  k.kcper = c->Cin / 32;
  ONDA_REQUIRE(c->stat_split >= 0 && c->stat_split < M);
  const L2Schedule q = l2_schedule(M, c->Cout, k.taps, c->Cin, true, stats ? (long long)c->stat_split : 0, c->plain_schedule != 0);
  k.tilesM = q.tilesM;
  k.tilesN = q.tilesN;
  const size_t limb_elems = (size_t)c->Cout * k.taps * c->Cin;  // weight planes are [Cout][taps*Cin]
  ONDA_REQUIRE(limb_elems * 4 < (1ull << 31));
  const unsigned x_bytes = 0, w_bytes = (unsigned)(limb_elems * 4);  // (x: ConvK.x_total, windows per tile)
  const unsigned xpl = 0, wpl = 0;  // (unused by the kernels since the limbs share a row)
  const int tiles = k.tilesM * k.tilesN;
  k.tiles_dp = tiles - q.rem;
  hipStream_t st = ONDA_STREAM(s);
#define L2_LAUNCH(WM_, WN_, ST_, OCC_)                                                                                       \
  do {                                                                                                                       \
    if (q.balanced) {                                                                                                        \
      hipLaunchKernelGGL((conv_l2_kernel<WM_, WN_, ST_, OCC_, true>), dim3(q.G), dim3(WM_ * WN_ * 64), 0, st, k, xpl, wpl, x_bytes, \
                         w_bytes, xamax, wamax);                                                                            \
      hipLaunchKernelGGL((conv_l2_fixup_kernel<64 * WM_, 64 * WN_>), dim3(q.rem_rows() * q.tilesN, q.sub), dim3(256), 0, st, k, q.G, \
                         q.tilesM);                                                                                           \
    } else {                                                                                                                 \
      hipLaunchKernelGGL((conv_l2_kernel<WM_, WN_, ST_, OCC_, false>), dim3(tiles), dim3(WM_ * WN_ * 64), 0, st, k, xpl, wpl,    \
                         x_bytes, w_bytes, xamax, wamax);                                                                   \
    }                                                                                                                        \
  } while (0)
  // the output as a buffer: last byte any tile can store (dense rows of ldy floats; scattered stride-2 gradients included)
  const long long y_rows = (long long)c->B * (c->out_os == 1 && c->Hf == c->Ho && c->Wf == c->Wo ? (long long)c->Ho * c->Wo : (long long)c->Hf * c->Wf);
  const long long y_total = ((y_rows - 1) * c->ldy + c->Cout) * 4;
  const bool dense_out = c->out_os == 1 && c->Hf == c->Ho && c->Wf == c->Wo;  // (buffer stores relative to the tile: any size)
  // conv_l2a_kernel takes the plain epilogue (with or without statistics).  Everything else -- scale / shift / residual / ReLU /
  // max|y|, limb-row outputs, scattered outputs -- runs on the 128 x 128 tile kernel: same statistic rows, no stream-K remainder.
  const bool plain_epi = lo == nullptr && scale == nullptr && shift == nullptr && !c->relu && yamax == nullptr;
  const int epi = plain_epi && residual == nullptr ? 0 : -1;
  if (q.stationary && dense_out && epi >= 0) {
    // 1 x 1, Cin 64 / 128 / 256: the rows in registers, the weights streamed (conv_l2a_kernel)
    ONDA_REQUIRE(c->pad == 0 && (c->stride == 1 || (c->Hi >= (c->Ho - 1) * c->stride + 1 && c->Wi >= (c->Wo - 1) * c->stride + 1)));
    const long long items = (long long)k.tilesM * k.tilesN;
    const int cus = conv_resident_workgroups() / 2;
    const int slots = cus;  // one workgroup of 8 waves per CU
    const int grid = (int)(items < slots ? items : slots);
#define L2A_LAUNCH(KB_, NST_, A2L_, EPI_) \
  hipLaunchKernelGGL((conv_l2a_kernel<KB_, NST_, A2L_, EPI_>), dim3(grid), dim3(512), 0, st, k, w_bytes, (unsigned)y_total, xamax, wamax)
    if (c->Cin == 256) L2A_LAUNCH(8, 3, true, 0);
    else if (c->Cin == 128) L2A_LAUNCH(4, 3, false, 0);
    else L2A_LAUNCH(2, 3, false, 0);
#undef L2A_LAUNCH
    return ONDA_LAUNCH_RESULT();
  }
  // short K loops (1 x 1 convolutions up to 1024 input channels) gain 6-17 % from the continuous stream; long ones lose
  // ~4 % against the slot-staggered kernel, whose per-tile start / end they amortise anyway (measured per shape, one process)
  const bool short_k = k.taps * k.kcper <= l2x_max_ksteps();
  if (short_k && q.variant == 0 && (dense_out || y_total < 0x7FFFF000ll)) {
    if (!q.balanced) k.tiles_dp = tiles;  // persistent either way: whole tiles only
    const int grid = tiles < q.G ? tiles : q.G;
    hipLaunchKernelGGL((conv_l2x_kernel<4, 2, 3, 2>), dim3(q.balanced ? q.G : grid), dim3(512), 0, st, k, xpl, wpl, x_bytes, w_bytes,
                       (unsigned)y_total, xamax, wamax);
    if (q.balanced)
      hipLaunchKernelGGL((conv_l2_fixup_kernel<256, 128>), dim3(q.rem_rows() * q.tilesN, q.sub), dim3(256), 0, st, k, q.G, q.tilesM);
    return ONDA_LAUNCH_RESULT();
  }
#ifdef ONDA_L2_ABLATIONS  // measurement builds only (tools/README.md): ONDA_L2_DBG = 1 no vmcnt waits, 2 no DMA in the K loop,
  {                       // 7 half of the LDS fragment reads, 8 none, 9 half of the DMA instructions, 10 neither DMA nor reads, 11 nor barriers -- wrong results, valid timings; 12 = stamps of one K-step's slots into the workspace
                          // (tools/l2_slot_stamps.py)
    static const int dbg = getenv("ONDA_L2_DBG") ? atoi(getenv("ONDA_L2_DBG")) : 0;
#define L2_DBG_LAUNCH(D_)                                                                                                      \
  do {                                                                                                                         \
    if (q.balanced) {                                                                                                          \
      hipLaunchKernelGGL((conv_l2_kernel<4, 2, 3, 2, true, D_>), dim3(q.G), dim3(512), 0, st, k, xpl, wpl, x_bytes, w_bytes, xamax, \
                         wamax);                                                                                               \
      hipLaunchKernelGGL((conv_l2_fixup_kernel<256, 128>), dim3(q.rem_rows() * q.tilesN, q.sub), dim3(256), 0, st, k, q.G, q.tilesM); \
    } else {                                                                                                                   \
      hipLaunchKernelGGL((conv_l2_kernel<4, 2, 3, 2, false, D_>), dim3(tiles), dim3(512), 0, st, k, xpl, wpl, x_bytes, w_bytes, xamax, \
                         wamax);                                                                                               \
    }                                                                                                                          \
    return ONDA_LAUNCH_RESULT();                                                                                               \
  } while (0)
    if (q.variant == 0 && dbg == 5) L2_DBG_LAUNCH(5);  // per-workgroup ticks of set-up / K loop / epilogue into the workspace (tools/l2_tile_stamps.py)
    if (q.variant == 0 && dbg == 1) L2_DBG_LAUNCH(1);
    if (q.variant == 0 && dbg == 2) L2_DBG_LAUNCH(2);
    if (q.variant == 0 && dbg == 7) L2_DBG_LAUNCH(7);
    if (q.variant == 0 && dbg == 8) L2_DBG_LAUNCH(8);
    if (q.variant == 0 && dbg == 9) L2_DBG_LAUNCH(9);
    if (q.variant == 0 && dbg == 10) L2_DBG_LAUNCH(10);
    if (q.variant == 0 && dbg == 11) L2_DBG_LAUNCH(11);
    if (q.variant == 0 && dbg == 12) L2_DBG_LAUNCH(12);
#undef L2_DBG_LAUNCH
  }
#endif
  if (q.variant == 0) L2_LAUNCH(4, 2, 3, 2);
  else if (q.variant == 1 && l2_small_ring2()) L2_LAUNCH(2, 2, 2, 2);
  else if (q.variant == 1) L2_LAUNCH(2, 2, 3, 1);
  else if (l2_small_ring2()) L2_LAUNCH(4, 1, 2, 2);
  else L2_LAUNCH(4, 1, 3, 1);
#undef L2_LAUNCH
  return ONDA_LAUNCH_RESULT();
}

int onda_conv2d_fwd_l2(const void* xl, int64_t xplane, const float* xamax, const void* w2, const float* wamax, float* y,
                       const float* scale, const float* shift, const float* residual, float* stats, int stats_rows, float* ws,
                       float* yamax, const OndaConv* c, onda_stream_t s) {
  return l2_fwd_impl(xl, xplane, xamax, w2, wamax, y, scale, shift, residual, stats, stats_rows, ws, yamax, c, s, nullptr);
}

int onda_conv2d_fwd_l2_limbs(const void* xl, int64_t xplane, const float* xamax, const void* w2, const float* wamax,
                             const float* scale, const float* shift, const OndaLimbOut* lo, float* ws, const OndaConv* c,
                             onda_stream_t s) {
  ONDA_REQUIRE(lo != nullptr);
  return l2_fwd_impl(xl, xplane, xamax, w2, wamax, nullptr, scale, shift, nullptr, nullptr, 2, ws, nullptr, c, s, lo);
}

// weight-gradient tile of the pre-split kernel for a (Cout, Cin) problem: 0 = 256 output x 128 input channels (8 waves),
// 1 = 128 x 128 (4 waves)
static unsigned long long* g_debug_stamps = nullptr;
/* diagnostics: device buffer of 4096 x 8 uint64 the weight-gradient kernel timestamps its phases into (NULL: off) */
void onda_debug_stamps(void* p) { g_debug_stamps = static_cast<unsigned long long*>(p); }

int onda_conv_wgrad_l2_variant(int Cout, int Cin) {
  (void)Cin;
  return Cout >= 256 ? 0 : 1;
}

}  // extern "C"

namespace {
// Input pixel of every (tap, output pixel) of a convolution geometry: what the weight-gradient kernel's compute slot reads
// instead of working it out (conv_wgrad_l2_kernel, MODE 2).  [taps][stride] int32, -1 in the padding and behind the last
// pixel.  The CALLER owns the table (OndaConv.pix_table / pix_stride; onda_conv2d_wgrad_l2_table_stride sizes it,
// onda_conv2d_wgrad_l2_table fills it with one launch): the library neither allocates nor synchronises.  A table built for a
// larger batch of the same geometry serves a smaller one (same stride argument: the rows of a tap are `stride` apart).
__global__ __launch_bounds__(256) void wgrad_pixel_table_kernel(int* __restrict__ t, long long stride, int B, int Hi, int Wi, int Ho,
                                                               int Wo, int kw, int cstride, int dil, int pad) {
  const long long M = (long long)B * Ho * Wo;
  const int tap = blockIdx.y, rr = tap / kw, ss = tap - rr * kw;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < stride; m += (long long)gridDim.x * 256) {
    int v = -1;
    if (m < M) {
      const int wo = (int)(m % Wo);
      const long long q = m / Wo;
      const int ho = (int)(q % Ho), b = (int)(q / Ho);
      const int hi = ho * cstride + rr * dil - pad, wi = wo * cstride + ss * dil - pad;
      if ((unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi) v = (b * Hi + hi) * Wi + wi;
    }
    t[(size_t)tap * stride + m] = v;
  }
}
}  // namespace

extern "C" {

static bool wgrad_l2_linear(const OndaConv* c) {  // x pixel = output pixel: no table
  return c->kh == 1 && c->kw == 1 && c->stride == 1 && c->pad == 0 && c->Hi == c->Ho && c->Wi == c->Wo;
}

/* int32 entries between two taps' rows of the pixel table of this geometry AT THIS BATCH (the table is taps * stride
 * entries), or 0 when the problem runs without one (1 x 1 stride-1 convolutions, the 128 x 128 tile) */
int64_t onda_conv2d_wgrad_l2_table_stride(const OndaConv* c) {
  if (!c || c->B <= 0 || c->Ho <= 0 || c->Wo <= 0) return 0;
  if (onda_conv_wgrad_l2_variant(c->Cout, c->Cin) != 0 || wgrad_l2_linear(c)) return 0;
  if ((long long)c->B * c->Hi * c->Wi >= (1ll << 31)) return 0;
  const long long M = (long long)c->B * c->Ho * c->Wo;
  return (M + 31) / 32 * 32 + 64;  // (a workgroup's last K-step may reach past M)
}

int onda_conv2d_wgrad_l2_table(const OndaConv* c, int32_t* table, onda_stream_t s) {
  ONDA_REQUIRE(c && table);
  const long long stride = onda_conv2d_wgrad_l2_table_stride(c);
  ONDA_REQUIRE(stride > 0);
  hipLaunchKernelGGL(wgrad_pixel_table_kernel, dim3(512, c->kh * c->kw), dim3(256), 0, ONDA_STREAM(s), table, stride, c->B, c->Hi, c->Wi,
                     c->Ho, c->Wo, c->kw, c->stride, c->dil, c->pad);
  return ONDA_LAUNCH_RESULT();
}

int onda_conv2d_wgrad_l2(const void* xl, int64_t xplane, const float* xamax, const void* dyl, int64_t dyplane, const float* dyamax,
                         float* slabs, int lddy, int splitk, const OndaConv* c, onda_stream_t s) {
  ONDA_REQUIRE(xl && dyl && xamax && dyamax && slabs && c && splitk >= 1);
  ONDA_REQUIRE(c->Cin % 8 == 0 && c->Cout % 8 == 0 && c->ldx % 32 == 0 && lddy % 32 == 0);
  // a caller's pixel table: at least this batch's rows per tap (a larger batch's table of the same geometry is fine)
  ONDA_REQUIRE(c->pix_table == nullptr || c->pix_stride >= ((long long)c->B * c->Ho * c->Wo + 31) / 32 * 32 + 64);
  (void)xplane;
  (void)dyplane;  // (limb rows: kept in the signature)
  if (!ONDA_ALIGNED16(xl) || !ONDA_ALIGNED16(dyl) || !ONDA_ALIGNED16(slabs)) return ONDA_EALIGN;
  const long long M = (long long)c->B * c->Ho * c->Wo;
  ONDA_REQUIRE(M > 0 && M < (1ll << 31));
  const long long x_total = (long long)c->B * c->Hi * c->Wi * c->ldx * 4;
  const long long dy_total = M * lddy * 4;
  WgradK k;
  k.x = static_cast<const float*>(xl); k.dy = static_cast<const float*>(dyl); k.slabs = slabs; k.c = *c;
  k.M = (int)M;
  k.lddy = lddy;
  k.splitk = splitk;
  k.stamps = g_debug_stamps;
  k.mchunk = (int)(((M + splitk - 1) / splitk + 31) / 32 * 32);
  ONDA_REQUIRE(k.mchunk / 32 <= 2048);  // the kernel lists a workgroup's live K-steps in LDS (MAX_KT); raise splitk beyond that
  {  // 32-bit offsets inside a workgroup's windows (its pixel range of dy; the images of x that range touches)
    const long long img = (long long)c->Hi * c->Wi * c->ldx * 4, per_wg = k.mchunk / ((long long)c->Ho * c->Wo) + 2;
    ONDA_REQUIRE((long long)k.mchunk * lddy * 4 < 0x7FFFF000ll && (per_wg < c->B ? per_wg : c->B) * img < 0x7FFFF000ll);
  }
  k.x_total = x_total;
  k.dy_total = dy_total;
  k.taps = c->kh * c->kw;
  const int variant = onda_conv_wgrad_l2_variant(c->Cout, c->Cin);
  const int TN = variant == 0 ? 256 : 128;
  k.tilesN = (c->Cout + TN - 1) / TN;
  k.tilesC = (c->Cin + 127) / 128;
  const unsigned grid = (unsigned)(k.tilesN * k.tilesC * k.taps * splitk);
  const unsigned xpl = 0, dypl = 0;
  static const int mode = getenv("ONDA_WGRAD_MODE") ? atoi(getenv("ONDA_WGRAD_MODE")) : 1;
  const bool linear = wgrad_l2_linear(c);
  if (variant == 0 && mode == 1 && linear)
    hipLaunchKernelGGL((conv_wgrad_l2_kernel<4, 2, 3, 2, 1>), dim3(grid), dim3(512), 0, ONDA_STREAM(s), k, xpl, dypl, 0u, 0u, xamax, dyamax);
  else if (variant == 0 && mode == 1 && c->pix_table != nullptr && (k.pix = c->pix_table, k.pix_stride = c->pix_stride, true))
    hipLaunchKernelGGL((conv_wgrad_l2_kernel<4, 2, 3, 2, 2>), dim3(grid), dim3(512), 0, ONDA_STREAM(s), k, xpl, dypl, 0u, 0u, xamax, dyamax);
  else if (variant == 0)
    hipLaunchKernelGGL((conv_wgrad_l2_kernel<4, 2, 3, 2>), dim3(grid), dim3(512), 0, ONDA_STREAM(s), k, xpl, dypl, 0u, 0u, xamax, dyamax);
  else
    hipLaunchKernelGGL((conv_wgrad_l2_kernel<2, 2, 2, 2>), dim3(grid), dim3(256), 0, ONDA_STREAM(s), k, xpl, dypl, 0u, 0u, xamax, dyamax);
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"
