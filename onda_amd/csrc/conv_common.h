// Shared between the fp32 and the split-precision convolution kernels: argument block, the
// stream-K partial-tile store and the common epilogue (statistics, scale/shift, residual, ReLU,
// slice / scatter store) for a 4-wave workgroup of 32x32 or 16x16 MFMA accumulator tiles.
#pragma once
#include "common.h"

struct ConvK {
  const float* x;
  const void* w;
  float* y;
  const float* scale;
  const float* shift;
  const float* res;
  float* stats;
  OndaConv c;
  int M, tilesM, tilesN, taps, kcper;
  float* ws;     // stream-K partial tiles [grid][2][BM*BN]
  int tiles_dp;  // tiles done one-per-workgroup before the stream-K remainder (multiple of the grid)
  float* amax = nullptr;  // optional: running max|y| of the stored output (amax_update, common.h)
  int skip_dead_taps = 1;  // whole tiles skip filter taps that only see padding (conv_l2.hip)
  int late_issue = 1;      // conv_l2x_kernel: second half of the waves issues its DMAs behind its MFMAs
  // ---- limb-plane OUTPUT (eval-mode conv + folded BatchNorm whose result feeds other convs; conv_l2.hip) ----------------
  _Float16* yl = nullptr;           // out planes [2][M][ldy] f16 (dense rows); when set, `y` is not written
  long long yplane = 0;             // f16 elements between the two planes
  float* ybound = nullptr;          // amax buffer that DEFINES the planes' scale: receives the a-priori bound
  const float* kb = nullptr;        // {max_c |scale_c| * sum_k |w_ck|, max_c |shift_c|}
  const float* xtrue = nullptr;     // amax buffer with the TRUE max|x| of the input (its planes may be scaled by a bound)
  const _Float16* resl = nullptr;   // residual as limb planes [2][M][ldr]
  long long resplane = 0;
  const float* res_amax = nullptr;  // scale-defining amax of the residual planes
  const float* res_true = nullptr;  // true max|residual|
  unsigned long long* stamps = nullptr;  // diagnostics (ONDA_L2X_STAMP=1, tools/l2x_stamps.py): s_memtime per workgroup,
                                         // [32] each: start, then (end of K loop, end of epilogue) per work item
  int stats_rows = 2;     // 2: stats[tile][sum, sumsq][Cout]; 4: also the per-channel min and max of the raw tile (conv_l2.hip)
  long long x_total = 0;  // conv_l2.hip: bytes of the whole input operand.  A tile addresses it with 32-bit offsets RELATIVE to the
                          // first image its rows touch (a window of < 2 GiB), so the operand itself may be larger
};

struct WgradK {
  const float* x;
  const float* dy;
  float* slabs;
  OndaConv c;
  int M, lddy, splitk, mchunk, tilesN, tilesC, taps;
  long long x_total = 0, dy_total = 0;  // bytes of the operands: a workgroup addresses them relative to its pixel range's start
  unsigned long long* stamps = nullptr;  // diagnostics (onda_debug_stamps, tools/wgrad_stamps.py): [8] s_memtime per workgroup
  // [taps][pix_stride] input pixel ((b*Hi + hi)*Wi + wi, or -1 in the padding) of output pixel m under each tap: a table per
  // convolution geometry, built once by the library (csrc/conv_l2.hip, wgrad_pixel_table)
  const int* pix = nullptr;
  long long pix_stride = 0;
};

constexpr int BK = 32;

// Accumulator fragment of one MF x MF MFMA tile: which (row, column) of the tile register e of a
// lane holds (32x32x* and 16x16x* MFMA result layouts).
template <int MF>
struct AccTile;
template <>
struct AccTile<32> {
  using T = f32x16;
  static constexpr int E = 16;
  static __device__ __forceinline__ int row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }
  static __device__ __forceinline__ int col(int lane) { return lane & 31; }
};
template <>
struct AccTile<16> {
  using T = f32x4;
  static constexpr int E = 4;
  static __device__ __forceinline__ int row(int e, int lane) { return 4 * (lane >> 4) + e; }
  static __device__ __forceinline__ int col(int lane) { return lane & 15; }
};

// raw accumulators of a partial (stream-K) tile -> slot[BM][BN]
template <int BN, int TM, int TN, int MF = 32>
__device__ __forceinline__ void conv_store_partial(float* slot, const typename AccTile<MF>::T (&acc)[TM][TN], int wm,
                                                   int wn, int lane) {
  using L = AccTile<MF>;
#pragma unroll
  for (int jn = 0; jn < TN; ++jn)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < L::E; ++e) {
        const int row = (wm * TM + i) * MF + L::row(e, lane);
        slot[row * BN + (wn * TN + jn) * MF + L::col(lane)] = acc[i][jn][e];
      }
}

// `red`: >= WAVES_M*BN*2 + 2*WAVES_M floats of LDS that no wave is still reading.
template <int BM, int BN, int TM, int TN, int WAVES_M, int MF = 32>
__device__ __forceinline__ void conv_epilogue(const ConvK& a, const typename AccTile<MF>::T (&acc)[TM][TN],
                                              float* red, int tile_m, int m0, int n0, int wm, int wn, int lane) {
  using L = AccTile<MF>;
  const OndaConv& c = a.c;
  const int t = threadIdx.x;
  if (a.stats != nullptr) {
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < L::E; ++e) {
          const float v = acc[i][jn][e];
          s1 += v;
          s2 += v * v;
        }
#pragma unroll
      for (int sh = MF; sh < 64; sh <<= 1) {  // lanes holding other rows of the same column
        s1 += __shfl_xor(s1, sh, 64);
        s2 += __shfl_xor(s2, sh, 64);
      }
      if (lane < MF) {
        const int col = (wn * TN + jn) * MF + lane;
        red[(wm * BN + col) * 2 + 0] = s1;
        red[(wm * BN + col) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (t < BN && n0 + t < c.Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WAVES_M; ++w_) {
        s1 += red[(w_ * BN + t) * 2 + 0];
        s2 += red[(w_ * BN + t) * 2 + 1];
      }
      a.stats[((size_t)tile_m * 2 + 0) * c.Cout + n0 + t] = s1;
      a.stats[((size_t)tile_m * 2 + 1) * c.Cout + n0 + t] = s2;
    }
  }

  const bool plain = (c.out_os == 1 && c.Hf == c.Ho && c.Wf == c.Wo);
  float mx = 0.f;
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int n = n0 + (wn * TN + jn) * MF + L::col(lane);
    if (n >= c.Cout) continue;
    const float sc = a.scale ? a.scale[n] : 1.f;
    const float sh = a.shift ? a.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < L::E; ++e) {
        const int m = m0 + (wm * TM + i) * MF + L::row(e, lane);
        if (m >= a.M) continue;
        float v = acc[i][jn][e] * sc + sh;
        if (a.res) v += a.res[(size_t)m * c.ldr + n];
        if (c.relu) v = fmaxf(v, 0.f);
        size_t orow = m;
        if (!plain) {
          const int wo = m % c.Wo, tq = m / c.Wo;
          const int ho = tq % c.Ho, b = tq / c.Ho;
          orow = ((size_t)b * c.Hf + (size_t)ho * c.out_os) * c.Wf + (size_t)wo * c.out_os;
        }
        a.y[orow * c.ldy + n] = v;
        mx = fmaxf(mx, fabsf(v));
      }
    }
  }
  if (a.amax != nullptr) {  // one atomic per tile: max over the workgroup's waves through LDS (past the statistics' scratch)
    float* ar = red + WAVES_M * BN * 2;
    mx = wave_max(mx);
    if ((t & 63) == 0) ar[t >> 6] = mx;
    __syncthreads();
    if (t == 0) {
      float m = ar[0];
#pragma unroll
      for (int w_ = 1; w_ < 2 * WAVES_M; ++w_) m = fmaxf(m, ar[w_]);
      if (m > 0.f)
        atomicMax(reinterpret_cast<unsigned*>(a.amax) + (blockIdx.x & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE, __float_as_uint(m));
    }
  }
}

// conv.hip: sums stream-K partial tiles and runs the epilogue for split tiles
int conv_launch_fixup(const ConvK& k, int G, bool wide, hipStream_t st);
int conv_launch_fixup_tile(const ConvK& k, int G, int BM, int BN, hipStream_t st);  // any of the tile shapes of conv_l2.hip
int conv_resident_workgroups();
int conv_sched_override();  // debugging aid: environment variable ONDA_CONV_SCHED (0 / unset = automatic)
