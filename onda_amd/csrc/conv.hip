// Convolutions of the DeepLabV2/ResNet-50 hot path as fp32 MFMA implicit GEMMs for gfx950.
//
// Data layout: activations NHWC (row m = one pixel, channels contiguous), weights packed
// [Cout][tap][Cin], so both GEMM operands are "row-major with K contiguous" and every
// global load is a 16-byte chunk of one pixel's channels / one filter's channels.
//
//   forward / data-gradient  out[m][n] = sum_{tap,c} X[pix(m,tap)][c] * W[n][tap][c]
//   weight-gradient          dW[n][tap][c] = sum_m dY[m][n] * X[pix(m,tap)][c]
//
// The contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per SIMD issue):
// LDS bandwidth is not the limiter at that rate, so tiles are register-staged
// (global_load_dwordx4 -> ds_write_b128) into padded rows and double-buffered with one
// barrier per 32-deep K step.  Workgroup = 256 threads = 4 waves, one wave per SIMD, two
// workgroups per CU.  The blockIdx -> tile map hands each XCD a contiguous band of
// M-tiles so the N-tiles that re-read one activation band share an L2.
#include <cstdlib>

#include "conv_common.h"

namespace {

constexpr int LDS_ROW = 36;  // 32 floats + 4 pad: ds_read_b128 of 16 distinct rows is conflict-free

// SK = false: one workgroup per output tile (grid = tiles).
// SK = true : "stream-K" -- the tiles x K-steps unit space is cut into gridDim.x equal contiguous
//             ranges (grid = resident workgroups), so every CU finishes together whatever the
//             tile count; a range end inside a tile leaves a raw partial tile in `ws`, summed in
//             fixed order by conv_fixup_kernel (deterministic, no atomics, no inter-block waits).
// (256, 2): two workgroups per CU = two waves per SIMD, i.e. at most 256 registers.  Without the second bound the stream-K
// instantiation -- the one most launches run -- grew to 296 registers: ONE wave per SIMD, nothing to cover a barrier, an LDS
// round trip or the next step's global loads with, SQ_VALU_MFMA_BUSY 0.66 of the SIMD cycles on a board at 950 W and
// 2.4 GHz (profiles/r06_d_f32_sq_counters.txt; round 6).
// (The template also instantiates as 256 x 128 tiles on 8 waves -- conv_big_tile below, off: measured slower.)
template <int BM, int BN, int WAVES_M, int WAVES_N, bool SK>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, 2) void conv_fwd_kernel(const ConvK a) {
  constexpr int NT = WAVES_M * WAVES_N * 64;   // threads: 8 per 32-channel row, NT / 8 rows per pass of the workgroup
  constexpr int RP = NT / 8;
  constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
  constexpr int AL = BM / RP, BL = BN / RP;
  constexpr int STAGE = (BM + BN) * LDS_ROW;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const OndaConv& c = a.c;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a
  // contiguous run of tiles (or of the unit space); N-tiles of one M-tile are adjacent in it.
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int KT = a.taps * a.kcper;
  // SK: a.tiles_dp tiles (whole rounds of the grid) are processed one per workgroup, all starting
  // at k = 0 together -- workgroups of an XCD then stream the SAME weight slices at the same time,
  // which is what keeps them in its L2.  Only the remaining (< gridDim.x) tiles are cut into
  // equal unit ranges over all workgroups.
  const int tiles_all = a.tilesM * a.tilesN;
  const int tiles_dp = SK ? a.tiles_dp : tiles_all;
  const long long U = (long long)(tiles_all - tiles_dp) * KT;  // units of the stream-K remainder
  long long u = SK ? swz * U / nblk : 0;
  const long long u_begin = u;
  const long long u_end = SK ? (swz + 1) * U / nblk : 0;
  int dp_tile = swz;  // next data-parallel tile of this workgroup
  const int ccol = (t & 7) * 4, rbase = t >> 3;

  while (dp_tile < tiles_dp || u < u_end) {
  const bool dp = dp_tile < tiles_dp;
  const int tile = dp ? dp_tile : tiles_dp + (int)(u / KT);
  const int k_begin = dp ? 0 : (int)(u - (long long)(tile - tiles_dp) * KT);
  const int k_end = dp ? KT : (int)min((long long)KT, k_begin + (u_end - u));
  const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int hi0[AL], wi0[AL], bH[AL];
#pragma unroll
  for (int u = 0; u < AL; ++u) {
    const int m = m0 + rbase + RP * u;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    const int wo = mm % c.Wo, tq = mm / c.Wo;
    const int ho = tq % c.Ho, b = tq / c.Ho;
    hi0[u] = vm ? ho * c.stride - c.pad : -(1 << 28);
    wi0[u] = wo * c.stride - c.pad;
    bH[u] = b * c.Hi;
  }
  const float* wrow[BL];
  bool vn[BL];
  const int wstride = a.taps * c.Cin;
#pragma unroll
  for (int u = 0; u < BL; ++u) {
    const int n = n0 + rbase + RP * u;
    vn[u] = n < c.Cout;
    wrow[u] = static_cast<const float*>(a.w) + (size_t)(vn[u] ? n : 0) * wstride + ccol;
  }

  // (Per-tile skipping of filter taps that only see padding, as conv_l2_kernel does it, was tried here in round 6: these problems
  //  run ONE round of 128 x 128 tiles, so the launch ends with the tiles in the middle of the image, which skip nothing.)
  long long aofs[AL];
  f32x4 ar[AL], br[BL];
  int tap = k_begin / a.kcper, c0 = (k_begin - tap * a.kcper) * BK;
  auto set_tap = [&](int tp) {
    const int rr = tp / c.kw, ss = tp - rr * c.kw;
#pragma unroll
    for (int u = 0; u < AL; ++u) {
      const int hi = hi0[u] + rr * c.dil, wi = wi0[u] + ss * c.dil;
      const bool ok = (unsigned)hi < (unsigned)c.Hi && (unsigned)wi < (unsigned)c.Wi;
      aofs[u] = ok ? ((long long)(bH[u] + hi) * c.Wi + wi) * c.ldx + ccol : -1;
    }
  };
  auto gload = [&]() {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < AL; ++u) ar[u] = aofs[u] >= 0 ? *reinterpret_cast<const f32x4*>(a.x + aofs[u] + c0) : z;
#pragma unroll
    for (int u = 0; u < BL; ++u)
      br[u] = vn[u] ? *reinterpret_cast<const f32x4*>(wrow[u] + tap * c.Cin + c0) : z;
  };
  auto sstore = [&](int buf) {
    float* Ab = lds + buf * STAGE;
    float* Bb = Ab + BM * LDS_ROW;
#pragma unroll
    for (int u = 0; u < AL; ++u) *reinterpret_cast<f32x4*>(&Ab[(rbase + RP * u) * LDS_ROW + ccol]) = ar[u];
#pragma unroll
    for (int u = 0; u < BL; ++u) *reinterpret_cast<f32x4*>(&Bb[(rbase + RP * u) * LDS_ROW + ccol]) = br[u];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  set_tap(tap);
  gload();
  __syncthreads();  // the previous tile's LDS reads (main loop / statistics) are done
  sstore(0);
  __syncthreads();

  for (int kt = k_begin; kt < k_end; ++kt) {
    const int cur = (kt - k_begin) & 1;
    const bool more = kt + 1 < k_end;
    if (more) {
      c0 += BK;
      if (c0 == c.Cin) {
        c0 = 0;
        ++tap;
        set_tap(tap);
      }
      gload();
    }
    const float* Ab = lds + cur * STAGE + (wm * TM * 32 + li) * LDS_ROW + 4 * lh;
    const float* Bb = lds + cur * STAGE + BM * LDS_ROW + (wn * TN * 32 + li) * LDS_ROW + 4 * lh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_ROW + 8 * j);
#pragma unroll
      for (int i = 0; i < TN; ++i) bf[i] = *reinterpret_cast<const f32x4*>(Bb + i * 32 * LDS_ROW + 8 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[jn][e], acc[i][jn], 0, 0, 0);
    }
    if (more) sstore(cur ^ 1);
    __syncthreads();
  }

  if (dp) dp_tile += nblk; else u += k_end - k_begin;
  if (SK && (k_begin != 0 || k_end != KT)) {
    // partial tile: raw accumulators to this block's slot (0 = its first segment, 1 = its last)
    float* slot = a.ws + ((size_t)swz * 2 + (u - (k_end - k_begin) == u_begin ? 0 : 1)) * (BM * BN);
    conv_store_partial<BN, TM, TN>(slot, acc, wm, wn, lane);
    continue;
  }
  conv_epilogue<BM, BN, TM, TN, WAVES_M>(a, acc, lds, tile_m, m0, n0, wm, wn, lane);
  }  // tile / segment loop
}

// Sums the partial tiles stream-K left in `ws` (fixed order: ascending workgroup) and runs the
// normal epilogue for every tile that was split.  One workgroup per output tile.
// Stage 1 (only when a remainder tile is cut into many pieces): sum the pieces of 8 tile rows per
// workgroup -- wide and shallow, so the many small reads run in parallel -- into sums[tile][BM][BN].
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_piece_sum_kernel(const ConvK a, int G, float* __restrict__ sums) {
  constexpr int ROWS = 1024 / BN;  // rows per workgroup: 256 threads x float4
  __shared__ int piece[1024];
  const int KT = a.taps * a.kcper;
  const long long U = (long long)(a.tilesM * a.tilesN - a.tiles_dp) * KT;
  const int lt = blockIdx.x;  // remainder-local tile
  const long long t0 = (long long)lt * KT, t1 = t0 + KT;
  const int vs = (int)(((t0 + 1) * G + U - 1) / U - 1), ve = (int)((t1 * G + U - 1) / U - 1);
  if (vs == ve) return;
  const int t = threadIdx.x, npieces = ve - vs + 1;
  for (int p = t; p < npieces; p += 256) {
    const int vb = vs + p;
    const long long b0 = (long long)vb * U / G, b1 = (long long)(vb + 1) * U / G;
    const long long g0 = max(b0, t0), g1 = min(b1, t1);
    piece[p] = g1 > g0 ? vb * 2 + (g0 == b0 ? 0 : 1) : -1;
  }
  __syncthreads();
  const int eo = (blockIdx.y * ROWS) * BN + t * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int p = 0; p < npieces; ++p) {
    const int pc = piece[p];
    if (pc >= 0) v += *reinterpret_cast<const f32x4*>(a.ws + (size_t)pc * (BM * BN) + eo);
  }
  *reinterpret_cast<f32x4*>(sums + (size_t)lt * (BM * BN) + eo) = v;
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_fixup_kernel(const ConvK a, int G, const float* __restrict__ sums) {
  constexpr int C4 = BN / 4;        // float4 columns of a tile row
  constexpr int RG = 256 / C4;      // row groups covered by the workgroup at once
  constexpr int RPT = BM / RG;      // rows per thread
  __shared__ f32x4 red[2][256];
  const OndaConv& c = a.c;
  const int KT = a.taps * a.kcper;
  const long long U = (long long)(a.tilesM * a.tilesN - a.tiles_dp) * KT;  // the stream-K remainder
  const int tile = a.tiles_dp + blockIdx.x;
  const long long t0 = (long long)blockIdx.x * KT, t1 = t0 + KT;
  const int vs = (int)(((t0 + 1) * G + U - 1) / U - 1), ve = (int)((t1 * G + U - 1) / U - 1);
  if (vs == ve) return;  // computed whole by one workgroup: its own epilogue already ran
  const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int t = threadIdx.x, col = (t % C4) * 4, rg = t / C4;
  const int n = n0 + col;
  const bool vn = n < c.Cout;  // Cout is a multiple of 4: the 4 columns are valid together
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (vn && a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
  if (vn && a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
  const bool plain = (c.out_os == 1 && c.Hf == c.Ho && c.Wf == c.Wo);

  // contributing workgroups (ascending = the fixed summation order) and the slot each used
  __shared__ int piece[1024];
  const int npieces = ve - vs + 1;  // <= gridDim of the conv kernel <= 1024 (checked on the host)
  for (int p = t; p < npieces; p += 256) {
    const int vb = vs + p;
    const long long b0 = (long long)vb * U / G, b1 = (long long)(vb + 1) * U / G;
    const long long g0 = max(b0, t0), g1 = min(b1, t1);
    piece[p] = g1 > g0 ? vb * 2 + (g0 == b0 ? 0 : 1) : -1;
  }
  __syncthreads();

  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 vmin = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f}, vmax = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
  float mx = 0.f;
#pragma unroll 2
  for (int i = 0; i < RPT; ++i) {
    const int row = rg + RG * i;
    const int eo = row * BN + col;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (sums != nullptr) {
      v = *reinterpret_cast<const f32x4*>(sums + (size_t)blockIdx.x * (BM * BN) + eo);
    } else {
#pragma unroll 4
      for (int p = 0; p < npieces; ++p) {
        const int pc = piece[p];
        if (pc >= 0) v += *reinterpret_cast<const f32x4*>(a.ws + (size_t)pc * (BM * BN) + eo);
      }
    }
    s1 += v;
    s2 += v * v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vmin[j] = fminf(vmin[j], v[j]);
      vmax[j] = fmaxf(vmax[j], v[j]);
    }
    const int m = m0 + row;
    if (m >= a.M || !vn) continue;
    f32x4 o = v * sc + sh;
    if (a.res) o += *reinterpret_cast<const f32x4*>(a.res + (size_t)m * c.ldr + n);
    if (c.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
    }
    size_t orow = m;
    if (!plain) {
      const int wo = m % c.Wo, tq = m / c.Wo;
      const int ho = tq % c.Ho, b = tq / c.Ho;
      orow = ((size_t)b * c.Hf + (size_t)ho * c.out_os) * c.Wf + (size_t)wo * c.out_os;
    }
    *reinterpret_cast<f32x4*>(a.y + orow * c.ldy + n) = o;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  }
  __shared__ float amax_red[4];
  if (a.amax != nullptr) amax_update_block(a.amax, mx, amax_red);
  if (a.stats != nullptr) {
    const int SR = a.stats_rows;
    red[0][t] = s1;
    red[1][t] = s2;
    __syncthreads();
    if (rg == 0 && vn) {
      for (int g = 1; g < RG; ++g) {
        s1 += red[0][g * C4 + t];
        s2 += red[1][g * C4 + t];
      }
      *reinterpret_cast<f32x4*>(a.stats + ((size_t)tile_m * SR + 0) * c.Cout + n) = s1;
      *reinterpret_cast<f32x4*>(a.stats + ((size_t)tile_m * SR + 1) * c.Cout + n) = s2;
    }
    if (SR == 4) {  // the extrema of the raw tile (conv_l2.hip / norm_l2.hip)
      __syncthreads();
      red[0][t] = vmin;
      red[1][t] = vmax;
      __syncthreads();
      if (rg == 0 && vn) {
        for (int g = 1; g < RG; ++g) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vmin[j] = fminf(vmin[j], red[0][g * C4 + t][j]);
            vmax[j] = fmaxf(vmax[j], red[1][g * C4 + t][j]);
          }
        }
        *reinterpret_cast<f32x4*>(a.stats + ((size_t)tile_m * 4 + 2) * c.Cout + n) = vmin;
        *reinterpret_cast<f32x4*>(a.stats + ((size_t)tile_m * 4 + 3) * c.Cout + n) = vmax;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradK a) {
  constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
  constexpr int SA = BM + 4, SB = BN + 4;
  constexpr int CHA = BM / 4, CHB = BN / 4;      // 16-byte chunks per tile row
  constexpr int RPA = 256 / CHA, RPB = 256 / CHB;  // tile rows covered by one pass of the block
  constexpr int AL = BK / RPA, BL = BK / RPB;
  constexpr int STAGE = BK * (SA + SB);
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const OndaConv& c = a.c;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // workgroups are dealt round-robin over the 8 XCDs: hand each XCD a contiguous band of the (pixel range, tap, channel tile)
  // order, so that the workgroups that stream the same dy rows (all input-channel tiles of a tap and pixel range) and the same
  // x rows find them in ONE L2 instead of eight (as conv_wgrad_l2_kernel does; round 6: this kernel read 3.1 TB/s from the
  // Infinity Cache for operands that eight L2s each fetched for themselves)
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
  const int n0 = tile_n * BM, c0 = tile_c * BN;
  const int mbeg = ks * a.mchunk;
  const int mend = min(a.M, mbeg + a.mchunk);
  const int KT = mend > mbeg ? (mend - mbeg + BK - 1) / BK : 0;
  const int rr = tap / c.kw, ss = tap - rr * c.kw;
  const int dh = rr * c.dil - c.pad, dw = ss * c.dil - c.pad;
  const bool direct = (c.kh == 1 && c.kw == 1 && c.stride == 1 && c.pad == 0);

  const int acol = (t % CHA) * 4, arow = t / CHA;
  const int bcol = (t % CHB) * 4, brow = t / CHB;
  const bool va = n0 + acol < c.Cout, vb = c0 + bcol < c.Cin;

  f32x4 ar[AL], br[BL];
  int p_wo[BL], p_ho[BL], p_b[BL];  // this thread's x rows of the K-step gload() fetches next (it is called once per K-step, in order)
#pragma unroll
  for (int u = 0; u < BL; ++u) {
    const int m = mbeg + brow + RPB * u;
    p_wo[u] = m % c.Wo;
    const int tq = m / c.Wo;
    p_ho[u] = tq % c.Ho;
    p_b[u] = tq / c.Ho;
  }
  auto gload = [&](int mb) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < AL; ++u) {
      const int m = mb + arow + RPA * u;
      ar[u] = (va && m < mend) ? *reinterpret_cast<const f32x4*>(a.dy + (size_t)m * a.lddy + n0 + acol) : z;
    }
#pragma unroll
    for (int u = 0; u < BL; ++u) {
      const int m = mb + brow + RPB * u;
      bool ok = vb && m < mend;
      size_t pix = m;
      if (!direct && ok) {
        // (b, ho, wo) of this thread's row, WALKED from K-step to K-step (round 6): four divisions per row and K-step were 600
        // VALU instructions in front of every K-step's MFMAs (3 x 3 filters 101 TFLOP/s, 1 x 1 -- no index arithmetic -- 120)
        const int hi = p_ho[u] * c.stride + dh, wi = p_wo[u] * c.stride + dw;
        ok = (unsigned)hi < (unsigned)c.Hi && (unsigned)wi < (unsigned)c.Wi;
        pix = ((size_t)p_b[u] * c.Hi + hi) * c.Wi + wi;
      }
      br[u] = ok ? *reinterpret_cast<const f32x4*>(a.x + pix * c.ldx + c0 + bcol) : z;
    }
  };
  auto advance = [&](int steps) {  // the walk, `steps` K-steps further
    if (direct || steps <= 0) return;
#pragma unroll
    for (int u = 0; u < BL; ++u) {
      p_wo[u] += steps * BK;
      while (p_wo[u] >= c.Wo) {
        p_wo[u] -= c.Wo;
        if (++p_ho[u] == c.Ho) {
          p_ho[u] = 0;
          ++p_b[u];
        }
      }
    }
  };
  // K-steps (32 pixels) whose output rows all map to input rows outside the image for this tap are exact zeros (whole rows of a
  // dilated tap: up to a quarter of the ASPP weight-gradient work): skipped, as conv_wgrad_l2_kernel does.  Output row ho is live
  // for this tap iff live_lo <= ho < live_hi; a dead step jumps straight to the step of the next live row.
  const int live_lo = dh >= 0 ? 0 : (-dh + c.stride - 1) / c.stride;
  const int live_hi = dh > c.Hi - 1 ? 0 : min(c.Ho, (c.Hi - 1 - dh) / c.stride + 1);
  const bool all_live = direct || (live_lo == 0 && live_hi == c.Ho);  // (the centre row of taps, undilated filters: nothing to test)
  // (output row, column) of the first pixel of the K-step `u_kt`, walked like the threads' own rows: no division per K-step
  int u_kt = 0, u_wo = mbeg % c.Wo, u_ho = (mbeg / c.Wo) % c.Ho;
  auto next_live = [&](int kt) {
    if (all_live) return kt;
    if (live_lo >= live_hi) return KT;
    for (;;) {
      while (u_kt < kt) {  // walk to K-step kt
        u_wo += BK;
        while (u_wo >= c.Wo) {
          u_wo -= c.Wo;
          if (++u_ho == c.Ho) u_ho = 0;
        }
        ++u_kt;
      }
      if (kt >= KT) return KT;
      int left = u_wo + min(BK, mend - (mbeg + kt * BK)), ho = u_ho;  // the rows this step touches
      bool lv = false;
      for (;;) {
        lv |= ho >= live_lo && ho < live_hi;
        left -= c.Wo;
        if (left <= 0) break;
        if (++ho == c.Ho) ho = 0;
      }
      if (lv) return kt;
      ++kt;
    }
  };

  auto sstore = [&](int buf) {
    float* Ab = lds + buf * STAGE;
    float* Bb = Ab + BK * SA;
#pragma unroll
    for (int u = 0; u < AL; ++u) *reinterpret_cast<f32x4*>(&Ab[(arow + RPA * u) * SA + acol]) = ar[u];
#pragma unroll
    for (int u = 0; u < BL; ++u) *reinterpret_cast<f32x4*>(&Bb[(brow + RPB * u) * SB + bcol]) = br[u];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int kt = next_live(0), walk = 0, cur = 0;
  if (kt < KT) {
    advance(kt - walk);
    walk = kt;
    gload(mbeg + kt * BK);
    sstore(0);
  }
  __syncthreads();
  while (kt < KT) {
    const int nx = next_live(kt + 1);
    const bool more = nx < KT;
    if (more) {
      advance(nx - walk);
      walk = nx;
      gload(mbeg + nx * BK);
    }
    const float* Ab = lds + cur * STAGE + lh * SA + wm * TM * 32 + li;
    const float* Bb = lds + cur * STAGE + BK * SA + lh * SB + wn * TN * 32 + li;
    // the operands of step kk + 1 are read while the four MFMAs of step kk run (two register sets): left to the compiler every
    // kk was "read, wait, multiply" -- sixteen exposed LDS round trips per K-step, MFMA busy 0.66 of the SIMD cycles (round 6)
    float af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = Ab[i * 32];
#pragma unroll
    for (int i = 0; i < TN; ++i) bf[0][i] = Bb[i * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int cu = kk & 1, nx = cu ^ 1;
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nx][i] = Ab[2 * (kk + 1) * SA + i * 32];
#pragma unroll
        for (int i = 0; i < TN; ++i) bf[nx][i] = Bb[2 * (kk + 1) * SB + i * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cu][i], bf[cu][jn], acc[i][jn], 0, 0, 0);
      // (the scheduler sinks the reads behind the MFMAs otherwise: reads of step kk + 1 first, then the MFMAs of step kk)
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    }
    if (more) sstore(cur ^ 1);
    __syncthreads();
    kt = nx;
    cur ^= 1;
  }

#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int cc = c0 + wn * TN * 32 + jn * 32 + li;
    if (cc >= c.Cin) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int n = n0 + wm * TM * 32 + i * 32 + row;
        if (n >= c.Cout) continue;
        a.slabs[(((size_t)ks * c.Cout + n) * a.taps + tap) * c.Cin + cc] = acc[i][jn][e];
      }
  }
}

// slabs k0 .. k1-1 of one 16-byte element group, summed in ascending order; eight loads in flight (a tail is padded with
// zeros, which add exactly)
__device__ __forceinline__ f32x4 slab_sum(const float* __restrict__ src, size_t total, int k0, int k1) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 s = zero;
  for (int k = k0; k < k1; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = k + j < k1 ? *reinterpret_cast<const f32x4*>(src + (size_t)(k + j) * total) : zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  return s;
}

// Filters with taps: slabs are [n][tap][cc], the gradient is [n][cc][tap].  A workgroup owns dw[n][cc0 .. cc0+128)[all taps]
// -- 128 * taps contiguous floats -- transposes through LDS and stores them as 16-byte vectors (a thread-per-element store
// leaves 4 of every 36 bytes, and the rest of each line to workgroups on other XCDs: partial-line writes all the way to HBM).
__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int splitk,
                                                                int Cout, int taps, int Cin, int Cout_real, int accumulate) {
  __shared__ __attribute__((aligned(16))) float tile[128 * 9];
  const int cbs = Cin / 128;
  const int n = blockIdx.x / cbs, cc0 = (blockIdx.x % cbs) * 128;
  if (n >= Cout_real) return;
  const size_t total = (size_t)Cout * taps * Cin;
  const int items = 32 * taps, t = threadIdx.x;
  // items t and t + 256 (taps <= 9: at most 288 items); both sums in flight together
  const int i1 = t + 256;
  const float* src0 = slabs + ((size_t)n * taps + (t >> 5)) * Cin + cc0 + 4 * (t & 31);
  const float* src1 = slabs + ((size_t)n * taps + (i1 >> 5)) * Cin + cc0 + 4 * (i1 & 31);
  const bool has0 = t < items, has1 = i1 < items;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 s0 = zero, s1 = zero;
  for (int k = 0; k < splitk; k += 8) {
    f32x4 v[8], u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = has0 && k + j < splitk ? *reinterpret_cast<const f32x4*>(src0 + (size_t)(k + j) * total) : zero;
      u[j] = has1 && k + j < splitk ? *reinterpret_cast<const f32x4*>(src1 + (size_t)(k + j) * total) : zero;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s0 += v[j];
      s1 += u[j];
    }
  }
  if (has0)
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[(4 * (t & 31) + j) * taps + (t >> 5)] = s0[j];
  if (has1)
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[(4 * (i1 & 31) + j) * taps + (i1 >> 5)] = s1[j];
  __syncthreads();
  f32x4* d = reinterpret_cast<f32x4*>(dw + ((size_t)n * Cin + cc0) * taps);
  for (int i = t; i < items; i += 256) {
    const f32x4 val = *reinterpret_cast<const f32x4*>(tile + 4 * i);
    d[i] = accumulate ? d[i] + val : val;
  }
}

// Sum of the split-K slabs [splitk][Cout][taps][Cin] into dw (OIHW).  A thread owns 4 consecutive elements (one 16-byte
// load per slab); P groups of threads share the slabs of those elements -- group p sums the contiguous slab range
// [p*S/P, (p+1)*S/P) in ascending order, the groups are added in ascending order through LDS: deterministic, and a 64 x 64
// filter with 128 slabs is no longer a chain of 128 dependent loads on 16 workgroups (31 us -> the launch floor).
template <int P>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int splitk, int Cout,
                                                           int taps, int Cin, int Cout_real, int Cin_real, int flat_k, int accumulate) {
  constexpr int TX = 256 / P;
  __shared__ f32x4 part[P > 1 ? P : 1][TX];
  const size_t total = (size_t)Cout * taps * Cin;
  const int tx = threadIdx.x % TX, p = threadIdx.x / TX;
  const size_t e = ((size_t)blockIdx.x * TX + tx) * 4;
  const int k0 = (int)((long long)splitk * p / P), k1 = (int)((long long)splitk * (p + 1) / P);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e < total) s = slab_sum(slabs + e, total, k0, k1);
  if (P > 1) {
    part[p][tx] = s;
    __syncthreads();
    if (p != 0) return;
#pragma unroll
    for (int i = 1; i < P; ++i) s += part[i][tx];
  }
  if (e >= total) return;
  const int cc = (int)(e % Cin);  // Cin % 4 == 0: the four elements share (n, tap)
  const size_t q = e / Cin;
  const int tap = (int)(q % taps);
  const int n = (int)(q / taps);
  if (n >= Cout_real) return;
  if (flat_k > 0) {  // stem: packed K index = tap7*Cin_real + c3, OIHW = [n][c3][tap7]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tap7 = (cc + j) / Cin_real, c3 = (cc + j) - tap7 * Cin_real;
      if (tap7 < flat_k) {
        float* d = dw + ((size_t)n * Cin_real + c3) * flat_k + tap7;
        *d = accumulate ? *d + s[j] : s[j];
      }
    }
  } else if (taps == 1 && Cin_real % 4 == 0) {
    if (cc < Cin_real) {
      f32x4* d = reinterpret_cast<f32x4*>(dw + (size_t)n * Cin_real + cc);
      *d = accumulate ? *d + s : s;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (cc + j < Cin_real) {
        float* d = dw + ((size_t)n * Cin_real + cc + j) * taps + tap;
        *d = accumulate ? *d + s[j] : s[j];
      }
  }
}

__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int taps,
                                int Cout_pad, int Kp) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)Cout_pad * Kp) return;
  const int k = (int)(e % Kp), n = (int)(e / Kp);
  float v = 0.f;
  if (n < Cout && k < taps * Cin) {
    const int tap = k / Cin, cc = k - tap * Cin;
    v = w[((size_t)n * Cin + cc) * taps + tap];
  }
  dst[e] = v;
}

__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int taps,
                                  int Cout_pad) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)Cin * taps * Cout_pad) return;
  const int n = (int)(e % Cout_pad);
  const size_t q = e / Cout_pad;
  const int tap = (int)(q % taps), cc = (int)(q / taps);
  dst[e] = n < Cout ? w[((size_t)n * Cin + cc) * taps + (taps - 1 - tap)] : 0.f;
}

__global__ void stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W, int Ho,
                                   int Wo, int Kp) {
  const int k4 = Kp / 4;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t M = (size_t)B * Ho * Wo;
  if (e >= M * k4) return;
  const int kq = (int)(e % k4);
  const size_t m = e / k4;
  const int wo = (int)(m % Wo);
  const size_t tq = m / Wo;
  const int ho = (int)(tq % Ho), b = (int)(tq / Ho);
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = kq * 4 + j;
    float val = 0.f;
    if (k < 147) {
      const int tap = k / 3, cc = k - tap * 3;
      const int r = tap / 7, s = tap - r * 7;
      const int hi = ho * 2 - 3 + r, wi = wo * 2 - 3 + s;
      if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) val = x[(((size_t)b * 3 + cc) * H + hi) * W + wi];
    }
    v[j] = val;
  }
  *reinterpret_cast<f32x4*>(col + m * Kp + kq * 4) = v;
}

}  // namespace

extern "C" {

int onda_conv_tiles_m(int M) { return (M + 127) / 128; }

// the exact-fp32 forward / data-gradient tile of an (M, Cout) problem: 256 x 128 on 8 waves when that fills the chip
static bool conv_big_tile(long long M, int Cout) {
  // OFF: built on the idea that these kernels wait for operand delivery like their f16x2 counterparts; measured, the 256 x 128 tile
  // is 2.4 % SLOWER at the step (292.8 / 293.2 against 286.0 / 286.5 ms, alternating runs) and 3 % per shape.  ONDA_F32_BIG_TILE=1.
  static const int on = getenv("ONDA_F32_BIG_TILE") ? atoi(getenv("ONDA_F32_BIG_TILE")) : 0;
  return on && Cout > 64 && ((M + 255) / 256) * ((Cout + 127) / 128) >= 200;
}
/* rows of the statistic partials onda_conv2d_fwd writes for an (M, Cout) problem (one per tile row of the kernel it runs on) */
int onda_conv_tiles_mc(int M, int Cout) { return conv_big_tile(M, Cout) ? (M + 255) / 256 : (M + 127) / 128; }

}  // extern "C"

int conv_launch_fixup(const ConvK& k, int G, bool wide, hipStream_t st) {
  const int tiles = k.tilesM * k.tilesN - k.tiles_dp;
  if (tiles <= 0) return ONDA_LAUNCH_RESULT();
  // few tiles cut into many pieces: sum the pieces with a wide launch first
  float* sums = (tiles * 4 < G) ? k.ws + (size_t)G * 2 * 128 * 128 : nullptr;
  if (wide) {
    if (sums) hipLaunchKernelGGL((conv_piece_sum_kernel<128, 128>), dim3(tiles, 128 * 128 / 1024), dim3(256), 0, st, k, G, sums);
    hipLaunchKernelGGL((conv_fixup_kernel<128, 128>), dim3(tiles), dim3(256), 0, st, k, G, sums);
  } else {
    if (sums) hipLaunchKernelGGL((conv_piece_sum_kernel<128, 64>), dim3(tiles, 128 * 64 / 1024), dim3(256), 0, st, k, G, sums);
    hipLaunchKernelGGL((conv_fixup_kernel<128, 64>), dim3(tiles), dim3(256), 0, st, k, G, sums);
  }
  return ONDA_LAUNCH_RESULT();
}

// the same for any tile shape (the pre-split kernels of conv_l2.hip: 256 x 128, 128 x 128, 256 x 64)
int conv_launch_fixup_tile(const ConvK& k, int G, int BM, int BN, hipStream_t st) {
  const int tiles = k.tilesM * k.tilesN - k.tiles_dp;
  if (tiles <= 0) return ONDA_LAUNCH_RESULT();
  float* sums = (tiles * 4 < G) ? k.ws + (size_t)G * 2 * BM * BN : nullptr;
#define FIXUP_TILE(BM_, BN_)                                                                                                  \
  do {                                                                                                                        \
    if (sums) hipLaunchKernelGGL((conv_piece_sum_kernel<BM_, BN_>), dim3(tiles, BM_ * BN_ / 1024), dim3(256), 0, st, k, G, sums); \
    hipLaunchKernelGGL((conv_fixup_kernel<BM_, BN_>), dim3(tiles), dim3(256), 0, st, k, G, sums);                            \
  } while (0)
  if (BM == 256 && BN == 128) FIXUP_TILE(256, 128);
  else if (BM == 128 && BN == 128) FIXUP_TILE(128, 128);
  else if (BM == 256 && BN == 64) FIXUP_TILE(256, 64);
  else if (BM == 128 && BN == 64) FIXUP_TILE(128, 64);
  else return ONDA_EINVAL;
#undef FIXUP_TILE
  return ONDA_LAUNCH_RESULT();
}

int conv_sched_override() {
  const char* e = getenv("ONDA_CONV_SCHED");
  return e ? atoi(e) : 0;
}

int conv_resident_workgroups() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    cached = 2 * cus;  // two 256-thread workgroups of the conv kernel fit one CU (LDS 73.7 KB, 176 registers)
  }
  return cached;
}

extern "C" {

int64_t onda_conv_ws_floats(void) { return (int64_t)conv_resident_workgroups() * 3 * 128 * 128; }  // 2 partial slots per workgroup + summed remainder tiles

int onda_conv2d_fwd(const float* x, const float* w, float* y, const float* scale, const float* shift,
                    const float* residual, float* stats, float* ws, const OndaConv* c, onda_stream_t s) {
  ONDA_REQUIRE(x && w && y && c);
  ONDA_REQUIRE(c->run_if == nullptr);  // device predicates: pre-split kernels only (conv_l2.hip)
  ONDA_REQUIRE(c->Cin > 0 && c->Cin % 32 == 0 && c->Cout > 0 && c->Cout % 4 == 0 && c->ldx % 4 == 0 && c->ldx >= c->Cin);
  ONDA_REQUIRE(c->kh >= 1 && c->kw >= 1 && c->stride >= 1 && c->dil >= 1 && c->out_os >= 1);
  if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(w)) return ONDA_EALIGN;
  if (ws && (c->ldy % 4 != 0 || !ONDA_ALIGNED16(y) || (residual && (c->ldr % 4 != 0 || !ONDA_ALIGNED16(residual)))))
    ws = nullptr;  // the fix-up pass stores 16-byte vectors; otherwise stay on the one-tile-per-workgroup path
  ConvK k;
  k.x = x; k.w = w; k.y = y; k.scale = scale; k.shift = shift; k.res = residual; k.stats = stats; k.ws = ws;
  k.c = *c;
  const long long M = (long long)c->B * c->Ho * c->Wo;
  ONDA_REQUIRE(M > 0 && M < (1ll << 31));
  k.M = (int)M;
  k.taps = c->kh * c->kw;
  k.kcper = c->Cin / 32;
  const bool big = conv_big_tile(M, c->Cout) && ws != nullptr;
  ONDA_REQUIRE(big || !conv_big_tile(M, c->Cout) || stats == nullptr);  // (the statistic rows follow onda_conv_tiles_mc)
  k.tilesM = big ? (k.M + 255) / 256 : (k.M + 127) / 128;
  const bool wide = c->Cout > 64;
  k.tilesN = wide ? (c->Cout + 127) / 128 : (c->Cout + 63) / 64;
  const int tiles = k.tilesM * k.tilesN, KT = k.taps * k.kcper, G = big ? conv_resident_workgroups() / 2 : conv_resident_workgroups();
  // One workgroup per tile wastes the last partial round of the G resident workgroups.  The
  // hybrid schedule runs the whole rounds tile-per-workgroup and spreads only the remaining
  // tiles evenly over all workgroups (stream-K); its price is the fix-up pass over those tiles.
  const int rem = tiles % G;
  k.tiles_dp = tiles - rem;
  const double t_tile_us = 2.0 * 128.0 * (wide ? 128.0 : 64.0) * k.taps * c->Cin / 0.2e6;  // one tile, half a CU, ~100 TF/s chip
  const double fix_us = 8.0 + (G + 2.0 * rem) * (big ? 0.06 : (wide ? 0.03 : 0.015));  // partial tiles written + read
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
  if (big) {
    if (balanced) {
      hipLaunchKernelGGL((conv_fwd_kernel<256, 128, 4, 2, true>), dim3(G), dim3(512), 0, st, k);
      return conv_launch_fixup_tile(k, G, 256, 128, st);
    }
    hipLaunchKernelGGL((conv_fwd_kernel<256, 128, 4, 2, false>), dim3(tiles), dim3(512), 0, st, k);
    return ONDA_LAUNCH_RESULT();
  }
  if (balanced) {
    if (wide)
      hipLaunchKernelGGL((conv_fwd_kernel<128, 128, 2, 2, true>), dim3(G), dim3(256), 0, st, k);
    else
      hipLaunchKernelGGL((conv_fwd_kernel<128, 64, 2, 2, true>), dim3(G), dim3(256), 0, st, k);
    return conv_launch_fixup(k, G, wide, st);
  } else if (wide) {
    hipLaunchKernelGGL((conv_fwd_kernel<128, 128, 2, 2, false>), dim3(tiles), dim3(256), 0, st, k);
  } else {
    hipLaunchKernelGGL((conv_fwd_kernel<128, 64, 2, 2, false>), dim3(tiles), dim3(256), 0, st, k);
  }
  return ONDA_LAUNCH_RESULT();
}

int onda_conv2d_wgrad(const float* x, const float* dy, float* slabs, int lddy, int splitk, const OndaConv* c,
                      onda_stream_t s) {
  ONDA_REQUIRE(x && dy && slabs && c && splitk >= 1);
  ONDA_REQUIRE(c->Cin % 4 == 0 && c->Cout % 4 == 0 && c->ldx % 4 == 0 && lddy % 4 == 0);
  if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(dy)) return ONDA_EALIGN;
  WgradK k;
  k.x = x; k.dy = dy; k.slabs = slabs; k.c = *c;
  const long long M = (long long)c->B * c->Ho * c->Wo;
  ONDA_REQUIRE(M > 0 && M < (1ll << 31));
  k.M = (int)M;
  k.lddy = lddy;
  k.splitk = splitk;
  k.mchunk = (int)(((M + splitk - 1) / splitk + 31) / 32 * 32);
  k.taps = c->kh * c->kw;
  if (c->Cout > 64 && c->Cin > 64) {
    k.tilesN = (c->Cout + 127) / 128;
    k.tilesC = (c->Cin + 127) / 128;
    hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2>), dim3(k.tilesN * k.tilesC * k.taps * splitk), dim3(256), 0,
                       ONDA_STREAM(s), k);
  } else {
    k.tilesN = (c->Cout + 63) / 64;
    k.tilesC = (c->Cin + 63) / 64;
    hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2>), dim3(k.tilesN * k.tilesC * k.taps * splitk), dim3(256), 0,
                       ONDA_STREAM(s), k);
  }
  return ONDA_LAUNCH_RESULT();
}

int onda_wgrad_reduce(const float* slabs, float* dw, int splitk, int Cout, int taps, int Cin, int Cout_real,
                      int Cin_real, int flat_k, int accumulate, onda_stream_t s) {
  ONDA_REQUIRE(slabs && dw && splitk >= 1 && Cin % 4 == 0);
  if (!ONDA_ALIGNED16(slabs) || (taps == 1 && flat_k == 0 && Cin_real % 4 == 0 && !ONDA_ALIGNED16(dw))) return ONDA_EALIGN;
  const size_t total = (size_t)Cout * taps * Cin;
  // slab groups per element: enough workgroups to fill the chip, at least two slabs per group
  int P = 1;
  while (P < 16 && total * P / 1024 < 1024 && splitk >= 8 * P) P *= 4;
#define WGRAD_REDUCE(P_)                                                                                                    \
  hipLaunchKernelGGL(wgrad_reduce_kernel<P_>, dim3((unsigned)((total / 4 + 256 / P_ - 1) / (256 / P_))), dim3(256), 0,      \
                     ONDA_STREAM(s), slabs, dw, splitk, Cout, taps, Cin, Cout_real, Cin_real, flat_k, accumulate)
  if (taps > 1 && taps <= 9 && flat_k == 0 && Cin % 128 == 0 && Cin_real == Cin && ONDA_ALIGNED16(dw) &&
      (P == 1 || (long long)Cout_real * (Cin / 128) >= 512)) {
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3((unsigned)(Cout_real * (Cin / 128))), dim3(256), 0, ONDA_STREAM(s), slabs, dw,
                       splitk, Cout, taps, Cin, Cout_real, accumulate);
  } else if (P == 1) WGRAD_REDUCE(1);
  else if (P == 4) WGRAD_REDUCE(4);
  else WGRAD_REDUCE(16);
#undef WGRAD_REDUCE
  return ONDA_LAUNCH_RESULT();
}

int onda_pack_weight_fwd(const float* w, float* dst, int Cout, int Cin, int taps, int Cout_pad, int Kp,
                         onda_stream_t s) {
  ONDA_REQUIRE(w && dst && Kp >= taps * Cin && Cout_pad >= Cout);
  const size_t total = (size_t)Cout_pad * Kp;
  hipLaunchKernelGGL(pack_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ONDA_STREAM(s), w, dst, Cout,
                     Cin, taps, Cout_pad, Kp);
  return ONDA_LAUNCH_RESULT();
}

int onda_pack_weight_dgrad(const float* w, float* dst, int Cout, int Cin, int taps, int Cout_pad, onda_stream_t s) {
  ONDA_REQUIRE(w && dst && Cout_pad >= Cout);
  const size_t total = (size_t)Cin * taps * Cout_pad;
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ONDA_STREAM(s), w, dst,
                     Cout, Cin, taps, Cout_pad);
  return ONDA_LAUNCH_RESULT();
}

int onda_stem_im2col(const float* x, float* col, int B, int H, int W, int Ho, int Wo, int Kp, onda_stream_t s) {
  ONDA_REQUIRE(x && col && Kp % 4 == 0 && Kp >= 147);
  const size_t total = (size_t)B * Ho * Wo * (Kp / 4);
  hipLaunchKernelGGL(stem_im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ONDA_STREAM(s), x, col, B,
                     H, W, Ho, Wo, Kp);
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"
