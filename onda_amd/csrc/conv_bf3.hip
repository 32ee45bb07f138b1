// fp32-accurate convolution on the bf16 matrix pipe ("split-3"): every fp32 operand is written as
// the exact sum of three bf16 limbs, x = x1 + x2 + x3 (8 + 8 + 8 mantissa bits), and the product
// a*b is evaluated as the six limb products whose weight is >= 2^-16,
//     a1*b1 + (a1*b2 + a2*b1) + (a1*b3 + a2*b2 + a3*b1),
// each exact in fp32 and accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  What is dropped
// (a2*b3 + a3*b2 + a3*b3 and the limb-3 rounding) is <= ~2^-23 |a||b| per product -- the size of
// an fp32 rounding error -- while the bf16 pipe runs 16x the fp32-MFMA rate, i.e. 16/6 = 2.7x per
// fp32-equivalent FLOP.  bf16 keeps fp32's exponent range, so no scaling is involved.
//
// Same implicit-GEMM geometry, hybrid stream-K schedule and epilogue as conv.hip.  Weights arrive
// pre-split ([3][rows][tap*Cin] bf16, onda_pack_weight_*_bf3) and reach LDS by LDS-DMA;
// activations are split by the VALU on their way into LDS.  MFMA shape: 16x16x32 rather than
// 32x32x16 -- same output tile per wave, same LDS bytes and MFMA cycles per K-step, but the chip
// holds a higher clock on it under load (MI355X_MICROARCH.md "DVFS give-back" item 7): +8-15 %.
//
// What was measured on the way (DESIGN.md section 6 has the numbers; tools/stamp_bf3.py,
// tools/micro/valu_vs_mfma.hip and tools/sq_summary.py reproduce them):
//   * SQ_VALU_MFMA_BUSY_CYCLES: the MFMA pipe is busy ~55 % of the forward kernel, ~45 % of the
//     weight-gradient kernel; in-kernel clock 2.15-2.3 GHz, so the rest is idle pipe, not DVFS;
//   * beside a wave that streams MFMAs another wave of the same SIMD gets NO VALU issue slots
//     (80 v_fma: 200 cycles alone, 1750 beside a 1555-cycle MFMA stream), so the second workgroup
//     of a CU hides barriers and memory waits, never the limb split;
//   * a weight stage requested one K-step ahead lands ~3300 cycles later (longer than a K-step);
//   * v_pk_add_f32 (what the SLP vectorizer makes of the split's subtractions) costs ~25 cycles
//     apiece between MFMAs of the same wave.
// A hand-scheduled K-step (split inside the wave's own MFMA stream) and an 8-wave 256 x 128
// ping-pong kernel with three weight stages built on those findings reached +3 % on the large
// shapes and lost 10-20 % on the small ones; they are in the history (commit 918e473), not here.

#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two floats -> two bf16 (round to nearest even) in one dword: a single v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// float4 -> three limbs, each 4 bf16 packed in 8 bytes (4.5 VALU per element)
__device__ __forceinline__ void split3(const f32x4 v, u32x2& l1, u32x2& l2, u32x2& l3) {
  unsigned p[2], q[2], r[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float x0 = v[2 * h], x1 = v[2 * h + 1];
    p[h] = cvt2(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, p[h] << 16);
    const float r1 = x1 - __builtin_bit_cast(float, p[h] & 0xFFFF0000u);
    q[h] = cvt2(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, q[h] << 16);
    const float s1 = r1 - __builtin_bit_cast(float, q[h] & 0xFFFF0000u);
    r[h] = cvt2(s0, s1);
  }
  l1 = u32x2{p[0], p[1]};
  l2 = u32x2{q[0], q[1]};
  l3 = u32x2{r[0], r[1]};
}

// Buffer resources: loads take a 32-bit per-lane byte offset plus a scalar offset, so the K
// loop needs no 64-bit address arithmetic, and an offset >= num_records reads as zero without
// a branch -- which is how padding taps, rows past M and channels past Cout are expressed.
constexpr unsigned OOB = 0x80000000u;     // every operand is < 2 GiB - 4 KiB (checked on the host)
constexpr unsigned CH_OOB = 0x7FFFF000u;  // second addend: row + channel never wraps, stays out of range
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// ---- forward / data gradient ----------------------------------------------------------------------
// 128 x BN tile, 4 waves of 64 x (BN/2), BK = 32: one v_mfma_f32_16x16x32_bf16 spans the whole
// K-step; lane l reads row l & 15, 16-byte chunk l >> 4 of a 16-row fragment block.
//   * weights: the pre-split tile never passes through VGPRs.  Each wave issues
//     `buffer_load_dwordx4 ... lds` (1 KiB per instruction, lane l -> LDS base + 16 l) into a
//     double-buffered, unpadded [3][BN][64 B] image one K-step ahead;
//   * activations: 16-byte buffer loads one K-step ahead into registers, split into limbs by the
//     VALU and stored ([3][128][64 B], single stage) between the two barriers of a K-step -- the
//     VGPR -> LDS store path, the slow side of the LDS, carries only these;
//   * both images are lane-linear 64-byte rows, so the bank-conflict fix is an XOR: LDS slot
//     (row, c') holds data chunk c' ^ swz_row(row) (applied on the SOURCE side of the DMA, on the
//     store address of the activation limbs, and again by the fragment read).  Every
//     ds_read_b128 lane group ({0-3, 12-15} chunk c with {4-11} chunk c + 1, and vice versa)
//     then hits 16 distinct 16-byte slots: SQ_LDS_BANK_CONFLICT = 0.
// 73.7 KB LDS, 2 workgroups / CU.
__device__ __forceinline__ int swz_row(int row) {
  const int q = (row >> 2) & 3;
  return q ^ ((q & 1) << 1) ^ ((row >> 1) & 1);
}

template <int BM, int BN, bool SK>
__global__ __launch_bounds__(256, 2) void conv_fwd_bf3_kernel(const ConvK a, unsigned limb_stride, unsigned x_bytes,
                                                              unsigned w_bytes) {
  constexpr int WAVES_M = 2, WAVES_N = 2;
  constexpr int MF = 16;
  constexpr int TM = BM / (MF * WAVES_M), TN = BN / (MF * WAVES_N);
  constexpr int AL = BM / 32;
  constexpr int PLANE_A = BM * 64;         // activation limb plane, 64-byte rows, swizzled
  constexpr int PLANE_B = BN * 64;         // weight limb plane, 64-byte rows, source-swizzled
  constexpr int A_BYTES = 3 * PLANE_A, B_STAGE = 3 * PLANE_B;
  constexpr int CHUNKS = BN / 16;          // 1-KiB DMA pieces per limb plane
  constexpr int DPW = 3 * CHUNKS / 4;      // DMA instructions per wave per K-step
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
        u32x2 l1, l2, l3;
        split3(ar[i], l1, l2, l3);
        const int row = rbase + 32 * i;
        const int off = row * 64 + ((((t & 7) >> 1) ^ swz_row(row)) << 4) + (t & 1) * 8;
        *reinterpret_cast<u32x2*>(lds + 0 * PLANE_A + off) = l1;
        *reinterpret_cast<u32x2*>(lds + 1 * PLANE_A + off) = l2;
        *reinterpret_cast<u32x2*>(lds + 2 * PLANE_A + off) = l3;
      }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

#if defined(ONDA_BF3_STAMP) || defined(ONDA_BF3_CLOCK)  // diagnostic builds only (tools/stamp_bf3.py)
    long long st[6] = {0, 0, 0, 0, 0, 0};
    long long st0 = __builtin_amdgcn_s_memtime();
    const long long clk0 = st0, rt0 = wall_clock64();
#endif
#ifdef ONDA_BF3_STAMP  // per-phase shader cycles of each wave (perturbs the kernel by ~10 %)
#define STAMP(i) { const long long st1 = __builtin_amdgcn_s_memtime(); st[i] += st1 - st0; st0 = st1; }
#else  // ONDA_BF3_CLOCK: two stamps per tile, for the in-kernel clock of the unperturbed loop
#define STAMP(i)
#endif
    __syncthreads();  // the previous segment's readers are done with every LDS region
    set_tap(tap);
    gload_a();
    dma_b(0);
    int cur = 0;
    STAMP(5)
    for (int kt = k_begin; kt < k_end; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's A rows and weight DMA have landed
      STAMP(0)
      __syncthreads();                                   // ... and everybody else's; A image is free
      STAMP(1)
      sstore_a();
      STAMP(2)
      __syncthreads();
      STAMP(3)
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
      STAMP(4)
      // lane l: row l & 15 of each 16-row block, data chunk l >> 4 (swizzle is the same for every block)
      const int frag = (lane & 15) * 64 + (((lane >> 4) ^ swz_row(lane & 15)) << 4);
      const unsigned char* Ab = lds + wm * TM * MF * 64 + frag;
      const unsigned char* Bb = lds + A_BYTES + cur * B_STAGE + wn * TN * MF * 64 + frag;
      // A limbs stay in registers; B limbs stream 3 -> 2 -> 1 (smallest products first)
      bf16x8 af[TM][3];
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][l] = *reinterpret_cast<const bf16x8*>(Ab + l * PLANE_A + i * MF * 64);
#pragma unroll
      for (int l = 2; l >= 0; --l) {
        bf16x8 bf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bb + l * PLANE_B + j * MF * 64);
#pragma unroll
        for (int la = 2 - l; la >= 0; --la)  // a_{la+1} * b_{l+1} with la + l <= 2
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][la], bf[j], acc[i][j], 0, 0, 0);
      }
      cur ^= 1;
      STAMP(5)
    }
#if defined(ONDA_BF3_STAMP) || defined(ONDA_BF3_CLOCK)
    if (lane == 0 && swz < 256)
      for (int i = 0; i < 6; ++i) a.ws[(swz * 4 + wave) * 8 + i] = (float)st[i];
    if (lane == 0 && swz < 256) {  // shader clock = d(memtime) / d(memrealtime) x 100 MHz
      a.ws[(swz * 4 + wave) * 8 + 6] = (float)(__builtin_amdgcn_s_memtime() - clk0);
      a.ws[(swz * 4 + wave) * 8 + 7] = (float)(wall_clock64() - rt0);
    }
#endif

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

// ---- weight gradient on the bf16 pipe ------------------------------------------------------------
// dW[n][tap][c] = sum_m dY[m][n] * X[pix(m,tap)][c]: both operands have the contraction index
// (the pixel m) as their SLOW axis in memory, while an MFMA fragment wants 8 consecutive k per
// lane.  The transposition happens in registers on the way into LDS (see WIDE below): a thread
// splits 8 consecutive pixels of a channel into limbs and writes each limb's 8 bf16 as one
// ds_write_b128 into the [channel row][k] image.  Threads 0-127 stage dY, 128-255 stage X.
// The consumer side is the forward kernel's: [row][k] limb planes, six limb products.

// LDS image of the 16x16x32 weight-gradient kernel: 64-byte rows; inside each 16-row block the
// row index is transposed as a 4 x 4 matrix and the 16-byte chunk index is XOR-ed with row bits
// {0,1} and {3,4}.  Conflict-free for the ds_read_b128 fragment read (lane l: row l & 15, chunk
// l >> 4) AND for both ds_write_b128 staging patterns (8 lanes on rows 4l + j, or on 8
// consecutive rows): SQ_LDS_BANK_CONFLICT 0.32 -> 0 of the LDS-active cycles.
__device__ __forceinline__ int wg_slot(int row, int chunk) {
  const int phys = (row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3);
  return phys * 64 + ((chunk ^ (row & 3) ^ ((row >> 3) & 3)) << 4);
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf3_kernel(const WgradK a, unsigned x_bytes, unsigned dy_bytes) {
  constexpr int WAVES_N = 2;
  constexpr int MF = 16;
  constexpr int TM = BM / (2 * MF), TN = BN / (2 * MF);
  constexpr int ROWS = BM + BN;
  constexpr int PLANE = ROWS * 64;  // 64-byte rows, placed by wg_slot()
  constexpr int CPT = (BM > BN ? BM : BN) / 32;  // channel columns per staging thread
  static_assert(BM == BN, "one staging half per operand");
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PLANE];
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
      u32x2 a1, a2, a3, b1, b2, b3;
      if constexpr (WIDE) {
        split3(f32x4{v[0][j], v[1][j], v[2][j], v[3][j]}, a1, a2, a3);
        split3(f32x4{v[4][j], v[5][j], v[6][j], v[7][j]}, b1, b2, b3);
      } else {
        split3(v[2 * j], a1, a2, a3);
        split3(v[2 * j + 1], b1, b2, b3);
      }
      const int row = (is_x ? BM : 0) + (WIDE ? 4 * cl + j : cl + 32 * j);
      unsigned char* dst = lds + wg_slot(row, kgroup);
      *reinterpret_cast<u32x4*>(dst + 0 * PLANE) = u32x4{a1[0], a1[1], b1[0], b1[1]};
      *reinterpret_cast<u32x4*>(dst + 1 * PLANE) = u32x4{a2[0], a2[1], b2[0], b2[1]};
      *reinterpret_cast<u32x4*>(dst + 2 * PLANE) = u32x4{a3[0], a3[1], b3[0], b3[1]};
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

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
    bf16x8 af[TM][3];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i][l] = *reinterpret_cast<const bf16x8*>(Ab + l * PLANE + i * MF * 64 + (frag ^ ((i & 1) << 5)));
#pragma unroll
    for (int l = 2; l >= 0; --l) {
      bf16x8 bf[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bb + l * PLANE + j * MF * 64 + (frag ^ ((j & 1) << 5)));
#pragma unroll
      for (int la = 2 - l; la >= 0; --la)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][la], bf[j], acc[i][j], 0, 0, 0);
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
        a.slabs[(((size_t)ks * c.Cout + n) * a.taps + tap) * c.Cin + cc] = acc[i][jn][e];
      }
  }
}

// OIHW fp32 -> limb planes dst[3][rows_pad][Kp] bf16.  dgrad = 0: row n, k = tap*Cin + c.
// dgrad = 1: row c, k = tap'*Cout_pad + n with the taps flipped (data-gradient operand).
__global__ void pack_bf3_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int Cout, int Cin, int taps,
                                int rows_pad, int Kp, int dgrad, int Cout_pad) {
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
  const __bf16 a = (__bf16)v;
  const float r1 = v - (float)a;
  const __bf16 b = (__bf16)r1;
  const float r2 = r1 - (float)b;
  dst[e] = a;
  dst[plane + e] = b;
  dst[2 * plane + e] = (__bf16)r2;
}

}  // namespace

extern "C" {

int onda_pack_weight_bf3(const float* w_oihw, void* dst, int Cout, int Cin, int taps, int rows_pad, int Kp, int dgrad,
                         int Cout_pad, onda_stream_t s) {
  ONDA_REQUIRE(w_oihw && dst && Kp % 8 == 0);
  ONDA_REQUIRE(dgrad ? (rows_pad >= Cin && Kp >= taps * Cout_pad && Cout_pad >= Cout) : (rows_pad >= Cout && Kp >= taps * Cin));
  const size_t plane = (size_t)rows_pad * Kp;
  hipLaunchKernelGGL(pack_bf3_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, ONDA_STREAM(s), w_oihw,
                     static_cast<__bf16*>(dst), Cout, Cin, taps, rows_pad, Kp, dgrad, Cout_pad);
  return ONDA_LAUNCH_RESULT();
}

int onda_conv2d_fwd_bf3(const float* x, const void* w3, float* y, const float* scale, const float* shift,
                        const float* residual, float* stats, float* ws, const OndaConv* c, onda_stream_t s) {
  ONDA_REQUIRE(x && w3 && y && c);
  ONDA_REQUIRE(c->run_if == nullptr);  // device predicates: pre-split kernels only (conv_l2.hip)
  ONDA_REQUIRE(c->Cin > 0 && c->Cin % 32 == 0 && c->Cout > 0 && c->Cout % 4 == 0 && c->ldx % 4 == 0 && c->ldx >= c->Cin);
  ONDA_REQUIRE(c->kh >= 1 && c->kw >= 1 && c->stride >= 1 && c->dil >= 1 && c->out_os >= 1);
  if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(w3)) return ONDA_EALIGN;
  if (ws && (c->ldy % 4 != 0 || !ONDA_ALIGNED16(y) || (residual && (c->ldr % 4 != 0 || !ONDA_ALIGNED16(residual)))))
    ws = nullptr;
  ConvK k;
  k.x = x; k.w = w3; k.y = y; k.scale = scale; k.shift = shift; k.res = residual; k.stats = stats; k.ws = ws;
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
  ONDA_REQUIRE(limb_elems * 6 < (1ull << 31));
  const unsigned limb_stride = (unsigned)limb_elems;
  const unsigned x_bytes = (unsigned)((size_t)c->B * c->Hi * c->Wi * c->ldx * 4), w_bytes = (unsigned)(limb_elems * 6);
  const int tiles = k.tilesM * k.tilesN, KT = k.taps * k.kcper, G = conv_resident_workgroups();
  const int rem = tiles % G;
  k.tiles_dp = tiles - rem;
  const double t_tile_us = 2.0 * 128.0 * (wide ? 128.0 : 64.0) * k.taps * c->Cin / 0.3e6;  // one tile, half a CU, ~150 TF/s chip
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
      hipLaunchKernelGGL((conv_fwd_bf3_kernel<128, 128, true>), dim3(G), dim3(256), 0, st, k, limb_stride, x_bytes, w_bytes);
    else
      hipLaunchKernelGGL((conv_fwd_bf3_kernel<128, 64, true>), dim3(G), dim3(256), 0, st, k, limb_stride, x_bytes, w_bytes);
    return conv_launch_fixup(k, G, wide, st);
  }
  if (wide)
    hipLaunchKernelGGL((conv_fwd_bf3_kernel<128, 128, false>), dim3(tiles), dim3(256), 0, st, k, limb_stride, x_bytes, w_bytes);
  else
    hipLaunchKernelGGL((conv_fwd_bf3_kernel<128, 64, false>), dim3(tiles), dim3(256), 0, st, k, limb_stride, x_bytes, w_bytes);
  return ONDA_LAUNCH_RESULT();
}

int onda_conv2d_wgrad_bf3(const float* x, const float* dy, float* slabs, int lddy, int splitk, const OndaConv* c,
                          onda_stream_t s) {
  ONDA_REQUIRE(x && dy && slabs && c && splitk >= 1);
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
    hipLaunchKernelGGL((conv_wgrad_bf3_kernel<128, 128>), dim3(k.tilesN * k.tilesC * k.taps * splitk), dim3(256), 0,
                       ONDA_STREAM(s), k, x_bytes, dy_bytes);
  } else {
    k.tilesN = (c->Cout + 63) / 64;
    k.tilesC = (c->Cin + 63) / 64;
    hipLaunchKernelGGL((conv_wgrad_bf3_kernel<64, 64>), dim3(k.tilesN * k.tilesC * k.taps * splitk), dim3(256), 0,
                       ONDA_STREAM(s), k, x_bytes, dy_bytes);
  }
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"

