// BatchNorm around the pre-split convolutions (conv_l2.hip): the kernels that PRODUCE a conv operand write it
// as limb rows (out[M][C / 32][2][32] f16: common.h limb_at; 4 bytes per element like fp32) instead of fp32 + a later
// split pass.  ("limb planes" below: the same two limbs, which lay in two planes until round 4.)
//
// A limb plane needs its tensor's power-of-two scale BEFORE the first element is written, i.e. an upper bound
// of max|out| from quantities that exist before the apply pass:
//   forward : the conv epilogue leaves per-channel min / max of the raw conv output next to sum / sum of squares
//             (stats[tile][4][C]); out = relu(gamma*(x-mean)*invstd + beta (+ res)) is monotone in x per channel,
//             so max|out| <= max_c max(|f_c(min_c)|, |f_c(max_c)|) (+ max|res|): exact without a residual;
//   backward: dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) with g = dout*[out > 0]:
//             |dx| <= |gamma*invstd| * (max|g| + |mean g| + max|xhat| * |mean g*xhat|) per channel; max|g| comes
//             out of the reduction pass that already reads every element.
// A bound instead of the maximum costs precision only at the far small end of a tensor (a bound 2^k too large:
// full two-limb accuracy down to 2^-(28-k) of the maximum instead of 2^-28; conv_h2.hip).
#include "common.h"

namespace {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr float LIMB2_SCALE = ONDA_LIMB2_SCALE, LIMB2_UNSCALE = 1.f / ONDA_LIMB2_SCALE;  // common.h

struct Scale2 {
  float s, inv;
};
__device__ __forceinline__ Scale2 scale_of(const float* __restrict__ amax) {
  const float m = amax_read(amax);
  int e = 0;
  if (m > 0.f && m < 3.0e38f) {
    int ex;
    frexpf(m, &ex);
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
// 8 scaled floats -> 8 + 8 f16 (16 bytes per limb)
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, u32x4& l1, u32x4& l2) {
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const float a = h < 2 ? v0[2 * h] : v1[2 * h - 4], b = h < 2 ? v0[2 * h + 1] : v1[2 * h - 3];
    const unsigned p = cvt2h(a, b);
    const f32x2 f = unpack2h(p);
    l1[h] = p;
    l2[h] = cvt2h((a - f[0]) * LIMB2_SCALE, (b - f[1]) * LIMB2_SCALE);
  }
}
// 8 + 8 f16 -> 8 floats (still scaled)
__device__ __forceinline__ void join8(const u32x4 l1, const u32x4 l2, f32x4& v0, f32x4& v1) {
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const f32x2 a = unpack2h(l1[h]), b = unpack2h(l2[h]);
    const float x0 = a[0] + b[0] * LIMB2_UNSCALE, x1 = a[1] + b[1] * LIMB2_UNSCALE;
    if (h < 2) {
      v0[2 * h] = x0;
      v0[2 * h + 1] = x1;
    } else {
      v1[2 * h - 4] = x0;
      v1[2 * h - 3] = x1;
    }
  }
}

#define LD4(p) (*reinterpret_cast<const f32x4*>(p))

// Store the two limbs of this thread's 8 channels (l1 at `o`, l2 32 f16 further) as WHOLE cache lines.  Four consecutive
// lanes own one 32-channel block = one 128-byte line [4 x l1 | 4 x l2]; stored as they stand, each of a lane's two stores
// covers half of its line (the limb planes of rounds 2-4 wrote whole lines; the limb rows cost bn_bwd_apply 7 % this way).
// So two neighbouring quads trade: the low quad's lanes store their own l1 and the HIGH quad's l1, the high quad's lanes the
// LOW quad's l2 and their own -- store 1 = the low block's whole line, store 2 = the high block's.  `whole`: every lane of
// the wave is in here with a valid item (wave-uniform); otherwise the plain two stores.
__device__ __forceinline__ void store_limb_lines(_Float16* o, u32x4 l1, u32x4 l2, bool whole) {
  if (!whole) {
    *reinterpret_cast<u32x4*>(o) = l1;
    *reinterpret_cast<u32x4*>(o + LIMB2_OFS) = l2;
    return;
  }
  const bool hi = (threadIdx.x & 4) != 0;
  const u32x4 send = hi ? l1 : l2;
  u32x4 recv;
#pragma unroll
  for (int j = 0; j < 4; ++j) recv[j] = (unsigned)__shfl_xor((int)send[j], 4, 64);
  const unsigned long long mine = reinterpret_cast<unsigned long long>(o);
  const unsigned long long partner = ((unsigned long long)(unsigned)__shfl_xor((int)(mine >> 32), 4, 64) << 32) |
                                     (unsigned)__shfl_xor((int)(unsigned)mine, 4, 64);
  _Float16* po = reinterpret_cast<_Float16*>(partner);
  _Float16* a1 = hi ? po + LIMB2_OFS : o;   // store 1: the low quad's line
  _Float16* a2 = hi ? o + LIMB2_OFS : po;   // store 2: the high quad's line
  *reinterpret_cast<u32x4*>(a1) = hi ? recv : l1;
  *reinterpret_cast<u32x4*>(a2) = hi ? l2 : recv;
}

// The big tensors of these passes are touched once per launch (8-20 bytes per element streamed through): with
// ONDA_NT_BN their loads and stores carry the non-temporal hint, so that they do not cycle through the L2.
// (A measurement switch: see DESIGN.md for what it measured.)
#ifndef ONDA_NT_BN
#define ONDA_NT_BN 0
#endif
template <class T>
__device__ __forceinline__ T ld_stream(const T* p) {  // (bit 0: loads)
  if constexpr (ONDA_NT_BN & 1) return __builtin_nontemporal_load(p); else return *p;
}
template <class T>
__device__ __forceinline__ void st_stream(T* p, T v) {  // (bit 1: stores)
  if constexpr (ONDA_NT_BN & 2) __builtin_nontemporal_store(v, p); else *p = v;
}
#define LD4S(p) ld_stream(reinterpret_cast<const f32x4*>(p))

// ---- a dependent pass without a launch boundary ---------------------------------------------------------------------------
// BatchNorm's statistics are a grid-wide dependency between two passes over the tensor: finalize (a few workgroups, a few
// microseconds of latency-bound table reduction) -> apply (thousands of workgroups).  As two launches the small one costs
// more in launch latency than in work (9 us per BatchNorm, 156 of them per adaptation step: round-3..5 reviews).  Fused: the
// FIRST `nfin` workgroups of the big launch do the small pass, publish it (release fence + one atomic increment each) and
// every workgroup waits for the count before it touches the results.  Workgroups are dispatched in ascending order (per XCD:
// ids k, k + 8, ...), so the ones everybody waits for are resident before any waiter and depend on nobody: no deadlock,
// whatever else shares the GPU.  The wait is bounded all the same (a GPU that hangs is worse than a wrong number the
// parity tests catch): after ~1 s it gives up and raises word 2 of the sync line.
// The counter needs a zeroed word that nobody else uses: floats 1..3 of the OUTPUT's amax buffer (ONDA_AMAX_FLOATS zeroed
// floats, written by exactly one producer, never reused; only every AMAX_STRIDE-th float of it carries a maximum).
__device__ __forceinline__ void grid_publish(float* amax_buf) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(reinterpret_cast<int*>(amax_buf) + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_wait(float* amax_buf, int target) {
  if (threadIdx.x == 0) {
    int* flag = reinterpret_cast<int*>(amax_buf) + 1;
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 21)) {  // ~1 s of s_sleep: never seen; do not hang the box
        reinterpret_cast<int*>(amax_buf)[2] = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("" ::: "memory");
}

// Column reduction of small partial tables [rows][NV][C] (conv tile statistics, the backward reduction's chunks), 16
// channels per 256-thread workgroup: thread t reads the 16-byte channel quad (t & 3) of the rows t >> 2, t >> 2 + 64, ...
// (64-byte segments instead of the 4-byte, row-strided reads of a wave per channel: 16 times fewer cache-line requests
// on tables of 0.5-3 MB), sums in double, folds the 16 row groups of a wave by butterfly and the four waves through
// LDS.  Vector v < NS is summed, the others are folded by min (v == NS, only when MINMAX) or max.  The result of channel
// ch0 + c lands in thread c < 16 (`sum[v]`, `ext[v - NS]`).  Every thread must call; deterministic (fixed order).
template <int NV, int NS, bool MINMAX>
__device__ __forceinline__ void reduce_rows16(const float* __restrict__ partials, int rows, int C, int ch0, double (&sum)[NS > 0 ? NS : 1],
                                              float (&ext)[NV - NS > 0 ? NV - NS : 1]) {
  constexpr int NE = NV - NS;
  __shared__ double lsum[4][NS > 0 ? NS : 1][16];
  __shared__ float lext[4][NE > 0 ? NE : 1][16];
  const int t = threadIdx.x, q = t & 3, rg = t >> 2, wave = t >> 6;
  const int col = ch0 + 4 * q;
  double s[NS > 0 ? NS : 1][4];
  float x[NE > 0 ? NE : 1][4];
#pragma unroll
  for (int v = 0; v < NS; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) s[v][j] = 0.0;
#pragma unroll
  for (int v = 0; v < NE; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) x[v][j] = MINMAX && v == 0 ? 3.0e38f : -3.0e38f;
  if (col < C) {
#pragma unroll 2
    for (int r = rg; r < rows; r += 64) {
      const float* p = partials + (size_t)r * NV * C + col;
      f32x4 val[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) val[v] = LD4(p + (size_t)v * C);
#pragma unroll
      for (int v = 0; v < NS; ++v)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[v][j] += (double)val[v][j];
#pragma unroll
      for (int v = 0; v < NE; ++v)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          x[v][j] = MINMAX && v == 0 ? fminf(x[v][j], val[NS + v][j]) : fmaxf(x[v][j], val[NS + v][j]);
    }
  }
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
    for (int v = 0; v < NS; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[v][j] += __shfl_xor(s[v][j], o, 64);
#pragma unroll
    for (int v = 0; v < NE; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float other = __shfl_xor(x[v][j], o, 64);
        x[v][j] = MINMAX && v == 0 ? fminf(x[v][j], other) : fmaxf(x[v][j], other);
      }
  }
  if ((t & 63) < 4) {
#pragma unroll
    for (int v = 0; v < NS; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) lsum[wave][v][4 * q + j] = s[v][j];
#pragma unroll
    for (int v = 0; v < NE; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) lext[wave][v][4 * q + j] = x[v][j];
  }
  __syncthreads();
  if (t < 16) {
#pragma unroll
    for (int v = 0; v < NS; ++v) sum[v] = ((lsum[0][v][t] + lsum[1][v][t]) + lsum[2][v][t]) + lsum[3][v][t];
#pragma unroll
    for (int v = 0; v < NE; ++v) {
      const float a = lext[0][v][t], b = lext[1][v][t], c = lext[2][v][t], d = lext[3][v][t];
      ext[v] = MINMAX && v == 0 ? fminf(fminf(a, b), fminf(c, d)) : fmaxf(fmaxf(a, b), fmaxf(c, d));
    }
  }
}

// Tile partials [tiles][4][C] = (sum, sum of squares, min, max) -> mean, invstd, running statistics, xhat_amax[C] =
// max|(x - mean) * invstd| (for the backward bound) and the bound of max|out| folded into out_amax.  16 channels per workgroup.
//
// ROW GROUPS (the student's source-replay and target micro-batches normalised in ONE pass, each with its own batch
// statistics; blockIdx.y = group): the GEMM rows [0, split) are group 0, [split, M) group 1; mean / invstd / xhat_amax are
// [2][C].  Partial row i covers the GEMM rows [i*bm, (i+1)*bm); the ONE row that straddles `split` (ts = split / bm, when
// split is not a multiple of bm -- every feature grid of this network leaves 4 rows over) is counted whole with group 1
// and corrected by the sums over its group-0 rows, read straight from the conv output y (a few rows x 16 channels per
// workgroup): group 0 adds them, group 1 subtracts them.  Extrema only have to BOUND a group's: group 1 keeps the whole
// row's.  (The host guarantees that the straddling tile is not a stream-K remainder tile: csrc/conv_l2.hip, l2_schedule.)
// run_group: the group whose statistics move the running buffers (-1: none).
struct BnFinalizeArgs {
  const float* partials;
  int tiles, C;
  double count;
  float eps;
  float *mean, *invstd, *rmean, *rvar;
  int64_t* nbt;
  float momentum;
  const float *gamma, *beta, *res_amax;
  int relu;
  float *xhat_amax, *out_amax;
  long long split;
  int bm, run_group;
  const float* y;
  int ldy;
};
// one workgroup (256 threads) = 16 channels `bx` of row group `g`
__device__ __forceinline__ void bn_finalize_block(const BnFinalizeArgs& a, const int bx, const int g) {
  const float* __restrict__ partials = a.partials;
  const int tiles = a.tiles, C = a.C, relu = a.relu, bm = a.bm, run_group = a.run_group, ldy = a.ldy;
  double count = a.count;
  const float eps = a.eps, momentum = a.momentum;
  float *mean = a.mean, *invstd = a.invstd, *rmean = a.rmean, *rvar = a.rvar, *xhat_amax = a.xhat_amax, *out_amax = a.out_amax;
  int64_t* nbt = a.nbt;
  const float* __restrict__ gamma = a.gamma;
  const float* __restrict__ beta = a.beta;
  const float* __restrict__ res_amax = a.res_amax;
  const float* __restrict__ y = a.y;
  const long long split = a.split;
  if (bx == 0 && g == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  int row0 = 0, nrows = tiles;
  if (split > 0) {
    const int ts = (int)(split / bm);
    if (g == 0) nrows = ts; else row0 = ts, nrows = tiles - ts;
    count = g == 0 ? (double)split : count - (double)split;
  }
  double sum[2];
  float ext[2];
  reduce_rows16<4, 2, true>(partials + (size_t)row0 * 4 * C, nrows, C, bx * 16, sum, ext);
  const int ch = bx * 16 + threadIdx.x;
  float bound = 0.f;
  if (threadIdx.x < 16 && ch < C) {
    if (split > 0 && split % bm != 0) {  // the group-0 rows of the straddling tile
      double c1 = 0.0, c2 = 0.0;
      float mn = 3.0e38f, mx = -3.0e38f;
      for (long long r = split - split % bm; r < split; ++r) {
        const float v = y[(size_t)r * ldy + ch];
        c1 += (double)v;
        c2 += (double)v * (double)v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
      }
      if (g == 0) {
        sum[0] += c1;
        sum[1] += c2;
        ext[0] = fminf(ext[0], mn);
        ext[1] = fmaxf(ext[1], mx);
      } else {
        sum[0] -= c1;
        sum[1] -= c2;
      }
    }
    const double mu = sum[0] / count;
    double var = sum[1] / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float muf = (float)mu, is = (float)(1.0 / sqrt(var + (double)eps));
    const float lo = (ext[0] - muf) * is, hi = (ext[1] - muf) * is;
    const float gm = gamma[ch], b = beta[ch];
    const float f_lo = lo * gm + b, f_hi = hi * gm + b;
    bound = relu ? fmaxf(0.f, fmaxf(f_lo, f_hi)) : fmaxf(fabsf(f_lo), fabsf(f_hi));  // behind a ReLU only the positive side counts
    mean[g * C + ch] = muf;
    invstd[g * C + ch] = is;
    xhat_amax[g * C + ch] = fmaxf(fabsf(lo), fabsf(hi));
    if (rmean && (split <= 0 || g == run_group)) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * muf;
      rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unbiased;
    }
  }
  if (res_amax != nullptr) bound += amax_read(res_amax);
  bound *= 1.0000005f;  // the apply pass rounds differently; the scale leaves a factor 2 of headroom anyway
  __shared__ float red[4];
  // one atomic per workgroup, slot by (bx, g) (amax_update_block takes blockIdx.x, which is something else in the fused launch)
  float m = wave_max(bound);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f)
      atomicMax(reinterpret_cast<unsigned*>(out_amax) + ((bx + 7 * g) & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE, __float_as_uint(m));
  }
}
__global__ __launch_bounds__(256) void bn_finalize_l2_kernel(const BnFinalizeArgs a) { bn_finalize_block(a, blockIdx.x, blockIdx.y); }

// out limbs = [relu]( (x - mean)*invstd*gamma + beta [+ residual limbs] ), 8 channels per thread.
// group1_at > 0: two row groups -- 8-channel items [0, group1_at) normalise with the statistics at mean / invstd, the others with
// those at mean + C / invstd + C (gamma / beta are shared).
// FUSED: the first `nfin` workgroups run bn_finalize_block first and everybody waits for them (grid_publish / grid_wait above):
// finalize + apply in ONE launch; mean / invstd / out_amax are then written by this very launch (no __restrict__ on them).
template <bool FUSED>
__global__ __launch_bounds__(256) void bn_apply_l2_kernel(const float* __restrict__ x, const float* mean, const float* invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const _Float16* __restrict__ res, size_t res_plane,
                                                          const float* __restrict__ res_amax, _Float16* __restrict__ out,
                                                          size_t out_plane, const float* out_amax, size_t total8, int C, int relu,
                                                          unsigned char* __restrict__ mask, size_t group1_at, const BnFinalizeArgs fin,
                                                          int nfin) {
  if constexpr (FUSED) {
    const int nb = (C + 15) / 16;
    if ((int)blockIdx.x < nfin) {
      bn_finalize_block(fin, blockIdx.x % nb, blockIdx.x / nb);
      grid_publish(fin.out_amax);
    }
    grid_wait(fin.out_amax, nfin);
  }
  const int c8 = C / 8;
  const float so = scale_of(out_amax).s;
  const float ri = res ? scale_of(res_amax).inv : 0.f;
  // the grid stride (a multiple of 256) is a multiple of c8 (a power of two <= 256 in this network; checked on the host):
  // a thread stays on its 8 channels, so their parameters are loaded once
  const size_t e0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int col = (int)(e0 % c8) * 8;
  const f32x4 ga0 = LD4(gamma + col), ga1 = LD4(gamma + col + 4);
  const f32x4 sc0 = LD4(invstd + col) * ga0, sc1 = LD4(invstd + col + 4) * ga1;
  const f32x4 mu0 = LD4(mean + col), mu1 = LD4(mean + col + 4), be0 = LD4(beta + col), be1 = LD4(beta + col + 4);
  const int g1 = group1_at > 0 ? C : 0;  // (one group: the second set is the first)
  const f32x4 tc0 = LD4(invstd + g1 + col) * ga0, tc1 = LD4(invstd + g1 + col + 4) * ga1;
  const f32x4 nu0 = LD4(mean + g1 + col), nu1 = LD4(mean + g1 + col + 4);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // item e = 8 channels `col` of row e / c8; in limb rows: first limbs at limb_at(row, col, C), second limbs 32 f16 further
  const size_t lcol = limb_at(0, col, C);
  auto lofs = [&](size_t e) { return (size_t)((unsigned)e / (unsigned)c8) * 2 * (size_t)C + lcol; };  // (items < 2^32: 32-bit division)
  auto one = [&](size_t e, f32x4 x0, f32x4 x1, u32x4 r1, u32x4 r2, bool whole) {
    const bool second = group1_at > 0 && e >= group1_at;
    f32x4 v0 = (x0 - (second ? nu0 : mu0)) * (second ? tc0 : sc0) + be0, v1 = (x1 - (second ? nu1 : mu1)) * (second ? tc1 : sc1) + be1;
    if (res) {
      f32x4 q0, q1;
      join8(r1, r2, q0, q1);
      v0 += q0 * ri;
      v1 += q1 * ri;
    }
    if (mask) {  // [out > 0], one bit per element (byte e = the 8 channels of this item): what the backward passes read
      unsigned bits = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) bits |= (v0[j] > 0.f ? 1u << j : 0u) | (v1[j] > 0.f ? 16u << j : 0u);
      mask[e] = (unsigned char)bits;
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v0[j] = fmaxf(v0[j], 0.f);
        v1[j] = fmaxf(v1[j], 0.f);
      }
    }
    u32x4 l1, l2;
    split8(v0 * so, v1 * so, l1, l2);
    store_limb_lines(out + lofs(e), l1, l2, whole);
  };
  size_t e = e0;
  // two independent items per iteration: all loads of both are issued before the first is used
  for (; e + stride < total8; e += 2 * stride) {
    const size_t f = e + stride;
    const f32x4 a0 = LD4S(x + e * 8), a1 = LD4S(x + e * 8 + 4), b0 = LD4S(x + f * 8), b1 = LD4S(x + f * 8 + 4);
    u32x4 ra1 = {}, ra2 = {}, rb1 = {}, rb2 = {};
    if (res) {
      ra1 = ld_stream(reinterpret_cast<const u32x4*>(res + lofs(e)));
      ra2 = ld_stream(reinterpret_cast<const u32x4*>(res + lofs(e) + LIMB2_OFS));
      rb1 = ld_stream(reinterpret_cast<const u32x4*>(res + lofs(f)));
      rb2 = ld_stream(reinterpret_cast<const u32x4*>(res + lofs(f) + LIMB2_OFS));
    }
    const bool whole = __ballot(1) == ~0ull;  // (the loop condition holds for every lane of this wave)
    one(e, a0, a1, ra1, ra2, whole);
    one(f, b0, b1, rb1, rb2, whole);
  }
  if (e < total8) {
    u32x4 r1 = {}, r2 = {};
    if (res) {
      r1 = *reinterpret_cast<const u32x4*>(res + lofs(e));
      r2 = *reinterpret_cast<const u32x4*>(res + lofs(e) + LIMB2_OFS);
    }
    one(e, LD4(x + e * 8), LD4(x + e * 8 + 4), r1, r2, __ballot(1) == ~0ull);
  }
}

// ---- backward ------------------------------------------------------------------------------------------------------------
// pass 1: per-channel partial sums of g and g*xhat and the partial max|g|; g = dout * [out > 0] (sign of the first limb:
// an element whose first limb rounds to zero is below 2^-24 of the tensor maximum... its second limb decides), dres = g
struct ColPlan3 {
  int cx, ry, gridx, chunks;
  int64_t rows_per_chunk;
};
static inline ColPlan3 col_plan3(int64_t M, int C) {
  ColPlan3 p;
  const int c4 = C / 4;
  p.cx = 1;
  while (p.cx < c4 && p.cx < 64) p.cx <<= 1;
  p.ry = 256 / p.cx;
  p.gridx = (c4 + p.cx - 1) / p.cx;
  int64_t target = 2048 / p.gridx;
  if (target < 1) target = 1;
  int64_t rpc = (M + target - 1) / target;
  const int64_t min_rows = (int64_t)p.ry * 8;
  if (rpc < min_rows) rpc = min_rows;
  p.rows_per_chunk = rpc;
  p.chunks = (int)((M + rpc - 1) / rpc);
  return p;
}

// g * [out > 0] from the FIRST limb of out (2 bytes per element).  The first limb of a positive element is zero only below
// 2^-40 of the tensor maximum (f16 subnormals reach 2^-24, the scale puts the maximum at 2^15): there the gradient is
// dropped, at the kink of the ReLU.
__device__ __forceinline__ f32x4 relu_mask4(const _Float16* __restrict__ out, const unsigned char* __restrict__ mask, size_t o, f32x4 g,
                                            int C) {
  if (mask != nullptr) {  // the bit mask of the apply pass: 1/8 byte per element instead of the 2 bytes of the first limb
    const unsigned bits = (unsigned)mask[o >> 3] >> ((o & 4) ? 4 : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = (bits >> j) & 1u ? g[j] : 0.f;
    return g;
  }
  const unsigned orow = (unsigned)(o / (unsigned)C);  // (o = row * C + channel)
  const u32x2 h1 = *reinterpret_cast<const u32x2*>(out + limb_at(orow, (int)(o - (size_t)orow * C), C));
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x2 a = unpack2h(h1[j]);
    g[2 * j] = a[0] > 0.f ? g[2 * j] : 0.f;
    g[2 * j + 1] = a[1] > 0.f ? g[2 * j + 1] : 0.f;
  }
  return g;
}

// Row groups (split > 0): chunks [0, chunks0) walk the rows [0, split) with the statistics of group 0, the others the rows
// [split, M) with those at mean + C / invstd + C -- a chunk never crosses the boundary.
__global__ __launch_bounds__(256) void bn_bwd_reduce_l2_kernel(const float* __restrict__ dout, const _Float16* __restrict__ out,
                                                               size_t out_plane, const float* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               float* __restrict__ dres, int64_t M, int C, int relu, int cx,
                                                               int64_t rows_per_chunk, float* __restrict__ partials,
                                                               const unsigned char* __restrict__ mask, int64_t split, int chunks0) {
  __shared__ f32x4 red[3][256];
  const int t = threadIdx.x;
  const int ry_n = 256 / cx;
  const int tx = t % cx, ty = t / cx;
  const int col = (blockIdx.x * cx + tx) * 4;
  const int chunk = blockIdx.y;
  const bool second = split > 0 && chunk >= chunks0;
  const int64_t r0 = second ? split + (int64_t)(chunk - chunks0) * rows_per_chunk : (int64_t)chunk * rows_per_chunk;
  const int64_t r1 = min(split > 0 && !second ? split : M, r0 + rows_per_chunk);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f};
  if (col < C) {
    const int gofs = second ? C : 0;
    const f32x4 mu = LD4(mean + gofs + col), is = LD4(invstd + gofs + col);
    for (int64_t r = r0 + ty; r < r1; r += ry_n) {
      const size_t o = (size_t)r * C + col;
      f32x4 g = LD4(dout + o);
      if (relu) g = relu_mask4(out, mask, o, g, C);
      if (dres) *reinterpret_cast<f32x4*>(dres + o) = g;
      const f32x4 xh = (LD4(x + o) - mu) * is;
      s1 += g;
      s2 += g * xh;
#pragma unroll
      for (int j = 0; j < 4; ++j) s3[j] = fmaxf(s3[j], fabsf(g[j]));
    }
  }
  red[0][t] = s1;
  red[1][t] = s2;
  red[2][t] = s3;
  __syncthreads();
  if (ty == 0 && col < C) {
    for (int y = 1; y < ry_n; ++y) {
      s1 += red[0][y * cx + tx];
      s2 += red[1][y * cx + tx];
      const f32x4 o3 = red[2][y * cx + tx];
#pragma unroll
      for (int j = 0; j < 4; ++j) s3[j] = fmaxf(s3[j], o3[j]);
    }
    float* dst = partials + ((size_t)chunk * 3) * C + col;
    *reinterpret_cast<f32x4*>(dst) = s1;
    *reinterpret_cast<f32x4*>(dst + C) = s2;
    *reinterpret_cast<f32x4*>(dst + 2 * C) = s3;
  }
}

// pass 2: chunk partials [chunks][3][C] -> sums[2][C] and the bound of max|dx| into dx_amax; 16 channels per workgroup.
// Row groups: blockIdx.y = group; its chunk rows are [0, chunks0) / [chunks0, chunks), its results go to sums + g*2*C, its
// statistics sit at invstd + g*C / xhat_amax + g*C.
struct BnBwdSumsArgs {
  const float* partials;
  int chunks, C;
  double inv_m0, inv_m1;
  const float *gamma, *invstd, *xhat_amax;
  float *sums, *dx_amax;
  int chunks0, groups;
};
__device__ __forceinline__ void bn_bwd_sums_block(const BnBwdSumsArgs& a, const int bx, const int g) {
  const int C = a.C;
  const int row0 = g == 0 ? 0 : a.chunks0, nrows = a.groups == 1 ? a.chunks : (g == 0 ? a.chunks0 : a.chunks - a.chunks0);
  const double inv_m = g == 0 ? a.inv_m0 : a.inv_m1;
  double sum[2];
  float ext[1];
  reduce_rows16<3, 2, false>(a.partials + (size_t)row0 * 3 * C, nrows, C, bx * 16, sum, ext);
  const int ch = bx * 16 + threadIdx.x;
  float bound = 0.f;
  if (threadIdx.x < 16 && ch < C) {
    a.sums[g * 2 * C + ch] = (float)sum[0];
    a.sums[g * 2 * C + C + ch] = (float)sum[1];
    bound = fabsf(a.gamma[ch] * a.invstd[g * C + ch]) *
            (ext[0] + (float)(fabs(sum[0]) * inv_m) + a.xhat_amax[g * C + ch] * (float)(fabs(sum[1]) * inv_m));
  }
  bound *= 1.000001f;
  __shared__ float red[4];
  float m = wave_max(bound);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f)
      atomicMax(reinterpret_cast<unsigned*>(a.dx_amax) + ((bx + 7 * g) & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE, __float_as_uint(m));
  }
}
__global__ __launch_bounds__(256) void bn_bwd_sums_l2_kernel(const BnBwdSumsArgs a) { bn_bwd_sums_block(a, blockIdx.x, blockIdx.y); }

// pass 3: dx limbs.  group1_at > 0: the 8-channel items from there on belong to row group 1 (statistics at + C, sums at + 2C,
// 1 / M of that group)
// FUSED: the first `nfin` workgroups reduce the chunk partials first (bn_bwd_sums_block) and everybody waits for them
template <bool FUSED>
__global__ __launch_bounds__(256) void bn_bwd_apply_l2_kernel(const float* __restrict__ dout, const _Float16* __restrict__ out,
                                                              size_t out_plane, const float* __restrict__ x,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* sums,
                                                              _Float16* __restrict__ dx, size_t dx_plane, const float* dx_amax,
                                                              size_t total8, int C, float inv_m, int relu,
                                                              const unsigned char* __restrict__ mask, size_t group1_at, float inv_m1,
                                                              const BnBwdSumsArgs fin, int nfin) {
  if constexpr (FUSED) {
    const int nb = (C + 15) / 16;
    if ((int)blockIdx.x < nfin) {
      bn_bwd_sums_block(fin, blockIdx.x % nb, blockIdx.x / nb);
      grid_publish(fin.dx_amax);
    }
    grid_wait(fin.dx_amax, nfin);
  }
  const int c8 = C / 8;
  const float sd = scale_of(dx_amax).s;
  const size_t e0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int col = (int)(e0 % c8) * 8;  // fixed per thread (see bn_apply_l2_kernel)
  f32x4 is[2][2], mu[2][2], gi[2][2], m1[2][2], m2[2][2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int go = g && group1_at > 0 ? C : 0;
    const float im = g && group1_at > 0 ? inv_m1 : inv_m;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cc = col + 4 * h;
      is[g][h] = LD4(invstd + go + cc);
      mu[g][h] = LD4(mean + go + cc);
      gi[g][h] = LD4(gamma + cc) * is[g][h];
      m1[g][h] = LD4(sums + 2 * go + cc) * im;
      m2[g][h] = LD4(sums + 2 * go + C + cc) * im;
    }
  }
  for (size_t e = e0; e < total8; e += (size_t)gridDim.x * blockDim.x) {
    const bool second = group1_at > 0 && e >= group1_at;
    f32x4 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const size_t o = e * 8 + 4 * h;
      f32x4 g = LD4S(dout + o);
      if (relu) g = relu_mask4(out, mask, o, g, C);
      const f32x4 xh = (LD4S(x + o) - (second ? mu[1][h] : mu[0][h])) * (second ? is[1][h] : is[0][h]);
      v[h] = (second ? gi[1][h] : gi[0][h]) * (g - (second ? m1[1][h] : m1[0][h]) - xh * (second ? m2[1][h] : m2[0][h]));
    }
    u32x4 l1, l2;
    split8(v[0] * sd, v[1] * sd, l1, l2);
    store_limb_lines(dx + (size_t)((unsigned)e / (unsigned)c8) * 2 * (size_t)C + limb_at(0, col, C), l1, l2, __ballot(1) == ~0ull);
  }
}

// SE gate applied to the ASPP concat, written as the bottleneck conv's OPERAND: out limbs = x * gate[b][c] scaled by the
// scale of max|x| (the gate is a sigmoid: |out| <= |x|, so the input's maximum bounds the output's).  No fp32 copy, no
// max pass, no split pass (framework/model/deeplabv2.py:99-114 -> :244).
__global__ __launch_bounds__(256) void chan_scale_limbs_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                               _Float16* __restrict__ out, size_t plane,
                                                               const float* __restrict__ amax, int64_t HW, int C, size_t total8) {
  const int c8 = C / 8;
  const float so = scale_of(amax).s;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total8; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % c8) * 8;
    const size_t b = (e / c8) / HW;
    const f32x4 v0 = LD4S(x + e * 8) * LD4(gate + b * C + col), v1 = LD4S(x + e * 8 + 4) * LD4(gate + b * C + col + 4);
    u32x4 l1, l2;
    split8(v0 * so, v1 * so, l1, l2);
    _Float16* o = out + (size_t)((unsigned)e / (unsigned)c8) * 2 * (size_t)C + limb_at(0, col, C);
    st_stream(reinterpret_cast<u32x4*>(o), l1);
    st_stream(reinterpret_cast<u32x4*>(o + LIMB2_OFS), l2);
  }
}

static inline unsigned ew_grid(size_t total) {
  size_t g = (total + 255) / 256;
#ifndef ONDA_EW_GRID_CAP
#define ONDA_EW_GRID_CAP 4096  // (8192: +0.2..0.7 ms per step on two boxes, 16384: +0.6; 1024..3072: within the noise of 4096)
#endif
  if (g > ONDA_EW_GRID_CAP) g = ONDA_EW_GRID_CAP;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" {

static BnFinalizeArgs finalize_args(const float* partials, int tiles, int C, int64_t count, float eps, float* mean, float* invstd,
                                    float* running_mean, float* running_var, int64_t* nbt, float momentum, const float* gamma,
                                    const float* beta, const float* res_amax, int relu, float* xhat_amax, float* out_amax,
                                    int64_t split, int tile_rows, int run_group, const float* y, int ldy) {
  return BnFinalizeArgs{partials, tiles, C, (double)count, eps, mean, invstd, running_mean, running_var, nbt, momentum, gamma, beta,
                        res_amax, relu, xhat_amax, out_amax, (long long)split, tile_rows > 0 ? tile_rows : 1, run_group, y, ldy};
}

int onda_bn_finalize_l2(const float* partials, int tiles, int C, int64_t count, float eps, float* mean, float* invstd,
                        float* running_mean, float* running_var, int64_t* nbt, float momentum, const float* gamma,
                        const float* beta, const float* res_amax, int relu, float* xhat_amax, float* out_amax, int64_t split,
                        int tile_rows, int run_group, const float* y, int ldy, onda_stream_t s) {
  ONDA_REQUIRE(partials && mean && invstd && gamma && beta && xhat_amax && out_amax && tiles >= 1 && C >= 4 && C % 4 == 0 && count >= 1);
  ONDA_REQUIRE(split >= 0 && split < count && (split == 0 || (tile_rows > 0 && y && ldy >= C && split / tile_rows < tiles)));
  if (!ONDA_ALIGNED16(partials)) return ONDA_EALIGN;
  hipLaunchKernelGGL(bn_finalize_l2_kernel, dim3((C + 15) / 16, split > 0 ? 2 : 1), dim3(256), 0, ONDA_STREAM(s),
                     finalize_args(partials, tiles, C, count, eps, mean, invstd, running_mean, running_var, nbt, momentum, gamma, beta,
                                   res_amax, relu, xhat_amax, out_amax, split, tile_rows, run_group, y, ldy));
  return ONDA_LAUNCH_RESULT();
}

int onda_chan_scale_limbs(const float* x, const float* gate, void* out, int64_t out_plane, const float* x_amax, int B, int64_t HW,
                          int C, onda_stream_t s) {
  ONDA_REQUIRE(x && gate && out && x_amax && C % 32 == 0 && B > 0 && HW > 0);
  (void)out_plane;
  if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(out) || !ONDA_ALIGNED16(gate)) return ONDA_EALIGN;
  const size_t total8 = (size_t)B * HW * C / 8;
  hipLaunchKernelGGL(chan_scale_limbs_kernel, dim3(ew_grid(total8)), dim3(256), 0, ONDA_STREAM(s), x, gate,
                     static_cast<_Float16*>(out), (size_t)out_plane, x_amax, HW, C, total8);
  return ONDA_LAUNCH_RESULT();
}

int onda_bn_apply_l2(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const void* res, int64_t res_plane, const float* res_amax, void* out, int64_t out_plane,
                     const float* out_amax, int64_t M, int C, int relu, uint8_t* relu_mask, int64_t split, onda_stream_t s) {
  ONDA_REQUIRE(x && mean && invstd && gamma && beta && out && out_amax && C % 32 == 0 && (!res || res_amax));
  ONDA_REQUIRE(256 % (C / 8) == 0 || (C / 8) % 256 == 0);  // a thread keeps its channel group across the grid stride
  ONDA_REQUIRE(split >= 0 && split < M);
  if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(out) || (res && !ONDA_ALIGNED16(res))) return ONDA_EALIGN;
  const size_t total8 = (size_t)M * C / 8;
  hipLaunchKernelGGL(bn_apply_l2_kernel<false>, dim3(ew_grid(total8)), dim3(256), 0, ONDA_STREAM(s), x, mean, invstd, gamma, beta,
                     static_cast<const _Float16*>(res), (size_t)res_plane, res_amax, static_cast<_Float16*>(out), (size_t)out_plane,
                     out_amax, total8, C, relu, relu_mask, (size_t)split * C / 8, BnFinalizeArgs{}, 0);
  return ONDA_LAUNCH_RESULT();
}

/* onda_bn_finalize_l2 + onda_bn_apply_l2 in ONE launch (grid_publish / grid_wait above): the first (C / 16) * groups
 * workgroups finalize, every workgroup waits for them.  out_amax: zeroed ONDA_AMAX_FLOATS floats as for the two calls; its
 * floats 1..3 serve as the hand-over's counter (zero before the call, not reusable after it). */
int onda_bn_train_l2(const float* x, const float* partials, int tiles, float eps, float* mean, float* invstd, float* running_mean,
                     float* running_var, int64_t* nbt, float momentum, const float* gamma, const float* beta, const void* res,
                     const float* res_amax, int relu, float* xhat_amax, void* out, float* out_amax, int64_t M, int C,
                     uint8_t* relu_mask, int64_t split, int tile_rows, int run_group, onda_stream_t s) {
  ONDA_REQUIRE(x && partials && mean && invstd && gamma && beta && xhat_amax && out && out_amax && tiles >= 1 && M >= 1);
  ONDA_REQUIRE(C % 32 == 0 && (!res || res_amax) && (256 % (C / 8) == 0 || (C / 8) % 256 == 0));
  ONDA_REQUIRE(split >= 0 && split < M && (split == 0 || (tile_rows > 0 && split / tile_rows < tiles)));
  if (!ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(out) || (res && !ONDA_ALIGNED16(res)) || !ONDA_ALIGNED16(partials)) return ONDA_EALIGN;
  const size_t total8 = (size_t)M * C / 8;
  const int nfin = (C + 15) / 16 * (split > 0 ? 2 : 1);
  const unsigned grid = ew_grid(total8) > (unsigned)nfin ? ew_grid(total8) : (unsigned)nfin;
  hipLaunchKernelGGL(bn_apply_l2_kernel<true>, dim3(grid), dim3(256), 0, ONDA_STREAM(s), x, mean, invstd, gamma, beta,
                     static_cast<const _Float16*>(res), (size_t)0, res_amax, static_cast<_Float16*>(out), (size_t)0, out_amax, total8, C,
                     relu, relu_mask, (size_t)split * C / 8,
                     finalize_args(partials, tiles, C, M, eps, mean, invstd, running_mean, running_var, nbt, momentum, gamma, beta, res_amax,
                                   relu, xhat_amax, out_amax, split, tile_rows, run_group, x, C),
                     nfin);
  return ONDA_LAUNCH_RESULT();
}

// chunk rows of the backward reduction with row groups: group 0 first, then group 1 (a chunk never crosses `split`)
static inline void bwd_chunks(const ColPlan3& p, int64_t M, int64_t split, int& chunks0, int& chunks) {
  if (split <= 0) {
    chunks0 = chunks = p.chunks;
    return;
  }
  chunks0 = (int)((split + p.rows_per_chunk - 1) / p.rows_per_chunk);
  chunks = chunks0 + (int)((M - split + p.rows_per_chunk - 1) / p.rows_per_chunk);
}

int64_t onda_bn_bwd_l2_ws(int64_t M, int C) {
  const ColPlan3 p = col_plan3(M, C);
  return (int64_t)(p.chunks + 1) * 3 * C + 4 * C;  // (+1 chunk row: two row groups round up separately; sums[2][2][C])
}

/* fuse_sums != 0: the chunk partials are reduced by the first workgroups of the apply launch (two launches instead of three);
 * dx_amax's floats 1..3 then serve as the hand-over's counter (see onda_bn_train_l2) */
int onda_bn_bwd_l2(const float* dout, const void* out, int64_t out_plane, const float* x, const float* mean, const float* invstd,
                   const float* gamma, const float* xhat_amax, void* dx, int64_t dx_plane, float* dx_amax, float* dres, float* ws,
                   int64_t M, int C, int relu, const uint8_t* relu_mask, int64_t split, int fuse_sums, onda_stream_t s) {
  ONDA_REQUIRE(dout && x && mean && invstd && gamma && xhat_amax && dx && dx_amax && ws && C % 32 == 0 && (!relu || out || relu_mask));
  ONDA_REQUIRE(256 % (C / 8) == 0 || (C / 8) % 256 == 0);
  ONDA_REQUIRE(split >= 0 && split < M);
  if (!ONDA_ALIGNED16(dout) || !ONDA_ALIGNED16(x) || !ONDA_ALIGNED16(dx) || (out && !ONDA_ALIGNED16(out))) return ONDA_EALIGN;
  const ColPlan3 p = col_plan3(M, C);
  int chunks0, chunks;
  bwd_chunks(p, M, split, chunks0, chunks);
  float* sums = ws + (size_t)(p.chunks + 1) * 3 * C;
  hipStream_t st = ONDA_STREAM(s);
  hipLaunchKernelGGL(bn_bwd_reduce_l2_kernel, dim3(p.gridx, chunks), dim3(256), 0, st, dout, static_cast<const _Float16*>(out),
                     (size_t)out_plane, x, mean, invstd, dres, M, C, relu, p.cx, p.rows_per_chunk, ws, relu_mask, split, chunks0);
  const double inv_m0 = 1.0 / (double)(split > 0 ? split : M), inv_m1 = 1.0 / (double)(M - split);
  const int groups = split > 0 ? 2 : 1;
  const BnBwdSumsArgs fin{ws, chunks, C, inv_m0, inv_m1, gamma, invstd, xhat_amax, sums, dx_amax, chunks0, groups};
  const size_t total8 = (size_t)M * C / 8;
  if (fuse_sums) {
    const int nfin = (C + 15) / 16 * groups;
    const unsigned grid = ew_grid(total8) > (unsigned)nfin ? ew_grid(total8) : (unsigned)nfin;
    hipLaunchKernelGGL(bn_bwd_apply_l2_kernel<true>, dim3(grid), dim3(256), 0, st, dout, static_cast<const _Float16*>(out),
                       (size_t)out_plane, x, mean, invstd, gamma, sums, static_cast<_Float16*>(dx), (size_t)dx_plane, dx_amax, total8, C,
                       (float)inv_m0, relu, relu_mask, (size_t)split * C / 8, (float)inv_m1, fin, nfin);
    return ONDA_LAUNCH_RESULT();
  }
  hipLaunchKernelGGL(bn_bwd_sums_l2_kernel, dim3((C + 15) / 16, groups), dim3(256), 0, st, fin);
  hipLaunchKernelGGL(bn_bwd_apply_l2_kernel<false>, dim3(ew_grid(total8)), dim3(256), 0, st, dout, static_cast<const _Float16*>(out),
                     (size_t)out_plane, x, mean, invstd, gamma, sums, static_cast<_Float16*>(dx), (size_t)dx_plane, dx_amax, total8,
                     C, (float)inv_m0, relu, relu_mask, (size_t)split * C / 8, (float)inv_m1, BnBwdSumsArgs{}, 0);
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"
