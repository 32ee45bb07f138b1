// HBM-bound pointwise / stencil kernels of the hot path: ceil-mode max-pool, the SE gate,
// channel scaling, align_corners bilinear upsampling (plain, gradient, fused argmax).
#include "common.h"

namespace {

#define LD4(p) (*reinterpret_cast<const f32x4*>(p))

// ---- MaxPool 3x3 / s2 / p1 / ceil_mode, NHWC -------------------------------------------
// yl != nullptr: the result goes out as LIMB ROWS (the operand format of the convolutions that read it, conv_l2.hip; scale from
// `amax` = max|x|: a maximum over windows of x cannot pass it) instead of fp32 -- no fp32 copy, no split pass behind the pool
typedef _Float16 mp_f16x2 __attribute__((ext_vector_type(2)));
typedef float mp_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned mp_u32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int B, int Hi, int Wi, int C,
                                                          int Ho, int Wo, _Float16* __restrict__ yl,
                                                          const float* __restrict__ amax) {
  const int c4 = C / 4;
  float so = 1.f;
  if (yl != nullptr) {  // 2^e with max|x| * 2^e in [2^14, 2^15) (conv_h2.hip)
    const float m = amax_read(amax);
    int ex = 0;
    if (m > 0.f && m < 3.0e38f) {
      frexpf(m, &ex);
      ex = 15 - ex;
      ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
    }
    so = ldexpf(1.f, ex);
  }
  const size_t total = (size_t)B * Ho * Wo * c4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % c4) * 4;
    size_t q = e / c4;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho), b = (int)(q / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int slot[4] = {0, 0, 0, 0};
    bool first = true;
    for (int r = 0; r < 3; ++r) {
      const int hi = ho * 2 - 1 + r;
      if ((unsigned)hi >= (unsigned)Hi) continue;
      for (int s = 0; s < 3; ++s) {
        const int wi = wo * 2 - 1 + s;
        if ((unsigned)wi >= (unsigned)Wi) continue;
        const f32x4 v = LD4(x + (((size_t)b * Hi + hi) * Wi + wi) * C + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // ATen's scan: strict '>' keeps the first maximum, NaN always wins
          if (first || v[j] > best[j] || v[j] != v[j]) {
            best[j] = v[j];
            slot[j] = r * 3 + s;
          }
        }
        first = false;
      }
    }
    if (yl != nullptr) {
      const f32x4 w = best * so;
      mp_u32x2 l1, l2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const mp_f16x2 p = __builtin_convertvector(mp_f32x2{w[2 * h], w[2 * h + 1]}, mp_f16x2);
        const mp_f32x2 f = __builtin_convertvector(p, mp_f32x2);
        l1[h] = __builtin_bit_cast(unsigned, p);
        l2[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(mp_f32x2{(w[2 * h] - f[0]) * ONDA_LIMB2_SCALE, (w[2 * h + 1] - f[1]) * ONDA_LIMB2_SCALE}, mp_f16x2));
      }
      _Float16* o = yl + limb_at(e / c4, col, C);
      *reinterpret_cast<mp_u32x2*>(o) = l1;
      *reinterpret_cast<mp_u32x2*>(o + LIMB2_OFS) = l2;
    } else {
      *reinterpret_cast<f32x4*>(y + e * 4) = best;
    }
    *reinterpret_cast<uint32_t*>(idx + e * 4) =
        (uint32_t)slot[0] | ((uint32_t)slot[1] << 8) | ((uint32_t)slot[2] << 16) | ((uint32_t)slot[3] << 24);
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int B, int Hi, int Wi, int C, int Ho,
                                                          int Wo) {
  const int c4 = C / 4;
  const size_t total = (size_t)B * Hi * Wi * c4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % c4) * 4;
    size_t q = e / c4;
    const int wi = (int)(q % Wi);
    q /= Wi;
    const int hi = (int)(q % Hi), b = (int)(q / Hi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int ho_lo = hi / 2, ho_hi = (hi + 1) / 2;  // windows with 2*ho-1 <= hi <= 2*ho+1
    const int wo_lo = wi / 2, wo_hi = (wi + 1) / 2;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      if (ho >= Ho) continue;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        if (wo >= Wo) continue;
        const int slot = (hi - (ho * 2 - 1)) * 3 + (wi - (wo * 2 - 1));
        const size_t o = (((size_t)b * Ho + ho) * Wo + wo) * C + col;
        const uint32_t pk = *reinterpret_cast<const uint32_t*>(idx + o);
        const f32x4 g = LD4(dy + o);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((int)((pk >> (8 * j)) & 0xff) == slot) acc[j] += g[j];
      }
    }
    *reinterpret_cast<f32x4*>(dx + e * 4) = acc;
  }
}

// ---- SE gate -------------------------------------------------------------------------------
// hidden = relu(W1 pooled + b1) (R rows), gate = sigmoid(W2 hidden + b2): two small launches that fill the chip (one wave
// per hidden unit and image; one thread per gate) instead of one workgroup per image walking both layers serially
// (193 us for a 0.8 MB problem).
__global__ __launch_bounds__(256) void se_fc1_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, float* __restrict__ hidden, int C, int R) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* p = pooled + (size_t)b * C;
  float s = 0.f;
  for (int ch = lane; ch < C; ch += 64) s += w1[(size_t)r * C + ch] * p[ch];  // same order of partial sums as before
  s = wave_sum(s);
  if (lane == 0) hidden[(size_t)b * R + r] = fmaxf(s + b1[r], 0.f);
}

__global__ __launch_bounds__(256) void se_fc2_kernel(const float* __restrict__ hidden, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, float* __restrict__ gate, int C, int R) {
  extern __shared__ float hid[];
  const int b = blockIdx.y, t = threadIdx.x;
  for (int r = t; r < R; r += 256) hid[r] = hidden[(size_t)b * R + r];
  __syncthreads();
  const int ch = blockIdx.x * 256 + t;
  if (ch >= C) return;
  float s = b2[ch];
  for (int r = 0; r < R; ++r) s += w2[(size_t)ch * R + r] * hid[r];
  gate[(size_t)b * C + ch] = 1.f / (1.f + expf(-s));
}

// dpre2[b][c] = dgate*gate*(1-gate);  dw2[c][r] = sum_b dpre2[b][c]*hidden[b][r]; db2[c]
__global__ void se_bwd_w2_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                 const float* __restrict__ hidden, float* __restrict__ dw2, float* __restrict__ db2,
                                 int B, int C, int R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * R) return;
  const int ch = i / R, r = i % R;
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = gate[(size_t)b * C + ch];
    const float d = dgate[(size_t)b * C + ch] * g * (1.f - g);
    s += d * hidden[(size_t)b * R + r];
    sb += d;
  }
  dw2[i] = s;
  if (r == 0) db2[ch] = sb;
}

// dpre1[b][r] = (sum_c dpre2[b][c]*w2[c][r]) * (hidden>0)   -> ws[b][r]
__global__ __launch_bounds__(256) void se_bwd_hidden_kernel(const float* __restrict__ dgate,
                                                            const float* __restrict__ gate,
                                                            const float* __restrict__ hidden,
                                                            const float* __restrict__ w2, float* __restrict__ dpre1,
                                                            int C, int R) {
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int r = wave; r < R; r += 4) {
    float s = 0.f;
    for (int ch = lane; ch < C; ch += 64) {
      const float g = gate[(size_t)b * C + ch];
      s += dgate[(size_t)b * C + ch] * g * (1.f - g) * w2[(size_t)ch * R + r];
    }
    s = wave_sum(s);
    if (lane == 0) dpre1[(size_t)b * R + r] = hidden[(size_t)b * R + r] > 0.f ? s : 0.f;
  }
}

// dw1[r][c] = sum_b dpre1[b][r]*pooled[b][c]; db1[r]; dpooled[b][c] = scale * sum_r dpre1[b][r]*w1[r][c]
__global__ void se_bwd_w1_kernel(const float* __restrict__ dpre1, const float* __restrict__ pooled,
                                 const float* __restrict__ w1, float* __restrict__ dw1, float* __restrict__ db1,
                                 float* __restrict__ dpooled, int B, int C, int R, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R * C) {
    const int r = i / C, ch = i % C;
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
      s += dpre1[(size_t)b * R + r] * pooled[(size_t)b * C + ch];
      sb += dpre1[(size_t)b * R + r];
    }
    dw1[i] = s;
    if (ch == 0) db1[r] = sb;
  }
  if (i < B * C) {
    const int b = i / C, ch = i % C;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += dpre1[(size_t)b * R + r] * w1[(size_t)r * C + ch];
    dpooled[i] = s * scale;
  }
}

__global__ __launch_bounds__(256) void chan_scale_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                         const float* __restrict__ add, float* __restrict__ out,
                                                         int64_t HW, int C, size_t total4) {
  const int c4 = C / 4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % c4) * 4;
    const int b = (int)((e / c4) / HW);
    f32x4 v = LD4(x + e * 4) * LD4(gate + (size_t)b * C + col);
    if (add) v += LD4(add + (size_t)b * C + col);
    *reinterpret_cast<f32x4*>(out + e * 4) = v;
  }
}

// ---- bilinear, align_corners=True ----------------------------------------------------------
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp lerp_index(int dst, float scale, int in_size) {
#pragma clang fp contract(off)  // the product is rounded before the subtraction below, as in ATen
  const float src = scale * (float)dst;
  Lerp o;
  o.i0 = (int)src;
  o.i1 = o.i0 + (o.i0 < in_size - 1 ? 1 : 0);
  o.l1 = src - (float)o.i0;
  o.l0 = 1.f - o.l1;
  return o;
}

// The confusion matrix (ARGMAX with `hist`) is counted per workgroup in LDS and flushed once: one global atomic per non-empty
// cell and workgroup instead of one per pixel (K * K counters shared by every pixel of the batch: 4.2 M adds to 361 addresses
// took 2.1 of the 20.3 ms of an 8-frame evaluation pass).  Integer adds: the result does not depend on the order.
constexpr int UPS_HIST_LDS = 1024;  // cells a workgroup counts in LDS (K <= 32); larger K: global atomics per pixel
template <bool ARGMAX>
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ logits, int ldl,
                                                       float* __restrict__ out, uint8_t* __restrict__ cls,
                                                       const uint8_t* __restrict__ gt, unsigned long long* __restrict__ hist,
                                                       int B, int h, int w, int K, int H, int W, float sy, float sx) {
  __shared__ unsigned lh[ARGMAX ? UPS_HIST_LDS : 1];
  const int KK = K * K;
  const bool local = ARGMAX && hist != nullptr && KK <= UPS_HIST_LDS;
  if (local) {
    for (int i = threadIdx.x; i < KK; i += blockDim.x) lh[i] = 0u;
    __syncthreads();
  }
  // rows of `logits` are ldl floats apart; whole 16-byte groups of classes when the padding allows (the head's rows are
  // 32 floats: 5 loads per corner for 19 classes instead of 19)
  const bool vec = (ldl & 3) == 0 && ((K + 3) & ~3) <= ldl && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
  const size_t total = (size_t)B * H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int X = (int)(e % W);
    const size_t q = e / W;
    const int Y = (int)(q % H), b = (int)(q / H);
    const Lerp ly = lerp_index(Y, sy, h), lx = lerp_index(X, sx, w);
    const float* p00 = logits + (((size_t)b * h + ly.i0) * w + lx.i0) * ldl;
    const float* p01 = logits + (((size_t)b * h + ly.i0) * w + lx.i1) * ldl;
    const float* p10 = logits + (((size_t)b * h + ly.i1) * w + lx.i0) * ldl;
    const float* p11 = logits + (((size_t)b * h + ly.i1) * w + lx.i1) * ldl;
    float best = -INFINITY;
    int arg = 0;
    auto take = [&](int k, float v) {
      if (ARGMAX) {
        if (v > best) {
          best = v;
          arg = k;
        }
      } else {
        out[(((size_t)b * K + k) * H + Y) * W + X] = v;
      }
    };
    if (vec) {
      for (int k = 0; k < K; k += 4) {
        const f32x4 a = LD4(p00 + k), c = LD4(p01 + k), d = LD4(p10 + k), f = LD4(p11 + k);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < K) take(k + j, ly.l0 * (lx.l0 * a[j] + lx.l1 * c[j]) + ly.l1 * (lx.l0 * d[j] + lx.l1 * f[j]));
      }
    } else {
      for (int k = 0; k < K; ++k) take(k, ly.l0 * (lx.l0 * p00[k] + lx.l1 * p01[k]) + ly.l1 * (lx.l0 * p10[k] + lx.l1 * p11[k]));
    }
    if (ARGMAX) {
      if (cls) cls[e] = (uint8_t)arg;
      if (hist) {  // fast_hist (func.py:77-79): rows = ground truth in [0,K), columns = prediction
        const int g = gt[e];
        if (g < K) {
          if (local) atomicAdd(&lh[g * K + arg], 1u);
          else atomicAdd(&hist[g * K + arg], 1ull);
        }
      }
    }
  }
  if (local) {
    __syncthreads();
    // every workgroup starts its flush at a different cell: the adds of one moment go to different addresses
    const int start = (int)((blockIdx.x * 37u) % (unsigned)KK);
    for (int i = threadIdx.x; i < KK; i += blockDim.x) {
      int cell = i + start;
      if (cell >= KK) cell -= KK;
      const unsigned n = lh[cell];
      if (n) atomicAdd(&hist[cell], (unsigned long long)n);
    }
  }
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dl,
                                                           int ldl, int B, int h, int w, int K, int H, int W, float sy,
                                                           float sx, float inv_sy, float inv_sx) {
  const size_t total = (size_t)B * h * w * K;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    size_t q = e / K;
    const int x = (int)(q % w);
    q /= w;
    const int y = (int)(q % h), b = (int)(q / h);
    // destination rows whose source coordinate falls in (y-1, y+1), with one row of slack
    int Y0 = (int)floorf((float)(y - 1) * inv_sy) - 1, Y1 = (int)ceilf((float)(y + 1) * inv_sy) + 1;
    int X0 = (int)floorf((float)(x - 1) * inv_sx) - 1, X1 = (int)ceilf((float)(x + 1) * inv_sx) + 1;
    Y0 = max(Y0, 0);
    X0 = max(X0, 0);
    Y1 = min(Y1, H - 1);
    X1 = min(X1, W - 1);
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
      const Lerp ly = lerp_index(Y, sy, h);
      const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
      if (wy == 0.f) continue;
      const float* row = dout + (((size_t)b * K + k) * H + Y) * W;
      float racc = 0.f;
      for (int X = X0; X <= X1; ++X) {
        const Lerp lx = lerp_index(X, sx, w);
        const float wx = (lx.i0 == x ? lx.l0 : 0.f) + (lx.i1 == x ? lx.l1 : 0.f);
        racc += wx * row[X];
      }
      acc += wy * racc;
    }
    dl[(((size_t)b * h + y) * w + x) * ldl + k] = acc;
  }
}

// ---- loss_calc(interp(logits), label): bilinear upsample (align_corners) -> cross-entropy, without the upsampled tensor ----
// The supervised step (segmentation.py:70-80) upsamples the [B,K,h,w] logits to the label resolution (159 MB at 512x1024,
// batch 4) only to reduce them to one scalar, and its backward pass writes and re-reads a gradient of the same size.  Here
// every output pixel's K logits are interpolated in registers from the low-resolution rows (4.3 MB: cache-resident) in both
// directions; only the labels (1 byte per pixel) are streamed.
constexpr int UCE_KMAX = 32;

// one thread per output pixel: -log softmax(v)[label] for valid labels; per-block partial (sum, count) into ws[block][2]
__global__ __launch_bounds__(256) void upsample_ce_fwd_kernel(const float* __restrict__ logits, int ldl,
                                                              const uint8_t* __restrict__ labels, float* __restrict__ ws, int B,
                                                              int h, int w, int K, int H, int W, float sy, float sx) {
  __shared__ float red[4];
  const size_t total = (size_t)B * H * W;
  float ce = 0.f, nv = 0.f;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int tl = labels[e];
    if (tl >= K) continue;  // 255 = ignore (loss.py:22-31)
    const int X = (int)(e % W);
    const size_t q = e / W;
    const int Y = (int)(q % H), b = (int)(q / H);
    const Lerp ly = lerp_index(Y, sy, h), lx = lerp_index(X, sx, w);
    const float* p00 = logits + (((size_t)b * h + ly.i0) * w + lx.i0) * ldl;
    const float* p01 = logits + (((size_t)b * h + ly.i0) * w + lx.i1) * ldl;
    const float* p10 = logits + (((size_t)b * h + ly.i1) * w + lx.i0) * ldl;
    const float* p11 = logits + (((size_t)b * h + ly.i1) * w + lx.i1) * ldl;
    float v[UCE_KMAX], m = -INFINITY, vt = 0.f;
#pragma unroll
    for (int k = 0; k < UCE_KMAX; ++k)
      if (k < K) {
        v[k] = ly.l0 * (lx.l0 * p00[k] + lx.l1 * p01[k]) + ly.l1 * (lx.l0 * p10[k] + lx.l1 * p11[k]);  // = upsample_kernel
        m = fmaxf(m, v[k]);
        if (k == tl) vt = v[k];
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < UCE_KMAX; ++k)
      if (k < K) sum += expf(v[k] - m);
    ce += m + logf(sum) - vt;
    nv += 1.f;
  }
  float r = block_sum_256(ce, red);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.x * 2] = r;
  r = block_sum_256(nv, red);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.x * 2 + 1] = r;
}

__global__ __launch_bounds__(256) void upsample_ce_finalize_kernel(const float* __restrict__ ws, int nblocks, float* __restrict__ result) {
  // fixed order: thread t sums the blocks t, t + 256, ... in double, then a fixed tree over the 256 threads
  __shared__ double sh[2][256];
  const int t = threadIdx.x;
  double s = 0.0, n = 0.0;
  for (int b = t; b < nblocks; b += 256) {
    s += (double)ws[(size_t)b * 2];
    n += (double)ws[(size_t)b * 2 + 1];
  }
  sh[0][t] = s;
  sh[1][t] = n;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      sh[0][t] += sh[0][t + o];
      sh[1][t] += sh[1][t + o];
    }
    __syncthreads();
  }
  if (t == 0) {
    result[0] = (float)(sh[0][0] / sh[1][0]);  // mean over the kept pixels; 0/0 = NaN as in the reference (loss.py:38 never fires)
    result[1] = (float)sh[1][0];
  }
}

// Gradient with respect to the LOW-resolution logits.  The interpolation weights are separable, so
//     dl[y][x][k] = sum_Y wy(Y, y) * ( sum_X wx(X, x) * g[Y][X][k] ),   g = a * (softmax(v) - onehot) at the kept pixels,
// is taken in two gather passes (fixed summation order: deterministic), each output pixel's softmax evaluated about once:
//   pass A, one workgroup per (image, output row Y, block of CW low-resolution columns): the row's interpolated low-res
//     rows r[x][k] = ly.l0 * p[y0][x][k] + ly.l1 * p[y1][x][k] go to LDS once, every output pixel of the block's span then
//     needs two of them (no global loads per pixel); its g[X][k] goes to LDS; threads (x, k) gather over their ~2 W/w
//     columns -> t[b][Y][x][k];
//   pass B, one thread per (b, y, x, k): gathers t over its ~2 H/h rows -> dl.
constexpr int UCE_SPAN = 640;  // output pixels of one pass-A workgroup (LDS: UCE_SPAN * K floats of g)

__global__ __launch_bounds__(256) void upsample_ce_bwd_rows_kernel(const float* __restrict__ logits, int ldl,
                                                                   const uint8_t* __restrict__ labels,
                                                                   const float* __restrict__ result,
                                                                   const float* __restrict__ gscale, float w_ce,
                                                                   float* __restrict__ tmp, int B, int h, int w, int K, int H, int W,
                                                                   float sy, float sx, float inv_sx, int CW) {
  extern __shared__ float uce_lds[];
  float* r = uce_lds;                    // [CW + 2][K]
  float* g = r + (CW + 2) * K;           // [UCE_SPAN][K]
  float* lw = g + UCE_SPAN * K;          // [UCE_SPAN]: l1 of the pixel's column interpolation
  int* li = reinterpret_cast<int*>(lw + UCE_SPAN);  // [UCE_SPAN]: its i0 (i1 = i0 + 1, or i0 at the right edge: l1 = 0 there)
  const int t = threadIdx.x;
  const int xb = blockIdx.x * CW, xe = min(w, xb + CW);  // low-resolution columns of this workgroup
  const int Y = blockIdx.y % H, b = blockIdx.y / H;
  const float nvalid = result[1];
  const float a_ce = nvalid > 0.f ? (gscale ? gscale[0] : 1.f) * w_ce / nvalid : 0.f;
  const Lerp ly = lerp_index(Y, sy, h);
  // output columns that read a column of [xb, xe): source coordinate in (xb - 1, xe), with one of slack on either side
  int X0 = (int)floorf((float)(xb - 1) * inv_sx) - 1, X1 = (int)ceilf((float)xe * inv_sx) + 1;
  X0 = max(X0, 0);
  X1 = min(X1, W - 1);
  const int span = X1 - X0 + 1;          // <= UCE_SPAN (the host picks CW)
  const int x_lo = max(xb - 1, 0), nxr = min(w, xe + 1) - x_lo;  // low-res columns those pixels interpolate between
  const float* p0 = logits + ((size_t)b * h + ly.i0) * w * ldl;
  const float* p1 = logits + ((size_t)b * h + ly.i1) * w * ldl;
  for (int idx = t; idx < nxr * K; idx += 256) {
    const int x = idx / K, k = idx - x * K;
    r[idx] = ly.l0 * p0[(size_t)(x_lo + x) * ldl + k] + ly.l1 * p1[(size_t)(x_lo + x) * ldl + k];
  }
  __syncthreads();
  for (int i = t; i < span; i += 256) {
    const int X = X0 + i;
    const Lerp lx = lerp_index(X, sx, w);
    li[i] = lx.i0;
    lw[i] = lx.i1 != lx.i0 ? lx.l1 : 0.f;
    float* gi = g + i * K;
    const int tl = labels[((size_t)b * H + Y) * W + X];
    const bool inside = lx.i0 >= x_lo && lx.i1 < x_lo + nxr;  // (slack pixels outside the block's rows contribute nothing)
    if (tl >= K || !inside) {
      for (int k = 0; k < K; ++k) gi[k] = 0.f;
      continue;
    }
    const float* r0 = r + (lx.i0 - x_lo) * K;
    const float* r1 = r + (lx.i1 - x_lo) * K;
    float v[UCE_KMAX], m = -INFINITY;
#pragma unroll
    for (int k = 0; k < UCE_KMAX; ++k)
      if (k < K) {
        v[k] = lx.l0 * r0[k] + lx.l1 * r1[k];
        m = fmaxf(m, v[k]);
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < UCE_KMAX; ++k)
      if (k < K) {
        v[k] = expf(v[k] - m);
        sum += v[k];
      }
    const float inv = a_ce / sum;
#pragma unroll
    for (int k = 0; k < UCE_KMAX; ++k)
      if (k < K) gi[k] = v[k] * inv - (k == tl ? a_ce : 0.f);
  }
  __syncthreads();
  float* out = tmp + (((size_t)b * H + Y) * w) * K;
  for (int idx = t; idx < (xe - xb) * K; idx += 256) {
    const int xr = idx / K, k = idx - xr * K, x = xb + xr;
    int A0 = (int)floorf((float)(x - 1) * inv_sx) - 1, A1 = (int)ceilf((float)(x + 1) * inv_sx) + 1;
    A0 = max(A0, X0);
    A1 = min(A1, X1);
    float acc = 0.f;
    for (int X = A0; X <= A1; ++X) {
      const int i = X - X0, i0 = li[i];
      const float l1 = lw[i];
      const float wx = i0 == x ? 1.f - l1 : (i0 + 1 == x ? l1 : 0.f);
      acc += wx * g[i * K + k];
    }
    out[(size_t)x * K + k] = acc;
  }
}

__global__ __launch_bounds__(256) void upsample_ce_bwd_cols_kernel(const float* __restrict__ tmp, float* __restrict__ dl, int ldl, int B,
                                                                   int h, int w, int K, int H, float sy, float inv_sy) {
  const size_t total = (size_t)B * h * w * ldl;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(e % ldl);
    size_t q = e / ldl;
    const int x = (int)(q % w);
    q /= w;
    const int y = (int)(q % h), b = (int)(q / h);
    float acc = 0.f;
    if (k < K) {
      int Y0 = (int)floorf((float)(y - 1) * inv_sy) - 1, Y1 = (int)ceilf((float)(y + 1) * inv_sy) + 1;
      Y0 = max(Y0, 0);
      Y1 = min(Y1, H - 1);
      for (int Y = Y0; Y <= Y1; ++Y) {
        const Lerp ly = lerp_index(Y, sy, h);
        const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
        if (wy != 0.f) acc += wy * tmp[(((size_t)b * H + Y) * w + x) * K + k];
      }
    }
    dl[e] = acc;  // (columns >= K: zero)
  }
}

static inline unsigned ew_grid(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// align_corners scale exactly as ATen computes it (float division, 0 when out size is 1)
static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

}  // namespace

extern "C" {

int onda_maxpool_fwd(const float* x, float* y, uint8_t* idx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                     onda_stream_t s) {
  ONDA_REQUIRE(x && y && idx && C % 4 == 0);
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, ONDA_STREAM(s), x, y, idx, B, Hi, Wi, C, Ho,
                     Wo, static_cast<_Float16*>(nullptr), static_cast<const float*>(nullptr));
  return ONDA_LAUNCH_RESULT();
}

int onda_maxpool_fwd_limbs(const float* x, const float* xamax, void* yl, uint8_t* idx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                           onda_stream_t s) {
  ONDA_REQUIRE(x && xamax && yl && idx && C % 32 == 0);
  if (!ONDA_ALIGNED16(yl)) return ONDA_EALIGN;
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, ONDA_STREAM(s), x, static_cast<float*>(nullptr), idx, B,
                     Hi, Wi, C, Ho, Wo, static_cast<_Float16*>(yl), xamax);
  return ONDA_LAUNCH_RESULT();
}

int onda_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                     onda_stream_t s) {
  ONDA_REQUIRE(dy && dx && idx && C % 4 == 0);
  const size_t total = (size_t)B * Hi * Wi * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, ONDA_STREAM(s), dy, idx, dx, B, Hi, Wi, C,
                     Ho, Wo);
  return ONDA_LAUNCH_RESULT();
}

int onda_se_fc_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2,
                   float* hidden, float* gate, int B, int C, int R, onda_stream_t s) {
  ONDA_REQUIRE(pooled && w1 && b1 && w2 && b2 && hidden && gate && R <= 4096);
  hipLaunchKernelGGL(se_fc1_kernel, dim3((R + 3) / 4, B), dim3(256), 0, ONDA_STREAM(s), pooled, w1, b1, hidden, C, R);
  hipLaunchKernelGGL(se_fc2_kernel, dim3((C + 255) / 256, B), dim3(256), R * sizeof(float), ONDA_STREAM(s), hidden, w2, b2, gate, C,
                     R);
  return ONDA_LAUNCH_RESULT();
}

int onda_se_fc_bwd(const float* dgate, const float* pooled, const float* hidden, const float* gate, const float* w1,
                   const float* w2, float* dw1, float* db1, float* dw2, float* db2, float* dpooled, float* ws,
                   float dpooled_scale, int B, int C, int R, onda_stream_t s) {
  ONDA_REQUIRE(dgate && pooled && hidden && gate && w1 && w2 && dw1 && db1 && dw2 && db2 && dpooled && ws);
  hipLaunchKernelGGL(se_bwd_w2_kernel, dim3((C * R + 255) / 256), dim3(256), 0, ONDA_STREAM(s), dgate, gate, hidden, dw2,
                     db2, B, C, R);
  hipLaunchKernelGGL(se_bwd_hidden_kernel, dim3(B), dim3(256), 0, ONDA_STREAM(s), dgate, gate, hidden, w2, ws, C, R);
  const int n = (R > B ? R : B) * C;
  hipLaunchKernelGGL(se_bwd_w1_kernel, dim3((n + 255) / 256), dim3(256), 0, ONDA_STREAM(s), ws, pooled, w1, dw1, db1,
                     dpooled, B, C, R, dpooled_scale);
  return ONDA_LAUNCH_RESULT();
}

int onda_chan_scale(const float* x, const float* gate, const float* add, float* out, int B, int64_t HW, int C,
                    onda_stream_t s) {
  ONDA_REQUIRE(x && gate && out && C % 4 == 0);
  const size_t total4 = (size_t)B * HW * C / 4;
  hipLaunchKernelGGL(chan_scale_kernel, dim3(ew_grid(total4)), dim3(256), 0, ONDA_STREAM(s), x, gate, add, out, HW, C,
                     total4);
  return ONDA_LAUNCH_RESULT();
}

int onda_upsample_fwd(const float* logits, int ldl, float* out, int B, int h, int w, int K, int H, int W,
                      onda_stream_t s) {
  ONDA_REQUIRE(logits && out && K <= ldl);
  hipLaunchKernelGGL((upsample_kernel<false>), dim3(ew_grid((size_t)B * H * W)), dim3(256), 0, ONDA_STREAM(s), logits,
                     ldl, out, (uint8_t*)nullptr, (const uint8_t*)nullptr, (unsigned long long*)nullptr, B, h, w, K, H, W,
                     ac_scale(h, H), ac_scale(w, W));
  return ONDA_LAUNCH_RESULT();
}

int onda_upsample_argmax(const float* logits, int ldl, uint8_t* cls, int B, int h, int w, int K, int H, int W,
                         onda_stream_t s) {
  ONDA_REQUIRE(logits && cls && K <= ldl && K <= 255);
  hipLaunchKernelGGL((upsample_kernel<true>), dim3(ew_grid((size_t)B * H * W)), dim3(256), 0, ONDA_STREAM(s), logits,
                     ldl, (float*)nullptr, cls, (const uint8_t*)nullptr, (unsigned long long*)nullptr, B, h, w, K, H, W,
                     ac_scale(h, H), ac_scale(w, W));
  return ONDA_LAUNCH_RESULT();
}

int onda_upsample_argmax_hist(const float* logits, int ldl, const uint8_t* labels, int64_t* hist, uint8_t* cls, int B,
                              int h, int w, int K, int H, int W, onda_stream_t s) {
  ONDA_REQUIRE(logits && labels && hist && K <= ldl && K <= 255);
  unsigned grid = ew_grid((size_t)B * H * W);
  if (K * K <= UPS_HIST_LDS && grid > 1024u) grid = 1024u;  // one flush of the workgroup's counts each: fewer, longer workgroups
  hipLaunchKernelGGL((upsample_kernel<true>), dim3(grid), dim3(256), 0, ONDA_STREAM(s), logits,
                     ldl, (float*)nullptr, cls, labels, reinterpret_cast<unsigned long long*>(hist), B, h, w, K, H, W,
                     ac_scale(h, H), ac_scale(w, W));
  return ONDA_LAUNCH_RESULT();
}

int64_t onda_upsample_ce_ws(int B, int H, int W) { return 2 * (int64_t)ew_grid((size_t)B * H * W); }

int onda_upsample_ce_fwd(const float* logits, int ldl, const uint8_t* labels, float* result, float* ws, int B, int h, int w, int K,
                         int H, int W, onda_stream_t s) {
  ONDA_REQUIRE(logits && labels && result && ws && K <= ldl && K <= UCE_KMAX && h > 1 && w > 1 && H > 1 && W > 1);
  const unsigned nb = ew_grid((size_t)B * H * W);
  hipLaunchKernelGGL(upsample_ce_fwd_kernel, dim3(nb), dim3(256), 0, ONDA_STREAM(s), logits, ldl, labels, ws, B, h, w, K, H, W,
                     ac_scale(h, H), ac_scale(w, W));
  hipLaunchKernelGGL(upsample_ce_finalize_kernel, dim3(1), dim3(256), 0, ONDA_STREAM(s), ws, (int)nb, result);
  return ONDA_LAUNCH_RESULT();
}

int64_t onda_upsample_ce_bwd_ws(int B, int w, int K, int H) { return (int64_t)B * H * w * K; }

int onda_upsample_ce_bwd(const float* logits, int ldl, const uint8_t* labels, const float* result, const float* gscale, float w_ce,
                         float* dlogits, float* ws, int B, int h, int w, int K, int H, int W, onda_stream_t s) {
  ONDA_REQUIRE(logits && labels && result && dlogits && ws && K <= ldl && K <= UCE_KMAX && h > 1 && w > 1 && H > 1 && W > 1);
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  // low-resolution columns per pass-A workgroup: as many as keep its output span within UCE_SPAN pixels
  int CW = (int)((UCE_SPAN - 6) * sx) - 2;
  if (CW > 64) CW = 64;
  ONDA_REQUIRE(CW >= 1);
  const size_t lds = ((size_t)(CW + 2) * K + (size_t)UCE_SPAN * K + 2 * UCE_SPAN) * sizeof(float);
  // 19 classes: 59 KB.  Every K the forward accepts (<= UCE_KMAX = 32: 95 KB) has to run backward as well (round-4 advisor): above
  // the default 64 KB limit of dynamic LDS the kernel asks for its size once (a CU of gfx950 has 160 KB)
  ONDA_REQUIRE(lds <= 160 * 1024);
  if (lds > 64 * 1024) {
    static size_t granted = 0;
    if (lds > granted) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(upsample_ce_bwd_rows_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      granted = lds;
    }
  }
  hipLaunchKernelGGL(upsample_ce_bwd_rows_kernel, dim3((w + CW - 1) / CW, B * H), dim3(256), lds, ONDA_STREAM(s), logits, ldl, labels,
                     result, gscale, w_ce, ws, B, h, w, K, H, W, sy, sx, 1.f / sx, CW);
  hipLaunchKernelGGL(upsample_ce_bwd_cols_kernel, dim3(ew_grid((size_t)B * h * w * ldl)), dim3(256), 0, ONDA_STREAM(s), ws, dlogits,
                     ldl, B, h, w, K, H, sy, 1.f / sy);
  return ONDA_LAUNCH_RESULT();
}

int onda_upsample_bwd(const float* dout, float* dlogits, int ldl, int B, int h, int w, int K, int H, int W,
                      onda_stream_t s) {
  ONDA_REQUIRE(dout && dlogits && K <= ldl && h > 1 && w > 1 && H > 1 && W > 1);
  const float sy = ac_scale(h, H), sx = ac_scale(w, W);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(ew_grid((size_t)B * h * w * K)), dim3(256), 0, ONDA_STREAM(s), dout,
                     dlogits, ldl, B, h, w, K, H, W, sy, sx, 1.f / sy, 1.f / sx);
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"
