// Per-pixel class-vector kernels: softmax statistics, the fused target losses and their
// gradient, prototype distances / pseudo-labels, prototype statistics and EMA, plus the
// multi-tensor SGD and teacher-EMA updates.  All HBM-bound; reductions are two-stage with a
// fixed order (bitwise reproducible).
#include <cstdlib>
#include "common.h"

namespace {

constexpr int KMAX = 32;              // class vectors are kept in registers, K <= 32
constexpr float LOG_CLAMP = -9.21034037f;  // log(1e-4): loss.py:104-106 clamps the one-hot to [1e-4, 1]

// result[j] = scale * sum over blocks of ws[block][j]; one workgroup of 64 * nvals threads: a wave per value, lanes strided
// over the blocks, fixed order (lane-local sums in block order, then the wave tree): bitwise reproducible
__global__ void sum_partials_kernel(const float* __restrict__ ws, int nblocks, int nvals, float scale,
                                    float* __restrict__ result) {
  const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (j >= nvals) return;
  double s = 0.0;
  for (int b = lane; b < nblocks; b += 64) s += (double)ws[(size_t)b * nvals + j];
  s = wave_sum_d(s);
  if (lane == 0) result[j] = (float)(s * (double)scale);
}

// ---- softmax statistics ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float* __restrict__ logits, int ldl,
                                                            float* __restrict__ probs, int ldp,
                                                            int32_t* __restrict__ argmax, float* __restrict__ ws,
                                                            int64_t N, int K) {
  __shared__ float red[4];
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float pmax = 0.f;
  if (n < N) {
    const float* row = logits + (size_t)n * ldl;
    float v[KMAX];
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        v[k] = row[k];
        if (v[k] > m) {
          m = v[k];
          am = k;
        }
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        v[k] = expf(v[k] - m);
        sum += v[k];
      }
    const float inv = 1.f / sum;
    if (probs) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) probs[(size_t)n * ldp + k] = v[k] / sum;
    }
    if (argmax) argmax[n] = am;
    pmax = inv;  // exp(0)/sum at the maximum
  }
  const float tot = block_sum_256(pmax, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = tot;
}

// ---- CE + RCE + MRKLD ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ logits, int ldl,
                                                           const int64_t* __restrict__ labels,
                                                           float* __restrict__ ws, int64_t N, int K) {
  __shared__ float red[4];
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float ce = 0.f, rce = 0.f, kld = 0.f, nv = 0.f, nm = 0.f;
  if (n < N) {
    const float* row = logits + (size_t)n * ldl;
    float v[KMAX];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        v[k] = row[k];
        m = fmaxf(m, v[k]);
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) sum += expf(v[k] - m);
    const float lse = m + logf(sum);
    const int64_t tl = labels[n];
    const bool valid = tl >= 0 && tl != 255 && tl < K;
    const bool mask = tl != 255;
    float others = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        const float lp = v[k] - lse;
        kld -= lp;
        if (valid && k == (int)tl) ce = -lp;
        if (mask && k != (int)tl) others += expf(lp);
      }
    rce = mask ? -LOG_CLAMP * others : 0.f;
    nv = valid ? 1.f : 0.f;
    nm = mask ? 1.f : 0.f;
  }
  float r;
  r = block_sum_256(ce, red);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.x * 8 + 0] = r;
  r = block_sum_256(rce, red);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.x * 8 + 1] = r;
  r = block_sum_256(kld, red);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.x * 8 + 2] = r;
  r = block_sum_256(nv, red);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.x * 8 + 3] = r;
  r = block_sum_256(nm, red);
  if (threadIdx.x == 0) ws[(size_t)blockIdx.x * 8 + 4] = r;
}

__global__ void seg_loss_finalize_kernel(const float* __restrict__ ws, int nblocks, double total_elems,
                                         float* __restrict__ result) {
  if (threadIdx.x != 0) return;
  double s[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < nblocks; ++b)
    for (int j = 0; j < 5; ++j) s[j] += (double)ws[(size_t)b * 8 + j];
  result[0] = (float)(s[0] / s[3]);  // mean over kept pixels; 0/0 = NaN as in the reference
  result[1] = (float)(s[1] / (s[4] + 1e-6));
  result[2] = (float)(s[2] / total_elems);
  result[3] = (float)s[3];
  result[4] = (float)s[4];
}

__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ logits, int ldl,
                                                           const int64_t* __restrict__ labels,
                                                           const float* __restrict__ result,
                                                           const float* __restrict__ gscale, float w_ce, float w_rce,
                                                           float w_reg, float* __restrict__ dl, int64_t N, int K) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float gs = gscale ? gscale[0] : 1.f;
  const float nvalid = result[3], nmask = result[4];
  const float a_ce = nvalid > 0.f ? gs * w_ce / nvalid : 0.f;
  const float a_rce = gs * w_rce * (-LOG_CLAMP) / (nmask + 1e-6f);
  const float a_reg = gs * w_reg / ((float)N * (float)K);
  const float* row = logits + (size_t)n * ldl;
  float v[KMAX];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      v[k] = row[k];
      m = fmaxf(m, v[k]);
    }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      v[k] = expf(v[k] - m);
      sum += v[k];
    }
  const int64_t tl = labels[n];
  const bool valid = tl >= 0 && tl != 255 && tl < K;
  const bool mask = tl != 255;
  float pt = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      v[k] = v[k] / sum;
      if (k == (int)tl) pt = v[k];
    }
  float* out = dl + (size_t)n * ldl;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k >= ldl) break;
    float g = 0.f;
    if (k < K) {
      const float onehot = (k == (int)tl) ? 1.f : 0.f;
      if (valid) g += a_ce * (v[k] - onehot);
      if (mask) g += a_rce * pt * (v[k] - onehot);
      g += a_reg * ((float)K * v[k] - 1.f);
    }
    out[k] = g;
  }
}

// ---- prototypes ---------------------------------------------------------------------------
__global__ void proto_sigma_kernel(const float* __restrict__ proto, const float* __restrict__ sqmean,
                                   const float* __restrict__ counter, float* __restrict__ sigma, int K, int C) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  float total = 0.f;
  for (int k = 0; k < K; ++k) total += counter[k];
  float gsq = 0.f, gm = 0.f;
  for (int k = 0; k < K; ++k) {
    gsq += sqmean[(size_t)k * C + ch] * counter[k] / total;
    gm += proto[(size_t)k * C + ch] * counter[k] / total;
  }
  sigma[ch] = sqrtf(gsq - gm * gm);
}

// The distance matrix by itself (prototype_handler.distance / mahalanobis_distance, :111-138): dist[n][k] = D[k] - min_k D,
// the direct form of proto_assign_kernel; one wave per pixel.
__global__ __launch_bounds__(256) void proto_distances_kernel(const float* __restrict__ feat, int ldf,
                                                              const float* __restrict__ proto,
                                                              const float* __restrict__ sigma, int mahalanobis,
                                                              float* __restrict__ dist, int64_t N, int K) {
  __shared__ __attribute__((aligned(16))) float sp[KMAX * 256];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < K * 64; i += 256) reinterpret_cast<f32x4*>(sp)[i] = reinterpret_cast<const f32x4*>(proto)[i];
  __syncthreads();
  f32x4 sg = {1.f, 1.f, 1.f, 1.f};
  if (mahalanobis) sg = *reinterpret_cast<const f32x4*>(sigma + lane * 4);
  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
    const f32x4 f = *reinterpret_cast<const f32x4*>(feat + (size_t)n * ldf + lane * 4);
    float mine = 0.f, dmin = INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        f32x4 df = f - reinterpret_cast<const f32x4*>(sp)[k * 64 + lane];
        if (mahalanobis) df = df / sg;
        const float d = sqrtf(wave_sum(df[0] * df[0] + df[1] * df[1] + df[2] * df[2] + df[3] * df[3]));
        dmin = fminf(dmin, d);
        if (k == lane) mine = d;
      }
    if (lane < K) dist[(size_t)n * K + lane] = mine - dmin;
  }
}

// One wave per pixel; lane l owns channels 4l..4l+3 (C == 256).
__global__ __launch_bounds__(256) void proto_assign_kernel(const float* __restrict__ feat, int ldf,
                                                           const float* __restrict__ prior, int ldp,
                                                           const float* __restrict__ proto,
                                                           const float* __restrict__ sigma, int mahalanobis,
                                                           float tau, float thresh, int64_t* __restrict__ labels,
                                                           float* __restrict__ soft, float* __restrict__ ws, int64_t N,
                                                           int K, const int* __restrict__ list = nullptr,
                                                           const int* __restrict__ list_len = nullptr) {
  // (list != nullptr: only the pixels list[0 .. *list_len) -- the close decisions of proto_assign_mfma_kernel)
  __shared__ __attribute__((aligned(16))) float sp[KMAX * 256];
  __shared__ float red[3][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < K * 64; i += 256) reinterpret_cast<f32x4*>(sp)[i] = reinterpret_cast<const f32x4*>(proto)[i];
  __syncthreads();
  f32x4 sg = {1.f, 1.f, 1.f, 1.f};
  if (mahalanobis) sg = *reinterpret_cast<const f32x4*>(sigma + lane * 4);
  float a_conf = 0.f, a_soft = 0.f, a_prior = 0.f;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  const int64_t count = list ? (int64_t)*list_len : N;
  for (int64_t i_ = (int64_t)blockIdx.x * 4 + wave; i_ < count; i_ += wstride) {
    const int64_t n = list ? (int64_t)list[i_] : i_;
    const f32x4 f = *reinterpret_cast<const f32x4*>(feat + (size_t)n * ldf + lane * 4);
    float d[KMAX];
    float dmin = INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        f32x4 df = f - reinterpret_cast<const f32x4*>(sp)[k * 64 + lane];
        if (mahalanobis) df = df / sg;
        float s = df[0] * df[0] + df[1] * df[1] + df[2] * df[2] + df[3] * df[3];
        s = wave_sum(s);
        d[k] = sqrtf(s);
        dmin = fminf(dmin, d[k]);
      }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        d[k] = expf(-(d[k] - dmin) / tau);
        sum += d[k];
      }
    float conf = 0.f, psum = 0.f, prmax = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        float p = d[k] / sum;
        conf = fmaxf(conf, p);
        if (prior) {
          const float pr = prior[(size_t)n * ldp + k];
          prmax = fmaxf(prmax, pr);
          p *= pr;
        }
        d[k] = p;
        psum += p;
      }
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
        d[k] = d[k] / psum;
        if (d[k] > best) {
          best = d[k];
          arg = k;
        }
      }
    if (lane == 0) labels[n] = best < thresh ? 255 : arg;
    if (soft) {
      float mine = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K && k == lane) mine = d[k];
      if (lane < K) soft[(size_t)n * K + lane] = mine;
    }
    a_conf += conf;
    a_soft += best;
    a_prior += prmax;
  }
  if (lane == 0) {
    red[0][wave] = a_conf;
    red[1][wave] = a_soft;
    red[2][wave] = a_prior;
  }
  __syncthreads();
  if (t < 3) ws[(size_t)blockIdx.x * 3 + t] = red[t][0] + red[t][1] + red[t][2] + red[t][3];
}

// ---- the same assignment with the feature <-> prototype contraction on the matrix cores -------------------------------------
//   D^2[n][k] = |f_n/s|^2 - 2 (f_n/s).(p_k/s) + |p_k/s|^2,   the [N x 256] x [256 x K<=32] product on v_mfma_f32_32x32x2_f32
// (fp32 operands, fp32 accumulate: exact products).  A wave takes 32 pixels: their scaled features are staged through LDS
// in two 128-channel halves (coalesced 16-byte loads, rows padded to 129 floats: conflict-free operand reads), the scaled
// prototypes live in LDS for the whole workgroup (rows of 257 floats).  K order: MFMA step s of a half multiplies channel
// s (lanes 0-31) and channel 64+s (lanes 32-63) -- any order is a valid order of the sum.  The accumulator tile goes
// through LDS once more so that lane p < 32 holds pixel p's K products and runs the softmax / prior / argmax / threshold
// exactly like the direct kernel above.
// The expanded form cancels where the direct form (what the reference computes, prototype_handler.py:111-138) does
// not: its distances carry an absolute error of ~1e-5.  Decisions that close are not trusted: a pixel whose two largest
// posteriors, or whose largest posterior and the threshold, lie within `margin` goes on a list and is redone by the
// direct kernel (proto_assign_list_kernel); everything else -- labels, soft map, monitor sums -- is final here.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PA_TS = 129, PA_PS = 257, PA_GS = 33;
constexpr int PA_WAVES = 6;  // 33 KB of prototypes + 6 x 16.6 KB of feature tiles = 133 KB: one workgroup per CU; 1 536 waves
                             // take the 1 049 pixel blocks of a 65 x 129 x 4 grid in ONE round (four waves per CU: two)
constexpr int PA_WAVE_FLOATS = 32 * PA_TS + 32;  // a wave's half-tile + its 32 squared feature norms

__global__ __launch_bounds__(64 * PA_WAVES) void proto_assign_mfma_kernel(const float* __restrict__ feat, int ldf,
                                                                const float* __restrict__ prior, int ldp,
                                                                const float* __restrict__ proto,
                                                                const float* __restrict__ sigma, int mahalanobis, float tau,
                                                                float thresh, float margin, int64_t* __restrict__ labels,
                                                                float* __restrict__ soft, float* __restrict__ ws,
                                                                int* __restrict__ flagged, int* __restrict__ nflagged,
                                                                int64_t N, int K) {
  __shared__ float sm[32 * PA_PS + 32 + PA_WAVES * PA_WAVE_FLOATS];
  __shared__ float red[3][PA_WAVES];
  float* ph = sm;                // [32][PA_PS]: prototypes / sigma, rows >= K zero
  float* pn = sm + 32 * PA_PS;   // [32]: their squared norms
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* tile = pn + 32 + wave * PA_WAVE_FLOATS;  // [32][PA_TS], later the product tile [32][PA_GS]
  for (int i = t; i < 32 * 256; i += 64 * PA_WAVES) {
    const int k = i >> 8, ch = i & 255;
    float v = 0.f;
    if (k < K) v = mahalanobis ? proto[k * 256 + ch] / sigma[ch] : proto[k * 256 + ch];
    ph[k * PA_PS + ch] = v;
  }
  __syncthreads();
  if (t < 256) {  // squared norms: 8 threads per class, 32 channels each, fixed order
    const int k = t >> 3, part = t & 7;
    float a = 0.f;
#pragma unroll 8
    for (int ch = part * 32; ch < part * 32 + 32; ++ch) a += ph[k * PA_PS + ch] * ph[k * PA_PS + ch];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 4, 64);
    if (part == 0) pn[k] = a;
  }
  __syncthreads();
  const int half_lane = lane >> 5, l32 = lane & 31;
  float a_conf = 0.f, a_soft = 0.f, a_prior = 0.f;
  const int64_t nblocks = (N + 31) / 32;
  for (int64_t blk = (int64_t)blockIdx.x * PA_WAVES + wave; blk < nblocks; blk += (int64_t)gridDim.x * PA_WAVES) {
    const int64_t n0 = blk * 32;
    float f2 = 0.f;  // lanes 0-31: squared norm of pixel l32's scaled features
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // this lane's pixel's priors, requested now, used after the contraction
    float pr_[KMAX];
    {
      const int64_t n = n0 + l32;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) pr_[k] = (prior != nullptr && k < K && n < N && half_lane == 0) ? prior[(size_t)n * ldp + k] : 1.f;
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      f32x4 sg = {1.f, 1.f, 1.f, 1.f};  // 1 / sigma (a reciprocal, not the direct kernel's division: close decisions are redone there)
      if (mahalanobis) sg = f32x4{1.f, 1.f, 1.f, 1.f} / *reinterpret_cast<const f32x4*>(sigma + half * 128 + l32 * 4);
      // stage 32 pixels x 128 channels: a 16-byte load per lane covers two pixel rows per instruction; all 16 loads of
      // the half go out before the first is used (a wave has about one block: nothing else hides the latency)
      f32x4 fv[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int64_t n = n0 + 2 * it + half_lane;
        fv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n < N) fv[it] = *reinterpret_cast<const f32x4*>(feat + (size_t)n * ldf + half * 128 + l32 * 4);
      }
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = 2 * it + half_lane;
        const f32x4 v = fv[it] * sg;
        float* dst = tile + row * PA_TS + l32 * 4;
        dst[0] = v[0];
        dst[1] = v[1];
        dst[2] = v[2];
        dst[3] = v[3];
      }
      __builtin_amdgcn_wave_barrier();  // one wave: its LDS instructions execute in order
      if (half_lane == 0) {  // the pixel's squared norm from its staged row (conflict-free reads, no cross-lane traffic)
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
        for (int ch = 0; ch < 128; ch += 2) {
          const float x0 = tile[l32 * PA_TS + ch], x1 = tile[l32 * PA_TS + ch + 1];
          a0 += x0 * x0;
          a1 += x1 * x1;
        }
        f2 += a0 + a1;
      }
      const float* ap = tile + l32 * PA_TS + half_lane * 64;
      const float* bp = ph + l32 * PA_PS + half * 128 + half_lane * 64;
#pragma unroll 8
      for (int s_ = 0; s_ < 64; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s_], bp[s_], acc, 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
    }
    // accumulator (row 8*(v/4) + 4*(lane/32) + v%4, column lane%32) -> [pixel][class] in LDS
#pragma unroll
    for (int v = 0; v < 16; ++v) tile[(8 * (v >> 2) + 4 * half_lane + (v & 3)) * PA_GS + l32] = acc[v];
    __builtin_amdgcn_wave_barrier();
    const int64_t n = n0 + l32;
    if (half_lane == 0 && n < N) {
      float d[KMAX];
      float dmin = INFINITY;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          d[k] = sqrtf(fmaxf(f2 + pn[k] - 2.f * tile[l32 * PA_GS + k], 0.f));
          dmin = fminf(dmin, d[k]);
        }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          d[k] = expf(-(d[k] - dmin) / tau);
          sum += d[k];
        }
      float conf = 0.f, psum = 0.f, prmax = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          float pp = d[k] / sum;
          conf = fmaxf(conf, pp);
          if (prior) {
            prmax = fmaxf(prmax, pr_[k]);
            pp *= pr_[k];
          }
          d[k] = pp;
          psum += pp;
        }
      float best = -INFINITY, second = -INFINITY;
      int arg = 0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          d[k] = d[k] / psum;
          if (d[k] > best) {
            second = best;
            best = d[k];
            arg = k;
          } else if (d[k] > second) {
            second = d[k];
          }
        }
      if (best - second < margin || fabsf(best - thresh) < margin || !(best == best)) {
        flagged[atomicAdd(nflagged, 1)] = (int)n;  // redone in the direct form
      } else {
        labels[n] = best < thresh ? 255 : arg;
        if (soft) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (k < K) soft[(size_t)n * K + k] = d[k];
        }
        a_conf += conf;
        a_soft += best;
        a_prior += prmax;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  a_conf = wave_sum(a_conf);
  a_soft = wave_sum(a_soft);
  a_prior = wave_sum(a_prior);
  if (lane == 0) {
    red[0][wave] = a_conf;
    red[1][wave] = a_soft;
    red[2][wave] = a_prior;
  }
  __syncthreads();
  if (t < 3) {
    float a = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < PA_WAVES; ++w_) a += red[t][w_];
    ws[(size_t)blockIdx.x * 3 + t] = a;
  }
}

constexpr int SUMS_BLOCKS = 256;

__global__ __launch_bounds__(256) void proto_class_sums_kernel(const float* __restrict__ feat, int ldf,
                                                               const int32_t* __restrict__ cls,
                                                               float* __restrict__ ws, int64_t N, int C, int K,
                                                               int64_t rows_per_block) {
  extern __shared__ float acc[];  // [2][K][C] + [K]
  float* cnt = acc + (size_t)2 * K * C;
  const int t = threadIdx.x;
  for (int i = t; i < 2 * K * C + K; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int64_t n0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t n1 = min(N, n0 + rows_per_block);
  for (int64_t n = n0; n < n1; ++n) {
    const int k = cls[n];
    if ((unsigned)k >= (unsigned)K) continue;
    for (int ch = t; ch < C; ch += 256) {
      const float v = feat[(size_t)n * ldf + ch];
      acc[(size_t)k * C + ch] += v;
      acc[(size_t)(K + k) * C + ch] += v * v;
    }
    if (t == 0) cnt[k] += 1.f;
  }
  __syncthreads();
  float* dst = ws + (size_t)blockIdx.x * (2 * K * C + K);
  for (int i = t; i < 2 * K * C + K; i += 256) dst[i] = acc[i];
}

__global__ void proto_sums_finalize_kernel(const float* __restrict__ ws, int nblocks, int KC2, int K,
                                           float* __restrict__ sums, float* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= KC2 + K) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += ws[(size_t)b * (KC2 + K) + i];
  if (i < KC2)
    sums[i] = s;
  else
    counts[i - KC2] = s;
}

__global__ void proto_ema_kernel(float* __restrict__ proto, float* __restrict__ sqmean, const float* __restrict__ sums,
                                 const float* __restrict__ counts, float lam, int K, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * C) return;
  const int k = i / C;
  const float n = counts[k];
  const float keep = n > 0.f ? lam : 1.f;
  const float denom = n > 0.f ? n : 1.f;
  proto[i] = proto[i] * keep + (1.f - keep) * (sums[i] / denom);
  sqmean[i] = sqmean[i] * keep + (1.f - keep) * (sums[(size_t)K * C + i] / denom);
}

__global__ void proto_append_kernel(float* __restrict__ proto, float* __restrict__ sqmean, float* __restrict__ counter,
                                    const float* __restrict__ sums, const float* __restrict__ counts, int K, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * C) return;
  const int k = i / C;
  const float n = counts[k];
  const float tot = counter[k] + n;
  const float denom = tot > 0.f ? tot : 1.f;
  proto[i] += (sums[i] - proto[i] * n) / denom;
  sqmean[i] += (sums[(size_t)K * C + i] - sqmean[i] * n) / denom;
}
__global__ void proto_append_counter_kernel(float* __restrict__ counter, const float* __restrict__ counts, int K) {
  const int k = threadIdx.x;
  if (k < K) counter[k] += counts[k];
}

// ---- multi-tensor optimizer / teacher ---------------------------------------------------------
// FLAT launches (see pack_h2_multi_kernel): an entry owns the blocks [first_block, first_block + ceil(n / MT_BLOCK)) of a 1-D
// grid and is found by bisection -- a (largest tensor) x (217 tensors) grid is nine tenths empty workgroups.
constexpr int MT_BLOCK = 4096;  // elements per workgroup: four 16-byte vectors per thread
template <class E>
__device__ __forceinline__ int mt_entry_of(const E* __restrict__ table, int n, int block) {
  // the block starts go to LDS first (one load per thread): ten dependent global loads per workgroup were most of a
  // 16 KB workgroup's life
  __shared__ int starts[1024];
  const bool in_lds = n <= 1024;
  if (in_lds) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) starts[i] = table[i].first_block;
    __syncthreads();
  }
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((in_lds ? starts[mid] : table[mid].first_block) <= block) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ void sgd_one(float& p, float& b, float g, float lr, float momentum, float wd, int times, int fresh) {
  if (fresh) b = 0.f;
  for (int r = 0; r < times; ++r) {
    const float d = g + wd * p;
    b = fresh ? d : b * momentum + d;
    p = p - lr * b;
  }
}

// grad_scale: every gradient element is multiplied by it on the way in (1 / world: the exchanged buffer holds rank SUMS, the
// division never becomes a pass of its own over 184 MB)
__global__ __launch_bounds__(256) void sgd_multi_kernel(const OndaSgdEntry* __restrict__ table, int n_entries, float momentum,
                                                        float wd, float grad_scale) {
  const OndaSgdEntry e = table[mt_entry_of(table, n_entries, blockIdx.x)];
  const int64_t i0 = (int64_t)(blockIdx.x - e.first_block) * MT_BLOCK, i1 = i0 + MT_BLOCK < e.n ? i0 + MT_BLOCK : e.n;
  const bool vec = ((reinterpret_cast<size_t>(e.p) | reinterpret_cast<size_t>(e.g) | reinterpret_cast<size_t>(e.buf)) & 15) == 0;
  if (vec && i1 - i0 == MT_BLOCK) {
#pragma unroll
    for (int k = 0; k < MT_BLOCK / 1024; ++k) {
      const int64_t i = i0 + (k * 256 + threadIdx.x) * 4;
      f32x4 p = *reinterpret_cast<const f32x4*>(e.p + i), b = e.fresh ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(e.buf + i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(e.g + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = p[j], bj = b[j];
        sgd_one(pj, bj, g[j] * grad_scale, e.lr, momentum, wd, e.times, e.fresh);
        p[j] = pj;
        b[j] = bj;
      }
      *reinterpret_cast<f32x4*>(e.p + i) = p;
      *reinterpret_cast<f32x4*>(e.buf + i) = b;
    }
    return;
  }
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    float p = e.p[i], b = e.fresh ? 0.f : e.buf[i];
    sgd_one(p, b, e.g[i] * grad_scale, e.lr, momentum, wd, e.times, e.fresh);
    e.p[i] = p;
    e.buf[i] = b;
  }
}

__global__ __launch_bounds__(256) void ema_multi_kernel(const OndaEmaEntry* __restrict__ table, int n_entries) {
  const OndaEmaEntry e = table[mt_entry_of(table, n_entries, blockIdx.x)];
  const int64_t i0 = (int64_t)(blockIdx.x - e.first_block) * MT_BLOCK, i1 = i0 + MT_BLOCK < e.n ? i0 + MT_BLOCK : e.n;
  const bool vec = ((reinterpret_cast<size_t>(e.k) | reinterpret_cast<size_t>(e.q)) & 15) == 0;
  if (vec && i1 - i0 == MT_BLOCK) {
#pragma unroll
    for (int k = 0; k < MT_BLOCK / 1024; ++k) {
      const int64_t i = i0 + (k * 256 + threadIdx.x) * 4;
      *reinterpret_cast<f32x4*>(e.k + i) = *reinterpret_cast<const f32x4*>(e.k + i) * e.keep + *reinterpret_cast<const f32x4*>(e.q + i) * e.blend;
    }
    return;
  }
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) e.k[i] = e.k[i] * e.keep + e.q[i] * e.blend;
}

}  // namespace

extern "C" {

int onda_softmax_stats(const float* logits, int ldl, float* probs, int ldp, int32_t* argmax, float* result, float* ws,
                       int64_t N, int K, onda_stream_t s) {
  ONDA_REQUIRE(logits && result && ws && K >= 1 && K <= KMAX && N >= 1);
  const int nb = (int)((N + 255) / 256);
  hipLaunchKernelGGL(softmax_stats_kernel, dim3(nb), dim3(256), 0, ONDA_STREAM(s), logits, ldl, probs, ldp, argmax, ws,
                     N, K);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, ONDA_STREAM(s), ws, nb, 1, (float)(1.0 / (double)N),
                     result);
  return ONDA_LAUNCH_RESULT();
}

int onda_seg_loss_fwd(const float* logits, int ldl, const int64_t* labels, float* result, float* ws, int64_t N, int K,
                      onda_stream_t s) {
  ONDA_REQUIRE(logits && labels && result && ws && K >= 1 && K <= KMAX && N >= 1);
  const int nb = (int)((N + 255) / 256);
  hipLaunchKernelGGL(seg_loss_fwd_kernel, dim3(nb), dim3(256), 0, ONDA_STREAM(s), logits, ldl, labels, ws, N, K);
  hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(64), 0, ONDA_STREAM(s), ws, nb, (double)N * K, result);
  return ONDA_LAUNCH_RESULT();
}

int onda_seg_loss_bwd(const float* logits, int ldl, const int64_t* labels, const float* result, const float* gscale,
                      float w_ce, float w_rce, float w_reg, float* dlogits, int64_t N, int K, onda_stream_t s) {
  ONDA_REQUIRE(logits && labels && result && dlogits && K >= 1 && K <= KMAX && ldl <= KMAX && N >= 1);
  const int nb = (int)((N + 255) / 256);
  hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3(nb), dim3(256), 0, ONDA_STREAM(s), logits, ldl, labels, result, gscale,
                     w_ce, w_rce, w_reg, dlogits, N, K);
  return ONDA_LAUNCH_RESULT();
}

int onda_proto_sigma(const float* proto, const float* sqmean, const float* counter, float* sigma, int K, int C,
                     onda_stream_t s) {
  ONDA_REQUIRE(proto && sqmean && counter && sigma);
  hipLaunchKernelGGL(proto_sigma_kernel, dim3((C + 255) / 256), dim3(256), 0, ONDA_STREAM(s), proto, sqmean, counter,
                     sigma, K, C);
  return ONDA_LAUNCH_RESULT();
}

// workspace of onda_proto_assign in units of 3 floats: partial sums of the MFMA pass (PA_GRID workgroups) and of the direct
// pass over the close decisions (PA_LIST_GRID), the list itself (N ints) and its length
constexpr int PA_GRID = 256, PA_LIST_GRID = 256;
static bool proto_direct_only() { return false; }  // (true: every pixel in the direct form -- the round-1 path the MFMA contraction replaced)
int onda_proto_assign_blocks(int64_t N) {
  int64_t nb = (N + 3) / 4;
  if (nb > 2048) nb = 2048;
  const int64_t mfma = PA_GRID + PA_LIST_GRID + (N + 4 + 2) / 3;
  return (int)(nb > mfma ? nb : mfma);
}

int onda_proto_assign(const float* feat, int ldf, const float* prior, int ldp, const float* proto, const float* sigma,
                      int mahalanobis, float tau, float thresh, int64_t* labels, float* soft, float* result, float* ws,
                      int64_t N, int C, int K, onda_stream_t s) {
  ONDA_REQUIRE(feat && proto && labels && result && ws && C == 256 && K >= 1 && K <= KMAX && N >= 1 && ldf % 4 == 0);
  ONDA_REQUIRE(!mahalanobis || sigma);
  if (!ONDA_ALIGNED16(feat) || !ONDA_ALIGNED16(proto)) return ONDA_EALIGN;
  if (proto_direct_only() || N >= (1ll << 31)) {  // the direct form everywhere (measurement / fallback for huge N)
    int64_t nb = (N + 3) / 4;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(proto_assign_kernel, dim3((unsigned)nb), dim3(256), 0, ONDA_STREAM(s), feat, ldf, prior, ldp, proto, sigma,
                       mahalanobis, tau, thresh, labels, soft, ws, N, K, nullptr, nullptr);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(192), 0, ONDA_STREAM(s), ws, (int)nb, 3, (float)(1.0 / (double)N),
                       result);
    return ONDA_LAUNCH_RESULT();
  }
  int* list = reinterpret_cast<int*>(ws + 3 * (PA_GRID + PA_LIST_GRID));
  int* list_len = list + N;
  hipError_t e = hipMemsetAsync(list_len, 0, sizeof(int), ONDA_STREAM(s));
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(proto_assign_mfma_kernel, dim3(PA_GRID), dim3(64 * PA_WAVES), 0, ONDA_STREAM(s), feat, ldf, prior, ldp, proto, sigma,
                     mahalanobis, tau, thresh, 1.0e-3f, labels, soft, ws, list, list_len, N, K);
  hipLaunchKernelGGL(proto_assign_kernel, dim3(PA_LIST_GRID), dim3(256), 0, ONDA_STREAM(s), feat, ldf, prior, ldp, proto, sigma,
                     mahalanobis, tau, thresh, labels, soft, ws + 3 * PA_GRID, N, K, list, list_len);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(192), 0, ONDA_STREAM(s), ws, PA_GRID + PA_LIST_GRID, 3,
                     (float)(1.0 / (double)N), result);
  return ONDA_LAUNCH_RESULT();
}

int onda_proto_distances(const float* feat, int ldf, const float* proto, const float* sigma, int mahalanobis, float* dist,
                         int64_t N, int C, int K, onda_stream_t s) {
  ONDA_REQUIRE(feat && proto && dist && N >= 1 && C == 256 && K >= 1 && K <= KMAX && ldf % 4 == 0 && (!mahalanobis || sigma));
  if (!ONDA_ALIGNED16(feat) || !ONDA_ALIGNED16(proto) || (sigma && !ONDA_ALIGNED16(sigma))) return ONDA_EALIGN;
  int64_t nb = (N + 3) / 4;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(proto_distances_kernel, dim3((unsigned)nb), dim3(256), 0, ONDA_STREAM(s), feat, ldf, proto, sigma, mahalanobis,
                     dist, N, K);
  return ONDA_LAUNCH_RESULT();
}

int64_t onda_proto_sums_ws(int64_t N, int C, int K) { return (int64_t)SUMS_BLOCKS * (2 * K * C + K); }

int onda_proto_class_sums(const float* feat, int ldf, const int32_t* cls, float* sums, float* counts, float* ws,
                          int64_t N, int C, int K, onda_stream_t s) {
  ONDA_REQUIRE(feat && cls && sums && counts && ws && N >= 1);
  const size_t lds = ((size_t)2 * K * C + K) * sizeof(float);
  ONDA_REQUIRE(lds <= 64 * 1024);
  const int64_t rpb = (N + SUMS_BLOCKS - 1) / SUMS_BLOCKS;
  hipLaunchKernelGGL(proto_class_sums_kernel, dim3(SUMS_BLOCKS), dim3(256), lds, ONDA_STREAM(s), feat, ldf, cls, ws, N,
                     C, K, rpb);
  const int tot = 2 * K * C + K;
  hipLaunchKernelGGL(proto_sums_finalize_kernel, dim3((tot + 255) / 256), dim3(256), 0, ONDA_STREAM(s), ws, SUMS_BLOCKS,
                     2 * K * C, K, sums, counts);
  return ONDA_LAUNCH_RESULT();
}

int onda_proto_ema(float* proto, float* sqmean, const float* sums, const float* counts, float lam, int K, int C,
                   onda_stream_t s) {
  ONDA_REQUIRE(proto && sqmean && sums && counts);
  hipLaunchKernelGGL(proto_ema_kernel, dim3((K * C + 255) / 256), dim3(256), 0, ONDA_STREAM(s), proto, sqmean, sums,
                     counts, lam, K, C);
  return ONDA_LAUNCH_RESULT();
}

int onda_proto_append(float* proto, float* sqmean, float* counter, const float* sums, const float* counts, int K, int C,
                      onda_stream_t s) {
  ONDA_REQUIRE(proto && sqmean && counter && sums && counts && K <= 64);
  hipLaunchKernelGGL(proto_append_kernel, dim3((K * C + 255) / 256), dim3(256), 0, ONDA_STREAM(s), proto, sqmean,
                     counter, sums, counts, K, C);
  hipLaunchKernelGGL(proto_append_counter_kernel, dim3(1), dim3(64), 0, ONDA_STREAM(s), counter, counts, K);
  return ONDA_LAUNCH_RESULT();
}

int onda_sgd_multi(const OndaSgdEntry* table, int n, float momentum, float weight_decay, float grad_scale, int64_t max_n,
                   onda_stream_t s) {
  ONDA_REQUIRE(table && n >= 1 && max_n >= 1 && max_n < (1ll << 31));  // max_n: total blocks (flat launch)
  hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)max_n), dim3(256), 0, ONDA_STREAM(s), table, n, momentum, weight_decay, grad_scale);
  return ONDA_LAUNCH_RESULT();
}

int onda_multi_tensor_block(void) { return MT_BLOCK; }

int onda_ema_multi(const OndaEmaEntry* table, int n, int64_t max_n, onda_stream_t s) {
  ONDA_REQUIRE(table && n >= 1 && max_n >= 1 && max_n < (1ll << 31));  // max_n: total blocks (flat launch)
  hipLaunchKernelGGL(ema_multi_kernel, dim3((unsigned)max_n), dim3(256), 0, ONDA_STREAM(s), table, n);
  return ONDA_LAUNCH_RESULT();
}

// ONDA_SRC_HASH: sha256 (first 16 hex digits) over csrc/*.hip, csrc/*.h and include/*.h at build time (onda_amd/build.py):
// whoever loads the library can check that it was compiled from the sources beside it (onda_amd.build.source_hash()).
#ifndef ONDA_SRC_HASH
#define ONDA_SRC_HASH "unknown"
#endif
float onda_limb2_scale(void) { return ONDA_LIMB2_SCALE; }

const char* onda_version(void) { return "onda_hip 0.2 (gfx950) src=" ONDA_SRC_HASH; }

}  // extern "C"
