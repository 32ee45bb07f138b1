// HBM-bound normalisation kernels: BatchNorm2d (batch / running statistics), GroupNorm,
// per-image column sums (SE pooling, bias gradients).  NHWC rows, 16-byte accesses, two-stage
// fixed-order reductions (block partials -> finalize) so results are bitwise reproducible.
#include "common.h"

namespace {

// ---- column reduce plan ------------------------------------------------------------------
// A block of 256 threads covers CX column quads (16 B each) x RY row lanes and walks a chunk
// of rows; partials[b][chunk][2][C].
struct ColPlan {
  int cx, ry, gridx, chunks;
  int64_t rows_per_chunk;
};

static inline ColPlan col_plan(int B, int64_t HW, int C) {
  ColPlan p;
  const int c4 = C / 4;
  p.cx = 1;
  while (p.cx < c4 && p.cx < 64) p.cx <<= 1;
  p.ry = 256 / p.cx;
  p.gridx = (c4 + p.cx - 1) / p.cx;
  int64_t target = 2048 / ((int64_t)B * p.gridx);
  if (target < 1) target = 1;
  int64_t rpc = (HW + target - 1) / target;
  const int64_t min_rows = (int64_t)p.ry * 8;
  if (rpc < min_rows) rpc = min_rows;
  p.rows_per_chunk = rpc;
  p.chunks = (int)((HW + rpc - 1) / rpc);
  return p;
}

template <class F>
__global__ __launch_bounds__(256) void colreduce_kernel(F f, int64_t HW, int C, int cx, int64_t rows_per_chunk,
                                                        float* __restrict__ partials) {
  __shared__ f32x4 red[2][256];
  const int t = threadIdx.x;
  const int ry_n = 256 / cx;
  const int tx = t % cx, ty = t / cx;
  const int col = (blockIdx.x * cx + tx) * 4;
  const int chunk = blockIdx.y, b = blockIdx.z;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  const int64_t r1 = min(HW, r0 + rows_per_chunk);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (col < C) {
    for (int64_t r = r0 + ty; r < r1; r += ry_n) {
      f32x4 a, bb;
      f(b, r, col, a, bb);
      s1 += a;
      s2 += bb;
    }
  }
  red[0][t] = s1;
  red[1][t] = s2;
  __syncthreads();
  if (ty == 0 && col < C) {
    for (int y = 1; y < ry_n; ++y) {
      s1 += red[0][y * cx + tx];
      s2 += red[1][y * cx + tx];
    }
    float* dst = partials + (((size_t)b * gridDim.y + chunk) * 2) * C + col;
    *reinterpret_cast<f32x4*>(dst) = s1;
    *reinterpret_cast<f32x4*>(dst + C) = s2;
  }
}

template <class F>
static int launch_colreduce(const F& f, int B, int64_t HW, int C, float* partials, const ColPlan& p, hipStream_t s) {
  dim3 grid(p.gridx, p.chunks, B);
  hipLaunchKernelGGL((colreduce_kernel<F>), grid, dim3(256), 0, s, f, HW, C, p.cx, p.rows_per_chunk, partials);
  return ONDA_LAUNCH_RESULT();
}

#define LD4(p) (*reinterpret_cast<const f32x4*>(p))

struct SqStats {  // a = x, b = x*x
  const float* x;
  int64_t HW;
  int ldx;
  __device__ void operator()(int b, int64_t r, int col, f32x4& a, f32x4& bb) const {
    a = LD4(x + ((size_t)b * HW + r) * ldx + col);
    bb = a * a;
  }
};

struct ProdSum {  // a = x (* y)
  const float* x;
  const float* y;
  int64_t HW;
  int ldx, ldy;
  __device__ void operator()(int b, int64_t r, int col, f32x4& a, f32x4& bb) const {
    a = LD4(x + ((size_t)b * HW + r) * ldx + col);
    if (y) a *= LD4(y + ((size_t)b * HW + r) * ldy + col);
    bb = f32x4{0.f, 0.f, 0.f, 0.f};
  }
};

struct BnBwdRed {  // g = dout*(out>0); a = g, b = g*xhat; optional dres = g
  const float* dout;
  const float* out;
  const float* x;
  const float* mean;
  const float* invstd;
  float* dres;
  int C, relu;
  __device__ void operator()(int, int64_t r, int col, f32x4& a, f32x4& bb) const {
    const size_t o = (size_t)r * C + col;
    f32x4 g = LD4(dout + o);
    if (relu) {
      const f32x4 ov = LD4(out + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = ov[j] > 0.f ? g[j] : 0.f;
    }
    if (dres) *reinterpret_cast<f32x4*>(dres + o) = g;
    const f32x4 xh = (LD4(x + o) - LD4(mean + col)) * LD4(invstd + col);
    a = g;
    bb = g * xh;
  }
};

struct GnBwdRed {
  const float* dout;
  const float* out;
  const float* x;
  const float* chmul;
  const float* mean;
  const float* rstd;
  int64_t HW;
  int lddo, ldo, ldx, C, groups, relu;
  __device__ void operator()(int b, int64_t r, int col, f32x4& a, f32x4& bb) const {
    const size_t row = (size_t)b * HW + r;
    f32x4 g = LD4(dout + row * lddo + col);
    if (relu) {
      const f32x4 ov = LD4(out + row * ldo + col);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = ov[j] > 0.f ? g[j] : 0.f;
    }
    if (chmul) g *= LD4(chmul + (size_t)b * C + col);
    const int grp = col / (C / groups);  // the 4 columns share a group (C/groups is a multiple of 4)
    const float mu = mean[b * groups + grp], rs = rstd[b * groups + grp];
    const f32x4 xh = (LD4(x + row * ldx + col) - mu) * rs;
    a = g;
    bb = g * xh;
  }
};

// ---- BatchNorm --------------------------------------------------------------------------
// One wave per channel: lanes stride over the tile partials, fp64 wave reduction.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int tiles, int C,
                                                          double count, float eps, float* mean, float* invstd,
                                                          float* rmean, float* rvar, int64_t* nbt, float momentum) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  if (ch >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int tl = lane; tl < tiles; tl += 64) {
    s1 += (double)partials[((size_t)tl * 2 + 0) * C + ch];
    s2 += (double)partials[((size_t)tl * 2 + 1) * C + ch];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane != 0) return;
  const double mu = s1 / count;
  double var = s2 / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[ch] = (float)mu;
  invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)mu;
    rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, float* __restrict__ out,
                                                       size_t total4, int C, int relu, float* __restrict__ amax) {
  const int c4 = C / 4;
  float mx = 0.f;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % c4) * 4;
    const f32x4 sc = LD4(invstd + col) * LD4(gamma + col);
    f32x4 v = (LD4(x + e * 4) - LD4(mean + col)) * sc + LD4(beta + col);
    if (res) v += LD4(res + e * 4);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + e * 4) = v;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  __shared__ float amax_red[4];
  if (amax != nullptr) amax_update_block(amax, mx, amax_red);
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                               float* scale, float* shift, int C) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  // same association as ATen's eval-mode batch_norm: (x - mean) * (gamma * invstd) + beta
  const float sc = gamma[ch] / sqrtf(rvar[ch] + eps);
  scale[ch] = sc;
  shift[ch] = beta[ch] - rmean[ch] * sc;
}

__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ partials, int chunks, int C,
                                                          float* sums) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < chunks; k += 64) {
    s1 += (double)partials[((size_t)k * 2 + 0) * C + ch];
    s2 += (double)partials[((size_t)k * 2 + 1) * C + ch];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane == 0) {
    sums[ch] = (float)s1;
    sums[C + ch] = (float)s2;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout,
                                                           const float* __restrict__ out, const float* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, float* __restrict__ dx,
                                                           size_t total4, int C, float inv_m, int relu,
                                                           float* __restrict__ amax) {
  const int c4 = C / 4;
  float mx = 0.f;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % c4) * 4;
    f32x4 g = LD4(dout + e * 4);
    if (relu) {
      const f32x4 ov = LD4(out + e * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = ov[j] > 0.f ? g[j] : 0.f;
    }
    const f32x4 is = LD4(invstd + col);
    const f32x4 xh = (LD4(x + e * 4) - LD4(mean + col)) * is;
    const f32x4 v = LD4(gamma + col) * is * (g - LD4(sums + col) * inv_m - xh * (LD4(sums + C + col) * inv_m));
    *reinterpret_cast<f32x4*>(dx + e * 4) = v;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  __shared__ float amax_red[4];
  if (amax != nullptr) amax_update_block(amax, mx, amax_red);
}

// ---- GroupNorm --------------------------------------------------------------------------
// One workgroup per (image, group): 256 threads stride over chunks x channels-of-the-group, waves combined in fixed order
// (one wave per pair walked 16+ dependent iterations: 17 us for 30 KB).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partials, int chunks, int B,
                                                          int C, int groups, double count, float eps, float* mean,
                                                          float* rstd) {
  __shared__ double red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = blockIdx.x;
  const int b = idx / groups, g = idx % groups, cpg = C / groups;
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < chunks * cpg; i += 256) {
    const int k = i / cpg, j = i - k * cpg;
    const size_t base = (((size_t)b * chunks + k) * 2) * C + g * cpg + j;
    s1 += (double)partials[base];
    s2 += (double)partials[base + C];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane == 0) {
    red[0][wave] = s1;
    red[1][wave] = s2;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  s1 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
  s2 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  const double mu = s1 / count;
  double var = s2 / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[idx] = (float)mu;
  rstd[idx] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ chmul, float* __restrict__ out,
                                                       int ldo, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int64_t HW, int C, int groups,
                                                       size_t total4, int relu, float* __restrict__ amax) {
  const int c4 = C / 4, cpg = C / groups;
  float mx = 0.f;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % c4) * 4;
    const size_t row = e / c4;
    const int b = (int)(row / HW);
    const int gi = b * groups + col / cpg;
    f32x4 v = (LD4(x + row * ldx + col) - mean[gi]) * rstd[gi] * LD4(gamma + col) + LD4(beta + col);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    if (chmul) v *= LD4(chmul + (size_t)b * C + col);
    *reinterpret_cast<f32x4*>(out + row * ldo + col) = v;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (amax != nullptr) amax_update(amax, mx);  // the output feeds a convolution (directly, or through the SE gate): leave max|out| behind
}

// partials[b][chunk][2][C] -> AB[b][2][C] (one wave per (b,c))
__global__ __launch_bounds__(256) void pair_finalize_kernel(const float* __restrict__ partials, int chunks, int B,
                                                            int C, float* __restrict__ AB) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B * C) return;
  const int b = i / C, ch = i % C;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < chunks; k += 64) {
    const size_t base = (((size_t)b * chunks + k) * 2) * C + ch;
    s1 += (double)partials[base];
    s2 += (double)partials[base + C];
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane == 0) {
    AB[((size_t)b * 2 + 0) * C + ch] = (float)s1;
    AB[((size_t)b * 2 + 1) * C + ch] = (float)s2;
  }
}

// AB[b][2][C] -> dgamma/dbeta and the per-(image, group) sums the dx formula needs
__global__ void gn_bwd_group_kernel(const float* __restrict__ AB, int B, int C, int groups,
                                    const float* __restrict__ gamma, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, float* __restrict__ gsum /*[B][groups][2]*/) {
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    float da = 0.f, db = 0.f;
    for (int b = 0; b < B; ++b) {
      db += AB[((size_t)b * 2 + 0) * C + ch];
      da += AB[((size_t)b * 2 + 1) * C + ch];
    }
    if (dgamma) dgamma[ch] = da;
    if (dbeta) dbeta[ch] = db;
  }
  const int cpg = C / groups;
  for (int i = threadIdx.x; i < B * groups; i += blockDim.x) {
    const int b = i / groups, g = i % groups;
    float ds = 0.f, dq = 0.f;
    for (int j = 0; j < cpg; ++j) {
      const int ch = g * cpg + j;
      ds += gamma[ch] * AB[((size_t)b * 2 + 0) * C + ch];
      dq += gamma[ch] * AB[((size_t)b * 2 + 1) * C + ch];
    }
    gsum[i * 2 + 0] = ds;
    gsum[i * 2 + 1] = dq;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dout, int lddo,
                                                           const float* __restrict__ out, int ldo,
                                                           const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ chmul,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gsum, float* __restrict__ dx,
                                                           int64_t HW, int C, int groups, size_t total4, float inv_n,
                                                           int relu) {
  const int c4 = C / 4, cpg = C / groups;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % c4) * 4;
    const size_t row = e / c4;
    const int b = (int)(row / HW);
    const int gi = b * groups + col / cpg;
    f32x4 g = LD4(dout + row * lddo + col);
    if (relu) {
      const f32x4 ov = LD4(out + row * ldo + col);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = ov[j] > 0.f ? g[j] : 0.f;
    }
    if (chmul) g *= LD4(chmul + (size_t)b * C + col);
    const float rs = rstd[gi];
    const f32x4 xh = (LD4(x + row * ldx + col) - mean[gi]) * rs;
    const f32x4 v = rs * (LD4(gamma + col) * g - (gsum[gi * 2] + xh * gsum[gi * 2 + 1]) * inv_n);
    *reinterpret_cast<f32x4*>(dx + row * C + col) = v;
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partials, int chunks, int B,
                                                              int C, float alpha, float* out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B * C) return;
  const int b = i / C, ch = i % C;
  double s = 0.0;
  for (int k = lane; k < chunks; k += 64) s += (double)partials[(((size_t)b * chunks + k) * 2) * C + ch];
  s = wave_sum_d(s);
  if (lane == 0) out[i] = (float)(s * (double)alpha);
}

static inline unsigned ew_grid(size_t total4) {
  size_t g = (total4 + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" {

int onda_bn_finalize(const float* partials, int tiles, int C, int64_t count, float eps, float* mean, float* invstd,
                     float* running_mean, float* running_var, int64_t* nbt, float momentum, onda_stream_t s) {
  ONDA_REQUIRE(partials && mean && invstd && tiles >= 1 && C >= 1 && count >= 1);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, ONDA_STREAM(s), partials, tiles, C,
                     (double)count, eps, mean, invstd, running_mean, running_var, nbt, momentum);
  return ONDA_LAUNCH_RESULT();
}

int onda_bn_stats(const float* x, int64_t M, int C, int ldx, float* partials, int* tiles_out, onda_stream_t s) {
  ONDA_REQUIRE(x && partials && tiles_out && C % 4 == 0 && ldx % 4 == 0);
  const ColPlan p = col_plan(1, M, C);
  *tiles_out = p.chunks;
  SqStats f{x, M, ldx};
  return launch_colreduce(f, 1, M, C, partials, p, ONDA_STREAM(s));
}

int onda_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const float* residual, float* out, int64_t M, int C, int relu, float* amax, onda_stream_t s) {
  ONDA_REQUIRE(x && mean && invstd && gamma && beta && out && C % 4 == 0);
  const size_t total4 = (size_t)M * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, ONDA_STREAM(s), x, mean, invstd, gamma, beta,
                     residual, out, total4, C, relu, amax);
  return ONDA_LAUNCH_RESULT();
}

int onda_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale,
                 float* shift, int C, onda_stream_t s) {
  ONDA_REQUIRE(gamma && beta && rm && rv && scale && shift);
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, ONDA_STREAM(s), gamma, beta, rm, rv, eps,
                     scale, shift, C);
  return ONDA_LAUNCH_RESULT();
}

int64_t onda_bn_bwd_ws(int64_t M, int C) {
  const ColPlan p = col_plan(1, M, C);
  return (int64_t)p.chunks * 2 * C + 2 * C;
}

int onda_bn_bwd(const float* dout, const float* out, const float* x, const float* mean, const float* invstd,
                const float* gamma, float* dx, float* dres, float* ws, int64_t M, int C, int relu, float* amax,
                onda_stream_t s) {
  ONDA_REQUIRE(dout && x && mean && invstd && gamma && dx && ws && C % 4 == 0 && (!relu || out));
  const ColPlan p = col_plan(1, M, C);
  float* sums = ws + (size_t)p.chunks * 2 * C;
  BnBwdRed f{dout, out, x, mean, invstd, dres, C, relu};
  int rc = launch_colreduce(f, 1, M, C, ws, p, ONDA_STREAM(s));
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, ONDA_STREAM(s), ws, p.chunks, C, sums);
  const size_t total4 = (size_t)M * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, ONDA_STREAM(s), dout, out, x, mean,
                     invstd, gamma, sums, dx, total4, C, (float)(1.0 / (double)M), relu, amax);
  return ONDA_LAUNCH_RESULT();
}

int64_t onda_gn_ws(int B, int64_t HW, int C) {
  const ColPlan p = col_plan(B, HW, C);
  return (int64_t)B * p.chunks * 2 * C + (int64_t)B * C * 2 + (int64_t)B * 2 * 128 + 64;
}

int onda_gn_fwd(const float* x, int ldx, const float* gamma, const float* beta, const float* chmul, float* out, int ldo,
                float* mean, float* rstd, float* ws, int B, int64_t HW, int C, int groups, float eps, int relu, float* amax,
                onda_stream_t s) {
  ONDA_REQUIRE(x && gamma && beta && out && mean && rstd && ws);
  ONDA_REQUIRE(C % groups == 0 && (C / groups) % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0);
  const ColPlan p = col_plan(B, HW, C);
  SqStats f{x, HW, ldx};
  int rc = launch_colreduce(f, B, HW, C, ws, p, ONDA_STREAM(s));
  if (rc) return rc;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(256), 0, ONDA_STREAM(s), ws, p.chunks, B,
                     C, groups, (double)HW * (C / groups), eps, mean, rstd);
  const size_t total4 = (size_t)B * HW * C / 4;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, ONDA_STREAM(s), x, ldx, gamma, beta, chmul,
                     out, ldo, mean, rstd, HW, C, groups, total4, relu, amax);
  return ONDA_LAUNCH_RESULT();
}

int onda_gn_bwd(const float* dout, int lddo, const float* out, int ldo, const float* x, int ldx, const float* gamma,
                const float* chmul, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                float* ws, int B, int64_t HW, int C, int groups, int relu, onda_stream_t s) {
  ONDA_REQUIRE(dout && x && gamma && mean && rstd && dx && ws && (!relu || out));
  ONDA_REQUIRE(C % groups == 0 && (C / groups) % 4 == 0 && ldx % 4 == 0 && lddo % 4 == 0);
  const ColPlan p = col_plan(B, HW, C);
  float* AB = ws + (size_t)B * p.chunks * 2 * C;
  float* gsum = AB + (size_t)B * 2 * C;
  GnBwdRed f{dout, out, x, chmul, mean, rstd, HW, lddo, ldo, ldx, C, groups, relu};
  int rc = launch_colreduce(f, B, HW, C, ws, p, ONDA_STREAM(s));
  if (rc) return rc;
  hipLaunchKernelGGL(pair_finalize_kernel, dim3((B * C + 3) / 4), dim3(256), 0, ONDA_STREAM(s), ws, p.chunks, B, C, AB);
  hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(1), dim3(256), 0, ONDA_STREAM(s), AB, B, C, groups, gamma, dgamma, dbeta,
                     gsum);
  const size_t total4 = (size_t)B * HW * C / 4;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, ONDA_STREAM(s), dout, lddo, out, ldo, x,
                     ldx, gamma, chmul, mean, rstd, gsum, dx, HW, C, groups, total4,
                     (float)(1.0 / ((double)HW * (C / groups))), relu);
  return ONDA_LAUNCH_RESULT();
}

int64_t onda_colsum_ws(int B, int64_t HW, int C) {
  const ColPlan p = col_plan(B, HW, C);
  return (int64_t)B * p.chunks * 2 * C;
}

int onda_colsum(const float* x, int ldx, const float* y, int ldy, float* out, float alpha, float* ws, int B,
                int64_t HW, int C, onda_stream_t s) {
  ONDA_REQUIRE(x && out && ws && C % 4 == 0 && ldx % 4 == 0 && (!y || ldy % 4 == 0));
  const ColPlan p = col_plan(B, HW, C);
  ProdSum f{x, y, HW, ldx, ldy};
  int rc = launch_colreduce(f, B, HW, C, ws, p, ONDA_STREAM(s));
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((B * C + 3) / 4), dim3(256), 0, ONDA_STREAM(s), ws, p.chunks, B, C,
                     alpha, out);
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"
