// Input pipeline on the GPU: the arithmetic of the reference's loader,
// framework/dataset/segmentation_db.py:56-99 + base_dataset.py:89-95 -- Pillow's antialiased BICUBIC
// resize (separable, 22-bit fixed-point coefficients, uint8 intermediate after the horizontal
// pass), RGB->BGR, ToTensor + Normalize, and the NEAREST label resizes with the id map.
// The coefficient / index tables are built on the host in double precision exactly as Pillow
// builds them (onda_amd/pipeline.py); the kernels are integer multiply-accumulate + clip and
// IEEE fp32 sub/div, so the results are bit-identical to the reference's CPU path.
// All three kernels are HBM-bound byte movers (6 MB in / 12.6 MB out per 2048x1024 frame).
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow Resample.c

__device__ __forceinline__ unsigned char clip8(int acc) {
  const int v = acc >> PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// in u8[H][Win][3] -> out u8[H][Wout][3]; out(y, xx, c) = clip8(2^21 + sum_k in(y, xmin + k, c) * kk[xx][k])
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ in,
                                                         unsigned char* __restrict__ out, int H, int Win, int Wout,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk,
                                                         int ksize) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)H * Wout) return;
  const int xx = (int)(e % Wout), y = (int)(e / Wout);
  const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  const unsigned char* p = in + ((size_t)y * Win + x0) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int i = 0; i < n; ++i) {
    const int w = k[i];
    s0 += p[3 * i + 0] * w;
    s1 += p[3 * i + 1] * w;
    s2 += p[3 * i + 2] * w;
  }
  unsigned char* o = out + e * 3;
  o[0] = clip8(s0);
  o[1] = clip8(s1);
  o[2] = clip8(s2);
}

// tmp u8[Hin][W][3] -> out f32[3][Hout][W]: vertical pass, then channel c of the output is input
// channel (flip ? 2 - c : c) as ((v / 255) - mean[c]) / std[c] in fp32 (ToTensor + Normalize)
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char* __restrict__ tmp, float* __restrict__ out,
                                                              int Hin, int W, int Hout, const int* __restrict__ bounds,
                                                              const int* __restrict__ kk, int ksize, float m0, float m1,
                                                              float m2, float d0, float d1, float d2, int flip) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)Hout * W) return;
  const int x = (int)(e % W), yy = (int)(e / W);
  const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = kk + (size_t)yy * ksize;
  const unsigned char* p = tmp + ((size_t)y0 * W + x) * 3;
  int s[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
  for (int i = 0; i < n; ++i) {
    const int w = k[i];
    const unsigned char* q = p + (size_t)i * W * 3;
    s[0] += q[0] * w;
    s[1] += q[1] * w;
    s[2] += q[2] * w;
  }
  const float mean[3] = {m0, m1, m2}, sd[3] = {d0, d1, d2};
  const size_t plane = (size_t)Hout * W;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)clip8(s[flip ? 2 - c : c]);
    out[c * plane + e] = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.f), mean[c]), sd[c]);
  }
}

// in u8[Hin][Win] -> out u8[Hout][Wout] = lut[in[ytab[y]][xtab[x]]]  (NEAREST resize + id map)
__global__ __launch_bounds__(256) void nearest_lut_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                          int Win, int Hout, int Wout, const int* __restrict__ xtab,
                                                          const int* __restrict__ ytab, const unsigned char* __restrict__ lut) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)Hout * Wout) return;
  const int x = (int)(e % Wout), y = (int)(e / Wout);
  out[e] = lut[in[(size_t)ytab[y] * Win + xtab[x]]];
}

inline unsigned blocks(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int onda_resample_h_u8(const unsigned char* in, unsigned char* out, int H, int Win, int Wout, const int* bounds,
                       const int* kk, int ksize, onda_stream_t s) {
  ONDA_REQUIRE(in && out && bounds && kk && H > 0 && Win > 0 && Wout > 0 && ksize > 0);
  hipLaunchKernelGGL(resample_h_kernel, dim3(blocks((long long)H * Wout)), dim3(256), 0, ONDA_STREAM(s), in, out, H, Win, Wout,
                     bounds, kk, ksize);
  return ONDA_LAUNCH_RESULT();
}

int onda_resample_v_norm(const unsigned char* tmp, float* out, int Hin, int W, int Hout, const int* bounds, const int* kk,
                         int ksize, const float* mean3, const float* std3, int flip, onda_stream_t s) {
  ONDA_REQUIRE(tmp && out && bounds && kk && mean3 && std3 && Hin > 0 && W > 0 && Hout > 0 && ksize > 0);
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3(blocks((long long)Hout * W)), dim3(256), 0, ONDA_STREAM(s), tmp, out, Hin, W,
                     Hout, bounds, kk, ksize, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], flip);
  return ONDA_LAUNCH_RESULT();
}

int onda_resize_nearest_lut(const unsigned char* in, unsigned char* out, int Win, int Hout, int Wout, const int* xtab,
                            const int* ytab, const unsigned char* lut256, onda_stream_t s) {
  ONDA_REQUIRE(in && out && xtab && ytab && lut256 && Win > 0 && Hout > 0 && Wout > 0);
  hipLaunchKernelGGL(nearest_lut_kernel, dim3(blocks((long long)Hout * Wout)), dim3(256), 0, ONDA_STREAM(s), in, out, Win, Hout,
                     Wout, xtab, ytab, lut256);
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"
