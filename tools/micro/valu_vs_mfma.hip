// Microbenchmark: how fast does a wave's VALU / LDS-store stream run while the OTHER wave on its
// SIMD issues v_mfma_f32_16x16x32_bf16 back to back?  512 threads: waves 0-3 MFMA, waves 4-7 filler.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_vs_mfma.hip -o tools/micro/valu_vs_mfma && ./valu_vs_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>  // filler: 0 v_fma_f32, 1 v_cvt_pk_bf16_f32 chain (the limb split), 2 ds_write_b64, 3 v_and/v_sub integer split
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int run_mfma, int run_fill) {
  __shared__ unsigned char lds[65536];
  const int t = threadIdx.x, wave = t >> 6;
  long long t0 = 0, t1 = 0;
  if (wave < 4) {
    if (!run_mfma) return;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (t + i)); b[i] = (__bf16)(0.002f * (t - i)); }
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + t] = s;
  } else {
    if (!run_fill) return;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (t + i);
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {  // split pairs into 3 bf16 limbs: cvt, sub, cvt, sub, cvt
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 x = {v[i], v[i + 1]};
          bf2 h1 = __builtin_convertvector(x, bf2);
          f2 r1 = x - __builtin_convertvector(h1, f2);
          bf2 h2 = __builtin_convertvector(r1, bf2);
          f2 r2 = r1 - __builtin_convertvector(h2, f2);
          bf2 h3 = __builtin_convertvector(r2, bf2);
          v[i] += (float)h1[0] + (float)h2[1] + (float)h3[0];
          v[i + 1] += (float)h3[1];
        }
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 12; ++i)
          *reinterpret_cast<u32x2*>(lds + ((t & 255) * 8 + i * 2048 + (it & 1) * 24576)) = u32x2{(unsigned)t + i, (unsigned)it};
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // truncation split with integer ops: and, sub, and, sub
          const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[i]) & 0xFFFF0000u);
          const float r1 = v[i] - h1;
          const float h2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u);
          const float r2 = r1 - h2;
          v[i] = h1 + h2 * 1.5f + r2 * 2.0f;
        }
      }
    }
    for (int i = 0; i < 16; ++i) s += v[i];
    if (MODE == 2) s += lds[t];
    t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + t] = s;
  }
  if ((t & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 200;
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int rm = cfg != 1, rf = cfg != 0;
    hipMemset(cyc, 0, 256 * 8 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, iters, rm, rf);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double m = 0, f = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : f) += h[b * 8 + w];
    printf("%-28s %-10s mfma wave: %7.1f cyc/iter (96 MFMA = 1536 ideal)   filler wave: %7.1f cyc/iter\n", name,
           cfg == 0 ? "mfma only" : cfg == 1 ? "fill only" : "both", m / 1024 / iters, f / 1024 / iters);
  }
}
int main() {
  run<0>("80 v_fma_f32");
  run<1>("limb split of 16 values");
  run<2>("12 ds_write_b64");
  run<3>("integer split of 16 values");
  return 0;
}
