// What the chip sustains on v_mfma_f32_16x16x32_f16 (and, second table, v_mfma_f32_32x32x16_f16) alone: operands in registers (random bit patterns or
// zeros), 16 independent accumulators per wave, 1 or 2 waves per SIMD on every CU, ~50 ms of back-to-back
// launches.  Reports TFLOP/s and the in-kernel clock (s_memtime / wall_clock64).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak && ./tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ seed, float* out, long long* cyc, int iters, int zero) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned r = seed[(t * 64 + i * 16 + j * 2) & 0xFFFFF];
      a[i][j] = zero ? (_Float16)0.f : (_Float16)(((int)(r & 0xFFFF) - 32768) / 4096.f);
      b[i][j] = zero ? (_Float16)0.f : (_Float16)(((int)(r >> 16) - 32768) / 4096.f);
    }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
  const long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[t] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = w1 - w0; }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the same loop on the 32 x 32 x 16 shape: twice the flops per instruction, half the operand registers read per flop
__global__ __launch_bounds__(256) void k32(const unsigned* __restrict__ seed, float* out, long long* cyc, int iters, int zero) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned r = seed[(t * 64 + i * 16 + j * 2) & 0xFFFFF];
      a[i][j] = zero ? (_Float16)0.f : (_Float16)(((int)(r & 0xFFFF) - 32768) / 4096.f);
      b[i][j] = zero ? (_Float16)0.f : (_Float16)(((int)(r >> 16) - 32768) / 4096.f);
    }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
  const long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[t] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
  unsigned* seed; float* out; long long* cyc;
  std::vector<unsigned> h(1 << 20);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
  hipMalloc(&seed, h.size() * 4); hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps)
    for (int zero = 0; zero <= 1; ++zero) {
      const int blocks = 256 * wps;  // 4 waves per block: wps blocks per CU -> wps waves per SIMD
      hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 16);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, seed, out, cyc, iters, zero);
      hipEventRecord(e0);
      const int reps = 10;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, seed, out, cyc, iters, zero);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> c(blocks * 2);
      hipMemcpy(c.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
      double ghz = 0; for (int i = 0; i < blocks; ++i) ghz += (double)c[2 * i] / c[2 * i + 1] * 0.1; ghz /= blocks;
      const double flops = (double)reps * blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
      printf("%d wave(s)/SIMD, %s operands: %7.1f TFLOP/s of f16 MFMA, in-kernel clock %.2f GHz, %.1f cycles per MFMA per SIMD\n", wps,
             zero ? "all-zero" : "random  ", flops / (ms * 1e-3) / 1e12, ghz, (double)c[0] / (iters * 16.0) / 1.0 / wps * wps);
      hipFree(out); hipFree(cyc);
    }
  for (int wps = 1; wps <= 2; ++wps)
    for (int zero = 0; zero <= 1; ++zero) {
      const int blocks = 256 * wps;
      hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 16);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, seed, out, cyc, iters, zero);
      hipEventRecord(e0);
      const int reps = 10;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, seed, out, cyc, iters, zero);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> c(blocks * 2);
      hipMemcpy(c.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
      double ghz = 0; for (int i = 0; i < blocks; ++i) ghz += (double)c[2 * i] / c[2 * i + 1] * 0.1; ghz /= blocks;
      const double flops = (double)reps * blocks * 4 * iters * 8 * (2.0 * 32 * 32 * 16);
      printf("32x32x16: %d wave(s)/SIMD, %s operands: %7.1f TFLOP/s of f16 MFMA, in-kernel clock %.2f GHz\n", wps,
             zero ? "all-zero" : "random  ", flops / (ms * 1e-3) / 1e12, ghz);
      hipFree(out); hipFree(cyc);
    }
  return 0;
}
