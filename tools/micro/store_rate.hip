// What a CU's output path sustains when a tile epilogue fires: every workgroup (8 waves, one per CU) issues 16 stores of
// 16 bytes per lane per wave = 128 KB, the way conv_l2's epilogue does, R times with `gap` K-step-like pauses between.
// Reports ticks (s_memtime) per 128 KB burst seen by the issuing wave, bytes per tick per CU, and the aggregate rate,
// for: G workgroups (256 = the whole chip, 32 = four per XCD), row-segment stores (4 rows x 256 B per instruction, row
// stride ldy) vs contiguous ones, plain vs non-temporal.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_rate.hip -o tools/micro/store_rate && ./tools/micro/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int ROWS>
__global__ __launch_bounds__(512) void k(float* __restrict__ y, long long ldy_floats, int rounds, int gap, long long* __restrict__ out,
                                         int tiles_n) {
  __shared__ unsigned char hold[100 * 1024];  // one workgroup per CU
  hold[threadIdx.x] = 0;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 v = {1.f * t, 2.f, 3.f, 4.f};
  long long burst = 0, first = 0, last = 0;
  for (int r = 0; r < rounds; ++r) {
    const long long tile = (long long)r * gridDim.x + blockIdx.x;
    const long long tm = tile / tiles_n, tn = tile % tiles_n;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (r == 0) first = t0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* dst;
        if (ROWS) {  // the epilogue's pattern: lane -> row 4q + (lane >> 4) of a 16-row block, columns 4 (lane & 15) .. +3
          const long long m = tm * 256 + (wm * 4 + i) * 16 + 4 * q + (lane >> 4);
          dst = y + m * ldy_floats + tn * 128 + wn * 64 + (lane & 15) * 4;
        } else {     // 1 KB contiguous per instruction
          dst = y + (tile * 8 + wave) * (16 * 256) + (i * 4 + q) * 256 + lane * 4;
        }
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst)); else *reinterpret_cast<f32x4*>(dst) = v;
        v[0] += 1.f;
      }
    const long long t1 = __builtin_amdgcn_s_memtime();
    burst += t1 - t0;
    last = t1;
    for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(127);  // ~8 k cycles of "K loop" per unit of 128
  }
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 3] = burst; out[(blockIdx.x * 8 + wave) * 3 + 1] = first; out[(blockIdx.x * 8 + wave) * 3 + 2] = last; }
}

int main() {
  const long long M = 33540, C = 2048;  // the 512 -> 2048 conv3 output: 132 x 16 tiles of 256 x 128
  float* y; hipMalloc(&y, (M + 256) * C * 4);
  long long* out; hipMalloc(&out, 256 * 8 * 3 * 8);
  std::vector<long long> h(256 * 8 * 3);
  printf("%-10s %-6s %4s %6s %4s | %10s %10s %10s | %8s\n", "pattern", "nt", "G", "rounds", "gap", "ticks/burst", "B/tick/CU", "span ticks", "wall us");
  for (int rows = 1; rows >= 0; --rows)
    for (int nt = 0; nt < 2; ++nt)
      for (int G : {256, 32})
        for (int gap : {0, 8}) {
          const int rounds = 8;
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          float ms = 0;
          for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (rows && nt) hipLaunchKernelGGL((k<1, 1>), dim3(G), dim3(512), 0, 0, y, C, rounds, gap, out, 16);
            else if (rows) hipLaunchKernelGGL((k<0, 1>), dim3(G), dim3(512), 0, 0, y, C, rounds, gap, out, 16);
            else if (nt) hipLaunchKernelGGL((k<1, 0>), dim3(G), dim3(512), 0, 0, y, C, rounds, gap, out, 16);
            else hipLaunchKernelGGL((k<0, 0>), dim3(G), dim3(512), 0, 0, y, C, rounds, gap, out, 16);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
          }
          hipMemcpy(h.data(), out, G * 8 * 3 * 8, hipMemcpyDeviceToHost);
          double burst = 0; long long lo = 1LL << 62, hi = 0;
          for (int i = 0; i < G * 8; ++i) { burst += (double)h[i * 3] / rounds; lo = std::min(lo, h[i * 3 + 1]); hi = std::max(hi, h[i * 3 + 2]); }
          burst /= G * 8;
          printf("%-10s %-6s %4d %6d %4d | %10.0f %10.2f %10lld | %8.1f   (%.2f TB/s over the launch)\n", rows ? "rows" : "contig", nt ? "nt" : "plain", G, rounds, gap,
                 burst, 131072.0 / burst, hi - lo, ms * 1e3, (double)G * rounds * 131072 / (ms * 1e-3) / 1e12);
        }
  return 0;
}
