// What bounds the K loop of the pre-split convolutions?  Every one of them (long K, short K, the 128 x 128 producer / consumer
// variant) runs its K-step at 16-21 bytes of LDS-DMA per clock and CU, a third of the 64 B/clk of a CU's vector L1.  This
// probe issues the convolution's LDS-DMA stream and NOTHING else -- no fragment reads, no MFMA: 8 waves per workgroup, one
// workgroup per CU, a ring of three 48 KB stages, six 1 KB buffer_load ... lds instructions per wave and step, two steps in
// flight, one barrier per step -- for different SOURCE patterns of the same byte count:
//   rows64 : a piece = 16 rows x 64 B, rows `ld` bytes apart (the NHWC limb planes as they are: 32 channels of a pixel = HALF
//            a 128-byte cache line per row; the other half belongs to the next K-step)
//   rows128: a piece = 8 rows x 128 B (whole lines: what a 64-channel K-step, or a channel-blocked layout, would fetch)
//   blocked: a piece = 1 KB contiguous (rows of a 32-channel block adjacent in memory: [C/32][pixel][32])
// Operands sized like layer3's 1 x 1 convolutions (M = 33 540 rows, 1024 channels: 137 MB of limb planes; weights 256 x 1024),
// every workgroup walking its own row tile along K like the kernels do, so the L2 / Infinity-Cache / HBM mix is the real one.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_rate.hip -o tools/micro/dma_rate && ./tools/micro/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// MODE 0 rows64, 1 rows128, 2 blocked.  A: [2 limbs][M][C] f16 (plane apart), B: [2][N][K] f16.
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const _Float16* __restrict__ A, const _Float16* __restrict__ B, unsigned a_plane, unsigned b_plane,
                                            unsigned a_bytes, unsigned b_bytes, int M, int C, int N, int tiles_n, int tiles, long long* __restrict__ out, int lda) {
  constexpr int STAGE = 48 * 1024, PLANE_A = 256 * 64, PLANE_B = 128 * 64, A_BYTES = 2 * PLANE_A;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, a_bytes), rb = make_rsrc(B, b_bytes);
  const int KT = C / 32;
  long long t_begin = __builtin_amdgcn_s_memtime();
  int st_issue = 0, issued = 0, done = 0;
  int tile = blockIdx.x, kt_i = 0;
  constexpr int AM = MODE == 1 ? 1 : (MODE == 2 || MODE == 6) ? 2 : MODE == 5 ? -1 : 0;   // A pattern (-1: not loaded)
  constexpr int BP = MODE == 1 ? 1 : (MODE == 2 || MODE == 3) ? 2 : (MODE == 4 || MODE == 6) ? -1 : 0;  // B pattern
  auto issue = [&]() {
    if (tile >= tiles) return;
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 128;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {  // two 16-row blocks of A per wave
        const int blk = wave * 2 + d;
        unsigned off;
        if (AM < 0) {
          off = OOB;  // (dropped by the hardware: the instruction is issued and counted, nothing is fetched)
        } else if (AM == 0) {  // lane -> row lane >> 2 of the block, 16-byte chunk lane & 3 of its 64 bytes
          const int m = m0 + blk * 16 + (lane >> 2);
          off = m < M ? (unsigned)((m * lda + kt_i * 32) * 2 + (lane & 3) * 16) : OOB;
        } else if (AM == 1) {  // 8 rows x 128 B: the block's rows 0-7 (d-th half by the K-step's parity: same bytes per step)
          const int m = m0 + blk * 16 + (kt_i & 1) * 8 + (lane >> 3);
          off = m < M ? (unsigned)((m * lda + (kt_i >> 1) * 64) * 2 + (lane & 7) * 16) : OOB;
        } else {  // channel-blocked layout [C/32][M][32]: 16 rows of one block are 1 KB contiguous
          const int m = m0 + blk * 16;
          off = m + 15 < M ? (unsigned)(((kt_i * M + m) * 32) * 2 + lane * 16) : OOB;
        }
        unsigned char* dst = lds + st_issue + l * PLANE_A + blk * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)dst, 16, off, l * a_plane, 0, 0);
      }
      {  // one 16-row block of B per wave (weights: always L2-resident, row stride 2 K bytes)
        const int blk = wave;
        unsigned off;
        if (BP < 0) {
          off = OOB;
        } else if (BP == 0) {
          const int n = n0 + blk * 16 + (lane >> 2);
          off = n < N ? (unsigned)((n * C + kt_i * 32) * 2 + (lane & 3) * 16) : OOB;
        } else if (BP == 1) {
          const int n = n0 + blk * 16 + (kt_i & 1) * 8 + (lane >> 3);
          off = n < N ? (unsigned)((n * C + (kt_i >> 1) * 64) * 2 + (lane & 7) * 16) : OOB;
        } else {
          const int n = n0 + blk * 16;
          off = n + 15 < N ? (unsigned)(((kt_i * N + n) * 32) * 2 + lane * 16) : OOB;
        }
        unsigned char* dst = lds + st_issue + A_BYTES + l * PLANE_B + blk * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)dst, 16, off, l * b_plane, 0, 0);
      }
    }
    st_issue = st_issue + STAGE == 3 * STAGE ? 0 : st_issue + STAGE;
    ++issued;
    if (++kt_i == KT) {
      kt_i = 0;
      tile += gridDim.x;
    }
  };
  issue();
  issue();
  while (done < issued) {
    if (issued > done + 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ++done;
    issue();
  }
  const long long t_end = __builtin_amdgcn_s_memtime();
  if (t == 0) {
    out[blockIdx.x * 2] = t_end - t_begin;
    out[blockIdx.x * 2 + 1] = done;
  }
  if (lds[t] == 77 && t_end == 0) out[0] = 1;  // keep the ring alive
}

int main() {
  const int M = 33540;
  struct Shape { int C, N; } shapes[] = {{1024, 256}, {256, 1024}, {512, 2048}};
  long long* out; hipMalloc(&out, 256 * 2 * 8);
  std::vector<long long> h(512);
  const char* names[] = {"rows64", "rows128", "blocked", "A64+Bblk", "A64 only", "B64 only", "Ablk only"};
  printf("%-8s %5s %5s | %8s %10s %12s %10s\n", "pattern", "C", "N", "wall us", "steps/WG", "ticks/step", "B/tick/CU");
  for (auto s : shapes) {
    const int maxpad = 256;
    const size_t a_plane = (size_t)(M + 256) * (s.C + maxpad), b_plane = (size_t)s.N * s.C;  // f16 elements per limb plane
    _Float16 *A, *B;
    hipMalloc(&A, a_plane * 2 * 2);
    hipMalloc(&B, b_plane * 2 * 2);
    hipMemset(A, 0, a_plane * 4);
    hipMemset(B, 0, b_plane * 4);
    const int tiles_n = s.N / 128, tiles = ((M + 255) / 256) * tiles_n;
    for (int pass = 0; pass < 7 + 5; ++pass) {
      // passes 0-6: the seven patterns at the dense row stride; 7-11: rows64 (A and B) with the A rows padded by 32 .. 256 halves
      const int mode = pass < 7 ? pass : 0;
      const int pads[] = {32, 64, 96, 128, 256};
      const int lda = s.C + (pass < 7 ? 0 : pads[pass - 7]);
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const unsigned ab = (unsigned)(a_plane * 4), bb = (unsigned)(b_plane * 4);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, A, B, (unsigned)(a_plane * 2), (unsigned)(b_plane * 2), ab, bb, M, s.C, s.N, tiles_n, tiles, out, lda);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, A, B, (unsigned)(a_plane * 2), (unsigned)(b_plane * 2), ab, bb, M, s.C, s.N, tiles_n, tiles, out, lda);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, A, B, (unsigned)(a_plane * 2), (unsigned)(b_plane * 2), ab, bb, M, s.C, s.N, tiles_n, tiles, out, lda);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, A, B, (unsigned)(a_plane * 2), (unsigned)(b_plane * 2), ab, bb, M, s.C, s.N, tiles_n, tiles, out, lda);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, A, B, (unsigned)(a_plane * 2), (unsigned)(b_plane * 2), ab, bb, M, s.C, s.N, tiles_n, tiles, out, lda);
        if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, A, B, (unsigned)(a_plane * 2), (unsigned)(b_plane * 2), ab, bb, M, s.C, s.N, tiles_n, tiles, out, lda);
        if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, A, B, (unsigned)(a_plane * 2), (unsigned)(b_plane * 2), ab, bb, M, s.C, s.N, tiles_n, tiles, out, lda);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
      }
      hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost);
      std::vector<double> per;
      long long steps = 0;
      for (int b = 0; b < 256; ++b) { per.push_back((double)h[2 * b] / (double)h[2 * b + 1]); steps += h[2 * b + 1]; }
      std::sort(per.begin(), per.end());
      char label[32];
      if (pass < 7) snprintf(label, sizeof label, "%s", names[mode]); else snprintf(label, sizeof label, "r64+%dh", pads[pass - 7]);
      printf("%-8s %5d %5d | %8.1f %10.1f %12.0f %10.1f\n", label, s.C, s.N, best * 1e3,
             steps / 256.0, per[128], 48.0 * 1024 / per[128]);
    }
    hipFree(A); hipFree(B);
  }
  return 0;
}
