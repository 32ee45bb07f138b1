// Shared helpers for the gfx950 kernels of libonda_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "onda_hip.h"

// "f16x2" limb planes: the second limb is stored times LIMB2_SCALE (2^11) so that it keeps 11 bits down to 2^-28 of a tensor's
// maximum; the cross products then carry that factor and live in accumulators of their own.  A measurement build with
// -DONDA_LIMB2_SCALE=1.f stores it un-scaled (what a single-accumulator kernel would need; DESIGN.md section 7, "Next (0)").
#ifndef ONDA_LIMB2_SCALE
#define ONDA_LIMB2_SCALE 2048.f
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LIMB ROWS (round 5): how an operand of the pre-split convolutions lies in memory.  Row r -- a pixel of an activation, an
// output channel of a packed weight -- holds, for every block of 32 channels, the 32 FIRST limbs followed by the 32 SECOND
// limbs: 128 bytes = one cache line per (row, K-step of 32 channels).  Until round 4 the two limbs were separate planes
// ([2][rows][ld]); a K-step then fetched two HALF lines per row, and tools/micro/dma_rate.hip shows what that costs: the
// convolutions' LDS-DMA stream alone (no MFMA, no fragment read) takes as long as the whole 1024 -> 256 1x1 kernel, bound
// by cache-line REQUESTS per CU, and runs 22-30 % faster with the same bytes as whole lines.
//   f16 index of (row r, channel c, first limb) = r * 2 * ld + (c / 32) * 64 + c % 32;  second limb: + LIMB2_OFS
// ld = channels per row (a multiple of 32); a row is 2 * ld f16 = 4 * ld bytes, like an fp32 row.
constexpr int LIMB2_OFS = 32;
__host__ __device__ __forceinline__ size_t limb_at(size_t row, int c, int ld) {
  return row * 2 * (size_t)ld + (size_t)((c >> 5) << 6) + (size_t)(c & 31);
}

#define ONDA_STREAM(s) (reinterpret_cast<hipStream_t>(s))
// Argument check at the top of every entry point.  It also drops any stale error another
// library left in this thread's HIP error slot, so that the hipGetLastError() after our own
// launches reports our launches only.
#define ONDA_REQUIRE(cond)                                                                              \
  do {                                                                                                  \
    (void)hipGetLastError();                                                                            \
    if (!(cond)) {                                                                                      \
      if (getenv("ONDA_DEBUG_REQUIRE")) fprintf(stderr, "onda_hip: %s:%d: requirement failed: %s\n", __FILE__, __LINE__, #cond); \
      return ONDA_EINVAL;                                                                               \
    }                                                                                                   \
  } while (0)
#define ONDA_ALIGNED16(p) ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)
#define ONDA_LAUNCH_RESULT() static_cast<int>(hipGetLastError())

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Running max|x| of a tensor for the "f16x2" convolutions.  The value lives in ONDA_AMAX_FLOATS device floats
// (all zero before the producer runs), used as ONDA_AMAX_SLOTS slots one 128-byte line apart: L2 serialises
// atomics per line, so the slots are what lets thousands of workgroups fold their maxima without queueing.
// A producer does one atomicMax on the bit pattern (monotonic for non-negative floats) per workgroup
// (amax_update_block) or per wave (amax_update); a consumer takes the maximum over the slots (amax_read).
// max is order-independent: the result is deterministic.  Every lane / thread must call these.
constexpr int AMAX_STRIDE = ONDA_AMAX_FLOATS / ONDA_AMAX_SLOTS;
__device__ __forceinline__ void amax_update(float* amax, float m) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f)
    atomicMax(reinterpret_cast<unsigned*>(amax) + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE,
              __float_as_uint(m));
}
// one atomic per 256-thread workgroup; `red`: 4 floats of LDS
__device__ __forceinline__ void amax_update_block(float* amax, float m, float* red) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f)
      atomicMax(reinterpret_cast<unsigned*>(amax) + (blockIdx.x & (ONDA_AMAX_SLOTS - 1)) * AMAX_STRIDE, __float_as_uint(m));
  }
}
__device__ __forceinline__ float amax_read(const float* __restrict__ amax) {
  static_assert(ONDA_AMAX_SLOTS == 64, "one slot per lane");
  return wave_max(amax[(threadIdx.x & 63) * AMAX_STRIDE]);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over the 256 threads of a block; result valid in thread 0. `red` holds >= 4 floats.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
