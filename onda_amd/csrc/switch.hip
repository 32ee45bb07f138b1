// The hybrid switch on the device: the one monitored series the static / dynamic decision reads ("prior static",
// framework/utils/monitoring.py:7-96) and the two-state machine that reads it (prototypes_hybrid_switch.py:22-34), as ONE
// small launch per adaptation step -- the decision never visits the host, so a step has no blocking read-back (and with
// several ranks no host in the path of the switch scalars' all-reduce).  The result is a device flag: the dynamic model's
// convolutions are launched with it as their predicate (OndaConv::run_if) and the prior is picked by select_prior_kernel.
//
// Bit-exactness.  The reference computes these statistics in float64 on the host with numpy; the arithmetic here repeats
// numpy's ORDER of operations, so the trajectories of fixture G5 are reproduced to the last bit:
//   * avg     = np.median(window): the middle element, or (lo + hi) / 2 of the two middle ones;
//   * exp     = (1 - c) * exp + c * v, products rounded before the sum (no fused multiply-add);
//   * dev_avg = level(w[1:]) - level(w[:-1]) once the window is full, level(w) = np.sum(taps * w) / np.sum(taps) with
//               numpy's pairwise summation (np_pairwise_sum below: 8 running partial sums over blocks of at most 128);
//               taps and np.sum(taps) come from the caller (np.hamming(limit - 1), or ones for "mean"), "median" levels are
//               medians of the two sub-windows.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_WINDOW = 1024;

// Float64 operations that must round exactly once each, as numpy's do.  Defined HERE, under this file's
// `fp contract(off)`: HIP's __dadd_rn / __dmul_rn are plain inline `a + b` / `a * b` from a header compiled with the
// default contraction mode, and after inlining the compiler fused them into FMAs (measured: the exponential average
// differed from numpy's in the last bit).
__device__ __forceinline__ double dadd(double a, double b) { return a + b; }
__device__ __forceinline__ double dsub(double a, double b) { return a - b; }
__device__ __forceinline__ double dmul(double a, double b) { return a * b; }
__device__ __forceinline__ double ddiv(double a, double b) { return a / b; }

// numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum, on products a[i] * b[i] (np.sum(taps * w): the product
// array is rounded element by element first).  Runs in one thread: 2 x 199 terms per step.
__device__ double np_pairwise_sum(const double* a, const double* b, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res = dadd(res, dmul(a[i], b[i]));
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = dmul(a[j], b[j]);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] = dadd(r[j], dmul(a[i + j], b[i + j]));
    double res = dadd(dadd(dadd(r[0], r[1]), dadd(r[2], r[3])), dadd(dadd(r[4], r[5]), dadd(r[6], r[7])));
    for (; i < n; ++i) res = dadd(res, dmul(a[i], b[i]));
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return dadd(np_pairwise_sum(a, b, n2), np_pairwise_sum(a + n2, b + n2, n - n2));
}

// sorted[] <- w[0..n) ascending, by rank (ties keep their order): every thread places its own elements
__device__ void rank_sort(const double* w, double* sorted, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = w[i];
    const bool vnan = v != v;
    int rank = 0;
    // a total order, so that every slot of sorted[] is written: NaN samples (a diverged model) sort behind everything
    for (int j = 0; j < n; ++j) {
      const double u = w[j];
      const bool unan = u != u;
      const bool less = unan ? false : (vnan ? true : u < v);
      const bool same = unan ? vnan : (!vnan && u == v);
      rank += less || (same && j < i);
    }
    sorted[rank] = v;
  }
  __syncthreads();
}
__device__ double median_of_sorted(const double* s, int n) {
  return (n & 1) ? s[n / 2] : ddiv(dadd(s[n / 2 - 1], s[n / 2]), 2.0);
}

// state (doubles): [0] exp, [1] avg, [2] dev_avg, [3] the confidence the switch looked at, [8 ..) the ring
// istate (int32): [0] count, [1] head, [2] current, [3] current_dev, [4] steps taken; flag[0] = current (the predicate)
__global__ __launch_bounds__(256) void switch_step_kernel(double* __restrict__ state, int* __restrict__ istate, const void* sample,
                                                          int sample_f64, const double* __restrict__ taps, OndaSwitchCfg cfg,
                                                          int* __restrict__ flag) {
  __shared__ double w[MAX_WINDOW], sorted[MAX_WINDOW];
  __shared__ double levels[2];
  __shared__ int meta[2];
  double* ring = state + 8;
  const int t = threadIdx.x, limit = cfg.limit;
  if (t == 0) {
    const double v = sample_f64 ? *static_cast<const double*>(sample) : (double)*static_cast<const float*>(sample);
    int count = istate[0], head = istate[1];
    if (count == 0) {
      ring[0] = v;
      state[0] = v;
      count = 1;
      head = 1 % limit;
    } else {
      ring[head] = v;
      head = (head + 1) % limit;
      count = count < limit ? count + 1 : limit;
      state[0] = dadd(dmul(cfg.one_minus_exp_const, state[0]), dmul(cfg.exp_const, v));
    }
    istate[0] = meta[0] = count;
    istate[1] = meta[1] = head;
    __threadfence_block();
  }
  __syncthreads();
  const int count = meta[0], head = meta[1];
  // the window, oldest first
  for (int i = t; i < count; i += blockDim.x) w[i] = count < limit ? ring[i] : ring[(head + i) % limit];
  __syncthreads();
  rank_sort(w, sorted, count);
  const double avg = median_of_sorted(sorted, count);
  __syncthreads();
  double dev = 0.0;
  if (count >= limit && limit > 1) {  // (uniform)
    const int n = limit - 1;
    if (cfg.level_kind == 1) {  // "median": medians of the window without its oldest / without its newest sample
      rank_sort(w + 1, sorted, n);
      const double a = median_of_sorted(sorted, n);
      __syncthreads();
      rank_sort(w, sorted, n);
      dev = dsub(a, median_of_sorted(sorted, n));
    } else {  // weighted level ("hamming", or "mean" with taps of one): one thread per level, in two waves
      if (t == 0) levels[0] = ddiv(np_pairwise_sum(taps, w + 1, n), cfg.taps_total);
      if (t == 64) levels[1] = ddiv(np_pairwise_sum(taps, w, n), cfg.taps_total);
      __syncthreads();
      dev = dsub(levels[0], levels[1]);
    }
  }
  if (t == 0) {
    const double conf = cfg.use_exp ? state[0] : avg;
    int current = istate[2], current_dev = istate[3];
    // a significant trend is remembered (static = 0); the reference's two one-sided tests, in its order
    // (prototypes_hybrid_switch.py:24-27: they differ from |dev| > thr for a negative threshold)
    if (dev > cfg.dev_threshold) current_dev = 0;
    else if (dev < -cfg.dev_threshold) current_dev = 1;
    if (conf < cfg.gray_lo) current = 1;
    else if (conf > cfg.gray_hi) current = 0;
    else current = current_dev;
    state[1] = avg;
    state[2] = dev;
    state[3] = conf;
    istate[2] = current;
    istate[3] = current_dev;
    istate[4] += 1;
    flag[0] = current;
  }
}

// out[i] = flag ? wb * b[i] : wa * a[i]  (a true select: `b` holds garbage when the predicated forward did not run)
__global__ __launch_bounds__(256) void select_prior_kernel(const int* __restrict__ flag, const float* __restrict__ a, float wa,
                                                           const float* __restrict__ b, float wb, float* __restrict__ out,
                                                           long long n) {
  const bool pick_b = flag[0] != 0;
  const float* src = pick_b ? b : a;
  const float wgt = pick_b ? wb : wa;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = wgt * src[i];
}

// out[0] = flag ? v[0] : NaN  (a monitor sample that exists only on one side of the switch)
__global__ void gate_scalar_kernel(const int* __restrict__ flag, const float* __restrict__ v, float* __restrict__ out) {
  out[0] = flag[0] != 0 ? v[0] : __builtin_nanf("");
}

}  // namespace

extern "C" {

int onda_switch_state_doubles(int limit) { return 8 + limit; }
int onda_switch_max_window(void) { return MAX_WINDOW; }

int onda_switch_step(double* state, int32_t* istate, const void* sample, int sample_f64, const double* taps, const OndaSwitchCfg* cfg,
                     int32_t* flag, onda_stream_t s) {
  ONDA_REQUIRE(state && istate && sample && cfg && flag && cfg->limit >= 1 && cfg->limit <= MAX_WINDOW);
  ONDA_REQUIRE(cfg->level_kind == 1 || taps != nullptr || cfg->limit == 1);
  hipLaunchKernelGGL(switch_step_kernel, dim3(1), dim3(256), 0, ONDA_STREAM(s), state, istate, sample, sample_f64, taps, *cfg, flag);
  return ONDA_LAUNCH_RESULT();
}

int onda_select_prior(const int32_t* flag, const float* a, float wa, const float* b, float wb, float* out, int64_t n,
                      onda_stream_t s) {
  ONDA_REQUIRE(flag && a && b && out && n > 0);
  const int blocks = (int)(n / 1024 + 1 > 1024 ? 1024 : n / 1024 + 1);
  hipLaunchKernelGGL(select_prior_kernel, dim3(blocks), dim3(256), 0, ONDA_STREAM(s), flag, a, wa, b, wb, out, (long long)n);
  return ONDA_LAUNCH_RESULT();
}

int onda_gate_scalar(const int32_t* flag, const float* v, float* out, onda_stream_t s) {
  ONDA_REQUIRE(flag && v && out);
  hipLaunchKernelGGL(gate_scalar_kernel, dim3(1), dim3(1), 0, ONDA_STREAM(s), flag, v, out);
  return ONDA_LAUNCH_RESULT();
}

}  // extern "C"
