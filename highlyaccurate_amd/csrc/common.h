// Shared helpers for libhla (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/hla.h"

void hla_set_error(const char* fmt, ...);

// measurement hooks (prof.hip)
enum { K_PACK = 0, K_CONV02, K_CONV_NT2, K_CONV_NT2_POOL, K_CONV_NT1, K_CONV_NT1_POOL, K_CONF, K_L2NORM,
       K_LM256, K_LM128, K_LM64, K_LM16, K_LMSOLVE, K_GRIDSAMPLE, K_LMBWD, K_WGRAD, K_ELEMWISE };
void hla_prof_begin(int id, double flops, double bytes, hipStream_t st);
void hla_prof_begin_dyn(int id, double flops, double bytes, hipStream_t st, const int* dev_live, int denom);
void hla_prof_end(hipStream_t st);

// per-wave cycle stamps of ONE conv3x3_kernel launch (tooling builds only: python -m highlyaccurate_amd.build --out=libhla_stamps.so
// -DHLA_CONV_STAMPS=1; tools/probes/conv_stamps.py): launch_conv hands `buf` to the launch whose ordinal since the last
// hla_debug_conv_stamps call equals `want`; every wave writes HLA_STAMP_N 64-bit words at ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave).
#ifndef HLA_CONV_STAMPS
#define HLA_CONV_STAMPS 0
#endif
#if HLA_CONV_STAMPS
constexpr int HLA_STAMP_N = 8;
struct HlaStampCfg { unsigned long long* buf; int want; int counter; unsigned grid_x, grid_y; };
extern HlaStampCfg g_hla_stamp;      // prof.hip
#endif

#define HLA_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      hla_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return HLA_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

static inline bool hla_dtype_ok(int dtype) { return dtype == HLA_F32 || dtype == HLA_BF16 || dtype == HLA_F16 || dtype == HLA_F16X3; }

#define HLA_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      hla_set_error(__VA_ARGS__);   \
      return HLA_ERR_ARG;           \
    }                               \
  } while (0)

// hipFuncSetAttribute applies to the CURRENT device: remember per (call site, device) that it was done.  Thread-safe; two
// threads racing on a first call both set the attribute (idempotent) before either marks it done.
struct HlaPerDeviceOnce {
  std::atomic<unsigned long long> mask{0};
  template <typename F> hipError_t run(F&& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = (d >= 0 && d < 64) ? 1ull << d : 0ull;      // (device ids >= 64: never remembered, f() is idempotent)
    if (mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = f();
    if (e == hipSuccess) mask.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

static inline size_t hla_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// compute units of the current device (cached per device id < 64; 256 on MI355X)
static inline int hla_num_cus() {
  static std::atomic<int> cache[64];
  int d = 0;
  (void)hipGetDevice(&d);
  if (d >= 0 && d < 64) {
    const int c = cache[d].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
  if (d >= 0 && d < 64) cache[d].store(n, std::memory_order_relaxed);
  return n;
}

// 64-lane butterfly sum (all lanes end with the total)
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
