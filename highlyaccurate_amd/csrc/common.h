// Shared helpers for libhla (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/hla.h"

void hla_set_error(const char* fmt, ...);

#define HLA_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      hla_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return HLA_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define HLA_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      hla_set_error(__VA_ARGS__);   \
      return HLA_ERR_ARG;           \
    }                               \
  } while (0)

static inline size_t hla_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

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
