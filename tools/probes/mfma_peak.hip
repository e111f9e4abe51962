// What does the matrix pipe of THIS chip sustain?  Back-to-back v_mfma_f32_32x32x16_{bf16,f16} with register-resident operands:
// no LDS, no memory, 2 waves per SIMD (the conv kernels' occupancy), every CU busy for milliseconds -- the ceiling any
// MFMA-bound kernel of libhla can reach at the clock the chip holds under that load.  Operand data: zeros, or random bits
// (power, hence clock, depends on toggling).   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int F16>
__global__ __launch_bounds__(256, 2) void burn(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = threadIdx.x;
  uint4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 4095]; b[i] = src[(t * 8 + 4 + i) & 4095]; }
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (F16)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 3]), __builtin_bit_cast(f16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
      else
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[blockIdx.x * 256 + t] = s;      // (keeps the loop alive)
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, grid = cus * 2, iters = 40000;
  uint4* src; float* out;
  hipMalloc(&src, 4096 * sizeof(uint4));
  hipMalloc(&out, (size_t)grid * 256 * 4);
  uint4* h = (uint4*)malloc(4096 * sizeof(uint4));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("device %s, %d CUs, clock %d MHz (nominal)\n", p.name, cus, p.clockRate / 1000);
  for (int mode = 0; mode < 3; ++mode) {          // 0: zeros, 1: small random bf16/fp16 values, 2: post-ReLU-like (half zeros)
    srand(1);
    for (int i = 0; i < 4096 * 4; ++i) {
      unsigned lo = (rand() & 0x7fff) | ((rand() & 1) << 15), hi = (rand() & 0x7fff) | ((rand() & 1) << 15);
      lo = (lo & 0x83ff) | 0x3800; hi = (hi & 0x83ff) | 0x3800;          // magnitudes ~0.5..1 in fp16, small normal numbers in bf16
      unsigned w = mode == 0 ? 0u : (lo | (hi << 16));
      if (mode == 2 && (rand() & 1)) w = 0u;
      ((unsigned*)h)[i] = w;
    }
    hipMemcpy(src, h, 4096 * sizeof(uint4), hipMemcpyHostToDevice);
    for (int f16 = 0; f16 < 2; ++f16) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (f16) hipLaunchKernelGGL(burn<1>, dim3(grid), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(burn<0>, dim3(grid), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
        if (rep == 2)
          printf("%s data %-22s: %8.3f ms  %7.1f TFLOP/s  = %.3f of 2500; implied MFMA clock %.0f MHz\n", f16 ? "f16 " : "bf16",
                 mode == 0 ? "zeros" : mode == 1 ? "random" : "random, half zeros", ms, flops / ms / 1e9,
                 flops / ms / 1e9 / 2500.0, flops / ms / 1e9 / 2500.0 * 2400.0);
      }
    }
  }
  return 0;
}
