// VERDICT r05 #5, the gate of a Winograd F(2x2, 3x3) main loop for the split-fp16 mode, measured as an UPPER BOUND: what MFMA rate
// does the matrix pipe reach when it is fed the way such a loop must feed it on this chip?  (timing only: no correct results)
//   per 16-channel stage a workgroup (8 waves, 1 per CU: 64 accumulator tiles of 32 x 32 = half the CU's registers) multiplies the 16
//   transform positions' [128 couts x 16 cin] weight blocks (hi, lo fp16 fragments, 131 KB per stage, streamed L2 -> registers: they do
//   not fit LDS next to the data and every wave needs its own) with the positions' [16 cin x 32 tiles] transformed-input blocks
//   (hi, lo fragments from LDS): 16 x 4 x 3 = 192 MFMAs per stage for 4 rows x 32 output pixels x 128 channels, where the direct
//   split kernel issues 9 x 4 x 4 x 3 = 432 for 128 pixels x 128 channels -- the 2.25 x.
//   mode 0: the weight stream + LDS fragment reads only;  mode 1: + per stage the transformed tile written to LDS (64 KB: 16 positions
//   x 32 tiles x 16 cin x (hi + lo)) and 16 VALU operations per transformed value pair standing in for the input transform and the split.
// effective direct-conv TFLOP/s = raw MFMA rate / 3 x 2.25.   hipcc --offload-arch=gfx950 -O3 -o wino_probe wino_stream_probe.hip && ./wino_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void wino_stream(const uint4* __restrict__ wts, float* __restrict__ out, int nstage, int ntile, const float* __restrict__ src) {
  extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 x 32 KB of B fragments: [buf][pos 16][hl 2][lane 64] x 16 B
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), pg = w & 3, mh = w >> 2;
  f32x16 acc[4][2];
  for (int k = 0; k < 8; ++k) {      // B fragments: small normal fp16 values (random-operand power, as mfma_peak's data 1)
    unsigned r[4];
    for (int q = 0; q < 4; ++q) { unsigned x = (t * 8 + k) * 4 + q; x ^= x << 13; x ^= x >> 7; x ^= x << 17; x *= 2654435761u; r[q] = ((x & 0x83ffu) | 0x3800u) | ((((x >> 16) & 0x83ffu) | 0x3800u) << 16); }
    *(uint4*)(lds + (k * 512 + t) * 16) = make_uint4(r[0], r[1], r[2], r[3]);
  }
  __syncthreads();
  for (int tile = 0; tile < ntile; ++tile) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][m][r] = 0.f;
    // weights: [(blockIdx.y half)][stage][pos 16][mtile 4][hl 2][lane] 16 B
    const uint4* wq = wts + (size_t)blockIdx.y * nstage * 16 * 4 * 2 * 64 + lane;
    uint4 a[2][2][2];      // [buffer][m][hl]
    auto loadA = [&](int buf, int sg, int p) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) a[buf][m][hl] = wq[((((size_t)sg * 16 + pg * 4 + p) * 4 + mh * 2 + m) * 2 + hl) * 64];
    };
    loadA(0, 0, 0);
    for (int sg = 0; sg < nstage; ++sg) {
      char* cur = lds + (sg & 1) * 32768;
      if (MODE == 1) {      // stand-in for the input transform + split + LDS write of the NEXT stage's tile: 64 KB by 512 threads
        char* nxt = lds + ((sg + 1) & 1) * 32768;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(t * 8 + k + sg * 4096) & 65535];
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
          // 16 values in, 16 out: ~32 adds + 16 x 2 split instructions per 16 values
          float u[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) u[k] = (v[k] - v[(k + 2) & 7]) + (v[(k + 1) & 7] - v[(k + 3) & 7]);
          _Float16 h[8], l[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { h[k] = (_Float16)(u[k] * 1024.f); l[k] = (_Float16)(u[k] * 1024.f - (float)h[k]); }
          *(uint4*)(nxt + ((rep * 2 + 0) * 512 + t) * 16 % 32768) = __builtin_bit_cast(uint4, h);
          *(uint4*)(nxt + ((rep * 2 + 1) * 512 + t) * 16 % 32768) = __builtin_bit_cast(uint4, l);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = u[k] + 1.f;
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        // next position's (or next stage's first) weight fragments in flight under this position's MFMAs
        if (p < 3) loadA((p + 1) & 1, sg, p + 1);
        else if (sg + 1 < nstage) loadA(0, sg + 1, 0);
        const uint4 bh = *(const uint4*)(cur + (((pg * 4 + p) * 2 + 0) * 64 + lane) * 16);
        const uint4 bl = *(const uint4*)(cur + (((pg * 4 + p) * 2 + 1) * 64 + lane) * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          MMA(acc[p][m], a[p & 1][m][0], bh);
          MMA(acc[p][m], a[p & 1][m][1], bh);
          MMA(acc[p][m], a[p & 1][m][0], bl);
        }
      }
      __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][m][r];
    if (s == 12345.678f) out[blockIdx.x * 512 + t] = s;
  }
}

int main() {
  const int nstage = 16, ntile = 16, cus = 256;      // conv12: 256 cin = 16 stages of 16; 2 cout blocks of 128
  const size_t wbytes = (size_t)2 * nstage * 16 * 4 * 2 * 64 * 16;      // 4.2 MB: both cout blocks
  uint4* w; float* out; float* src;
  hipMalloc(&w, wbytes); hipMalloc(&out, (size_t)cus * 2 * 512 * 4); hipMalloc(&src, 65536 * 4);
  unsigned* h = (unsigned*)malloc(wbytes);
  unsigned long long x = 88172645463325252ull;
  for (size_t i = 0; i < wbytes / 4; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; unsigned lo = ((unsigned)x & 0x83ffu) | 0x3800u, hi = ((unsigned)(x >> 16) & 0x83ffu) | 0x3800u; h[i] = lo | (hi << 16); }
  hipMemcpy(w, h, wbytes, hipMemcpyHostToDevice);
  hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)wino_stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)wino_stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) {
        if (mode == 0) hipLaunchKernelGGL(wino_stream<0>, dim3(cus / 2, 2), dim3(512), 65536, 0, w, out, nstage, ntile, src);
        else hipLaunchKernelGGL(wino_stream<1>, dim3(cus / 2, 2), dim3(512), 65536, 0, w, out, nstage, ntile, src);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfma = 10.0 * cus * ntile * nstage * 192.0;
      const double raw = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
      printf("mode %d: %.3f ms per launch, raw MFMA %.0f TFLOP/s, effective direct-conv rate %.0f TFLOP/s (x 2.25 / 3); weight stream %.2f TB/s L2 -> CUs\n",
             mode, ms / 10, raw, raw / 3 * 2.25, 10.0 * cus * ntile * nstage * 131072.0 / (ms * 1e-3) / 1e12);
    }
  return hipGetLastError() != hipSuccess;
}
