// Probe (round 6): the split-K reduction of the weight gradients (reduce_partials4_kernel) against variants with more k-lanes /
// more loads in flight.  hipcc --offload-arch=gfx950 -O3 -o reduce_probe reduce_probe.hip && ./reduce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// shipped form: 32 float4 columns x 8 k-lanes per block, unroll 4
__global__ __launch_bounds__(256) void red_a(const float4* __restrict__ part, float4* __restrict__ out, size_t n4, int K) {
  __shared__ float4 sh[8][32];
  const int col = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + col;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
#pragma unroll 4
    for (int k = kl; k < K; k += 8) { const float4 v = part[(size_t)k * n4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  }
  sh[kl][col] = s;
  __syncthreads();
  if (kl == 0 && i < n4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) { const float4 v = sh[k][col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    out[i] = s;
  }
}

// KL k-lanes x (256 / KL) columns per block; every thread issues up to U loads before it adds
template <int KL, int U>
__global__ __launch_bounds__(256) void red_b(const float4* __restrict__ part, float4* __restrict__ out, size_t n4, int K) {
  constexpr int COLS = 256 / KL;
  __shared__ float4 sh[KL][COLS];
  const int col = threadIdx.x % COLS, kl = threadIdx.x / COLS;
  const size_t i = (size_t)blockIdx.x * COLS + col;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    int k = kl;
    for (; k + (U - 1) * KL < K; k += U * KL) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = part[(size_t)(k + u * KL) * n4 + i];
#pragma unroll
      for (int u = 0; u < U; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < K; k += KL) { const float4 v = part[(size_t)k * n4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  }
  sh[kl][col] = s;
  __syncthreads();
  if (kl == 0 && i < n4) {
#pragma unroll
    for (int k = 1; k < KL; ++k) { const float4 v = sh[k][col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    out[i] = s;
  }
}

template <typename F> float time_it(F f, int it = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < it; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / it;
}

int main() {
  struct Case { const char* name; int cout, cin, ks; } cases[] = {
    {"64x64", 64, 64, 256}, {"128x64", 128, 64, 128}, {"64x192", 64, 192, 85}, {"128x128", 128, 128, 64},
    {"128x384", 128, 384, 21}, {"256x128", 256, 128, 32}, {"256x256", 256, 256, 16}};
  float4 *part, *out, *flush;
  hipMalloc(&part, (size_t)40 << 20); hipMalloc(&out, (size_t)4 << 20); hipMalloc(&flush, (size_t)512 << 20);
  hipMemset(part, 0, (size_t)40 << 20);
  for (auto& c : cases) {
    const size_t n4 = (size_t)c.cout * c.cin * 9 / 4; const int K = c.ks;
    printf("%-8s n4 %7zu K %3d  %5.1f MB:", c.name, n4, K, n4 * 16.0 * K / 1e6);
    auto run = [&](const char* nm, auto launch) {
      // cold-ish: rewrite the partials before every timed launch would need a writer; time back-to-back (L2/MALL-warm) AND after a flush
      float warm = time_it([&] { launch(); });
      float cold = 0; const int it = 8;
      for (int i = 0; i < it; ++i) {
        hipMemsetAsync(flush, i, (size_t)512 << 20);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); cold += ms * 1e3f / it;
      }
      printf("  %s %5.1f/%5.1f", nm, warm, cold);
    };
    run("A", [&] { hipLaunchKernelGGL(red_a, dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, 0, part, out, n4, K); });
    run("B8x8", [&] { hipLaunchKernelGGL((red_b<8, 8>), dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, 0, part, out, n4, K); });
    run("B16x8", [&] { hipLaunchKernelGGL((red_b<16, 8>), dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, 0, part, out, n4, K); });
    run("B32x8", [&] { hipLaunchKernelGGL((red_b<32, 8>), dim3((unsigned)((n4 + 7) / 8)), dim3(256), 0, 0, part, out, n4, K); });
    run("B4x8", [&] { hipLaunchKernelGGL((red_b<4, 8>), dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, 0, part, out, n4, K); });
    run("B2x8", [&] { hipLaunchKernelGGL((red_b<2, 8>), dim3((unsigned)((n4 + 127) / 128)), dim3(256), 0, 0, part, out, n4, K); });
    run("B1x16", [&] { hipLaunchKernelGGL((red_b<1, 16>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, part, out, n4, K); });
    printf("\n");
  }
  return 0;
}
