// Pins the LDS-DMA semantics the conv halo loader relies on (gfx950):  buffer_load_dwordx4 ... offen lds
//   * lane L's 16 bytes land at LDS address M0 + 16 L (wave-uniform base, lane-linear image);
//   * a lane whose offset is out of the buffer's range writes ZEROS (the descriptor's range check), so the zero border of a
//     convolution needs no separate fill;
//   * completion is counted on vmcnt.
// hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe lds_dma_probe.hip && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned* src, int nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2 * 256];
  const int t = threadIdx.x;
  for (int i = t; i < 512; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  i32x4 rsrc;
  const unsigned long long base = (unsigned long long)src;
  rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
  rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));      // stride 0
  rsrc[2] = __builtin_amdgcn_readfirstlane(nbytes);
  rsrc[3] = 0x00020000;
  // lanes 0..15 fetch 16-B piece (15 - lane) (a permuted source: the LDS image stays lane-linear); 16..63 are out of range
  const unsigned voff = t < 16 ? (unsigned)(15 - t) * 16u : 0x80000000u;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds + 256));    // second KiB
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(dst) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = t; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  unsigned h[64], *d, *o, r[512];
  for (int i = 0; i < 64; ++i) h[i] = 0x1000 + i;
  hipMalloc(&d, 256); hipMalloc(&o, 2048);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 256, o);
  hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 256; ++i) ok &= r[i] == 0xdeadbeefu;                       // first KiB untouched
  for (int L = 0; L < 64; ++L)
    for (int k = 0; k < 4; ++k) {
      const unsigned want = L < 16 ? 0x1000u + (15 - L) * 4 + k : 0u;
      if (r[256 + L * 4 + k] != want) { ok = 0; printf("lane %d word %d: got %08x want %08x\n", L, k, r[256 + L * 4 + k], want); }
    }
  printf(ok ? "lds_dma_probe OK: lane-linear image at M0, out-of-range lanes wrote zeros\n" : "lds_dma_probe FAILED\n");
  return !ok;
}
