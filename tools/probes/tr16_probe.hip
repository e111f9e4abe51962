// What does ds_read_b64_tr_b16 return?  LDS holds lds[i] = i (16-bit); every lane passes its own address.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(int mode, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int elem;   // element (16-bit) index this lane points at
  if (mode == 0) elem = 0;                         // uniform address
  else if (mode == 1) elem = l * 4;                // lane-linear 8 B each
  else if (mode == 2) elem = (l & 15) * 64 + (l >> 4) * 4;   // 16 rows of 64 elems; lane group picks a 4-col segment
  else elem = (l >> 2 & 3) * 64 + (l & 3) * 4 + (l >> 4) * 16;  // group: 4 rows x 16 cols, lane t -> row t>>2, colseg t&3
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + elem));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  lane %2d: %5d %5d %5d %5d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 4 == 3) ? "\n" : " |"); }
  }
  return 0;
}
