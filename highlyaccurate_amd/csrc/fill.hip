// hla_zero_fill: several (strided) regions of device memory cleared by ONE launch (include/hla.h).
#include "common.h"

struct FillTable { hla_fill_region r[16]; };

static __global__ __launch_bounds__(256) void zero_fill_kernel(FillTable t) {
  const hla_fill_region& r = t.r[blockIdx.y];
  if ((((size_t)r.ptr | r.chunk_bytes | r.stride_bytes) & 15) == 0) {      // (per region: uniform)
    const size_t v_per = r.chunk_bytes / 16, total = v_per * (size_t)r.n_chunks;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
      const size_t c = i / v_per, o = i - c * v_per;
      *(uint4*)((char*)r.ptr + c * r.stride_bytes + o * 16) = z;
    }
  } else {      // a map whose size is not a multiple of four floats (odd level sizes: [B,h,w] confidence gradients): word stores
    const size_t v_per = r.chunk_bytes / 4, total = v_per * (size_t)r.n_chunks;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
      const size_t c = i / v_per, o = i - c * v_per;
      *(unsigned*)((char*)r.ptr + c * r.stride_bytes + o * 4) = 0u;
    }
  }
}

extern "C" int hla_zero_fill(const hla_fill_region* regions, int n_regions, int max_blocks, hla_stream_t stream) {
  HLA_REQUIRE(n_regions >= 0 && n_regions <= 16, "hla_zero_fill: at most 16 regions");
  if (n_regions == 0) return HLA_OK;
  HLA_REQUIRE(regions, "hla_zero_fill: null argument");
  FillTable t{};
  size_t most = 0;
  for (int i = 0; i < n_regions; ++i) {
    const hla_fill_region& r = regions[i];
    HLA_REQUIRE(r.n_chunks >= 0 && (r.n_chunks == 0 || r.ptr), "hla_zero_fill: region %d: null pointer", i);
    HLA_REQUIRE(((size_t)r.ptr | r.chunk_bytes | r.stride_bytes) % 4 == 0, "hla_zero_fill: region %d: pointer, chunk and stride must be multiples of 4 bytes", i);
    t.r[i] = r;
    const size_t v = r.chunk_bytes / 16 * (size_t)r.n_chunks + 1;
    most = v > most ? v : most;
  }
  if (most == 0) return HLA_OK;
  // (enough blocks for the largest region to keep every CU's store queue busy; small regions finish in their first blocks)
  size_t gx = (most + 256 * 16 - 1) / (256 * 16);
  gx = gx > 2048 ? 2048 : (gx < 1 ? 1 : gx);
  if (max_blocks > 0 && gx > (size_t)max_blocks) gx = (size_t)max_blocks;
  hla_prof_begin(K_ELEMWISE, 0, 0, (hipStream_t)stream);
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)gx, n_regions), dim3(256), 0, (hipStream_t)stream, t);
  hla_prof_end((hipStream_t)stream);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}
