// Stand-alone bilinear sampler with analytic Jacobian: jacobian.py:138-205.
// (The LM loop uses the fused kernel in lm_solve.hip; this is the operator-level entry point that
// mirrors the reference's `grid_sample(image, optical, jac)` for callers that want the maps.)
#include "common.h"

__global__ __launch_bounds__(256) void grid_sample_kernel(const float* __restrict__ img, const float* __restrict__ opt,
                                                          const float* __restrict__ jac, float* __restrict__ out,
                                                          float* __restrict__ jout, int N, int C, int IH, int IW,
                                                          int HW, int M) {
  const size_t total = (size_t)N * HW * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const size_t pix = e / C;              // n*HW + p
    const int n = (int)(pix / HW);
    const float ix = opt[pix * 2 + 0], iy = opt[pix * 2 + 1];
    const float lx = (float)(IW - 1), ly = (float)(IH - 1);
    const bool inb = (ix >= 0.f) && (ix <= lx) && (iy >= 0.f) && (iy <= ly);
    float v = 0.f, ddx = 0.f, ddy = 0.f;
    if (inb) {
      const float x0 = floorf(ix), y0 = floorf(iy);
      const float x1 = fminf(x0 + 1.f, lx), y1 = fminf(y0 + 1.f, ly);
      const float wx0 = x1 - ix, wx1 = ix - x0, wy0 = y1 - iy, wy1 = iy - y0;
      const float* b = img + (size_t)n * IH * IW * C + c;
      const float nw = b[((size_t)y0 * IW + (size_t)x0) * C], ne = b[((size_t)y0 * IW + (size_t)x1) * C];
      const float sw = b[((size_t)y1 * IW + (size_t)x0) * C], se = b[((size_t)y1 * IW + (size_t)x1) * C];
      v = nw * (wx0 * wy0) + ne * (wx1 * wy0) + sw * (wx0 * wy1) + se * (wx1 * wy1);
      ddx = -wy0 * nw + wy0 * ne - wy1 * sw + wy1 * se;
      ddy = -wx0 * nw - wx1 * ne + wx0 * sw + wx1 * se;
    }
    out[e] = v;
    if (jout) {
      for (int m = 0; m < M; ++m) {
        const float* j = jac + ((size_t)m * N * HW + pix) * 2;
        jout[(size_t)m * total + e] = ddx * j[0] + ddy * j[1];
      }
    }
  }
}

extern "C" int hla_grid_sample(const float* image, const float* optical, const float* jac, float* out, float* jac_out,
                               int N, int C, int IH, int IW, int H, int W, int M, hla_stream_t stream) {
  HLA_REQUIRE(image && optical && out, "hla_grid_sample: null argument");
  HLA_REQUIRE(N > 0 && C > 0 && IH > 0 && IW > 0 && H > 0 && W > 0, "hla_grid_sample: bad sizes");
  HLA_REQUIRE((jac == nullptr) == (jac_out == nullptr), "hla_grid_sample: jac and jac_out go together");
  const size_t total = (size_t)N * H * W * C;
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(grid_sample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, image, optical, jac, out,
                     jac_out, N, C, IH, IW, H * W, jac ? M : 0);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}
