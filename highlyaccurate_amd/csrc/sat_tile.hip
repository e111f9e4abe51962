// Dataset-side satellite-tile geometry on the GPU: the reference prepares every sample's tile with a chain
// of Pillow calls on the CPU (KITTI_dataset.py:128-157 rotate -> AFFINE shift -> AFFINE random shift -> rotate -> centre
// crop -> ToTensor; Ford_dataset.py:185-209 the same with the first two swapped).  Every stage output pixel depends on at
// most 4 pixels of the stage before it, so the cropped result is evaluated LAZILY per output pixel (<= 16 source pixels)
// with Pillow's exact arithmetic at every stage -- 16.16 fixed point for the NEAREST rotations, double precision + uint8
// truncation for the BILINEAR shifts -- and is bit-identical to materialising the four intermediate images.
// HBM-bound and tiny: 3 source bytes x <= 16 taps in, 12 bytes out per pixel; no intermediate image is ever written.
#include "common.h"

struct TileStage {          // one resampling stage of one sample
  double kind;              // 0: nearest, 16.16 fixed point (coefficients are integers stored as doubles); 1: bilinear
  double c[6];
  double pad;
};

struct TileArgs {
  const unsigned char* src;   // [B,S,S,3] uint8
  const TileStage* st;        // [B,4]
  float* out;                 // [B,3,crop,crop] fp32 = value / 255 (ToTensor)
  int B, S, crop, top, nstage;
};

struct Rgb { int r, g, b; };

template <int K>
__device__ Rgb tile_eval(const unsigned char* img, const TileStage* st, int S, int x, int y) {
#pragma clang fp contract(off)      // Pillow's C code runs without FMA contraction: keep every multiply and add separate
  if constexpr (K == 0) {
    const unsigned char* p = img + ((size_t)y * S + x) * 3;
    return Rgb{p[0], p[1], p[2]};
  } else {
    const TileStage& s = st[K - 1];
    if (s.kind == 0.0) {
      const long long a0 = (long long)s.c[0], a1 = (long long)s.c[1], a2 = (long long)s.c[2];
      const long long a3 = (long long)s.c[3], a4 = (long long)s.c[4], a5 = (long long)s.c[5];
      const long long xi = (a2 + a1 * y + a0 * x) >> 16, yi = (a5 + a4 * y + a3 * x) >> 16;
      if (xi < 0 || xi >= S || yi < 0 || yi >= S) return Rgb{0, 0, 0};
      return tile_eval<K - 1>(img, st, S, (int)xi, (int)yi);
    }
    const double xx = x + 0.5, yy = y + 0.5;
    double xin = s.c[0] * xx + s.c[1] * yy + s.c[2];
    double yin = s.c[3] * xx + s.c[4] * yy + s.c[5];
    if (xin < 0.0 || xin >= (double)S || yin < 0.0 || yin >= (double)S) return Rgb{0, 0, 0};
    xin -= 0.5; yin -= 0.5;
    const double fx = floor(xin), fy = floor(yin);
    const int x0 = (int)fx, y0 = (int)fy;
    const double dx = xin - fx, dy = yin - fy;
    const int cx0 = min(max(x0, 0), S - 1), cx1 = min(max(x0 + 1, 0), S - 1);
    const int cy0 = min(max(y0, 0), S - 1), cy1 = min(max(y0 + 1, 0), S - 1);
    const bool below = (y0 + 1) >= 0 && (y0 + 1) < S;
    Rgb p[4];
#pragma unroll 1
    for (int t = 0; t < 4; ++t) p[t] = tile_eval<K - 1>(img, st, S, (t & 1) ? cx1 : cx0, (t & 2) ? cy1 : cy0);
    auto lerp2 = [&](int v00, int v01, int v10, int v11) {
#pragma clang fp contract(off)
      const double v1 = (double)v00 + ((double)v01 - (double)v00) * dx;
      const double v2 = below ? (double)v10 + ((double)v11 - (double)v10) * dx : v1;
      return (int)(unsigned char)(v1 + (v2 - v1) * dy);          // Pillow truncates
    };
    return Rgb{lerp2(p[0].r, p[1].r, p[2].r, p[3].r), lerp2(p[0].g, p[1].g, p[2].g, p[3].g), lerp2(p[0].b, p[1].b, p[2].b, p[3].b)};
  }
}

__global__ __launch_bounds__(256) void sat_tile_kernel(TileArgs a) {
  const size_t n = (size_t)a.B * a.crop * a.crop;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int xo = (int)(i % a.crop), yo = (int)((i / a.crop) % a.crop), b = (int)(i / ((size_t)a.crop * a.crop));
    const unsigned char* img = a.src + (size_t)b * a.S * a.S * 3;
    const TileStage* st = a.st + (size_t)b * 4;
    const Rgb v = tile_eval<4>(img, st, a.S, xo + a.top, yo + a.top);
    float* o = a.out + ((size_t)b * 3 * a.crop + yo) * a.crop + xo;
    const size_t plane = (size_t)a.crop * a.crop;
    o[0] = (float)v.r / 255.0f; o[plane] = (float)v.g / 255.0f; o[2 * plane] = (float)v.b / 255.0f;
  }
}

extern "C" int hla_sat_tile(const unsigned char* src, const double* stages, float* out, int B, int S, int crop,
                            hla_stream_t stream) {
  HLA_REQUIRE(src && stages && out, "hla_sat_tile: null argument");
  HLA_REQUIRE(B > 0 && S > 0 && S < 32768 && crop > 0 && crop <= S, "hla_sat_tile: bad sizes (B %d, S %d, crop %d)", B, S, crop);
  TileArgs a{};
  a.src = src; a.st = (const TileStage*)stages; a.out = out; a.B = B; a.S = S; a.crop = crop; a.nstage = 4;
  a.top = (int)nearbyint((S - crop) / 2.0);             // TF.center_crop: int(round((S - crop) / 2.0)), ties to even
  const size_t n = (size_t)B * crop * crop;
  const int grid = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
  hla_prof_begin(K_ELEMWISE, 0, (double)n * (3.0 * 16 + 12), (hipStream_t)stream);
  hipLaunchKernelGGL(sat_tile_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  hla_prof_end((hipStream_t)stream);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}

// ---------------------------------------------------------------------------------------------
// Ground image: torchvision Resize([256,1024]) + ToTensor (KITTI_dataset.py:300-311, Ford_dataset.py:141-155) =
// Pillow's antialiased bilinear resample: two separable passes with integer taps scaled by 2^22 and a uint8 intermediate.
struct ResampleArgs {
  const unsigned char* src;   // pass 1: [B,H,W,3]; pass 2: [B,H,ow,3] (the intermediate)
  unsigned char* mid;         // pass 1 output, or null
  float* out;                 // pass 2 output [B,3,oh,ow] = value / 255, or null
  const int* bounds;          // [n_out][2] first input index, tap count
  const int* taps;            // [n_out][ksize]
  int B, H, W, n_out, ksize;
};

__global__ __launch_bounds__(256) void resample_h_kernel(ResampleArgs a) {
  const size_t n = (size_t)a.B * a.H * a.n_out;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int xx = (int)(i % a.n_out);
    const size_t row = i / a.n_out;                       // b*H + y
    const int x0 = a.bounds[xx * 2], cnt = a.bounds[xx * 2 + 1];
    const unsigned char* p = a.src + (row * a.W + x0) * 3;
    long long s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int k = 0; k < cnt; ++k) {
      const long long w = a.taps[xx * a.ksize + k];
      s0 += w * p[k * 3]; s1 += w * p[k * 3 + 1]; s2 += w * p[k * 3 + 2];
    }
    unsigned char* o = a.mid + i * 3;
    o[0] = (unsigned char)min(max(s0 >> 22, 0LL), 255LL);
    o[1] = (unsigned char)min(max(s1 >> 22, 0LL), 255LL);
    o[2] = (unsigned char)min(max(s2 >> 22, 0LL), 255LL);
  }
}

__global__ __launch_bounds__(256) void resample_v_kernel(ResampleArgs a) {
  // a.W = intermediate width (= final width), a.H = input height, a.n_out = output height
  const size_t n = (size_t)a.B * a.n_out * a.W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % a.W), yy = (int)((i / a.W) % a.n_out), b = (int)(i / ((size_t)a.W * a.n_out));
    const int y0 = a.bounds[yy * 2], cnt = a.bounds[yy * 2 + 1];
    const unsigned char* p = a.src + (((size_t)b * a.H + y0) * a.W + x) * 3;
    long long s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int k = 0; k < cnt; ++k) {
      const long long w = a.taps[yy * a.ksize + k];
      const unsigned char* q = p + (size_t)k * a.W * 3;
      s0 += w * q[0]; s1 += w * q[1]; s2 += w * q[2];
    }
    const size_t plane = (size_t)a.n_out * a.W;
    float* o = a.out + (size_t)b * 3 * plane + (size_t)yy * a.W + x;
    o[0] = (float)min(max(s0 >> 22, 0LL), 255LL) / 255.0f;
    o[plane] = (float)min(max(s1 >> 22, 0LL), 255LL) / 255.0f;
    o[2 * plane] = (float)min(max(s2 >> 22, 0LL), 255LL) / 255.0f;
  }
}

extern "C" int hla_resize_bilinear(const unsigned char* src, const int* hbounds, const int* htaps, int hksize,
                                   const int* vbounds, const int* vtaps, int vksize, unsigned char* mid, float* out, int B,
                                   int H, int W, int out_h, int out_w, hla_stream_t stream) {
  HLA_REQUIRE(src && hbounds && htaps && vbounds && vtaps && mid && out, "hla_resize_bilinear: null argument");
  HLA_REQUIRE(B > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0 && hksize > 0 && vksize > 0, "hla_resize_bilinear: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  ResampleArgs a{};
  a.src = src; a.mid = mid; a.bounds = hbounds; a.taps = htaps; a.B = B; a.H = H; a.W = W; a.n_out = out_w; a.ksize = hksize;
  size_t n = (size_t)B * H * out_w;
  hla_prof_begin(K_ELEMWISE, 0, (double)B * H * ((double)W + out_w) * 3, st);
  hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535)), dim3(256), 0, st, a);
  ResampleArgs v{};
  v.src = mid; v.out = out; v.bounds = vbounds; v.taps = vtaps; v.B = B; v.H = H; v.W = out_w; v.n_out = out_h; v.ksize = vksize;
  n = (size_t)B * out_h * out_w;
  hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535)), dim3(256), 0, st, v);
  hla_prof_end(st);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}
