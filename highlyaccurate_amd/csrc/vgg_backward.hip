// Backward of the VGG16-U-Net extractor (what autograd does in the reference through VGG.py:121-203):
//   * L2_norm backward (VGG.py:511-514)
//   * data gradients: every conv's dgrad is the SAME MFMA implicit-GEMM kernel as the forward pass, run on
//     transposed + tap-flipped packed weights, with the ReLU mask / gradient fan-in / "sum 2x2" (nearest-upsample
//     backward) fused into its epilogue and the max-pool routing (argmax) fused into its loader
//   * weight gradients: wgrad_kernel, an MFMA GEMM that contracts over PIXELS.  Both operands are then "k-major"
//     in NHWC (pixels are rows); for bf16 the fragments are fetched with gfx950's transpose read
//     ds_read_b64_tr_b16 straight from the same [pixel][channel] LDS tiles the forward kernel uses
//     (tools/probes/tr16_probe.hip documents the lane semantics), for fp32 a lane needs one element per MFMA
//     so plain ds_read_b32 suffices.  Bias gradients ride along as one extra MFMA against an all-ones fragment.
#include <vector>
#include <stdlib.h>
#include "conv_kernels.h"
#include "vgg_layers.h"

// (frag_kmajor / frag_ones / KStep: conv_kernels.h -- the fused conv0 weight gradient in the data-gradient epilogue uses them too)

// ---------------------------------------------------------------------------------------------
constexpr int WG_TH = 4;                                  // pixel tile of the weight-gradient kernels: 4 rows x 32 px
struct WgradArgs {
  const void* x1; const void* x2;      // conv input (stored post-ReLU activations); virtual upsample+concat as forward
  const void* g;                       // d(loss)/d(conv output) NHWC T [B,H,W,Cout], or the pooled map's gradient
  const unsigned char* g_unpool;       //   [B,H/2,W/2,Cout] + forward argmax (virtual unpool) when non-null
  float* part;                         // [KS][Cout][Cin][9] partial sums
  float* bpart;                        // [KS][Cout] partial bias gradients, or null
  int C1, C2, up1, B, H, W, Cout, Cin, tiles_x, tiles_y, ntile, KS;
  int row_begin;                       // first pixel row that carries gradient (even with g_unpool); rows above are skipped
  const int* dyn;                      // data-dependent launch (bwd_fan_kernel): base of the device-side tables, or null
  int dyn_desc;                        //   int offset of {live tiles per sample, list offset, band table of g or -1}: only the
                                       //   listed tiles are visited, and g reads as zero outside the part its producer wrote
};

// tile enumeration of the weight-gradient kernels
struct WgTiles {
  const int* dyn; int nl, list, gb, ntile, tiles_x, tiles_y, row_begin, H, W, gsh;
  __device__ __forceinline__ WgTiles(const int* dyn_, int desc, int H_, int W_, int row_begin_, int tiles_x_, int tiles_y_,
                                     int ntile_all, int B, int gsh_)
      : dyn(dyn_), nl(0), list(0), gb(-1), ntile(ntile_all), tiles_x(tiles_x_), tiles_y(tiles_y_), row_begin(row_begin_),
        H(H_), W(W_), gsh(gsh_) {
    if (!dyn) return;
    nl = dyn[desc]; list = dyn[desc + 1]; gb = dyn[desc + 2];
    ntile = nl * B;
  }
  // origin of a tile and the column interval [gx0, gx1) of this launch's coordinates in which g may be read
  __device__ __forceinline__ void origin(int tile, int& b, int& y0, int& x0, int& gx0, int& gx1) const {
    gx0 = 0; gx1 = W;
    if (dyn) {
      b = tile / nl;
      const int e = dyn[list + tile % nl];
      y0 = (e >> 16) * WG_TH; x0 = (e & 0xffff) * 32;
      if (gb >= 0) {
        const int band = (y0 >> gsh) >> 3;
        gx0 = dyn[gb + 2 * band] << gsh;
        gx1 = min(dyn[gb + 2 * band + 1] << gsh, W);
      }
    } else {
      int q = tile;
      x0 = (q % tiles_x) * 32; q /= tiles_x;
      y0 = row_begin + (q % tiles_y) * WG_TH;
      b = q / tiles_y;
    }
  }
};

template <typename T> constexpr int wg_stride() { return 64 * (int)sizeof(T) + 16; }
template <typename T> constexpr int wg_lds_bytes() { return ((WG_TH + 2) * HWID + WG_TH * 32) * wg_stride<T>(); }

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
  constexpr int EPL = 16 / sizeof(T), STR = wg_stride<T>(), PPX = 64 * (int)sizeof(T) / 16;   // 16-B pieces per pixel
  constexpr int XPIX = (WG_TH + 2) * HWID, GPIX = WG_TH * 32, KPX = KStep<T>::PX;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;
  char* Gs = smem + XPIX * STR;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, ct = wv >> 1, it = wv & 1;
  const int ks = blockIdx.x, ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const bool first = ci0 < a.C1;
  const T* xsrc = first ? (const T*)a.x1 : (const T*)a.x2;
  const int Cs = first ? a.C1 : a.C2, coff = first ? ci0 : ci0 - a.C1, sh = (first && a.up1) ? 1 : 0;
  const int Hs = a.H >> sh, Ws = a.W >> sh;
  const int gsh = a.g_unpool ? 1 : 0, Hg = a.H >> gsh, Wg = a.W >> gsh;
  const bool want_bias = a.bpart && blockIdx.y == 0 && it == 0;

  f32x16 acc[9], accb;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;
  const uint4 ones = frag_ones<T>();

  // Register-staged tile loads: all of a tile's 16-B pieces are requested back to back (memory-level parallelism).  (Issuing
  // the NEXT tile's requests before the current tile's MFMA phase needs the staging registers live across it: one workgroup
  // per CU, measured slower.)
  constexpr int NX = (XPIX * PPX + 255) / 256, NG = GPIX * PPX / 256, PSTEP = 256 / PPX;
  const int part = t % PPX, pix0 = t / PPX;
  uint4 xr[NX], gr[NG];
  unsigned long long gid[NG];
  const WgTiles tl(a.dyn, a.dyn_desc, a.H, a.W, a.row_begin, a.tiles_x, a.tiles_y, a.ntile, a.B, a.g_unpool ? 1 : 0);
  int gx0, gx1;                         // (set by every tile_origin call: the interval of the tile being loaded)
  auto tile_origin = [&](int tile, int& b, int& y0, int& x0) { tl.origin(tile, b, y0, x0, gx0, gx1); };
  // raw buffer loads through one descriptor per operand and sample: a piece outside the image / the written part of g gets an
  // offset beyond the range and reads zeros -- no branch around a load (hipcc ends every branch-guarded load's block with an
  // s_waitcnt vmcnt(0): the exact-fp32 kernel's 29 loads per tile were 29 dependent round trips)
  constexpr int OOB = (int)0x80000000;
  const size_t xs_bytes = (size_t)Hs * Ws * Cs * sizeof(T), gs_bytes = (size_t)Hg * Wg * a.Cout * sizeof(T);
  auto rsrc = [](const void* base, size_t bytes) __attribute__((always_inline)) {
    const unsigned long long p = (unsigned long long)base;
    const void* pu = (const void*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p));
    return __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
  auto load_x = [&](int tile, int lo, int hi) {
    int b, y0, x0;
    tile_origin(tile, b, y0, x0);
    const __amdgpu_buffer_rsrc_t rx = rsrc((const char*)xsrc + (size_t)b * xs_bytes, xs_bytes);
#pragma unroll
    for (int i = 0; i < NX; ++i) {                     // input halo tile, zero outside the image
      if (i < lo || i >= hi) continue;
      const int pix = pix0 + i * PSTEP;
      const int hy = pix / HWID, hx = pix - hy * HWID, y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = pix < XPIX && y >= 0 && y < a.H && x >= 0 && x < a.W;
      const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (((y >> sh) * Ws + (x >> sh)) * Cs + coff + part * EPL) * (int)sizeof(T) : OOB, 0, 0);
      xr[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  auto load_g = [&](int tile, int lo, int hi) {
    int b, y0, x0;
    tile_origin(tile, b, y0, x0);
    const __amdgpu_buffer_rsrc_t rg = rsrc((const char*)a.g + (size_t)b * gs_bytes, gs_bytes);
    const __amdgpu_buffer_rsrc_t ri = rsrc(a.g_unpool ? a.g_unpool + (size_t)b * (gs_bytes / sizeof(T)) : (const unsigned char*)a.g,
                                           a.g_unpool ? gs_bytes / sizeof(T) : 0);
#pragma unroll
    for (int i = 0; i < NG; ++i) {                     // output-gradient tile (virtual unpool: + the forward argmax)
      if (i < lo || i >= hi) continue;
      const int pix = pix0 + i * PSTEP;
      const int y = y0 + pix / 32, x = x0 + pix % 32;
      const bool ok = y < a.H && x >= gx0 && x < gx1;
      const int e0 = ok ? ((y >> gsh) * Wg + (x >> gsh)) * a.Cout + co0 + part * EPL : OOB;
      const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e0 * (int)sizeof(T) : OOB, 0, 0);
      gr[i] = make_uint4(v.x, v.y, v.z, v.w);
      unsigned long long id = 0;                       // (no unpool: a zero-sized descriptor, reads 0)
      if constexpr (EPL == 8) { const u32x2w w = __builtin_amdgcn_raw_buffer_load_b64(ri, e0, 0, 0); id = (unsigned long long)w.x | ((unsigned long long)w.y << 32); }
      else id = __builtin_amdgcn_raw_buffer_load_b32(ri, e0, 0, 0);
      gid[i] = id;
    }
  };
  auto store_x = [&](int lo, int hi) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (i < lo || i >= hi) continue;
      const int pix = pix0 + i * PSTEP;
      if (pix < XPIX) *(uint4*)(Xs + pix * STR + part * 16) = xr[i];
    }
  };
  auto store_g = [&](int tile, int lo, int hi) {
    int b, y0, x0;
    tile_origin(tile, b, y0, x0);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      if (i < lo || i >= hi) continue;
      const int pix = pix0 + i * PSTEP;
      uint4 v = gr[i];
      if (a.g_unpool) {                                // keep the elements whose forward argmax is this (y&1, x&1)
        const unsigned pos = (((y0 + pix / 32) & 1) << 1) | ((x0 + pix % 32) & 1);
        T ev[EPL];
        unsigned char id[8];
        __builtin_memcpy(ev, &v, 16);
        __builtin_memcpy(id, &gid[i], 8);
#pragma unroll
        for (int k = 0; k < EPL; ++k) if (id[k] != pos) ev[k] = (T)0.f;
        __builtin_memcpy(&v, ev, 16);
      }
      *(uint4*)(Gs + pix * STR + part * 16) = v;
    }
  };

  for (int tile = ks; tile < tl.ntile; tile += a.KS) {
    if (sizeof(T) == 2) {
      load_x(tile, 0, NX); load_g(tile, 0, NG);
      __syncthreads();                                 // previous tile fully consumed
      store_x(0, NX);
      store_g(tile, 0, NG);
    } else {                                           // fp32 (parity mode): batches of 4 pieces = 16 staging registers
      __syncthreads();
#pragma unroll
      for (int lo = 0; lo < NX; lo += 4) { load_x(tile, lo, lo + 4); store_x(lo, lo + 4); }
#pragma unroll
      for (int lo = 0; lo < NG; lo += 4) { load_g(tile, lo, lo + 4); store_g(tile, lo, lo + 4); }
    }
    __syncthreads();
    // G fragments of the whole tile stay in registers; every X fragment (halo row rho, column shift kx, K-step kk) is
    // fetched ONCE and feeds the up to three taps ky with r = rho - ky inside the tile  (halves the LDS reads per MFMA)
    // (K-steps are taken two at a time so that the resident G fragments cost 32 VGPRs for every dtype)
#pragma unroll 1
    for (int kk0 = 0; kk0 < 32 / KPX; kk0 += 2) {
      uint4 Af[WG_TH][2];
#pragma unroll
      for (int r = 0; r < WG_TH; ++r)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          Af[r][kk] = frag_kmajor<T>(Gs, STR, r * 32 + (kk0 + kk) * KPX, ct * 32, lane);
          if (want_bias) mma16<T>(accb, Af[r][kk], ones);
        }
#pragma unroll
      for (int rho = 0; rho < WG_TH + 2; ++rho) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const uint4 Bf = frag_kmajor<T>(Xs, STR, rho * HWID + kx + (kk0 + kk) * KPX, it * 32, lane);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const int r = rho - ky;
              if (r >= 0 && r < WG_TH) mma16<T>(acc[ky * 3 + kx], Af[r][kk], Bf);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // D[i = co][j = ci]: lane -> ci = ci0 + it*32 + (lane&31); reg r -> co = co0 + ct*32 + (r&3) + 8(r>>2) + 4(lane>>5)
  const int ci = ci0 + it * 32 + (lane & 31), g5 = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * g5;
    float* o = a.part + (((size_t)ks * a.Cout + co) * a.Cin + ci) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) o[tap] = acc[tap][r];
    if (want_bias && (lane & 31) == 0) a.bpart[(size_t)ks * a.Cout + co] = accb[r];
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad_dma_kernel: the same contraction for the 16-bit types with the tiles fetched HBM -> LDS by LDS-DMA and the NEXT tile's
// loads in flight under the current tile's MFMA phase.  wgrad_kernel stages a tile through 44 registers and runs
// load -> barrier -> LDS write -> barrier -> MFMA with nothing but the co-resident workgroup to overlap the phases; here the X
// halo tile is double-buffered, and the G tile -- whose fragments all sit in registers for the whole MFMA phase -- is refilled in
// place as soon as every wave holds them: 2 x 28 KB + 16 KB = 72 KB, still two workgroups per CU, and no staging registers.
// LDS image: 128-B pixels with NO pad (a DMA's image is lane-linear); the transposing reads stay conflict-free through an XOR of
// the 16-B chunk index with 2 (pixel & 3), applied to the SOURCE chunk a lane fetches and to the read address: a 16-lane group
// of ds_read_b64_tr_b16 touches 4 consecutive pixels x 32 B, which the key spreads over 4 different 32-B segments of the two
// 128-B bank halves.  Zero fill (outside the image / the written part of g) by the buffer descriptors' range check.
// Plain launches only (no virtual unpool: that loader masks elements on the way into LDS).
#ifndef HLA_WGRAD_DMA
#define HLA_WGRAD_DMA 1
#endif
constexpr int WGD_XBLK = 28, WGD_GBLK = 16;             // 1-KiB blocks (8 pixels) per X buffer (204 pixels + pad) / G buffer
constexpr int wgd_lds_bytes() { return (2 * WGD_XBLK + WGD_GBLK) * 1024; }

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(WgradArgs a) {
  static_assert(sizeof(T) == 2, "16-bit types");
  constexpr int XPIX = (WG_TH + 2) * HWID, XB = WGD_XBLK * 1024, NXJ = WGD_XBLK / 4, NGJ = WGD_GBLK / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Gs = smem + 2 * XB;
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), ct = wv >> 1, it = wv & 1;
  const int ks = blockIdx.x, ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const bool first = ci0 < a.C1;
  const char* xsrc = first ? (const char*)a.x1 : (const char*)a.x2;
  const int Cs = first ? a.C1 : a.C2, coff = first ? ci0 : ci0 - a.C1, sh = (first && a.up1) ? 1 : 0;
  const int Hs = a.H >> sh, Ws = a.W >> sh;
  const bool want_bias = a.bpart && blockIdx.y == 0 && it == 0;

  f32x16 acc[9], accb;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;
  const uint4 ones = frag_ones<T>();

  // what a lane fetches: block k = wv + 4 j of a buffer, pixel 8 k + (lane >> 3) of the tile, LDS chunk position lane & 7, which holds
  // channel chunk (lane & 7) ^ 2 (pixel & 3)
  const int q = lane >> 3, cp = lane & 7;
  int xhy[NXJ], xhx[NXJ], xc[NXJ];
#pragma unroll
  for (int j = 0; j < NXJ; ++j) {
    const int p = 8 * (wv + 4 * j) + q;
    xhy[j] = p < XPIX ? p / HWID : 1 << 20;            // (past the tile: never inside the image -> zeros into the pad)
    xhx[j] = p % HWID;
    xc[j] = (cp ^ (2 * (p & 3))) * 16;
  }
  typedef int rsrc_t __attribute__((ext_vector_type(4)));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem) + wv * 1024;
  const WgTiles tl(a.dyn, a.dyn_desc, a.H, a.W, a.row_begin, a.tiles_x, a.tiles_y, a.ntile, a.B, 0);
  auto dma_tile = [&](int tile, int xbuf) __attribute__((always_inline)) {
    int b, y0, x0, gx0, gx1;
    tl.origin(tile, b, y0, x0, gx0, gx1);
    const unsigned long long px = (unsigned long long)xsrc + (size_t)b * Hs * Ws * Cs * 2;
    const unsigned long long pg = (unsigned long long)a.g + (size_t)b * a.H * a.W * a.Cout * 2;
    rsrc_t rx, rg;
    rx[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)px); rx[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(px >> 32));
    rx[2] = __builtin_amdgcn_readfirstlane(Hs * Ws * Cs * 2); rx[3] = 0x00020000;
    rg[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pg); rg[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(pg >> 32));
    rg[2] = __builtin_amdgcn_readfirstlane(a.H * a.W * a.Cout * 2); rg[3] = 0x00020000;
    int ox[NXJ], og[NGJ];
#pragma unroll
    for (int j = 0; j < NXJ; ++j) {
      const int y = y0 - 1 + xhy[j], x = x0 - 1 + xhx[j];
      const bool ok = y >= 0 && y < a.H && x >= 0 && x < a.W;
      ox[j] = ok ? (((y >> sh) * Ws + (x >> sh)) * Cs + coff) * 2 + xc[j] : (int)0x80000000;
    }
#pragma unroll
    for (int j = 0; j < NGJ; ++j) {
      const int p = 8 * (wv + 4 * j) + q, y = y0 + p / 32, x = x0 + p % 32;
      const bool ok = y < a.H && x >= gx0 && x < gx1;
      og[j] = ok ? ((y * a.W + x) * a.Cout + co0) * 2 + (cp ^ (2 * (p & 3))) * 16 : (int)0x80000000;
    }
    const unsigned dx = lds0 + xbuf * XB, dg = lds0 + 2 * XB;
    const int zero = 0;
    unsigned keep;
    static_assert(NXJ == 7 && NGJ == 4, "asm below");
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %8, %9 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %8, %9 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %8, %9 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %8, %9 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, %9 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, %9 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %8, %9 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(ox[0]), "v"(ox[1]), "v"(ox[2]), "v"(ox[3]), "v"(ox[4]), "v"(ox[5]), "v"(ox[6]), "s"(rx), "s"(zero), "s"(dx)
                 : "memory", "scc");
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(og[0]), "v"(og[1]), "v"(og[2]), "v"(og[3]), "s"(rg), "s"(zero), "s"(dg)
                 : "memory", "scc");
  };
  // read side: byte offset of this lane's 8 bytes inside a buffer for a fragment whose first pixel px0 has px0 & 3 == m, WITHOUT
  // px0 * 128 (compile-time: it goes into the instruction's offset): row part + swizzled column part
  const int tq = (lane & 15) >> 2, g5 = lane >> 5;
  int scG[4], scX[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int key = (2 * ((m + tq) & 3)) << 4;
    const int rowb = (tq + 8 * g5) * 128;
    scG[m] = rowb + ((((ct * 32 + 16 * ((lane >> 4) & 1)) + 4 * (lane & 3)) * 2) ^ key);
    scX[m] = rowb + ((((it * 32 + 16 * ((lane >> 4) & 1)) + 4 * (lane & 3)) * 2) ^ key);
  }
  auto frag = [&](const char* buf, int px0, const int (&sc)[4]) __attribute__((always_inline)) {
    const char* p = buf + px0 * 128 + sc[px0 & 3];
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 4 * 128));
    uint4 r;
    __builtin_memcpy(&r.x, &lo, 8);
    __builtin_memcpy(&r.z, &hi, 8);
    return r;
  };

  int tile = ks, cur = 0;
  if (tile < tl.ntile) dma_tile(tile, 0);
  for (; tile < tl.ntile; tile += a.KS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of X(tile) and G(tile) has landed ...
    __syncthreads();                                       // ... and so has everyone's; the previous tile's MFMA phase is over
    const char* Xs = smem + cur * XB;
    uint4 Af[WG_TH][2];
#pragma unroll
    for (int r = 0; r < WG_TH; ++r)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Af[r][kk] = frag(Gs, r * 32 + kk * 16, scG);
    __syncthreads();                                       // every wave holds its G fragments: the G buffer can be refilled
    if (tile + a.KS < tl.ntile) dma_tile(tile + a.KS, cur ^ 1);
    if (want_bias) {
#pragma unroll
      for (int r = 0; r < WG_TH; ++r)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) mma16<T>(accb, Af[r][kk], ones);
    }
#pragma unroll
    for (int rho = 0; rho < WG_TH + 2; ++rho) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const uint4 Bf = frag(Xs, rho * HWID + kx + kk * 16, scX);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int r = rho - ky;
            if (r >= 0 && r < WG_TH) mma16<T>(acc[ky * 3 + kx], Af[r][kk], Bf);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    cur ^= 1;
  }
  const int ci = ci0 + it * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * g5;
    float* o = a.part + (((size_t)ks * a.Cout + co) * a.Cin + ci) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) o[tap] = acc[tap][r];
    if (want_bias && (lane & 31) == 0) a.bpart[(size_t)ks * a.Cout + co] = accb[r];
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad_ws_kernel: the 16-bit weight gradient, WAVE-SPECIALISED (round 5; the split-mode twin is wgrad_split_ws_kernel below, which
// explains the scheme).  One 8-wave workgroup per CU: waves 4-7 fetch a 4-row tile's pieces two tiles ahead through registers
// (raw buffer loads, zero fill by the descriptors' range check, the virtual unpool's argmax mask applied on the way into LDS) and
// write them into the idle one of two LDS buffers; waves 0-3 -- one per SIMD -- only read fragments and multiply, the next halo
// row's fragments requested ahead of the current row's MFMAs.  One barrier per tile.  Plain AND un-pooling launches (wgrad_kernel
// and wgrad_dma_kernel ran load -> barrier -> MFMA phases in every wave: 0.30 of the MFMA peak where the forward reaches 0.48).
#ifndef HLA_WGRAD_WS
#define HLA_WGRAD_WS 1
#endif
template <typename T> constexpr int wg_ws_buf_bytes() { return ((WG_TH + 2) * HWID + WG_TH * 32) * wg_stride<T>(); }
template <typename T> constexpr int wg_ws_lds_bytes() { return 2 * wg_ws_buf_bytes<T>(); }

template <typename T>
__global__ __launch_bounds__(512, 1) void wgrad_ws_kernel(WgradArgs a) {
  static_assert(sizeof(T) == 2, "16-bit types");
  constexpr int STR = wg_stride<T>(), PPX = 8, XPIX = (WG_TH + 2) * HWID, GPIX = WG_TH * 32, KPX = 16, BUFB = wg_ws_buf_bytes<T>();
  constexpr int oX = 0, oG = XPIX * STR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wv >= 4;
  const int ks = blockIdx.x, ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const bool first = ci0 < a.C1;
  const T* xsrc = first ? (const T*)a.x1 : (const T*)a.x2;
  const int Cs = first ? a.C1 : a.C2, coff = first ? ci0 : ci0 - a.C1, sh = (first && a.up1) ? 1 : 0;
  const int Hs = a.H >> sh, Ws = a.W >> sh;
  const int gsh = a.g_unpool ? 1 : 0, Hg = a.H >> gsh, Wg = a.W >> gsh;
  const WgTiles tl(a.dyn, a.dyn_desc, a.H, a.W, a.row_begin, a.tiles_x, a.tiles_y, a.ntile, a.B, a.g_unpool ? 1 : 0);
  const int ntile = tl.ntile;
  const int t_first = ks < ntile ? ks : -1;
  auto next_tile = [&](int tt) { tt += a.KS; return tt < ntile ? tt : -1; };

  if (loader) {
    const int tl_ = t - 256, part = tl_ % PPX, pix0 = tl_ / PPX;
    constexpr int NX = (XPIX * PPX + 255) / 256, NG = GPIX * PPX / 256, PSTEP = 256 / PPX;
    static_assert(GPIX * PPX % 256 == 0, "gradient tile pieces per thread");
    constexpr int OOB = (int)0x80000000;
    const size_t xs_bytes = (size_t)Hs * Ws * Cs * 2, gs_bytes = (size_t)Hg * Wg * a.Cout * 2;
    auto rsrc = [](const void* base, size_t bytes) __attribute__((always_inline)) {
      const unsigned long long p = (unsigned long long)base;
      const void* pu = (const void*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p));
      return __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    struct Stage { u32x4 xr[NX]; u32x4 gr[NG]; u32x2 gid[NG]; int ypar, xpar; };
    int xrel[NX], xhyx[NX], grel[NG];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int pix = pix0 + k * PSTEP, hy = pix / HWID, hx = pix - hy * HWID;
      xhyx[k] = pix < XPIX ? (hy << 8) | hx : (200 << 8);
      xrel[k] = (((hx - 1) >> sh) * Cs + coff + part * 8) * 2;      // column part (x0 is a multiple of 32); the row part per tile
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int pix = pix0 + k * PSTEP;
      grel[k] = (((pix / 32) >> gsh) * Wg + ((pix % 32) >> gsh)) * a.Cout + co0 + part * 8;
    }
    auto issue = [&](int tile, Stage& S) __attribute__((always_inline)) {
      int b = 0, x0 = 0, gx0 = 0, gx1 = 0, y0 = 0;
      const bool live = tile >= 0;
      if (live) tl.origin(tile, b, y0, x0, gx0, gx1);
      S.ypar = y0; S.xpar = x0;
      const __amdgpu_buffer_rsrc_t rx = rsrc((const char*)xsrc + (size_t)b * xs_bytes, xs_bytes);
      const __amdgpu_buffer_rsrc_t rg = rsrc((const char*)a.g + (size_t)b * gs_bytes, gs_bytes);
      const __amdgpu_buffer_rsrc_t ri = rsrc(a.g_unpool ? a.g_unpool + (size_t)b * (gs_bytes / 2) : (const unsigned char*)a.g, a.g_unpool ? gs_bytes / 2 : 0);
      // source row of halo row hy: (y0 - 1 + hy) >> sh = ((y0 - 1) >> sh) + ((hy + ((y0 - 1) & sh)) >> sh) -- the static first row of a
      // trimmed ground launch may be odd, so the tile origin's parity under the upsample shift is carried (ypar)
      const int ylo = y0 - 1, ypar = ylo & sh, rowb = Ws * Cs * 2;
      const int xbase = __builtin_amdgcn_readfirstlane(((ylo >> sh) * Ws + (x0 >> sh)) * Cs * 2);
      const int gbase = __builtin_amdgcn_readfirstlane(((y0 >> gsh) * Wg + (x0 >> gsh)) * a.Cout);
      const int hy_lo = live ? max(0, 1 - y0) : 255, hy_hi = a.H - y0 + 1, hx_lo = max(0, 1 - x0), hx_hi = a.W - x0 + 1;
      const int gy_hi = live ? a.H - y0 : 0, gx_lo = gx0 - x0, gx_hi = gx1 - x0;
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        const int hy = xhyx[k] >> 8, hx = xhyx[k] & 255;
        const bool ok = hy >= hy_lo && hy < hy_hi && hx >= hx_lo && hx < hx_hi;
        S.xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xbase + ((hy + ypar) >> sh) * rowb + xrel[k] : OOB, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int pix = pix0 + k * PSTEP, py = pix / 32, px = pix % 32;
        const bool ok = py < gy_hi && px >= gx_lo && px < gx_hi;
        const int e0 = ok ? gbase + grel[k] : OOB;
        S.gr[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e0 * 2 : OOB, 0, 0);
        S.gid[k] = __builtin_amdgcn_raw_buffer_load_b64(ri, e0, 0, 0);       // (no unpool: a zero-sized descriptor, reads 0)
      }
    };
    auto commit = [&](const Stage& S, int buf) __attribute__((always_inline)) {
      char* base = smem + buf * BUFB;
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        const int pix = pix0 + k * PSTEP;
        if (pix < XPIX) *(u32x4*)(base + oX + pix * STR + part * 16) = S.xr[k];
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int pix = pix0 + k * PSTEP;
        u32x4 v = S.gr[k];
        if (a.g_unpool) {      // keep the elements whose forward argmax is this (y&1, x&1): (id ^ pos) - 1 is negative only for a match
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          const unsigned pos = ((((S.ypar + pix / 32) & 1) << 1) | ((S.xpar + pix % 32) & 1)) * 0x01010101u;
          const unsigned m0 = S.gid[k].x ^ pos, m1 = S.gid[k].y ^ pos;
          auto keep = [](unsigned m, unsigned sel) {
            s16x2 w2 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, m, sel));
            w2 = (w2 - (short)1) >> 15;
            return __builtin_bit_cast(unsigned, w2);
          };
          v.x &= keep(m0, 0x0c010c00u); v.y &= keep(m0, 0x0c030c02u);
          v.z &= keep(m1, 0x0c010c00u); v.w &= keep(m1, 0x0c030c02u);
        }
        *(u32x4*)(base + oG + pix * STR + part * 16) = v;
      }
    };
    Stage A, Bq;
    int ta = t_first, tb = ta >= 0 ? next_tile(ta) : -1;
    issue(ta, A);
    issue(tb, Bq);
    commit(A, 0);
    __syncthreads();                                     // barrier 0: buffer 0 holds the first tile
    int cur = 0;
    while (ta >= 0) {
      int tc = tb >= 0 ? next_tile(tb) : -1;
      issue(tc, A);
      commit(Bq, cur ^ 1);
      __syncthreads();
      ta = tb; tb = tc; cur ^= 1;
      if (ta < 0) break;
      tc = tb >= 0 ? next_tile(tb) : -1;
      issue(tc, Bq);
      commit(A, cur ^ 1);
      __syncthreads();
      ta = tb; tb = tc; cur ^= 1;
    }
    return;
  }

  // ---------------- matrix waves
  const int ct = wv >> 1, it = wv & 1;
  const bool want_bias = a.bpart && blockIdx.y == 0 && it == 0;
  f32x16 acc[9], accb;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;
  const uint4 ones = frag_ones<T>();
  __syncthreads();                                       // barrier 0
  int cur = 0;
  for (int tile = t_first; tile >= 0; tile = next_tile(tile)) {
    const char* Xs = smem + cur * BUFB + oX;
    const char* Gs = smem + cur * BUFB + oG;
    // the G fragments of the whole tile (4 rows x 2 K-steps) stay in registers; halo row rho's X fragments (3 column shifts x 2
    // K-steps) are requested one row ahead of the MFMAs that consume them and feed the up to three taps ky with r = rho - ky
    uint4 Af[WG_TH][2], Bf[2][3][2];
#pragma unroll
    for (int r = 0; r < WG_TH; ++r)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Af[r][kk] = frag_kmajor<T>(Gs, STR, r * 32 + kk * KPX, ct * 32, lane);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Bf[0][kx][kk] = frag_kmajor<T>(Xs, STR, kx + kk * KPX, it * 32, lane);
#pragma unroll
    for (int rho = 0; rho < WG_TH + 2; ++rho) {
      if (rho + 1 < WG_TH + 2) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) Bf[(rho + 1) & 1][kx][kk] = frag_kmajor<T>(Xs, STR, (rho + 1) * HWID + kx + kk * KPX, it * 32, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (rho == 0 && want_bias) {
#pragma unroll
        for (int r = 0; r < WG_TH; ++r)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) mma16<T>(accb, Af[r][kk], ones);
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int r = rho - ky;
            if (r >= 0 && r < WG_TH) mma16<T>(acc[ky * 3 + kx], Af[r][kk], Bf[rho & 1][kx][kk]);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                     // the loaders have filled the other buffer; this one is free
    cur ^= 1;
  }
  const int ci = ci0 + it * 32 + (lane & 31), g5 = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * g5;
    float* o = a.part + (((size_t)ks * a.Cout + co) * a.Cin + ci) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) o[tap] = acc[tap][r];
    if (want_bias && (lane & 31) == 0) a.bpart[(size_t)ks * a.Cout + co] = accb[r];
  }
}

// ---------------------------------------------------------------------------------------------
// Split-fp16 weight gradient (precision 'fp16x3'): the same contraction over pixels with both operands fed to the matrix cores
// as hi + lo = fp16(s v) + fp16(s v - hi): G X ~= Ghi Xhi + Glo Xhi + Ghi Xlo, three v_mfma_f32_32x32x16_f16 per product, fp32
// accumulate -- fp32-class gradients at a third of the fp16 MFMA rate instead of the exact-fp32 kernels' sixteenth.
// Storage stays fp32 (the maps a split-mode forward / backward keep); a tile's 16-B pieces are split where they enter LDS, into
// an fp16 hi plane and an fp16 lo plane per operand, each laid out like the f16 kernel's tile, so the k-major fragments come
// from the same transpose reads.  Scales: ONE power of two per operand for the whole launch, from the maximum over the batch
// of the per-sample maxima their producers recorded (the gradient is a sum over the batch, so a batch-wide scale costs no
// accuracy where it matters: elements below 2^-17 of the batch maximum keep an absolute error of 2^-39 of it); exact to undo.
struct WgradSplitExtra {
  const unsigned* amax_x1; const unsigned* amax_x2; const unsigned* amax_g;   // [B] fp32 bit patterns of max |.| per sample
};
constexpr int WGS_STR = 64 * 2 + 16;                       // fp16 plane row stride (as wg_stride<f16>)
// A tile of the (shared) tile lists is WG_TH = 4 rows x 32 pixels; with two fp16 planes per operand that is 96 KB of LDS and one
// workgroup per CU, whose load and MFMA phases then run strictly one after the other (measured: 1.9 ms per launch, 6x the bf16
// kernel for 3x its MFMAs).  The tile is therefore walked as two HALVES of WGS_TH = 2 rows: 58 KB, two workgroups per CU.
constexpr int WGS_TH = 2;
constexpr int wgs_lds_bytes() { return 2 * ((WGS_TH + 2) * HWID + WGS_TH * 32) * WGS_STR; }

static __global__ __launch_bounds__(256, 2) void wgrad_split_kernel(WgradArgs a, WgradSplitExtra sx) {
  typedef f16 H;
  constexpr int STR = WGS_STR, PPX = 16;                   // 16-B fp32 pieces per pixel (64 channels)
  constexpr int XPIX = (WGS_TH + 2) * HWID, GPIX = WGS_TH * 32, KPX = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xh = smem;
  char* Xl = Xh + XPIX * STR;
  char* Gh = Xl + XPIX * STR;
  char* Gl = Gh + GPIX * STR;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, ct = wv >> 1, it = wv & 1;
  const int ks = blockIdx.x, ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const bool first = ci0 < a.C1;
  const float* xsrc = first ? (const float*)a.x1 : (const float*)a.x2;
  const int Cs = first ? a.C1 : a.C2, coff = first ? ci0 : ci0 - a.C1, sh = (first && a.up1) ? 1 : 0;
  const int Hs = a.H >> sh, Ws = a.W >> sh;
  const int gsh = a.g_unpool ? 1 : 0, Hg = a.H >> gsh, Wg = a.W >> gsh;
  const bool want_bias = a.bpart && blockIdx.y == 0 && it == 0;
  // launch-wide scales (uniform)
  unsigned mx = 0, mg = 0;
  const unsigned* ax = first ? sx.amax_x1 : sx.amax_x2;
  for (int b = 0; b < a.B; ++b) { mx = max(mx, ax[b]); mg = max(mg, sx.amax_g[b]); }
  const float s_x = split_scale(mx), s_g = split_scale(mg);

  f32x16 acc[9], accb;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;
  const uint4 ones = frag_ones<f16>();

  constexpr int NX = (XPIX * PPX + 255) / 256, NG = GPIX * PPX / 256, PSTEP = 256 / PPX;
  static_assert(GPIX * PPX % 256 == 0, "gradient tile pieces per thread");
  const int part = t % PPX, pix0 = t / PPX;
  const WgTiles tl(a.dyn, a.dyn_desc, a.H, a.W, a.row_begin, a.tiles_x, a.tiles_y, a.ntile, a.B, a.g_unpool ? 1 : 0);
  // Tile loads: raw buffer loads through one descriptor per operand and sample (base = the sample's map, range = its bytes).  A piece
  // outside the image / the written part of g gets an offset beyond the range and reads as ZERO: no branch around a load, so all of a
  // half tile's 13 (+ 4 argmax) loads are in flight together.  (With `if (inside) v = *p` hipcc put each load into its own exec-masked
  // block with an s_waitcnt vmcnt(0) at its end: 13 dependent memory round trips per half tile -- the kernel ran at 0.37 of its
  // MFMA ceiling where the forward kernels reach 0.55.)
  constexpr int OOB = (int)0x80000000;
  const size_t xs_bytes = (size_t)Hs * Ws * Cs * 4, gs_bytes = (size_t)Hg * Wg * a.Cout * 4;
  auto rsrc = [](const void* base, size_t bytes) __attribute__((always_inline)) {
    const unsigned long long p = (unsigned long long)base;
    const void* pu = (const void*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p));
    return __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  for (int tile2 = 2 * ks; tile2 < 2 * tl.ntile; tile2 += (tile2 & 1) ? 2 * a.KS - 1 : 1) {      // (tile, half 0), (tile, half 1), next tile
    int b, y0, x0, gx0, gx1;
    tl.origin(tile2 >> 1, b, y0, x0, gx0, gx1);
    y0 += (tile2 & 1) * WGS_TH;
    if (y0 >= a.H) continue;                           // (uniform: the lower half of a tile at the image's last rows)
    const __amdgpu_buffer_rsrc_t rx = rsrc((const char*)xsrc + (size_t)b * xs_bytes, xs_bytes);
    const __amdgpu_buffer_rsrc_t rg = rsrc((const char*)a.g + (size_t)b * gs_bytes, gs_bytes);
    const __amdgpu_buffer_rsrc_t ri = rsrc(a.g_unpool ? a.g_unpool + (size_t)b * (gs_bytes / 4) : (const unsigned char*)a.g, a.g_unpool ? gs_bytes / 4 : 0);
    u32x4 xr[NX], gr[NG];
    unsigned gid[NG];
#pragma unroll
    for (int k = 0; k < NX; ++k) {                     // input halo tile, zero outside the image
      const int pix = pix0 + k * PSTEP;
      const int hy = pix / HWID, hx = pix - hy * HWID, y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = pix < XPIX && y >= 0 && y < a.H && x >= 0 && x < a.W;
      const int off = ok ? (((y >> sh) * Ws + (x >> sh)) * Cs + coff + part * 4) * 4 : OOB;
      xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {                     // output-gradient tile (virtual unpool: + the forward argmax)
      const int pix = pix0 + k * PSTEP;
      const int y = y0 + pix / 32, x = x0 + pix % 32;
      const bool ok = y < a.H && x >= gx0 && x < gx1;
      const int e0 = ok ? ((y >> gsh) * Wg + (x >> gsh)) * a.Cout + co0 + part * 4 : OOB;
      gr[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e0 * 4 : OOB, 0, 0);
      gid[k] = a.g_unpool ? __builtin_amdgcn_raw_buffer_load_b32(ri, e0, 0, 0) : 0u;
    }
    __syncthreads();                                   // previous half tile fully consumed (the loads above are in flight across it)
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int pix = pix0 + k * PSTEP;
      uint2 hi, lw;
      split4(__uint_as_float(xr[k].x), __uint_as_float(xr[k].y), __uint_as_float(xr[k].z), __uint_as_float(xr[k].w), s_x, hi, lw);
      if (pix < XPIX) { *(uint2*)(Xh + pix * STR + part * 8) = hi; *(uint2*)(Xl + pix * STR + part * 8) = lw; }
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int pix = pix0 + k * PSTEP;
      float e0 = __uint_as_float(gr[k].x), e1 = __uint_as_float(gr[k].y), e2 = __uint_as_float(gr[k].z), e3 = __uint_as_float(gr[k].w);
      if (a.g_unpool) {                                // keep the elements whose forward argmax is this (y&1, x&1)
        const unsigned pos = (((y0 + pix / 32) & 1) << 1) | ((x0 + pix % 32) & 1);
        if ((gid[k] & 0xff) != pos) e0 = 0.f;
        if (((gid[k] >> 8) & 0xff) != pos) e1 = 0.f;
        if (((gid[k] >> 16) & 0xff) != pos) e2 = 0.f;
        if ((gid[k] >> 24) != pos) e3 = 0.f;
      }
      uint2 hi, lw;
      split4(e0, e1, e2, e3, s_g, hi, lw);
      *(uint2*)(Gh + pix * STR + part * 8) = hi; *(uint2*)(Gl + pix * STR + part * 8) = lw;
    }
    __syncthreads();
    // one K-step (16 pixels) at a time: the G fragments of the tile's four rows (hi and lo: 32 registers) stay resident while
    // every X fragment (halo row rho, column shift kx) is fetched once and feeds the up to three taps ky that use it
#pragma unroll 1
    for (int kk = 0; kk < 32 / KPX; ++kk) {
      uint4 Ah[WGS_TH], Al[WGS_TH];
#pragma unroll
      for (int r = 0; r < WGS_TH; ++r) {
        Ah[r] = frag_kmajor<H>(Gh, STR, r * 32 + kk * KPX, ct * 32, lane);
        Al[r] = frag_kmajor<H>(Gl, STR, r * 32 + kk * KPX, ct * 32, lane);
        if (want_bias) { mma16<H>(accb, Ah[r], ones); mma16<H>(accb, Al[r], ones); }
      }
#pragma unroll
      for (int rho = 0; rho < WGS_TH + 2; ++rho) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const uint4 Bh = frag_kmajor<H>(Xh, STR, rho * HWID + kx + kk * KPX, it * 32, lane);
          const uint4 Bl = frag_kmajor<H>(Xl, STR, rho * HWID + kx + kk * KPX, it * 32, lane);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int r = rho - ky;
            if (r >= 0 && r < WGS_TH) {
              mma16<H>(acc[ky * 3 + kx], Ah[r], Bh);
              mma16<H>(acc[ky * 3 + kx], Al[r], Bh);
              mma16<H>(acc[ky * 3 + kx], Ah[r], Bl);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // D[i = co][j = ci]: lane -> ci = ci0 + it*32 + (lane&31); reg r -> co = co0 + ct*32 + (r&3) + 8(r>>2) + 4(lane>>5)
  const float inv = 1.f / (s_x * s_g), invg = 1.f / s_g;
  const int ci = ci0 + it * 32 + (lane & 31), g5 = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * g5;
    float* o = a.part + (((size_t)ks * a.Cout + co) * a.Cin + ci) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) o[tap] = acc[tap][r] * inv;
    if (want_bias && (lane & 31) == 0) a.bpart[(size_t)ks * a.Cout + co] = accb[r] * invg;
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad_split_ws_kernel: the same contraction, WAVE-SPECIALISED (round 5).  wgrad_split_kernel runs load -> split -> barrier ->
// MFMA in every wave, with nothing but the second resident workgroup to overlap the phases: 0.47 of its MFMA ceiling where the
// forward kernels reach 0.55.  Here a workgroup is EIGHT waves on one CU: waves 4-7 are LOADERS (they fetch a half tile's fp32
// pieces two half tiles ahead, split them into the fp16 hi / lo planes and write them into the idle LDS buffer -- the ~400 VALU
// instructions per half tile that used to sit between two MFMA phases), waves 0-3 are the MATRIX waves (one per SIMD: transposing
// LDS reads and MFMAs only, fragments requested one halo row ahead of the MFMAs that consume them).  A matrix wave and a loader
// share each SIMD, so the split's VALU work and the global-load latency run under the MFMAs instead of between them.  One
// workgroup barrier per half tile; LDS: two buffers of {Xh, Xl, Gh, Gl} = 115 KB, one workgroup per CU, 256 registers per wave.
// Same tile lists, same order of every partial sum's terms as wgrad_split_kernel (bit-identical partials for the same KS).
#ifndef HLA_WGRAD_SPLIT_WS
#define HLA_WGRAD_SPLIT_WS 1
#endif
#ifndef HLA_WS_ABL
#define HLA_WS_ABL 0
#endif
constexpr int wgs_ws_buf_bytes() { return 2 * ((WGS_TH + 2) * HWID + WGS_TH * 32) * WGS_STR; }
constexpr int wgs_ws_lds_bytes() { return 2 * wgs_ws_buf_bytes(); }

static __global__ __launch_bounds__(512, 1) void wgrad_split_ws_kernel(WgradArgs a, WgradSplitExtra sx) {
  typedef f16 H;
  constexpr int STR = WGS_STR, PPX = 16;                   // 16-B fp32 pieces per pixel (64 channels)
  constexpr int XPIX = (WGS_TH + 2) * HWID, GPIX = WGS_TH * 32, KPX = 16, BUFB = wgs_ws_buf_bytes();
  constexpr int oXh = 0, oXl = XPIX * STR, oGh = 2 * XPIX * STR, oGl = 2 * XPIX * STR + GPIX * STR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wv >= 4;                             // wave-uniform role
  const int ks = blockIdx.x, ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const bool first = ci0 < a.C1;
  const float* xsrc = first ? (const float*)a.x1 : (const float*)a.x2;
  const int Cs = first ? a.C1 : a.C2, coff = first ? ci0 : ci0 - a.C1, sh = (first && a.up1) ? 1 : 0;
  const int Hs = a.H >> sh, Ws = a.W >> sh;
  const int gsh = a.g_unpool ? 1 : 0, Hg = a.H >> gsh, Wg = a.W >> gsh;
  const WgTiles tl(a.dyn, a.dyn_desc, a.H, a.W, a.row_begin, a.tiles_x, a.tiles_y, a.ntile, a.B, a.g_unpool ? 1 : 0);
  // the half-tile walk of this k-slice: (tile, half 0), (tile, half 1), next tile; the lower half of a tile at the image's last
  // rows may be empty.  Both roles step through it identically (they meet at one barrier per half tile).
  const int n2 = 2 * tl.ntile;
  auto half_y0 = [&](int t2, int& b, int& x0, int& gx0, int& gx1) {
    int y0;
    tl.origin(t2 >> 1, b, y0, x0, gx0, gx1);
    return y0 + (t2 & 1) * WGS_TH;
  };
  auto next_t2 = [&](int t2) {                             // the next non-empty half tile after t2, or -1
    for (;;) {
      t2 += (t2 & 1) ? 2 * a.KS - 1 : 1;
      if (t2 >= n2) return -1;
      int b, x0, g0, g1;
      if (half_y0(t2, b, x0, g0, g1) < a.H) return t2;
    }
  };
  int t_first = 2 * ks;
  if (t_first >= n2) t_first = -1;                         // (the upper half of a listed tile is never empty)

  if (loader) {
    // ---------------- loader waves: global -> registers (two half tiles in flight) -> split -> LDS planes of the idle buffer
    const int tl_ = t - 256, part = tl_ % PPX, pix0 = tl_ / PPX;
    constexpr int NX = (XPIX * PPX + 255) / 256, NG = GPIX * PPX / 256, PSTEP = 256 / PPX;
    static_assert(GPIX * PPX % 256 == 0, "gradient tile pieces per thread");
    unsigned mx = 0, mg = 0;
    const unsigned* ax = first ? sx.amax_x1 : sx.amax_x2;
    for (int b = 0; b < a.B; ++b) { mx = max(mx, ax[b]); mg = max(mg, sx.amax_g[b]); }
    const float s_x = split_scale(mx), s_g = split_scale(mg);
    constexpr int OOB = (int)0x80000000;
    const size_t xs_bytes = (size_t)Hs * Ws * Cs * 4, gs_bytes = (size_t)Hg * Wg * a.Cout * 4;
    auto rsrc = [](const void* base, size_t bytes) __attribute__((always_inline)) {
      const unsigned long long p = (unsigned long long)base;
      const void* pu = (const void*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p));
      return __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    struct Stage { u32x4 xr[NX]; u32x4 gr[NG]; unsigned gid[NG]; int ypar, xpar; };
    // A piece's place in the half tile is fixed for the thread's life: its halo row / column (hy, hx) and, relative to the tile's
    // origin, the COLUMN part of its byte offset in the source map (the origin's column is a multiple of 32 and splits off exactly,
    // also through the nearest-upsample shift; its row may be odd and does not).  Per half tile a piece then costs a shift-multiply-add,
    // the bounds compares and a select.
    int xrel[NX], xhyx[NX], grel[NG];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int pix = pix0 + k * PSTEP, hy = pix / HWID, hx = pix - hy * HWID;
      xhyx[k] = pix < XPIX ? (hy << 8) | hx : (200 << 8);                          // (a row no image has: never valid)
      xrel[k] = (((hx - 1) >> sh) * Cs + coff + part * 4) * 4;      // column part (arithmetic shift: floor; x0 is a multiple of 32)
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int pix = pix0 + k * PSTEP;
      grel[k] = (((pix / 32) >> gsh) * Wg + ((pix % 32) >> gsh)) * a.Cout + co0 + part * 4;
    }
    // every load is issued unconditionally (a half tile that does not exist gets out-of-range offsets everywhere and reads
    // zeros): straight-line code, so the compiler can COUNT the loads in flight and wait for one stage while the next one's are
    // still outstanding.  (A branch around the loads makes the wait a vmcnt(0).)
    auto issue = [&](int t2, Stage& S) __attribute__((always_inline)) {
      int b = 0, x0 = 0, gx0 = 0, gx1 = 0, y0 = 0;
      const bool live = t2 >= 0;
      if (live) y0 = half_y0(t2, b, x0, gx0, gx1);
      S.ypar = y0; S.xpar = x0;
      const __amdgpu_buffer_rsrc_t rx = rsrc((const char*)xsrc + (size_t)b * xs_bytes, xs_bytes);
      const __amdgpu_buffer_rsrc_t rg = rsrc((const char*)a.g + (size_t)b * gs_bytes, gs_bytes);
      const __amdgpu_buffer_rsrc_t ri = rsrc(a.g_unpool ? a.g_unpool + (size_t)b * (gs_bytes / 4) : (const unsigned char*)a.g, a.g_unpool ? gs_bytes / 4 : 0);
      // uniform: the tile origin's offset (y0 is even and x0 a multiple of 32, so the upsample / unpool shifts split off), and the
      // valid ranges of hy / hx (input halo) and of the gradient tile's rows / columns
      // source row of halo row hy: (y0 - 1 + hy) >> sh = ((y0 - 1) >> sh) + ((hy + ((y0 - 1) & sh)) >> sh) -- the static first row of a
      // trimmed ground launch may be odd, so the origin's parity under the upsample shift is carried (ypar); columns split off
      // exactly (x0 is a multiple of 32)
      const int ylo = y0 - 1, ypar = ylo & sh, rowb = Ws * Cs * 4;
      const int xbase = __builtin_amdgcn_readfirstlane(((ylo >> sh) * Ws + (x0 >> sh)) * Cs * 4);
      const int gbase = __builtin_amdgcn_readfirstlane(((y0 >> gsh) * Wg + (x0 >> gsh)) * a.Cout);
      const int hy_lo = live ? max(0, 1 - y0) : 255, hy_hi = a.H - y0 + 1, hx_lo = max(0, 1 - x0), hx_hi = a.W - x0 + 1;
      const int gy_hi = live ? a.H - y0 : 0, gx_lo = gx0 - x0, gx_hi = gx1 - x0;
#pragma unroll
      for (int k = 0; k < NX; ++k) {                   // input halo tile, zero outside the image
        const int hy = xhyx[k] >> 8, hx = xhyx[k] & 255;
        const bool ok = hy >= hy_lo && hy < hy_hi && hx >= hx_lo && hx < hx_hi;
        S.xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xbase + ((hy + ypar) >> sh) * rowb + xrel[k] : OOB, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) {                   // output-gradient tile (virtual unpool: + the forward argmax)
        const int pix = pix0 + k * PSTEP, py = pix / 32, px = pix % 32;
        const bool ok = py < gy_hi && px >= gx_lo && px < gx_hi;
        const int e0 = ok ? gbase + grel[k] : OOB;
        S.gr[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e0 * 4 : OOB, 0, 0);
        S.gid[k] = __builtin_amdgcn_raw_buffer_load_b32(ri, e0, 0, 0);      // (no unpool: a zero-sized descriptor, reads 0)
      }
    };
    auto commit = [&](const Stage& S, int buf) __attribute__((always_inline)) {
      char* base = smem + buf * BUFB;
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        const int pix = pix0 + k * PSTEP;
        uint2 hi, lw;
#if HLA_WS_ABL == 2      // timing-only: no split arithmetic
        hi = make_uint2(S.xr[k].x, S.xr[k].y); lw = make_uint2(S.xr[k].z, S.xr[k].w);
#else
        split4(__uint_as_float(S.xr[k].x), __uint_as_float(S.xr[k].y), __uint_as_float(S.xr[k].z), __uint_as_float(S.xr[k].w), s_x, hi, lw);
#endif
        if (pix < XPIX) { *(uint2*)(base + oXh + pix * STR + part * 8) = hi; *(uint2*)(base + oXl + pix * STR + part * 8) = lw; }
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int pix = pix0 + k * PSTEP;
        float e0 = __uint_as_float(S.gr[k].x), e1 = __uint_as_float(S.gr[k].y), e2 = __uint_as_float(S.gr[k].z), e3 = __uint_as_float(S.gr[k].w);
        if (a.g_unpool) {                              // keep the elements whose forward argmax is this (y&1, x&1)
          const unsigned pos = (((S.ypar + pix / 32) & 1) << 1) | ((S.xpar + pix % 32) & 1);
          if ((S.gid[k] & 0xff) != pos) e0 = 0.f;
          if (((S.gid[k] >> 8) & 0xff) != pos) e1 = 0.f;
          if (((S.gid[k] >> 16) & 0xff) != pos) e2 = 0.f;
          if ((S.gid[k] >> 24) != pos) e3 = 0.f;
        }
        uint2 hi, lw;
#if HLA_WS_ABL == 2
        hi = make_uint2(__float_as_uint(e0), __float_as_uint(e1)); lw = make_uint2(__float_as_uint(e2), __float_as_uint(e3));
#else
        split4(e0, e1, e2, e3, s_g, hi, lw);
#endif
        *(uint2*)(base + oGh + pix * STR + part * 8) = hi; *(uint2*)(base + oGl + pix * STR + part * 8) = lw;
      }
    };
    Stage A, Bq;
    int ta = t_first, tb = ta >= 0 ? next_t2(ta) : -1;
    issue(ta, A);
    issue(tb, Bq);
    commit(A, 0);
    __syncthreads();                                     // barrier 0: buffer 0 holds the first half tile
    int cur = 0;
#if HLA_WS_ABL == 3      // timing only: the loaders do nothing but keep the barrier count
    while (ta >= 0) { __syncthreads(); ta = next_t2(ta); }
    return;
#endif
    // at the top: the matrix waves work on half tile `ta` in buffer `cur`; Bq holds (in flight) the loads of `tb`; A is free
    while (ta >= 0) {
      int tc = tb >= 0 ? next_t2(tb) : -1;
      issue(tc, A);
      commit(Bq, cur ^ 1);
      __syncthreads();
      ta = tb; tb = tc; cur ^= 1;
      if (ta < 0) break;
      tc = tb >= 0 ? next_t2(tb) : -1;
      issue(tc, Bq);
      commit(A, cur ^ 1);
      __syncthreads();
      ta = tb; tb = tc; cur ^= 1;
    }
    return;
  }

  // ---------------- matrix waves
  const int ct = wv >> 1, it = wv & 1;
  const bool want_bias = a.bpart && blockIdx.y == 0 && it == 0;
  unsigned mx = 0, mg = 0;
  {
    const unsigned* ax = first ? sx.amax_x1 : sx.amax_x2;
    for (int b = 0; b < a.B; ++b) { mx = max(mx, ax[b]); mg = max(mg, sx.amax_g[b]); }
  }
  const float s_x = split_scale(mx), s_g = split_scale(mg);
  f32x16 acc[9], accb;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;
  const uint4 ones = frag_ones<f16>();
  __syncthreads();                                       // barrier 0
  int cur = 0;
  for (int t2 = t_first; t2 >= 0; t2 = next_t2(t2)) {
    const char* base = smem + cur * BUFB;
    const char *Xh = base + oXh, *Xl = base + oXl, *Gh = base + oGh, *Gl = base + oGl;
    // A flat walk over the half tile's 8 steps (K-step kk = 16 pixels, halo row rho): the G fragments of a K-step (two rows, hi and
    // lo) stay resident while its four halo rows pass; the X fragments of step s + 1 -- and, in a K-step's last row, the G
    // fragments of the next one -- are REQUESTED AT THE TOP of step s, ahead of its 9-18 MFMAs (this wave has the SIMD's matrix pipe
    // to itself: nothing else covers the LDS latency; left to the scheduler the requests sank to just before the step's last MFMA
    // and every step started with an LDS round trip: 0.77 of the pipe with the loaders idle).  A fragment feeds the up to two taps
    // ky that use it.  Term order per accumulator as in wgrad_split_kernel: (hi hi, lo hi, hi lo) per (kk, rho, kx, ky).
    constexpr int NSTEP = (32 / KPX) * (WGS_TH + 2);
    uint4 Ah[2][WGS_TH], Al[2][WGS_TH], Bh[2][3], Bl[2][3];
#if HLA_WS_ABL != 1
#pragma unroll
    for (int r = 0; r < WGS_TH; ++r) {
      Ah[0][r] = frag_kmajor<H>(Gh, STR, r * 32, ct * 32, lane);
      Al[0][r] = frag_kmajor<H>(Gl, STR, r * 32, ct * 32, lane);
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      Bh[0][kx] = frag_kmajor<H>(Xh, STR, kx, it * 32, lane);
      Bl[0][kx] = frag_kmajor<H>(Xl, STR, kx, it * 32, lane);
    }
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int kk = st / (WGS_TH + 2), rho = st % (WGS_TH + 2);
      if (st + 1 < NSTEP) {
        const int kn = (st + 1) / (WGS_TH + 2), rn = (st + 1) % (WGS_TH + 2);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          Bh[(st + 1) & 1][kx] = frag_kmajor<H>(Xh, STR, rn * HWID + kx + kn * KPX, it * 32, lane);
          Bl[(st + 1) & 1][kx] = frag_kmajor<H>(Xl, STR, rn * HWID + kx + kn * KPX, it * 32, lane);
        }
        if (rn == 0) {
#pragma unroll
          for (int r = 0; r < WGS_TH; ++r) {
            Ah[kn & 1][r] = frag_kmajor<H>(Gh, STR, r * 32 + kn * KPX, ct * 32, lane);
            Al[kn & 1][r] = frag_kmajor<H>(Gl, STR, r * 32 + kn * KPX, ct * 32, lane);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (rho == 0 && want_bias) {
#pragma unroll
        for (int r = 0; r < WGS_TH; ++r) { mma16<H>(accb, Ah[kk & 1][r], ones); mma16<H>(accb, Al[kk & 1][r], ones); }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int r = rho - ky;
          if (r >= 0 && r < WGS_TH) {
            mma16<H>(acc[ky * 3 + kx], Ah[kk & 1][r], Bh[st & 1][kx]);
            mma16<H>(acc[ky * 3 + kx], Al[kk & 1][r], Bh[st & 1][kx]);
            mma16<H>(acc[ky * 3 + kx], Ah[kk & 1][r], Bl[st & 1][kx]);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    __syncthreads();                                     // the loaders have filled the other buffer; this one is free
    cur ^= 1;
  }
  // D[i = co][j = ci]: lane -> ci = ci0 + it*32 + (lane&31); reg r -> co = co0 + ct*32 + (r&3) + 8(r>>2) + 4(lane>>5)
  const float inv = 1.f / (s_x * s_g), invg = 1.f / s_g;
  const int ci = ci0 + it * 32 + (lane & 31), g5 = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * g5;
    float* o = a.part + (((size_t)ks * a.Cout + co) * a.Cin + ci) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) o[tap] = acc[tap][r] * inv;
    if (want_bias && (lane & 31) == 0) a.bpart[(size_t)ks * a.Cout + co] = accb[r] * invg;
  }
}

// per-sample max |x| of an fp32 map (fp32 bit pattern, atomicMax into a zeroed word): the scale of a gradient map that no
// convolution epilogue produced (the L2-norm backward's outputs, with the confidence heads' contribution added)
static __global__ __launch_bounds__(256) void absmax_map_kernel(const float* __restrict__ x, size_t per_sample, size_t skip, int nblk,
                                                                unsigned* __restrict__ amax) {
  const int b = blockIdx.x / nblk, k = blockIdx.x % nblk;
  const float4* p = (const float4*)(x + (size_t)b * per_sample);
  float m = 0.f;
  for (size_t i = skip / 4 + (size_t)k * 256 + threadIdx.x; i < per_sample / 4; i += (size_t)nblk * 256) {
    const float4 v = p[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  m = wave_max_f32(m);
  if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __hip_atomic_load(amax + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(amax + b, __float_as_uint(m));
}

// conv0: dW0[co][k = c*9+tap] over the NCHW fp32 input (k padded to 32 as one "ci tile").
struct Wgrad0Args {
  const float* x;        // [B,3,H,W], channel planes x_plane elements apart
  size_t x_plane;
  const void* g;         // d(loss)/d(conv0 pre-activation) NHWC T [B,H,W,64]
  float* part;           // [KS][2 row-halves][64][32]
  float* bpart;          // [KS][2][64]
  int B, H, W, tiles_x, tiles_y, ntile, KS;
  int row_begin;         // first pixel row that carries gradient
  const int* dyn;        // as WgradArgs::dyn / dyn_desc
  int dyn_desc;
};

template <typename T>
__global__ __launch_bounds__(256) void wgrad0_kernel(Wgrad0Args a) {
  constexpr int EPL = 16 / sizeof(T), STR = wg_stride<T>(), PPX = 64 * (int)sizeof(T) / 16, KPX = KStep<T>::PX;
  constexpr int IW = 48;   // plane row pitch (32 + 2 halo + K-step overrun)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Gs = smem;
  float* in = (float*)(smem + WG_TH * 32 * STR);     // [3][WG_TH+2][IW]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, ct = wv & 1, half = wv >> 1;
  const int ks = blockIdx.x;
  f32x16 acc, accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
  const uint4 ones = frag_ones<T>();
  const int j = lane & 31, g5 = lane >> 5;            // B operand: column j = k index (c, ky, kx)
  const int jc = j < 27 ? j / 9 : 0, jky = (j % 9) / 3, jkx = j % 3;
  const WgTiles tl(a.dyn, a.dyn_desc, a.H, a.W, a.row_begin, a.tiles_x, a.tiles_y, a.ntile, a.B, 0);
  for (int tile = ks; tile < tl.ntile; tile += a.KS) {
    int b, y0, x0, gx0, gx1;
    tl.origin(tile, b, y0, x0, gx0, gx1);
    __syncthreads();
    // (every load of a tile is requested before the first LDS write: as two rolled loops this was 14 dependent
    //  load -> wait -> write round trips per tile against a few microseconds of MFMA work -- the kernel ran at 68 TF)
    constexpr int NI = (3 * (WG_TH + 2) * IW + 255) / 256, NG = WG_TH * 32 * PPX / 256;
    static_assert(WG_TH * 32 * PPX % 256 == 0, "gradient tile pieces per thread");
    float vi[NI];
    uint4 vg[NG];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int e = t + it * 256;
      const int c = e / ((WG_TH + 2) * IW), r = e % ((WG_TH + 2) * IW), iy = r / IW, ix = r % IW;
      const int y = y0 - 1 + iy, x = x0 - 1 + ix;
      vi[it] = 0.f;
      if (e < 3 * (WG_TH + 2) * IW && ix < HWID && y >= 0 && y < a.H && x >= 0 && x < a.W)
        vi[it] = a.x[((size_t)b * 3 + c) * a.x_plane + (size_t)y * a.W + x];
    }
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int e = t + it * 256;
      const int pix = e / PPX, part = e % PPX;
      const int y = y0 + pix / 32, x = x0 + pix % 32;
      vg[it] = make_uint4(0, 0, 0, 0);
      if (y < a.H && x >= gx0 && x < gx1)
        vg[it] = *(const uint4*)((const T*)a.g + (((size_t)b * a.H + y) * a.W + x) * 64 + part * EPL);
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int e = t + it * 256;
      if (e < 3 * (WG_TH + 2) * IW) in[e] = vi[it];
    }
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int e = t + it * 256;
      *(uint4*)(Gs + (e / PPX) * STR + (e % PPX) * 16) = vg[it];
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < WG_TH / 2; ++rr) {
      const int r = half * (WG_TH / 2) + rr;
#pragma unroll
      for (int kk = 0; kk < 32 / KPX; ++kk) {
        const uint4 A = frag_kmajor<T>(Gs, STR, r * 32 + kk * KPX, ct * 32, lane);
        T e[EPL];
        const float* row = in + (jc * (WG_TH + 2) + r + jky) * IW + jkx + kk * KPX;
#pragma unroll
        for (int jj = 0; jj < EPL; ++jj) {
          // bf16: k = 8*g5 + jj ; fp32: k = 2*jj + g5   (pixel offset inside the K-step, see KStep)
          const int k = sizeof(T) == 2 ? 8 * g5 + jj : 2 * jj + g5;
          e[jj] = (T)(j < 27 ? row[k] : 0.f);
        }
        mma16<T>(acc, A, __builtin_bit_cast(uint4, e));
        mma16<T>(accb, A, ones);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * g5;
    a.part[(((size_t)ks * 2 + half) * 64 + co) * 32 + j] = acc[r];
    if (j == 0) a.bpart[((size_t)ks * 2 + half) * 64 + co] = accb[r];
  }
}

// out[i] = sum_k part[k][i]  (fixed order: deterministic).  Generic 2-D gather: element i = (row, col) with col < out_inner,
// input row pitch in_inner (conv0: 32 -> 27).  16 columns x 16 k-lanes per block: the bias / conv0 / conf-head reductions
// have few columns and up to 2048 partial rows, so the parallelism has to come from k (a single thread walking all of
// K costs one load latency per row: 0.5 ms for K = 512).  Launch with (n + 15) / 16 blocks.
static __global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              size_t n, int K, int in_stride_inner, int out_inner, int in_inner) {
  __shared__ float sh[16][17];
  const int col = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + col;
  float s = 0.f;
  if (i < n) {
    const size_t src = (i / out_inner) * in_inner + i % out_inner;
#pragma unroll 8
    for (int k = kl; k < K; k += 16) s += part[(size_t)k * in_stride_inner + src];
  }
  sh[kl][col] = s;
  __syncthreads();
  if (kl == 0 && i < n) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += sh[k][col];
    out[i] = s;
  }
}

// contiguous case, n % 4 == 0: 32 float4 columns x 8 k-lanes per block; each k-lane sums its rows in order, the 8 lane
// sums are then added in order (deterministic), with 8x more loads in flight than the generic kernel
static __global__ __launch_bounds__(256) void reduce_partials4_kernel(const float4* __restrict__ part, float4* __restrict__ out,
                                                                      size_t n4, int K) {
  __shared__ float4 sh[8][32];
  const int col = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + col;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
#pragma unroll 4
    for (int k = kl; k < K; k += 8) {
      const float4 v = part[(size_t)k * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sh[kl][col] = s;
  __syncthreads();
  if (kl == 0 && i < n4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) { const float4 v = sh[k][col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    out[i] = s;
  }
}

// first level of a two-level fixed-order reduction over MANY partial rows (the fused conv0 weight gradient: one [64][32] row per
// data-gradient workgroup, up to 32768 of them): slab s = blockIdx.x sums rows s, s + gridDim.x, ... in that order into out2[s].
// K on the device for a data-dependent launch: n_live (the first int of its ConvDyn) x B workgroups wrote a row.
static __global__ __launch_bounds__(256) void reduce_rows_kernel(const float4* __restrict__ part, float4* __restrict__ out2, int n4, int K,
                                                                 const int* __restrict__ dyn_nlive, int B) {
  const int Kd = dyn_nlive ? *dyn_nlive * B : K, s = blockIdx.x, ns = gridDim.x;
  for (int c = threadIdx.x; c < n4; c += 256) {
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = s; k < Kd; k += ns) {
      const float4 v = part[(size_t)k * n4 + c];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    out2[(size_t)s * n4 + c] = sum;
  }
}

// ---------------------------------------------------------------------------------------------
// L2_norm backward: y = a*x with a = 1/||x||  =>  dx = a*dy - a^3 * (x . dy) * x      (per sample)
static __global__ __launch_bounds__(256) void l2bwd_dot_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        double* __restrict__ part, size_t per_sample, int nblk) {
  __shared__ double sh[4];
  const int b = blockIdx.x / nblk, k = blockIdx.x % nblk;
  const float4* px = (const float4*)(x + (size_t)b * per_sample);
  const float4* pd = (const float4*)(dy + (size_t)b * per_sample);
  float s = 0.f;
  double sd = 0.0;
  int cnt = 0;
  for (size_t i = (size_t)k * 256 + threadIdx.x; i < per_sample / 4; i += (size_t)nblk * 256) {
    const float4 u = px[i], v = pd[i];
    s += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    if (++cnt == 64) { sd += (double)s; s = 0.f; cnt = 0; }
  }
  sd += (double)s;
  sd = wave_sum_f64(sd);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sd;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)b * nblk + k] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// `rowiv` (optional, ORTHO only): per map row the column interval of the pixels where dy is not exactly zero, over the whole
// batch, kept as atomicMax targets {-lo, hi} (memset to 0x80 bytes = "nothing yet") -- the seed of the data-dependent trimming.
template <typename T, bool ORTHO>
__global__ __launch_bounds__(256) void l2bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const double* __restrict__ part, const double* __restrict__ inv,
                                                          T* __restrict__ out, size_t per_sample, int nblk,
                                                          int* __restrict__ rowiv = nullptr, int C = 1, int Wl = 1,
                                                          size_t skip = 0,        // floats at the start of a sample that are neither
                                                                                 // read nor written (the caller's static first rows)
                                                          unsigned* __restrict__ amax = nullptr) {   // split mode: [B] atomicMax target
                                                                                 // for max |out| per sample (the scale the map's consumer
                                                                                 // uses), instead of a separate pass over the map
  const int b = blockIdx.x / nblk, k = blockIdx.x % nblk;
  float mx = 0.f;
  double dot = 0.0;
  if (!ORTHO)
    for (int i = 0; i < nblk; ++i) dot += part[(size_t)b * nblk + i];
  const double al = inv[b];
  const float c1 = (float)al, c3 = (float)(al * al * al * dot);
  const float4* px = (const float4*)(x + (size_t)b * per_sample);
  const float4* pd = (const float4*)(dy + (size_t)b * per_sample);
  T* po = out + (size_t)b * per_sample;
  if (ORTHO && rowiv) {
    // Row tracking: the block takes a CONTIGUOUS slice of the sample (a few map rows), collects their intervals in LDS and
    // touches the global counters once per row at the end.  (History: one filter read + atomic per non-zero vector made the
    // counters an L2 hot spot -- 0.18 -> 2.6 ms for the three maps; one per wave and iteration still cost +0.45 ms.)
    constexpr int MAXR = 64;
    __shared__ int siv[2 * MAXR];
    const size_t n4 = per_sample / 4, c = (n4 + nblk - 1) / nblk, start = (size_t)k * c, end = start + c < n4 ? start + c : n4;
    const int r0 = (int)(start * 4 / (unsigned)C) / Wl, r1 = end > start ? (int)((end * 4 - 1) / (unsigned)C) / Wl : r0;
    const bool local = r1 - r0 < MAXR;                                    // block-uniform
    for (int e = threadIdx.x; e < 2 * MAXR; e += 256) siv[e] = INT_MIN;
    __syncthreads();
    for (size_t i = start + threadIdx.x; i < end; i += 256) {
      const float4 v = pd[i];
      store4(po + i * 4, c1 * v.x, c1 * v.y, c1 * v.z, c1 * v.w);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(c1 * v.x), fabsf(c1 * v.y))), fmaxf(fabsf(c1 * v.z), fabsf(c1 * v.w)));
      const bool nz = v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
      const unsigned long long m = __ballot(nz);
      if (m) {
        // a wave's 64 vectors are consecutive in memory (1-4 pixels), so they almost always lie in ONE row: reduce the columns
        // across the wave and let one lane update that row
        const int pix = (int)(i * 4 / (unsigned)C), py = pix / Wl, pxx = pix - py * Wl;
        const int first = __ffsll((long long)m) - 1;
        const int py0 = __shfl(py, first, 64);
        int nlo = nz ? -pxx : INT_MIN, hi = nz ? pxx + 1 : INT_MIN, row = py;
        if (__ballot(nz && py != py0) == 0) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) { nlo = max(nlo, __shfl_xor(nlo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
          if ((int)(threadIdx.x & 63) != first) nlo = hi = INT_MIN;
          row = py0;
        }
        if (nlo != INT_MIN) {
          if (local) { atomicMax(&siv[2 * (row - r0)], nlo); atomicMax(&siv[2 * (row - r0) + 1], hi); }
          else { atomicMax(rowiv + 2 * row, nlo); atomicMax(rowiv + 2 * row + 1, hi); }
        }
      }
    }
    __syncthreads();
    if (local && (int)threadIdx.x <= r1 - r0 && siv[2 * threadIdx.x + 1] != INT_MIN) {
      int* r = rowiv + 2 * (r0 + threadIdx.x);
      // (the plain reads only filter: a stale value costs one redundant atomic, never a missed one)
      if (siv[2 * threadIdx.x] > __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(r, siv[2 * threadIdx.x]);
      if (siv[2 * threadIdx.x + 1] > __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(r + 1, siv[2 * threadIdx.x + 1]);
    }
    if (amax) {
      mx = wave_max_f32(mx);
      if ((threadIdx.x & 63) == 0 && mx > 0.f && __float_as_uint(mx) > __hip_atomic_load(amax + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(amax + b, __float_as_uint(mx));      // (the plain read only filters)
    }
    return;
  }
  for (size_t i = skip / 4 + (size_t)k * 256 + threadIdx.x; i < per_sample / 4; i += (size_t)nblk * 256) {
    const float4 v = pd[i];
    if (ORTHO) {       // HLA_VGG_BWD_SCALE_INVARIANT: x . dy = 0 analytically, dx = dy / ||x||; x is not read
      store4(po + i * 4, c1 * v.x, c1 * v.y, c1 * v.z, c1 * v.w);
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(c1 * v.x), fabsf(c1 * v.y))), fmaxf(fabsf(c1 * v.z), fabsf(c1 * v.w)));
    } else {
      const float4 u = px[i];
      store4(po + i * 4, c1 * v.x - c3 * u.x, c1 * v.y - c3 * u.y, c1 * v.z - c3 * u.z, c1 * v.w - c3 * u.w);
    }
  }
  if (ORTHO && amax) {
    mx = wave_max_f32(mx);
    if ((threadIdx.x & 63) == 0 && mx > 0.f && __float_as_uint(mx) > __hip_atomic_load(amax + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(amax + b, __float_as_uint(mx));
  }
}

// level 4: gradient w.r.t. the conv2 output = unpool(gradient of the pooled map, forward argmax) + gradient arriving through
// the conv_dec3 skip connection (already ReLU-masked).  One pass at full resolution, 16 B per thread.
template <typename T>
__global__ __launch_bounds__(256) void unpool_add_kernel(const T* __restrict__ gp, const unsigned char* __restrict__ idx,
                                                         const T* __restrict__ skip, T* __restrict__ out, int B, int H, int W) {
  constexpr int EPL = 16 / sizeof(T), C = 64, PPX = C / EPL;
  const size_t n = (size_t)B * H * W * PPX;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int part = (int)(i % PPX);
    const size_t pix = i / PPX;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const size_t b = pix / ((size_t)W * H);
    const size_t e0 = ((b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * C + part * EPL;
    const unsigned pos = ((y & 1) << 1) | (x & 1);
    uint4 g = *(const uint4*)(gp + e0), sk = *(const uint4*)(skip + pix * C + part * EPL);
    T ge[EPL], se[EPL];
    unsigned char id[EPL];
    __builtin_memcpy(ge, &g, 16); __builtin_memcpy(se, &sk, 16); __builtin_memcpy(id, idx + e0, EPL);
#pragma unroll
    for (int k = 0; k < EPL; ++k) ge[k] = (T)((id[k] == pos ? to_f32(ge[k]) : 0.f) + to_f32(se[k]));
    __builtin_memcpy(&g, ge, 16);
    *(uint4*)(out + pix * C + part * EPL) = g;
  }
}

// ---------------------------------------------------------------------------------------------
// Confidence head backward (VGG.py:62-76,160-162):  conf = sigmoid(-s),  s = sigmoid(z),  z = conv3x3(relu(x), w)
//   dz = -d_conf * conf*(1-conf) * s*(1-s)   with s recovered from conf:  e^s = (1-conf)/conf
static __global__ __launch_bounds__(256) void conf_dz_kernel(const float* __restrict__ conf, const float* __restrict__ d_conf,
                                                      float* __restrict__ dz, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float c = conf[i];
    const float s = logf((1.f - c) / c);
    dz[i] = -d_conf[i] * c * (1.f - c) * s * (1.f - s);
  }
}

// One pass over relu(x) [B,H,W,C]:  gx[q,c] += (x>0) * sum_tap w[c,tap]*dz[q-off(tap)]   (data gradient, added into the
// gradient of the raw map) and per-block partial sums of  dw[c,tap] = sum_q relu(x)[q,c]*dz[q-off(tap)].
// G = C/EPL lanes share a pixel; the 256/G pixel slots of a block are folded with shuffles + LDS at the end.
template <typename T>
__global__ __launch_bounds__(256) void conf_bwd_kernel(const T* __restrict__ act, const float* __restrict__ w,
                                                       const float* __restrict__ dz, T* __restrict__ gx,
                                                       float* __restrict__ part, int B, int H, int W, int C) {
  constexpr int EPL = 16 / sizeof(T);
  extern __shared__ float sh[];   // [4][9*C]
  const int G = C / EPL, ppb = 256 / G, gi = threadIdx.x % G;
  float wr[9][EPL], acc[9][EPL];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < EPL; ++k) { wr[t][k] = w[(gi * EPL + k) * 9 + t]; acc[t][k] = 0.f; }
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * ppb + threadIdx.x / G; pix < npix; pix += (size_t)gridDim.x * ppb) {
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    float dzm[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
      dzm[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? dz[(long)pix - ((long)(t / 3 - 1) * W + (t % 3 - 1))] : 0.f;
    }
    const uint4 raw = *(const uint4*)(act + pix * C + gi * EPL);
    uint4 graw = *(const uint4*)(gx + pix * C + gi * EPL);
    T a[EPL], g[EPL];
    __builtin_memcpy(a, &raw, 16);
    __builtin_memcpy(g, &graw, 16);
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const float av = to_f32(a[k]);
      float d = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) { acc[t][k] += av * dzm[t]; d += wr[t][k] * dzm[t]; }
      g[k] = (T)(to_f32(g[k]) + (av > 0.f ? d : 0.f));
    }
    __builtin_memcpy(&graw, g, 16);
    *(uint4*)(gx + pix * C + gi * EPL) = graw;
  }
  // fold the pixel slots: lanes with equal gi inside a wave, then the 4 waves
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      float v = acc[t][k];
      for (int o = G; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      acc[t][k] = v;
    }
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  if (G >= 64 ? true : ln < G) {
    // G == 64 (fp32, C = 256): every lane of the wave owns distinct channels
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < EPL; ++k) sh[wv * 9 * C + ((ln % G) * EPL + k) * 9 + t] = acc[t][k];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 9 * C; e += 256)
    part[(size_t)blockIdx.x * 9 * C + e] = (sh[e] + sh[9 * C + e]) + (sh[2 * 9 * C + e] + sh[3 * 9 * C + e]);
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// Data-dependent trimming.  The satellite branch's incoming gradient is what the LM loop's bilinear taps scattered: for KITTI
// geometry ~10 % of the texels, inside the fan the camera sees (apex at the map centre, opening towards +u).  l2bwd_apply
// records, per ROW of each returned map, the column interval of its non-zero gradient (over the batch); bwd_fan_kernel pushes
// the three interval sets through the layer graph exactly -- a 3x3 conv: union of the three neighbouring rows, one pixel wider;
// the 2x upsample's sum-pool: rows pairwise, halved; an unpool: doubled; a fan-in: union -- and derives, for every gradient map,
// the part its producer WRITES: per band of 8 rows one column interval rounded out to the 32-px tiles (DYN_BANDS).  From those
// it builds the list of live tiles of every data- and weight-gradient launch.  A launch visits only its live tiles and reads a
// source as zero outside the part its producer wrote.  Nothing is promised by the caller and nothing is approximated: outside
// its support a gradient is exactly zero, so no value changes (only the weight-gradient summation order).
enum { DC_10, DC_9U, DC_9S, DC_8, DC_7U, DC_7S, DC_6, DC_5, DC_4, DC_3, DC_2, DC_1, DC_N };
enum { DW_10, DW_9, DW_8, DW_7, DW_6, DW_5, DW_4, DW_3, DW_2, DW_1, DW_0, DW_N };
// gradient maps (their resolution divisor): what a dgrad launch writes / a later launch reads
enum { M_X21, M_D2A, M_X3P, M_X18, M_D1A, M_X8P, M_X15, M_A12, M_A10, M_X8, M_A5, M_X3, M_A0, M_N };
struct DynLayout {
  int seed[3];              // per-row {-lo, hi} atomicMax targets of d_feat[l] (memset to 0x80 bytes = "nothing yet")
  int seed_ints;
  int bands[M_N];           // per map: [ceil(rows / 8)] x {lo, hi}
  int conv_desc[DC_N];      // ConvDyn
  int conv_list[DC_N];
  int wg_desc[DW_N];        // {n_live, list offset, band offset of g or -1}
  int wg_list[DW_N];
  int used;
  int total;                // ints
};
static DynLayout dyn_layout(int H, int W) {
  DynLayout L{};
  int o = 0;
  auto take = [&](int n) { const int r = o; o += (n + 3) & ~3; return r; };
  const int hs[3] = {H / 8, H / 4, H / 2};
  for (int l = 0; l < 3; ++l) L.seed[l] = take(2 * hs[l]);
  L.seed_ints = o;
  const int mdiv[M_N] = {2, 2, 2, 4, 4, 4, 8, 4, 4, 4, 2, 2, 1};
  for (int m = 0; m < M_N; ++m) L.bands[m] = take(2 * ((H / mdiv[m] + 7) / 8));
  const int cdiv[DC_N] = {2, 2, 2, 4, 4, 4, 4, 4, 4, 2, 2, 1};        // resolution divisor of each launch
  for (int i = 0; i < DC_N; ++i) {
    L.conv_desc[i] = take(4);
    L.conv_list[i] = take(((H / cdiv[i] + 7) / 8) * ((W / cdiv[i] + 31) / 32));
  }
  const int wdiv[DW_N] = {2, 2, 4, 4, 4, 4, 4, 2, 2, 1, 1};
  for (int i = 0; i < DW_N; ++i) {
    L.wg_desc[i] = take(4);
    L.wg_list[i] = take(((H / wdiv[i] + WG_TH - 1) / WG_TH) * ((W / wdiv[i] + 31) / 32));
  }
  L.used = take(4);         // [0] = 1 when the last call trimmed
  L.total = o;
  return L;
}

typedef int2 IV;                                    // column interval [x, y); empty = {1 << 29, 0}
__device__ __forceinline__ IV iv_empty() { return make_int2(1 << 29, 0); }
__device__ __forceinline__ bool iv_none(const IV& a) { return a.x >= a.y; }
__device__ __forceinline__ IV iv_union(const IV& a, const IV& b) { return make_int2(min(a.x, b.x), max(a.y, b.y)); }

// one block of 1024 threads, thread = row.  H <= 1024 (the host falls back to the dense walk otherwise).
static __global__ __launch_bounds__(1024) void bwd_fan_kernel(int* __restrict__ dyn, DynLayout L, int H, int W) {
  __shared__ IV A[1024], Bv[1024], C9[512], C7[256];
  const int t = threadIdx.x;
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
  auto seed = [&](int l, int row, int Wn) {
    const int* r = dyn + L.seed[l] + 2 * row;
    if (r[1] < 0) return iv_empty();
    const IV v = make_int2(max(-r[0], 0), min(r[1], Wn));
    return iv_none(v) ? iv_empty() : v;
  };
  auto at = [&](const IV* src, int row, int n) { return (row >= 0 && row < n) ? src[row] : iv_empty(); };
  auto grow = [&](const IV* src, int row, int n, int Wn) {
    IV v = iv_union(iv_union(at(src, row - 1, n), at(src, row, n)), at(src, row + 1, n));
    if (iv_none(v)) return iv_empty();
    return make_int2(max(v.x - 1, 0), min(v.y + 1, Wn));
  };
  auto down2 = [&](const IV* src, int row, int n) {
    const IV v = iv_union(at(src, 2 * row, n), at(src, 2 * row + 1, n));
    return iv_none(v) ? iv_empty() : make_int2(v.x >> 1, (v.y + 1) >> 1);
  };
  auto up2 = [&](const IV* src, int row, int n) {
    const IV v = at(src, row >> 1, n);
    return iv_none(v) ? iv_empty() : make_int2(2 * v.x, 2 * v.y);
  };
  // what the producer of map m writes: per band of 8 rows the support rounded out to 32-px tiles
  auto bands = [&](int m, const IV* src, int n, int Wn) {
    __syncthreads();
    if (t < (n + 7) / 8) {
      IV v = iv_empty();
      for (int r = 0; r < 8; ++r) v = iv_union(v, at(src, 8 * t + r, n));
      if (iv_none(v)) v = make_int2(0, 0);
      else v = make_int2(v.x & ~31, min((v.y + 31) & ~31, Wn));
      dyn[L.bands[m] + 2 * t] = v.x;
      dyn[L.bands[m] + 2 * t + 1] = v.y;
    }
    __syncthreads();
  };
  if (t < H2) A[t] = seed(2, t, W2);                          bands(M_X21, A, H2, W2);
  if (t < H2) Bv[t] = grow(A, t, H2, W2);                     bands(M_D2A, Bv, H2, W2);
  if (t < H2) C9[t] = grow(Bv, t, H2, W2);                    bands(M_X3P, C9, H2, W2);     // conv_dec2.1^T of g_d2a, still at H/2
  if (t < H4) A[t] = iv_union(seed(1, t, W4), down2(C9, t, H2));   bands(M_X18, A, H4, W4);
  if (t < H4) Bv[t] = grow(A, t, H4, W4);                     bands(M_D1A, Bv, H4, W4);
  if (t < H4) C7[t] = grow(Bv, t, H4, W4);                    bands(M_X8P, C7, H4, W4);
  if (t < H8) A[t] = iv_union(seed(0, t, W8), down2(C7, t, H4));   bands(M_X15, A, H8, W8);
  if (t < H4) Bv[t] = up2(A, t, H8);                          __syncthreads();               // virtual unpool of g_x15
  if (t < H4) A[t] = grow(Bv, t, H4, W4);                     bands(M_A12, A, H4, W4);
  if (t < H4) Bv[t] = grow(A, t, H4, W4);                     bands(M_A10, Bv, H4, W4);
  if (t < H4) A[t] = iv_union(grow(Bv, t, H4, W4), C7[t]);    bands(M_X8, A, H4, W4);        // + the x8 skip branch
  if (t < H2) Bv[t] = up2(A, t, H4);                          __syncthreads();
  if (t < H2) A[t] = grow(Bv, t, H2, W2);                     bands(M_A5, A, H2, W2);
  if (t < H2) Bv[t] = iv_union(grow(A, t, H2, W2), C9[t]);    bands(M_X3, Bv, H2, W2);       // + the x3 skip branch
  if (t < H) A[t] = up2(Bv, t, H2);                           __syncthreads();
  if (t < H) Bv[t] = grow(A, t, H, W);                        bands(M_A0, Bv, H, W);
  __threadfence_block();
  __syncthreads();

  // ---- live-tile lists: one wave per launch.  entry = (tile row << 16) | tile column
  const int lane = t & 63, wv = t >> 6;
  // dgrad launch i: out map, resolution divisor, pooled (the sum-pool launches run at twice the out map's resolution),
  // source map (-1: dense, written by l2bwd_apply), add map (-1: none / dense)
  const int c_out[DC_N] = {M_D2A, M_X18, M_X3P, M_D1A, M_X15, M_X8P, M_A12, M_A10, M_X8, M_A5, M_X3, M_A0};
  const int c_div[DC_N] = {2, 2, 2, 4, 4, 4, 4, 4, 4, 2, 2, 1};
  const int c_pool[DC_N] = {0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0};
  const int c_src[DC_N] = {-1, M_D2A, M_D2A, M_X18, M_D1A, M_D1A, M_X15, M_A12, M_A10, M_X8, M_A5, M_X3};
  const int c_add[DC_N] = {-1, -1, -1, -1, -1, -1, -1, -1, M_X8P, -1, M_X3P, -1};
  // wgrad launch i: gradient map, launch resolution divisor, g at half the launch resolution (virtual unpool)
  const int w_g[DW_N] = {M_X21, M_D2A, M_X18, M_D1A, M_X15, M_A12, M_A10, M_X8, M_A5, M_X3, M_A0};
  const int w_div[DW_N] = {2, 2, 4, 4, 4, 4, 4, 2, 2, 1, 1};
  const int w_sh[DW_N] = {0, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0};
  for (int job = wv; job < DC_N + DW_N; job += 16) {
    const bool isw = job >= DC_N;
    const int i = isw ? job - DC_N : job;
    const int div = isw ? w_div[i] : c_div[i], th = isw ? WG_TH : 8;
    const int Hl = H / div, Wl = W / div, tiles_y = (Hl + th - 1) / th, tiles_x = (Wl + 31) / 32;
    const int map = isw ? w_g[i] : c_out[i];
    const int* bd = dyn + L.bands[map];
    int* list = dyn + (isw ? L.wg_list[i] : L.conv_list[i]);
    int base = 0;
    for (int ty0 = 0; ty0 < tiles_y; ty0 += 64) {
      const int ty = ty0 + lane;
      int tx0 = 0, tx1 = 0;
      if (ty < tiles_y) {
        // band of the map this tile row belongs to, and the map -> launch column scale
        const int sh = isw ? w_sh[i] : 0, pool = isw ? 0 : c_pool[i];
        const int band = pool ? (ty * 8 / 2) >> 3 : ((ty * th) >> sh) >> 3;
        const int lo = bd[2 * band], hi = bd[2 * band + 1];
        const int up = pool ? 1 : sh;                        // map columns -> launch columns: << up
        tx0 = (lo << up) / 32;
        tx1 = min(((hi << up) + 31) / 32, tiles_x);
      }
      const int cnt = max(tx1 - tx0, 0);
      int inc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
      int w = base + inc - cnt;
      for (int tx = tx0; tx < tx1; ++tx) list[w++] = (ty << 16) | tx;
      base += __shfl(inc, 63, 64);
    }
    if (lane == 0) {
      int* d = dyn + (isw ? L.wg_desc[i] : L.conv_desc[i]);
      d[0] = base;
      d[1] = isw ? L.wg_list[i] : L.conv_list[i];
      if (isw) {
        d[2] = map == M_X21 ? -1 : L.bands[map];             // g_x21 is dense (l2bwd_apply writes every element)
        d[3] = 0;
      } else {
        d[2] = c_src[i] < 0 ? -1 : L.bands[c_src[i]];
        d[3] = c_add[i] < 0 ? -1 : L.bands[c_add[i]];
      }
    }
  }
}

struct BwdPlan {
  size_t g_x21, g_d2a, g_x18, l2_18, g_d1a, g_x15, l2_15, g_a12, g_a10, g_x8, g_x8p, g_a5, g_x3, g_x3p, g_a0;
  size_t dot, part, bpart, dz, dyn;
  size_t gamax;                                 // split mode: [kGradAmaxSlots][B] per-sample max |gradient map| (fp32 bits)
  size_t g_x24, g_d3a, g_x2p, g_c2, l2_21;      // level 4 only
  size_t total;
};

// split mode: one per-sample maximum per gradient map that a data- or weight-gradient launch reads
enum { GA_X21 = 0, GA_D2A, GA_X18, GA_X3P, GA_D1A, GA_X15, GA_X8P, GA_A12, GA_A10, GA_X8, GA_A5, GA_X3, GA_A0, GA_X24, GA_D3A,
       GA_X2P, GA_C2, kGradAmaxSlots = 24 };

static std::atomic<unsigned long long> g_wgrad_dma_ok{0};     // per device: wgrad_dma_kernel's LDS request was accepted
// per device (bit = device id; ids >= 64 never set a bit and take the fallback kernels): the wave-specialised weight-gradient
// kernel's LDS request was accepted.  One word per kernel -- wgrad_split_ws_kernel asks for 115 KB, wgrad_ws_kernel<T> for 96 KB:
// a device or partition mode may grant one and refuse the other.
static std::atomic<unsigned long long> g_wgrad_split_ws_ok{0};
static std::atomic<unsigned long long> g_wgrad_ws_ok{0};
static inline unsigned long long dev_bit(int d) { return (d >= 0 && d < 64) ? 1ull << d : 0ull; }
static int wgrad_ksplit(int Cout, int Cin, int ntile, int resident = 512) {
  const int pairs = (Cout / 64) * (Cin / 64);
  int ks = resident / (pairs > 0 ? pairs : 1);     // 512 workgroups = one resident generation (2 per CU)
  if (ks < 1) ks = 1;
  if (ks > ntile) ks = ntile;
  return ks;
}

static void bwd_plan(int B, int H, int W, int dtype, BwdPlan* p, bool level4 = false) {
  const size_t es = hla_elem_bytes(dtype);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += hla_align_up(bytes, 256); return r; };
  const size_t P = (size_t)B * H * W;
  p->g_x21 = take(P / 4 * 64 * es);  p->g_d2a = take(P / 4 * 64 * es);
  p->g_x18 = take(P / 16 * 128 * es); p->l2_18 = take(P / 16 * 128 * es); p->g_d1a = take(P / 16 * 128 * es);
  p->g_x15 = take(P / 64 * 256 * es); p->l2_15 = take(P / 64 * 256 * es);
  p->g_a12 = take(P / 16 * 256 * es); p->g_a10 = take(P / 16 * 256 * es);
  p->g_x8 = take(P / 16 * 128 * es);  p->g_x8p = take(P / 16 * 128 * es);
  p->g_a5 = take(P / 4 * 128 * es);
  p->g_x3 = take(P / 4 * 64 * es);    p->g_x3p = take(P / 4 * 64 * es);
  p->g_a0 = take(P * 64 * es);
  p->dot = take((size_t)B * 64 * sizeof(double));
  size_t maxpart = (size_t)1024 * 2 * 64 * 32 * 4;      // conv0
  const int hs[11] = {1, 1, 2, 2, 4, 4, 4, 4, 4, 2, 2};  // resolution divisor of each layer's output
  for (int l = 1; l < (level4 ? kAllLayers : kPackedLayers); ++l) {
    const int div = l < 11 ? hs[l] : 1;
    const int h = H / div, w = W / div;
    const int ntile = B * ((h + WG_TH - 1) / WG_TH) * ((w + 31) / 32);
    const size_t sz = (size_t)wgrad_ksplit(kLayers[l].cout, kLayers[l].cin, ntile) * kLayers[l].cout * kLayers[l].cin * 9 * 4;
    if (sz > maxpart) maxpart = sz;
  }
  p->part = take(maxpart);
  p->bpart = take((size_t)2048 * 256 * 4);
  p->dz = take((level4 ? P : P / 4) * sizeof(float));
  p->dyn = take((size_t)dyn_layout(H, W).total * sizeof(int));
  p->gamax = take((size_t)kGradAmaxSlots * B * sizeof(unsigned));
  p->g_x24 = p->g_d3a = p->g_x2p = p->g_c2 = p->l2_21 = 0;
  if (level4) {
    p->g_x24 = take(P * 64 * es); p->g_d3a = take(P * 64 * es); p->g_x2p = take(P * 64 * es); p->g_c2 = take(P * 64 * es);
    p->l2_21 = take(P / 4 * 64 * es);
  }
  p->total = o;
}

template <typename T>
void vgg_pack_all_T(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st) {
  if constexpr (Prec<T>::SPLIT) {
    // split mode: (hi, lo) fp16 fragments of s_w * w, transposed; the tail behind the last layer holds the per-layer scales
    // (float[16]) and 32 scratch words for the |w| maxima, as in the forward packing (vgg.hip)
    float* tail = (float*)(packed + packed_offset(kAllLayers, dtype));
    unsigned* scratch = (unsigned*)tail + 32;
    (void)hipMemsetAsync(tail, 0, kPackTailBytes, st);
    SplitPackTable tb{};
    int n = 0;
    for (int l = 1; l < kAllLayers; ++l) {          // conv0 needs no data gradient
      if (l >= kPackedLayers && !prm->w[l]) continue;
      // transposed conv: Cout' = cin, Cin' = cout (pack mode 2 transposes and flips the taps)
      tb.w[n] = prm->w[l]; tb.off[n] = packed_offset(l, dtype); tb.cout[n] = kLayers[l].cin; tb.cin[n] = kLayers[l].cout;
      tb.first[n] = 2; tb.slot[n] = l;
      ++n;
    }
    hipLaunchKernelGGL(absmax_multi_kernel, dim3(64, n), dim3(256), 0, st, tb, scratch, (unsigned*)nullptr);
    hipLaunchKernelGGL(pack_weights_split_multi_kernel, dim3(256, n), dim3(256), 0, st, tb, packed, (const unsigned*)scratch, tail);
  } else {
    PackTable tb{};
    int n = 0;
    for (int l = 1; l < kAllLayers; ++l) {          // conv0 needs no data gradient
      if (l >= kPackedLayers && !prm->w[l]) continue;
      // transposed conv: Cout' = cin, Cin' = cout
      tb.w[n] = prm->w[l]; tb.off[n] = packed_offset(l, dtype); tb.cout[n] = kLayers[l].cin; tb.cin[n] = kLayers[l].cout;
      tb.first[n] = 2;
      ++n;
    }
    hipLaunchKernelGGL((pack_weights_multi_kernel<T>), dim3(256, n), dim3(256), 0, st, tb, packed);
  }
}

template <typename T>
int vgg_backward_t(const float* x, size_t x_plane, const hla_vgg_params* prm, const char* packedT, int dtype, const char* fw,
                          const float* const feat[4], const double* inv_norm, const float* const d_feat[4],
                          const float* const conf[4], const float* const d_conf[4], const hla_vgg_grads* gr, char* bw, const BwdPlan& bp, int B, int H, int W, int flags, int first_row8, hipStream_t st) {
  const bool level4 = bp.g_x24 != 0;
  // conv0's weight gradient inside the epilogue of conv2's data gradient (conv_epilogue_wg0); the two-phase flag keeps the
  // stored map + wgrad0_kernel (A/B and tests; level 4 reads conv2's gradient from a materialised map anyway)
  const bool fuse0 = !level4 && !(flags & (HLA_VGG_BWD_WGRAD_TWO_PHASE | HLA_VGG_BWD_WGRAD0_UNFUSED));
  const int NL = level4 ? 4 : 3;
  VggPlan fp;
  vgg_plan(B, H, W, dtype, true, &fp, level4);
  constexpr int KC = SB / (int)sizeof(T);
  constexpr bool SPLIT = Prec<T>::SPLIT;
  using ET = std::conditional_t<SPLIT, float, T>;      // element type of the stored maps (split mode: fp32): the elementwise kernels
  static HlaPerDeviceOnce attr_once;
  HLA_CHECK_HIP(attr_once.run([] {
    if constexpr (SPLIT) {
      // (the wave-specialised form needs 115 KB of dynamic LDS: where the device refuses it, the launches below fall back to
      //  wgrad_split_kernel -- ws_ok, per device)
      if (HLA_WGRAD_SPLIT_WS) {
        int d = 0;
        (void)hipGetDevice(&d);
        const bool ok = hipFuncSetAttribute((const void*)wgrad_split_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wgs_ws_lds_bytes()) == hipSuccess;
        if (ok) g_wgrad_split_ws_ok.fetch_or(dev_bit(d)); else (void)hipGetLastError();
      }
      return hipFuncSetAttribute((const void*)wgrad_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wgs_lds_bytes());
    }
    else {
      if constexpr (sizeof(T) == 2 && HLA_WGRAD_DMA) {      // 72 KB: refused (a partition mode with a smaller per-workgroup limit) ->
        int d = 0;                                          // the plain launches fall back to wgrad_kernel<T>
        (void)hipGetDevice(&d);
        const bool ok = hipFuncSetAttribute((const void*)wgrad_dma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, wgd_lds_bytes()) == hipSuccess;
        if (ok) g_wgrad_dma_ok.fetch_or(dev_bit(d)); else (void)hipGetLastError();
      }
      if constexpr (sizeof(T) == 2 && HLA_WGRAD_WS) {      // 96 KB of dynamic LDS: refused -> the launches fall back (g_wgrad_ws_ok)
        int d = 0;
        (void)hipGetDevice(&d);
        const bool ok = hipFuncSetAttribute((const void*)wgrad_ws_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_ws_lds_bytes<T>()) == hipSuccess;
        if (ok) g_wgrad_ws_ok.fetch_or(dev_bit(d)); else (void)hipGetLastError();
      }
      return hipFuncSetAttribute((const void*)wgrad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_lds_bytes<T>());
    }
  }));
  auto F = [&](size_t off) { return (const void*)(fw + off); };
  auto G = [&](size_t off) { return (void*)(bw + off); };
  // split mode: per-sample maxima of the forward's activations (recorded by its epilogues) and of this pass's gradient maps
  // (recorded by the data-gradient epilogues, or by absmax_map_kernel for the maps no convolution produced); the transposed
  // packing's per-layer weight scales sit behind the last packed layer
  const unsigned* fam = (const unsigned*)(fw + fp.amax);
  unsigned* gam = (unsigned*)(bw + bp.gamax);
  auto FA = [&](int slot) { return (SPLIT && slot >= 0) ? fam + (size_t)slot * B : (const unsigned*)nullptr; };
  auto GA = [&](int slot) { return (SPLIT && slot >= 0) ? gam + (size_t)slot * B : (unsigned*)nullptr; };
  const float* wtailT = (const float*)(packedT + packed_offset(kAllLayers, dtype));
  if (SPLIT) HLA_CHECK_HIP(hipMemsetAsync(gam, 0, (size_t)kGradAmaxSlots * B * sizeof(unsigned), st));

  // ---- L2_norm backward of the three returned maps
  const size_t per[4] = {(size_t)(H / 8) * (W / 8) * 256, (size_t)(H / 4) * (W / 4) * 128, (size_t)(H / 2) * (W / 2) * 64,
                         (size_t)H * W * 64};
  // at level 4 x21 also feeds conv_dec3, so its L2-norm gradient goes to a side buffer and is merged by that dgrad's epilogue
  void* l2out[4] = {G(bp.l2_15), G(bp.l2_18), level4 ? G(bp.l2_21) : G(bp.g_x21), G(bp.g_x24)};
  // data-dependent trimming (see bwd_fan_kernel): needs exact zeros outside the support, i.e. the one-pass L2 backward, and no
  // confidence-head gradient (which is dense).  Not combined with the caller's static first rows (the ground branch: every
  // column of its bottom half carries gradient, so there is nothing to gain).
  const bool dynamic = !level4 && (flags & HLA_VGG_BWD_SCALE_INVARIANT) && !(flags & HLA_VGG_BWD_DENSE) && !(conf && d_conf) &&
                       first_row8 == 0 && H <= 1024;
  const DynLayout dl = dyn_layout(H, W);
  int* dynp = (int*)(bw + bp.dyn);
  if (dynamic) HLA_CHECK_HIP(hipMemsetAsync(dynp, 0x80, (size_t)dl.seed_ints * sizeof(int), st));
  HLA_CHECK_HIP(hipMemsetAsync(dynp + dl.used, dynamic ? 1 : 0, sizeof(int), st));
  // Row ranges.  first_row8 = f > 0 is the caller's promise that d_feat[0..2] are zero above rows f / 2f / 4f (the LM loop
  // only ever reads rows h_l/2.. of the ground maps, so that is where its gradient lives).  The gradient of every activation
  // is then exactly zero above a first row that follows from the layer graph -- a 3x3 conv widens the support by one row, a 2x
  // upsample / 2x2 pool halves / doubles it -- and every launch below starts at that row (ConvArgs::row_begin, even where a
  // pool is involved), reading its sources as zero above THEIR first row (src_row_lo / add_row_lo: those rows are never
  // written).  With confidence heads (d_conf) the support of the three raw-map gradients starts one row higher.
  // n_* : first row of a gradient map in its own resolution; all zero (= no trimming) when f == 0.
  const int f = (level4 || !(flags & HLA_VGG_BWD_SCALE_INVARIANT)) ? 0 : first_row8;   // (the two-pass L2 backward leaves
                                                                                       //  rounding noise above the support)
  const int wc = (conf && d_conf) ? 1 : 0;
  auto ev = [](int v) { return v < 0 ? 0 : (v & ~1); };
  auto nn = [](int v) { return v < 0 ? 0 : v; };
  const int n_x21 = f ? 4 * f - wc : 0;                 // H/2
  const int n_d2a = nn(n_x21 - 1);                      // H/2
  const int rb_up18 = ev(n_d2a - 1);                    // H/2 rows of the pool_sum launch that produces g_x18
  const int n_x18 = rb_up18 / 2;                        // H/4   (<= 2f - wc: covers x18's own L2 / conf gradient)
  const int n_x3p = nn(n_d2a - 1);                      // H/2
  const int n_d1a = nn(n_x18 - 1);                      // H/4
  const int rb_up15 = ev(n_d1a - 1);
  const int n_x15 = rb_up15 / 2;                        // H/8
  const int n_x8p = nn(n_d1a - 1);                      // H/4
  const int n_a12 = nn(2 * n_x15 - 1);                  // H/4
  const int n_a10 = nn(n_a12 - 1);
  const int n_x8 = nn(n_a10 - 1);                       // H/4
  const int n_a5 = nn(2 * n_x8 - 1);                    // H/2
  const int n_x3 = nn(n_a5 - 1);                        // H/2
  const int n_a0 = nn(2 * n_x3 - 1);                    // H
  // ... and the L2-norm backward below starts at n_x15 / n_x18 / n_x21 as well: the rows of d_feat[l] above f * 2^l - 2 are neither
  // read nor written (the caller need not even zero them)
  const int l2_row0[4] = {n_x15, n_x18, n_x21, 0};
  bool l2_recorded_amax = false;
  for (int l = 0; l < NL; ++l) {
    int nblk = (int)(per[l] / 4 / 256 / 8);
    nblk = nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
    double* part = (double*)(bw + bp.dot);
    if (flags & HLA_VGG_BWD_SCALE_INVARIANT) {
      const int Cl[4] = {256, 128, 64, 64};
      const size_t skip = (size_t)l2_row0[l] * (W >> (3 - l)) * Cl[l];
      hla_prof_begin(K_ELEMWISE, 0, (double)B * (per[l] - skip) * (4 + sizeof(T)), st);
      // split mode: the finest map's gradient (g_x21 / g_x24) is READ by a convolution, which needs its per-sample maximum; the
      // pass that writes it records it, unless the confidence heads still add into it (then absmax_map below does)
      unsigned* am = (SPLIT && l == NL - 1 && !(conf && d_conf)) ? GA(level4 ? GA_X24 : GA_X21) : (unsigned*)nullptr;
      l2_recorded_amax = l2_recorded_amax || am != nullptr;
      hipLaunchKernelGGL((l2bwd_apply_kernel<ET, true>), dim3(B * nblk), dim3(256), 0, st, feat[l], d_feat[l], part,
                         inv_norm + (size_t)l * B, (ET*)l2out[l], per[l], nblk, dynamic ? dynp + dl.seed[l] : (int*)nullptr,
                         Cl[l], W >> (3 - l), skip, am);
    } else {
      hla_prof_begin(K_ELEMWISE, 0, (double)B * per[l] * (16 + sizeof(T)), st);
      hipLaunchKernelGGL(l2bwd_dot_kernel, dim3(B * nblk), dim3(256), 0, st, feat[l], d_feat[l], part, per[l], nblk);
      hipLaunchKernelGGL((l2bwd_apply_kernel<ET, false>), dim3(B * nblk), dim3(256), 0, st, feat[l], d_feat[l], part,
                         inv_norm + (size_t)l * B, (ET*)l2out[l], per[l], nblk);
    }
    hla_prof_end(st);
  }

  if (dynamic) hipLaunchKernelGGL(bwd_fan_kernel, dim3(1), dim3(1024), 0, st, dynp, dl, H, W);

  // ---- confidence heads (only the ground branch with using_weight=1 ever has d_conf): adds into the raw-map gradients
  if (conf && d_conf) {
    const ET* acts[4] = {(const ET*)(fw + fp.x15r), (const ET*)(fw + fp.x18r), (const ET*)(fw + fp.x21r), (const ET*)(fw + fp.x24r)};
    const int Cs[4] = {256, 128, 64, 64}, hs[4] = {H / 8, H / 4, H / 2, H}, wsz[4] = {W / 8, W / 4, W / 2, W};
    for (int l = 0; l < NL; ++l) {
      if (!d_conf[l]) continue;
      constexpr int EPL = 16 / (int)sizeof(ET);
      const int ppb = 256 / (Cs[l] / EPL);
      const size_t npix = (size_t)B * hs[l] * wsz[l];
      const int grid = (int)((npix + ppb - 1) / ppb < 1024 ? (npix + ppb - 1) / ppb : 1024);
      float* dz = (float*)(bw + bp.dz);
      hla_prof_begin(K_ELEMWISE, 4.0 * 9 * Cs[l] * (double)npix, (double)npix * (3 * Cs[l] * sizeof(ET) + 12), st);
      hipLaunchKernelGGL(conf_dz_kernel, dim3((unsigned)((npix + 255) / 256 < 2048 ? (npix + 255) / 256 : 2048)), dim3(256), 0, st,
                         conf[l], d_conf[l], dz, npix);
      hipLaunchKernelGGL((conf_bwd_kernel<ET>), dim3(grid), dim3(256), 4 * 9 * Cs[l] * sizeof(float), st, acts[l],
                         prm->w[13 + l], (const float*)dz, (ET*)l2out[l], (float*)(bw + bp.part), B, hs[l], wsz[l], Cs[l]);
      const size_t n = (size_t)9 * Cs[l];
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, (const float*)(bw + bp.part),
                         gr->dw[13 + l], n, grid, (int)n, 1, 1);
      hla_prof_end(st);
    }
  }

  // split mode: the scale of the two gradient maps that convolutions READ but no convolution wrote (at level 3: g_x21; l2_18 /
  // l2_15 are only ever added in an epilogue, in fp32)
  auto absmax_map = [&](const void* map, size_t per_sample, size_t skip, int slot) {
    int nblk = (int)(per_sample / 4 / 256 / 8);
    nblk = nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
    hla_prof_begin(K_ELEMWISE, 0, (double)B * (per_sample - skip) * 4, st);
    hipLaunchKernelGGL(absmax_map_kernel, dim3(B * nblk), dim3(256), 0, st, (const float*)map, per_sample, skip, nblk, GA(slot));
    hla_prof_end(st);
  };
  if (SPLIT && !l2_recorded_amax) {
    if (level4) absmax_map(G(bp.g_x24), per[3], 0, GA_X24);
    else absmax_map(G(bp.g_x21), per[2], (size_t)l2_row0[2] * (W / 2) * 64, GA_X21);
  }

  // ---- helpers
  // data gradient of layer l restricted to its input channels [c0, c0+n): a forward conv on the transposed weights
  // (ga_src / ga_out, split mode: the amax slots of the gradient map it reads and of the one it writes)
  bool launch_ok = true;
  auto dgrad = [&](int l, int c0, int n, const void* gsrc, const unsigned char* unpool, int Hout, int Wout, void* out,
                   const void* mask, const void* add, bool pool_sum, int row_begin = 0, int src_lo = 0, int add_lo = 0, int dc = -1,
                   int ga_src = -1, int ga_out = -1, bool fuse_wg0 = false) {
    ConvArgs a{};
    if (fuse_wg0) {      // conv2's data gradient contracted into conv0's weight gradient in its epilogue; `out` receives the partials
      a.wg0_x = x; a.wg0_x_plane = x_plane ? x_plane : (size_t)H * W; a.wg0_part = (float*)out;
    }
    a.amax1 = GA(ga_src); a.amax_out = GA(ga_out); a.wscale = SPLIT ? wtailT + l : nullptr;
    a.dyn = (dynamic && dc >= 0) ? dynp : nullptr; a.dyn_desc = dc >= 0 ? dl.conv_desc[dc] : 0;
    a.src1 = gsrc; a.C1 = kLayers[l].cout; a.unpool_idx = unpool;
    const int nstage = kLayers[l].cout / KC;
    a.wpk = (const uint4*)(packedT + packed_offset(l, dtype)) + (size_t)(c0 / 32) * nstage * 18 * 64;
    a.out_act = fuse_wg0 ? nullptr : out; a.mask_act = mask; a.add_src = add; a.pool_sum = pool_sum ? 1 : 0;
    a.B = B; a.H = Hout; a.W = Wout; a.Cout = n; a.relu_act = 0;
    a.row_begin = row_begin > 0 ? row_begin : 0; a.src_row_lo = src_lo > 0 ? src_lo : 0; a.add_row_lo = add_lo > 0 ? add_lo : 0;
    if (!launch_conv<T, true>(st, a, pool_sum)) launch_ok = false;
  };
  // (fa1 / fa2 / ga, split mode: the amax slots of its input activation(s) and of the gradient map)
  auto wgrad = [&](int l, const void* x1, int C1, const void* x2, int C2, int up1, const void* g, const unsigned char* unpool,
                   int Hout, int Wout, int row_begin = 0, int dw = -1, int fa1 = -1, int fa2 = -1, int ga = -1) {
    WgradArgs a{};
    a.dyn = (dynamic && dw >= 0) ? dynp : nullptr; a.dyn_desc = dw >= 0 ? dl.wg_desc[dw] : 0;
    a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.up1 = up1; a.g = g; a.g_unpool = unpool;
    a.B = B; a.H = Hout; a.W = Wout; a.Cout = kLayers[l].cout; a.Cin = kLayers[l].cin;
    a.row_begin = row_begin > 0 ? row_begin : 0;
    a.tiles_x = (Wout + 31) / 32; a.tiles_y = (Hout - a.row_begin + WG_TH - 1) / WG_TH; a.ntile = B * a.tiles_x * a.tiles_y;
    a.KS = wgrad_ksplit(a.Cout, a.Cin, a.ntile);
    bool ws = false, dma_ok = false;
    {
      int d = 0;
      (void)hipGetDevice(&d);
      dma_ok = (g_wgrad_dma_ok.load() & dev_bit(d)) != 0;
    }
    if constexpr (SPLIT || (sizeof(T) == 2 && HLA_WGRAD_WS)) {      // wave-specialised kernels: one 8-wave workgroup per CU -> 256
      int d = 0;                                                    // workgroups are one resident generation
      (void)hipGetDevice(&d);
      ws = (SPLIT ? HLA_WGRAD_SPLIT_WS : HLA_WGRAD_WS) && ((SPLIT ? g_wgrad_split_ws_ok : g_wgrad_ws_ok).load() & dev_bit(d)) != 0 &&
           !(flags & HLA_VGG_BWD_WGRAD_TWO_PHASE);
      if (ws) a.KS = wgrad_ksplit(a.Cout, a.Cin, a.ntile, 256);
    }
    a.part = (float*)(bw + bp.part);
    a.bpart = (kLayers[l].has_bias && gr->db[l]) ? (float*)(bw + bp.bpart) : nullptr;
    const double P = (double)B * (Hout - a.row_begin) * Wout;
    hla_prof_begin_dyn(K_WGRAD, 2.0 * 9 * a.Cin * a.Cout * P, P * (a.Cin + a.Cout) * sizeof(T), st,
                       a.dyn ? a.dyn + a.dyn_desc : nullptr, a.tiles_x * a.tiles_y);
    if constexpr (SPLIT) {
      WgradSplitExtra ex{FA(fa1), FA(fa2), GA(ga)};
      if (ws) hipLaunchKernelGGL(wgrad_split_ws_kernel, dim3(a.KS, a.Cin / 64, a.Cout / 64), dim3(512), wgs_ws_lds_bytes(), st, a, ex);
      else hipLaunchKernelGGL(wgrad_split_kernel, dim3(a.KS, a.Cin / 64, a.Cout / 64), dim3(256), wgs_lds_bytes(), st, a, ex);
    } else if constexpr (sizeof(T) == 2 && HLA_WGRAD_DMA) {
      if (ws) hipLaunchKernelGGL((wgrad_ws_kernel<T>), dim3(a.KS, a.Cin / 64, a.Cout / 64), dim3(512), wg_ws_lds_bytes<T>(), st, a);
      else if (!unpool && dma_ok) hipLaunchKernelGGL((wgrad_dma_kernel<T>), dim3(a.KS, a.Cin / 64, a.Cout / 64), dim3(256), wgd_lds_bytes(), st, a);
      else hipLaunchKernelGGL((wgrad_kernel<T>), dim3(a.KS, a.Cin / 64, a.Cout / 64), dim3(256), wg_lds_bytes<T>(), st, a);
    } else {
      hipLaunchKernelGGL((wgrad_kernel<T>), dim3(a.KS, a.Cin / 64, a.Cout / 64), dim3(256), wg_lds_bytes<T>(), st, a);
    }
    hla_prof_end(st);
    const size_t n = (size_t)a.Cout * a.Cin * 9;
    hla_prof_begin(K_ELEMWISE, 0, (double)n * 4 * (a.KS + 1), st);
    hipLaunchKernelGGL(reduce_partials4_kernel, dim3((unsigned)((n / 4 + 31) / 32)), dim3(256), 0, st, (const float4*)a.part,
                       (float4*)gr->dw[l], n / 4, a.KS);
    hla_prof_end(st);
    if (a.bpart)
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((a.Cout + 15) / 16), dim3(256), 0, st, a.bpart, gr->db[l], (size_t)a.Cout, a.KS, a.Cout, 1, 1);
  };
  const unsigned char* idx3 = (const unsigned char*)(fw + fp.idx3);
  const unsigned char* idx8 = (const unsigned char*)(fw + fp.idx8);
  const unsigned char* idx15 = (const unsigned char*)(fw + fp.idx15);
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;

  // ---- decoder 3 (VGG.py:153-155, level 4 only; zero-padded to 64 channels, the host slices the weight gradients)
  if (level4) {
    dgrad(12, 0, 64, G(bp.g_x24), nullptr, H, W, G(bp.g_d3a), F(fp.d3a), nullptr, false, 0, 0, 0, -1, GA_X24, GA_D3A);
    wgrad(12, F(fp.d3a), 64, nullptr, 0, 0, G(bp.g_x24), nullptr, H, W, 0, -1, AM_D3A, -1, GA_X24);
    dgrad(11, 0, 64, G(bp.g_d3a), nullptr, H, W, G(bp.g_x21), F(fp.x21r), G(bp.l2_21), true, 0, 0, 0, -1, GA_D3A, GA_X21);     // up(x21) branch
    dgrad(11, 64, 64, G(bp.g_d3a), nullptr, H, W, G(bp.g_x2p), F(fp.x2r), nullptr, false, 0, 0, 0, -1, GA_D3A, GA_X2P);         // x2 skip branch
    wgrad(11, F(fp.x21r), 64, F(fp.x2r), 64, 1, G(bp.g_d3a), nullptr, H, W, 0, -1, AM_X21, AM_X2, GA_D3A);
  }
  // ---- decoder 2 (VGG.py:148-151)
  dgrad(10, 0, 64, G(bp.g_x21), nullptr, H2, W2, G(bp.g_d2a), F(fp.d2a), nullptr, false, n_d2a, n_x21, 0, DC_10, GA_X21, GA_D2A);   // (g_x21 is unwritten above n_x21)
  wgrad(10, F(fp.d2a), 64, nullptr, 0, 0, G(bp.g_x21), nullptr, H2, W2, n_x21, DW_10, AM_D2A, -1, GA_X21);
  dgrad(9, 0, 128, G(bp.g_d2a), nullptr, H2, W2, G(bp.g_x18), F(fp.x18r), G(bp.l2_18), true, rb_up18, n_d2a, 0, DC_9U, GA_D2A, GA_X18);   // up(x18) branch
  dgrad(9, 128, 64, G(bp.g_d2a), nullptr, H2, W2, G(bp.g_x3p), F(fp.x3), nullptr, false, n_x3p, n_d2a, 0, DC_9S, GA_D2A, GA_X3P);          // x3 skip branch
  wgrad(9, F(fp.x18r), 128, F(fp.x3), 64, 1, G(bp.g_d2a), nullptr, H2, W2, n_d2a, DW_9, AM_X18, AM_X3, GA_D2A);
  // ---- decoder 1 (VGG.py:144-146)
  dgrad(8, 0, 128, G(bp.g_x18), nullptr, H4, W4, G(bp.g_d1a), F(fp.d1a), nullptr, false, n_d1a, n_x18, 0, DC_8, GA_X18, GA_D1A);
  wgrad(8, F(fp.d1a), 128, nullptr, 0, 0, G(bp.g_x18), nullptr, H4, W4, n_x18, DW_8, AM_D1A, -1, GA_X18);
  dgrad(7, 0, 256, G(bp.g_d1a), nullptr, H4, W4, G(bp.g_x15), F(fp.x15r), G(bp.l2_15), true, rb_up15, n_d1a, 0, DC_7U, GA_D1A, GA_X15);   // up(x15) branch
  dgrad(7, 256, 128, G(bp.g_d1a), nullptr, H4, W4, G(bp.g_x8p), F(fp.x8), nullptr, false, n_x8p, n_d1a, 0, DC_7S, GA_D1A, GA_X8P);        // x8 skip branch
  wgrad(7, F(fp.x15r), 256, F(fp.x8), 128, 1, G(bp.g_d1a), nullptr, H4, W4, n_d1a, DW_7, AM_X15, AM_X8, GA_D1A);
  // ---- encoder block 2 (VGG.py:136-141); conv14 is followed by the pool (no ReLU in between)
  dgrad(6, 0, 256, G(bp.g_x15), idx15, H4, W4, G(bp.g_a12), F(fp.a12), nullptr, false, n_a12, 2 * n_x15, 0, DC_6, GA_X15, GA_A12);
  wgrad(6, F(fp.a12), 256, nullptr, 0, 0, G(bp.g_x15), idx15, H4, W4, 2 * n_x15, DW_6, AM_A12, -1, GA_X15);
  dgrad(5, 0, 256, G(bp.g_a12), nullptr, H4, W4, G(bp.g_a10), F(fp.a10), nullptr, false, n_a10, n_a12, 0, DC_5, GA_A12, GA_A10);
  wgrad(5, F(fp.a10), 256, nullptr, 0, 0, G(bp.g_a12), nullptr, H4, W4, n_a12, DW_5, AM_A10, -1, GA_A12);
  dgrad(4, 0, 128, G(bp.g_a10), nullptr, H4, W4, G(bp.g_x8), F(fp.x8), G(bp.g_x8p), false, n_x8, n_a10, n_x8p, DC_4, GA_A10, GA_X8);
  wgrad(4, F(fp.x8), 128, nullptr, 0, 0, G(bp.g_a10), nullptr, H4, W4, n_a10, DW_4, AM_X8, -1, GA_A10);
  // ---- encoder block 1
  dgrad(3, 0, 128, G(bp.g_x8), idx8, H2, W2, G(bp.g_a5), F(fp.a5), nullptr, false, n_a5, 2 * n_x8, 0, DC_3, GA_X8, GA_A5);
  wgrad(3, F(fp.a5), 128, nullptr, 0, 0, G(bp.g_x8), idx8, H2, W2, 2 * n_x8, DW_3, AM_A5, -1, GA_X8);
  dgrad(2, 0, 64, G(bp.g_a5), nullptr, H2, W2, G(bp.g_x3), F(fp.x3), G(bp.g_x3p), false, n_x3, n_a5, n_x3p, DC_2, GA_A5, GA_X3);
  wgrad(2, F(fp.x3), 64, nullptr, 0, 0, G(bp.g_a5), nullptr, H2, W2, n_a5, DW_2, AM_X3, -1, GA_A5);
  // ---- encoder block 0
  if (level4) {      // the conv2 output also fed conv_dec3: materialise unpool(g_x3) + skip gradient once
    const size_t n = (size_t)B * H * W * (64 * sizeof(ET) / 16);
    hla_prof_begin(K_ELEMWISE, 0, (double)B * H * W * 64 * sizeof(ET) * 2.25, st);
    hipLaunchKernelGGL((unpool_add_kernel<ET>), dim3((unsigned)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535)), dim3(256), 0, st,
                       (const ET*)G(bp.g_x3), idx3, (const ET*)G(bp.g_x2p), (ET*)G(bp.g_c2), B, H, W);
    hla_prof_end(st);
    if (SPLIT) absmax_map(G(bp.g_c2), (size_t)H * W * 64, 0, GA_C2);
    dgrad(1, 0, 64, G(bp.g_c2), nullptr, H, W, G(bp.g_a0), F(fp.a0), nullptr, false, 0, 0, 0, -1, GA_C2, GA_A0);
    wgrad(1, F(fp.a0), 64, nullptr, 0, 0, G(bp.g_c2), nullptr, H, W, 0, -1, AM_A0, -1, GA_C2);
  } else {
    dgrad(1, 0, 64, G(bp.g_x3), idx3, H, W, G(bp.g_a0), F(fp.a0), nullptr, false, n_a0, 2 * n_x3, 0, DC_1, GA_X3, fuse0 ? -1 : GA_A0, fuse0);
    wgrad(1, F(fp.a0), 64, nullptr, 0, 0, G(bp.g_x3), idx3, H, W, 2 * n_x3, DW_1, AM_A0, -1, GA_X3);
  }
  if (fuse0) {
    // conv0's weight gradient left the data gradient's epilogue as one [64][32] partial per workgroup, in the place of the map
    // (8 KB per 8 x 32-pixel tile against the map's 32 / 64 KB): 1024 slabs, then the generic gather (k < 27: dW0, column 27: db0)
    const int tiles = ((W + 31) / 32) * ((H - (n_a0 > 0 ? n_a0 : 0) + 7) / 8) * B, NS = 1024;
    const bool dyn1 = dynamic;
    float* slab = (float*)(bw + bp.part);
    hla_prof_begin(K_ELEMWISE, 0, (double)tiles * 8192.0 * (dyn1 ? 0.5 : 1.0), st);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(NS), dim3(256), 0, st, (const float4*)G(bp.g_a0), (float4*)slab, 512, tiles,
                       dyn1 ? dynp + dl.conv_desc[DC_1] : (const int*)nullptr, B);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((64 * 27 + 15) / 16), dim3(256), 0, st, (const float*)slab, gr->dw[0], (size_t)64 * 27, NS, 64 * 32, 27, 32);
    if (gr->db[0]) hipLaunchKernelGGL(reduce_partials_kernel, dim3(4), dim3(256), 0, st, (const float*)slab + 27, gr->db[0], (size_t)64, NS, 64 * 32, 1, 32);
    hla_prof_end(st);
  } else {
    Wgrad0Args a{};
    a.x = x; a.x_plane = x_plane ? x_plane : (size_t)H * W; a.g = G(bp.g_a0); a.B = B; a.H = H; a.W = W;
    a.row_begin = level4 ? 0 : n_a0;
    a.dyn = dynamic ? dynp : nullptr; a.dyn_desc = dl.wg_desc[DW_0];
    a.tiles_x = (W + 31) / 32; a.tiles_y = (H - a.row_begin + WG_TH - 1) / WG_TH; a.ntile = B * a.tiles_x * a.tiles_y;
    a.KS = a.ntile < 1024 ? a.ntile : 1024;
    a.part = (float*)(bw + bp.part); a.bpart = (float*)(bw + bp.bpart);
    // (split mode: conv0's weight gradient -- K = pixels, N = 27 -- runs on the exact-fp32 kernel: 34 GFLOP, 0.2 ms at B = 32)
    const int lds = WG_TH * 32 * wg_stride<ET>() + 3 * (WG_TH + 2) * 48 * 4;
    const double P = (double)B * (H - a.row_begin) * W;
    hla_prof_begin_dyn(K_WGRAD, 2.0 * 27 * 64 * P, P * (12 + 64 * sizeof(ET)), st, a.dyn ? a.dyn + a.dyn_desc : nullptr,
                       a.tiles_x * a.tiles_y);
    hipLaunchKernelGGL((wgrad0_kernel<ET>), dim3(a.KS), dim3(256), lds, st, a);
    hla_prof_end(st);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((64 * 27 + 15) / 16), dim3(256), 0, st, a.part, gr->dw[0], (size_t)64 * 27, a.KS * 2, 64 * 32, 27, 32);
    if (gr->db[0]) hipLaunchKernelGGL(reduce_partials_kernel, dim3(4), dim3(256), 0, st, a.bpart, gr->db[0], (size_t)64, a.KS * 2, 64, 1, 1);
  }
  HLA_CHECK_HIP(hipGetLastError());
  return launch_ok ? HLA_OK : HLA_ERR_ARG;
}

#if HLA_TU_DTYPE >= 0
template void vgg_pack_all_T<TuT>(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st);
template int vgg_backward_t<TuT>(const float* x, size_t x_plane, const hla_vgg_params* prm, const char* packedT, int dtype, const char* fw, const float* const feat[3], const double* inv_norm, const float* const d_feat[3], const float* const conf[3], const float* const d_conf[3], const hla_vgg_grads* gr, char* bw, const BwdPlan& bp, int B, int H, int W, int flags, int first_row8, hipStream_t st);
#else
#define HLA_EXTERN_T(T) \
  extern template void vgg_pack_all_T<T>(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st); \
  extern template int vgg_backward_t<T>(const float* x, size_t x_plane, const hla_vgg_params* prm, const char* packedT, int dtype, const char* fw, const float* const feat[3], const double* inv_norm, const float* const d_feat[3], const float* const conf[3], const float* const d_conf[3], const hla_vgg_grads* gr, char* bw, const BwdPlan& bp, int B, int H, int W, int flags, int first_row8, hipStream_t st);
HLA_EXTERN_T(float) HLA_EXTERN_T(bf16) HLA_EXTERN_T(f16) HLA_EXTERN_T(split32)

extern "C" size_t hla_vgg_bwd_workspace_bytes(int B, int H, int W, int level, int dtype) {
  BwdPlan p;
  bwd_plan(B, H, W, dtype, &p, level == 4);
  return p.total;
}

// (HLA_F16X3: fp32 storage in the HLA_F32 workspace layout; data and weight gradients on split-fp16 kernels, conv0's weight
// gradient and the elementwise passes on the fp32 ones)
static inline int bwd_dtype(int dtype) { return dtype; }

extern "C" size_t hla_vgg_packed_weight_T_bytes(int dtype) {
  return packed_offset(kAllLayers, dtype) + (dtype == HLA_F16X3 ? kPackTailBytes : 0);
}

extern "C" int hla_vgg_pack_weights_T(const hla_vgg_params* params, void* packed, int dtype, hla_stream_t stream) {
  HLA_REQUIRE(params && packed, "hla_vgg_pack_weights_T: null argument");
  HLA_REQUIRE(hla_dtype_ok(dtype), "hla_vgg_pack_weights_T: bad dtype %d", dtype);
  dtype = bwd_dtype(dtype);
  if (dtype == HLA_BF16) vgg_pack_all_T<bf16>(params, (char*)packed, dtype, (hipStream_t)stream);
  else if (dtype == HLA_F16) vgg_pack_all_T<f16>(params, (char*)packed, dtype, (hipStream_t)stream);
  else if (dtype == HLA_F16X3) vgg_pack_all_T<split32>(params, (char*)packed, dtype, (hipStream_t)stream);
  else vgg_pack_all_T<float>(params, (char*)packed, dtype, (hipStream_t)stream);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}

extern "C" int hla_vgg_backward(const float* x, size_t x_plane, const hla_vgg_params* params, const void* packed_weights_T,
                                const void* fwd_workspace, const float* const feat[4], const double* inv_norm,
                                const float* const d_feat[4], const float* const conf[4], const float* const d_conf[4],
                                const hla_vgg_grads* grads, void* workspace, size_t workspace_bytes, int B, int H, int W,
                                int level, int dtype, int flags, int first_row8, hla_stream_t stream) {
  HLA_REQUIRE(x && params && packed_weights_T && fwd_workspace && feat && inv_norm && d_feat && grads && workspace,
              "hla_vgg_backward: null argument");
  HLA_REQUIRE(hla_dtype_ok(dtype), "hla_vgg_backward: bad dtype %d", dtype);
  dtype = bwd_dtype(dtype);
  HLA_REQUIRE(level == 3 || level == 4, "hla_vgg_backward: level must be 3 or 4");
  HLA_REQUIRE((size_t)H * W * 64 * 4 < ((size_t)1 << 31), "hla_vgg_backward: image too large (H*W must be below 2^23 pixels)");
  HLA_REQUIRE(x_plane == 0 || x_plane >= (size_t)H * W, "hla_vgg_backward: x_plane (%zu) must be 0 or >= H*W", x_plane);
  HLA_REQUIRE(first_row8 == 0 || (first_row8 >= 4 && first_row8 < H / 8), "hla_vgg_backward: first_row8 must be 0 or in [4, H/8)");
  const int NLc = level == 4 ? 4 : 3;
  HLA_REQUIRE(B > 0 && H % 8 == 0 && W % 8 == 0, "hla_vgg_backward: H and W must be multiples of 8");
  for (int l = 0; l < kPackedLayers; ++l) HLA_REQUIRE(grads->dw[l], "hla_vgg_backward: dw[%d] missing", l);
  HLA_REQUIRE(!d_conf || conf, "hla_vgg_backward: d_conf given without conf");
  if (d_conf)
    for (int l = 0; l < NLc; ++l)
      HLA_REQUIRE(!d_conf[l] || (conf[l] && grads->dw[13 + l]), "hla_vgg_backward: d_conf[%d] needs conf[%d] and dw[%d]", l, l, 13 + l);
  if (level == 4)
    HLA_REQUIRE(feat[3] && d_feat[3] && params->w[11] && params->w[12] && grads->dw[11] && grads->dw[12],
                "hla_vgg_backward: level 4 needs feat[3], d_feat[3], the padded conv_dec3 weights and dw[11], dw[12] ([64,128,3,3], [64,64,3,3])");
  BwdPlan bp;
  bwd_plan(B, H, W, dtype, &bp, level == 4);
  if (workspace_bytes < bp.total) {
    hla_set_error("hla_vgg_backward: workspace %zu < %zu", workspace_bytes, bp.total);
    return HLA_ERR_WORKSPACE;
  }
  if (dtype == HLA_BF16)
    return vgg_backward_t<bf16>(x, x_plane, params, (const char*)packed_weights_T, dtype, (const char*)fwd_workspace, feat, inv_norm,
                                d_feat, conf, d_conf, grads, (char*)workspace, bp, B, H, W, flags, first_row8, (hipStream_t)stream);
  if (dtype == HLA_F16)
    return vgg_backward_t<f16>(x, x_plane, params, (const char*)packed_weights_T, dtype, (const char*)fwd_workspace, feat, inv_norm,
                               d_feat, conf, d_conf, grads, (char*)workspace, bp, B, H, W, flags, first_row8, (hipStream_t)stream);
  if (dtype == HLA_F16X3)
    return vgg_backward_t<split32>(x, x_plane, params, (const char*)packed_weights_T, dtype, (const char*)fwd_workspace, feat, inv_norm,
                                   d_feat, conf, d_conf, grads, (char*)workspace, bp, B, H, W, flags, first_row8, (hipStream_t)stream);
  return vgg_backward_t<float>(x, x_plane, params, (const char*)packed_weights_T, dtype, (const char*)fwd_workspace, feat, inv_norm,
                               d_feat, conf, d_conf, grads, (char*)workspace, bp, B, H, W, flags, first_row8, (hipStream_t)stream);
}
extern "C" int hla_vgg_backward_live_tiles(const void* workspace, int B, int H, int W, int level, int dtype, long long* live,
                                           long long* total) {
  HLA_REQUIRE(workspace && live && total, "hla_vgg_backward_live_tiles: null argument");
  HLA_REQUIRE(hla_dtype_ok(dtype), "hla_vgg_backward_live_tiles: bad dtype %d", dtype);
  BwdPlan bp;
  bwd_plan(B, H, W, bwd_dtype(dtype), &bp, level == 4);
  const DynLayout L = dyn_layout(H, W);
  std::vector<int> h(L.total);
  HLA_CHECK_HIP(hipMemcpy(h.data(), (const char*)workspace + bp.dyn, (size_t)L.total * sizeof(int), hipMemcpyDeviceToHost));
  *live = *total = 0;
  if ((h[L.used] & 0xff) != 1) return HLA_OK;
  const int cdiv[DC_N] = {2, 2, 2, 4, 4, 4, 4, 4, 4, 2, 2, 1}, wdiv[DW_N] = {2, 2, 4, 4, 4, 4, 4, 2, 2, 1, 1};
  for (int i = 0; i < DC_N; ++i) {
    *live += h[L.conv_desc[i]];
    *total += (long long)((H / cdiv[i] + 7) / 8) * ((W / cdiv[i] + 31) / 32);
  }
  for (int i = 0; i < DW_N; ++i) {
    *live += h[L.wg_desc[i]];
    *total += (long long)((H / wdiv[i] + WG_TH - 1) / WG_TH) * ((W / wdiv[i] + 31) / 32);
  }
  return HLA_OK;
}
#endif
