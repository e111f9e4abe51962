// Device code shared by the VGG forward (vgg.hip) and backward (vgg_backward.hip) translation units.
#pragma once
// VGG16-U-Net feature extractor on gfx950 matrix cores: VGG.py:13-203, L2_norm VGG.py:511-514.
//
// Layout: activations NHWC; T = bf16 / f16 (throughput modes, fp32 accumulate), float (exact-fp32 MFMA) or split32
//   ("fp16x3", the matched-accuracy throughput mode): activations and weights stay fp32 in HBM, and every operand is fed
//   to the matrix cores as hi + lo = fp16(s x) + fp16(s x - hi) with a power-of-two scale s per tensor and sample, so that
//   a product costs three v_mfma_f32_32x32x16_f16 (hi hi + hi lo + lo hi, fp32 accumulate) instead of sixteen
//   v_mfma_f32_32x32x2_f32 passes: fp32-class results (hi + lo carries 23 significand bits, the dropped lo lo term is
//   2^-24 relative) at 1/3 of the fp16 MFMA rate instead of 1/16.
//
// conv3x3_kernel -- implicit GEMM computed as D^T = W * X^T, so a lane owns one output pixel and four
//   consecutive output channels per accumulator quad:
//     M (MFMA rows) = 32 output channels   A operand = weight fragment, straight from L2/L1: weights are
//                                          pre-packed in fragment order, so a fragment is one coalesced 1 KiB load
//     N (MFMA cols) = 32 consecutive pixels of one image row; B operand = pixel fragment from the LDS halo tile
//     K             = 9 taps x Cin, walked as  stage (64 B of channels) -> tap -> 2 k-groups of 16 B per lane
//   The (TH+2)x34 input halo tile of a stage lands in LDS once (64-B pixels, XOR-swizzled 16-B slots: conflict-free
//   ds_read_b128) and is reused by all 9 taps; LDS is double-buffered, one barrier per stage.  16-bit types: the next stage's
//   tile goes HBM -> LDS directly (buffer_load ... lds, requested mid-stage, awaited with a counted vmcnt before the stage's
//   barrier; the zero border comes from the buffer descriptor's range check); the 4-byte types and the un-pooling loader of the
//   backward stage it through registers.  Weights are prefetched 1-2 taps ahead into a register ring.
//   The loader handles "virtual concat + nearest 2x upsample" (VGG.py:144-151) without materialising it.
//   The epilogue fuses bias, 2x2 max-pool, ReLU, the raw fp32 feature copy and its per-sample sum of squares.
// conv02_kernel -- conv0 (3->64 on the NCHW fp32 input, K = 27 padded to 32) computed by MFMA directly into the
//   LDS halo tile of conv2, then conv2 + pool: the 64-channel full-resolution map never touches HBM.
#include "common.h"
#include <type_traits>

typedef __bf16 bf16;
typedef _Float16 f16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

// storage type of the split-fp16 mode: an fp32 in memory, (hi, lo) fp16 pairs in LDS and in the packed weights
struct split32 {
  float v;
  split32() = default;
  __host__ __device__ explicit split32(float f) : v(f) {}
  __host__ __device__ explicit operator float() const { return v; }
};
// CEPL = elements of one 16-B MFMA operand fragment (what the matrix core consumes); 16 / sizeof(T) = elements of one 16-B
// piece of an activation map in memory.  They differ only in split mode.
template <typename T> struct Prec { static constexpr bool SPLIT = false; static constexpr int CEPL = 16 / (int)sizeof(T); };
template <> struct Prec<split32> { static constexpr bool SPLIT = true; static constexpr int CEPL = 8; };

// Power-of-two scale that maps a tensor whose largest magnitude has the fp32 bit pattern `amax_bits` into [2^14, 2^15]: hi
// never overflows fp16 (65504) and lo = s x - hi stays a NORMAL fp16 for every |x| > 2^-17 max|x| (smaller values keep an
// absolute error of 2^-25 / s, i.e. 2^-39 of the maximum).  Scaling by a power of two is exact, so the choice of s never
// changes a result as long as nothing leaves the normal range.  The exponent is clamped to [-60, 60] so that the product of
// an activation scale and a weight scale, and its reciprocal, stay finite.
__device__ __host__ __forceinline__ float split_scale(unsigned amax_bits) {
  int E = (int)(amax_bits >> 23) & 0xff;
  E = E < 81 ? 81 : (E > 201 ? 201 : E);
  const unsigned bits = (unsigned)(127 + 141 - E) << 23;      // 2^(14 - (E - 127))
  float f;
  __builtin_memcpy(&f, &bits, 4);
  return f;
}
// four fp32 values -> four (hi, lo) fp16 pairs of s x
// Eight instructions: v_fma_mix{lo,hi}_f16 forms fp16(x * s + 0) (s is a power of two: the product is exact in fp32, one rounding to
// nearest even) straight into one half of the destination, and fp16(x * s - hi) with the fp16 hi read as the addend (the fp32 FMA
// result is exact before its one rounding).  Value for value what the plain C++ below computes -- hipcc needs 12-16 instructions
// for it (packed multiplies, cvt_pk, converts back, packs) -- and the split is what the loaders of every split-mode kernel spend
// their VALU time on.  -DHLA_SPLIT4_ASM=0 builds the C++ form (tools/probes/bitcmp_libs.py compares the two bit for bit).
#ifndef HLA_SPLIT4_ASM
#define HLA_SPLIT4_ASM 1
#endif
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, float s, uint2& hi, uint2& lo) {
#if HLA_SPLIT4_ASM
  unsigned h01, h23, l01, l23;
  asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
      "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
      "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
      "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
      "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(s));
  hi = make_uint2(h01, h23);
  lo = make_uint2(l01, l23);
#else
  x0 *= s; x1 *= s; x2 *= s; x3 *= s;
  const f16 h0 = (f16)x0, h1 = (f16)x1, h2 = (f16)x2, h3 = (f16)x3;     // round to nearest even
  const f16x4 h = {h0, h1, h2, h3};
  const f16x4 l = {(f16)(x0 - (float)h0), (f16)(x1 - (float)h1), (f16)(x2 - (float)h2), (f16)(x3 - (float)h3)};
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
#endif
}

// Every dtype's kernels are compiled in their own translation unit (build.py passes -DHLA_TU_DTYPE=0|1|2); the
// dispatcher TU (-1, the default) only declares them.
#ifndef HLA_TU_DTYPE
#define HLA_TU_DTYPE -1
#endif
#if HLA_TU_DTYPE == 0
typedef float TuT;
#elif HLA_TU_DTYPE == 1
typedef bf16 TuT;
#elif HLA_TU_DTYPE == 2
typedef f16 TuT;
#elif HLA_TU_DTYPE == 3
typedef split32 TuT;
#endif

template <typename T> __device__ __forceinline__ void mma16(f32x16& acc, const uint4& w, const uint4& p);
template <> __device__ __forceinline__ void mma16<bf16>(f32x16& acc, const uint4& w, const uint4& p) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, p), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<f16>(f32x16& acc, const uint4& w, const uint4& p) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, p), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<split32>(f32x16& acc, const uint4& w, const uint4& p) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, p), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x16& acc, const uint4& w, const uint4& p) {
  // element t of both fragments: channels {8q+t (lanes 0-31), 8q+4+t (lanes 32-63)}
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(p.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(p.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(p.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(p.w), acc, 0, 0, 0);
}

// ---- k-major operand fragments (weight-gradient kernels, vgg_backward.hip; the fused conv0 weight gradient below)
typedef short v4s __attribute__((ext_vector_type(4)));

// fragment of a k-major LDS tile T[k = pixel][i = channel] (row stride `stride` bytes) in MFMA operand order:
// lane l -> row i = ch0 + (l & 31), its 16 bytes = the K-step's k values of its half (see mma16<T>).
//   bf16: K-step = 16 pixels, lane group g = l>>5 holds k = 8g..8g+7
//   fp32: K-step =  8 pixels, element t of lane group g is k = 2t+g
template <typename T> struct KStep;
template <> struct KStep<bf16> { static constexpr int PX = 16; };
template <> struct KStep<f16> { static constexpr int PX = 16; };
template <> struct KStep<float> { static constexpr int PX = 8; };

template <typename T> __device__ __forceinline__ uint4 frag_kmajor(const char* tile, int stride, int px0, int ch0, int lane);
template <> __device__ __forceinline__ uint4 frag_kmajor<bf16>(const char* tile, int stride, int px0, int ch0, int lane) {
  const int t = lane & 15, i0 = ch0 + 16 * ((lane >> 4) & 1), k0 = px0 + 8 * (lane >> 5);
  const char* p = tile + (k0 + (t >> 2)) * stride + (i0 + 4 * (t & 3)) * 2;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 4 * stride));
  uint4 r;
  __builtin_memcpy(&r.x, &lo, 8);
  __builtin_memcpy(&r.z, &hi, 8);
  return r;
}
template <> __device__ __forceinline__ uint4 frag_kmajor<f16>(const char* tile, int stride, int px0, int ch0, int lane) {
  return frag_kmajor<bf16>(tile, stride, px0, ch0, lane);     // same 16-bit transpose read
}
template <> __device__ __forceinline__ uint4 frag_kmajor<float>(const char* tile, int stride, int px0, int ch0, int lane) {
  const char* p = tile + (px0 + (lane >> 5)) * stride + (ch0 + (lane & 31)) * 4;
  uint4 r;
  r.x = *(const unsigned*)(p); r.y = *(const unsigned*)(p + 2 * stride);
  r.z = *(const unsigned*)(p + 4 * stride); r.w = *(const unsigned*)(p + 6 * stride);
  return r;
}
template <typename T> __device__ __forceinline__ uint4 frag_ones();
template <> __device__ __forceinline__ uint4 frag_ones<bf16>() { return make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u); }
template <> __device__ __forceinline__ uint4 frag_ones<f16>() { return make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u); }
template <> __device__ __forceinline__ uint4 frag_ones<float>() { return make_uint4(0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u); }

__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
  const bf16 h[4] = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
  *(uint2*)p = __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ void store4(f16* p, float a, float b, float c, float d) {
  const f16 h[4] = {(f16)a, (f16)b, (f16)c, (f16)d};
  *(uint2*)p = __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *(float4*)p = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(split32* p, float a, float b, float c, float d) {
  *(float4*)p = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(unsigned char* p, float a, float b, float c, float d) {
  *(unsigned*)p = (unsigned)a | ((unsigned)b << 8) | ((unsigned)c << 16) | ((unsigned)d << 24);
}
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(split32 v) { return v.v; }
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Remap the linear id so that every XCD works on
// a CONTIGUOUS run of tiles: vertically adjacent tiles, which share two halo rows, then hit the same L2.  Bijective for
// any grid size (cdna_hip_programming.md "XCD swizzle must be bijective").  Measured: +1 % in one A/B, within noise in the next.
__device__ __forceinline__ int xcd_contiguous(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

struct ConvArgs {
  const void* src1;   // NHWC T, C1 channels; at half resolution when up1
  const void* src2;   // NHWC T, C2 channels (virtual concat after src1), or null
  const uint4* wpk;   // fragment-packed weights
  const float* bias;  // [Cout] or null
  void* out_act;      // NHWC T [B,Ho,Wo,Cout] (post-ReLU when relu_act) or null
  float* out_raw;     // NHWC fp32 (pre-ReLU) or null; T elements instead when raw16 (HLA_VGG_FEAT16)
  int raw16;
  double* sumsq;      // [B, tiles_per_img * gridDim.y] sum of squares of out_raw, or null
  int C1, C2, up1;
  int B, H, W, Cout;
  int relu_act;
  int tiles_x, tiles_y;
  int row_begin;      // first output row this launch computes (even for POOL); rows above it are left untouched.  The halo row
                      // row_begin-1 of the sources is read as it lies in memory (the caller guarantees it was written)
  int src_row_lo;     // source rows above this one (in THIS launch's full-resolution row coordinates) read as zero: the
                      // backward's gradient maps are exactly zero -- and unwritten -- above their first row
  int add_row_lo;     // likewise for add_src, in output row coordinates
  const int* dyn;     // data-dependent launch (vgg_backward.hip, bwd_fan_kernel): base of the device-side tables, or null
  int dyn_desc;       //   int offset of this launch's ConvDyn in them
  // --- training / backward extras (all optional) ---
  const unsigned char* unpool_idx;  // src1 is a max-pooled map's gradient at half resolution [B,H/2,W/2,C1] and this is
                                    // the forward argmax (0..3): the loader routes it to full resolution (virtual unpool)
  const void* mask_act;             // NHWC T, output shape: out = act > 0 ? out : 0   (ReLU backward)
  const void* add_src;              // NHWC T, output shape: out += add                (gradient fan-in)
  unsigned char* idx_out;           // POOL (max): argmax position 2*row+col of every pooled element (forward, training)
  int pool_sum;                     // POOL epilogue sums the 2x2 block instead of max (backward of nearest upsample)
  // --- split-fp16 mode (T = split32) ---
  const unsigned* amax1;            // [B] fp32 bit pattern of max |src1| per sample (written by the producer's epilogue)
  const unsigned* amax2;            // likewise for src2, or null; the two sources share one scale
  unsigned* amax_out;               // [B] atomicMax target for max |out_act| per sample (zeroed before the forward), or null
  const float* wscale;              // the power-of-two scale baked into wpk (device scalar written by the packer)
  // --- conv0's weight gradient fused into the epilogue of conv2's data gradient (EPI_DGRAD_WG0; UNPOOL, Cout = 64 only):
  //     the masked gradient tile is contracted with the image patch on the spot and never stored (out_act is not written)
  const float* wg0_x;               // [B,3,H,W] NCHW fp32 network input, channel planes wg0_x_plane elements apart, or null
  size_t wg0_x_plane;
  float* wg0_part;                  // [workgroup][64 co][32]: k = c*9 + tap in columns 0..26, the bias gradient in column 27
#if HLA_CONV_STAMPS
  unsigned long long* stamps;       // tooling build: per-wave cycle stamps of this launch, or null (common.h)
#endif
};
#if HLA_CONV_STAMPS
// s_memtime (shader-clock cycles), HW_ID / XCC_ID: which CU / SIMD / slot the wave ran on
#define HLA_STAMP(K) do { if (a.stamps) { const unsigned long long ts_ = __builtin_readcyclecounter(); \
    if ((threadIdx.x & 63) == 0) a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * HLA_STAMP_N + (K)] = ts_; } } while (0)
#define HLA_STAMP_HWID() do { if (a.stamps && (threadIdx.x & 63) == 0) { \
    const unsigned long long hw_ = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xc_ = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); \
    a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * HLA_STAMP_N + 7] = hw_ | (xc_ << 32); } } while (0)
#else
#define HLA_STAMP(K) do {} while (0)
#define HLA_STAMP_HWID() do {} while (0)
#endif

// Data-dependent trimming (the backward of a branch whose incoming gradient has a small footprint, vgg_backward.hip): tables in
// DEVICE memory, written by an earlier kernel on the same stream.  Only the `n_live` tiles of the list are computed (the same
// ones for every sample); a source pixel reads as zero unless its column lies in the interval its producer WROTE for that
// band of 8 source rows (`src_bands`: {lo, hi} pairs, multiples of 32 in the source's own coordinates), add_src likewise.
struct ConvDyn {
  int n_live;         // live tiles per sample
  int list;           // int offset of the list: entry = (tile row << 16) | tile column
  int src_bands;      // int offset of the source's band table, or -1: the source is dense
  int add_bands;      // likewise for add_src (output coordinates)
};
struct PixBox { int y0, y1, x0, x1; };

constexpr int HWID = 34;   // halo tile width in pixels
constexpr int SB = 64;     // bytes of channels per pixel per pipeline stage
constexpr int PSTR = 64;   // LDS bytes per halo pixel: its 64 B of channels and no pad.  ds_read_b128 is conflict-free through an XOR
                           // swizzle: 16-B slot s of the pixel in halo column hx is stored at slot s ^ halo_key(hx).  A lane group of a
                           // b128 read covers 16 pixels of one halo row with x mod 16 all distinct (MI355X_MICROARCH.md, LDS table)
                           // and one slot s: (pixel mod 4 = which quarter of the 256-B bank row, swizzled slot) is then distinct for
                           // all 16.  (Round 2 padded the pixel to 80 B instead: 25 % more LDS, which capped the 64-channel kernels
                           // at two workgroups per CU.)
__device__ __forceinline__ int halo_key(int hx) { return (hx >> 2) & 3; }
// byte offset of 16-B slot `slot` of halo pixel `pix` (= hy * HWID + hx) inside a stage buffer
__device__ __forceinline__ int halo_off(int pix, int hx, int slot) { return pix * PSTR + ((slot ^ halo_key(hx)) << 4); }
// The next stage's halo tile is fetched in two halves through the SAME staging registers: half 0 is requested at tap HALO_TAP0,
// written to the idle buffer at tap HALO_TAP1, where half 1 is requested; half 1 is written after the stage's last MFMA.
// (measured, same-box A/B of (TAP0, TAP1): (2, 6) beats (1, 5) by 3 % on the pooled 64-channel-wave-tile kernel and by 1 % on the
// un-pooled one; (0, 4) the same; (3, 7) and (1, 6) lose 1-4 %; (3, 6) costs the 32-channel wave tile 20 %)
constexpr int HALO_TAP0 = 2, HALO_TAP1 = 6;
// The 16-bit plain kernels fetch a stage's halo tile HBM -> LDS directly (conv3x3_kernel, DMA): -DHLA_CONV_HALO_DMA=0 builds the
// register-staged loader for them too (same-box A/B); the tap at whose weight loads the tile is requested (0 / 1 / 3 measure the same).
#ifndef HLA_CONV_HALO_DMA
#define HLA_CONV_HALO_DMA 1
#endif
constexpr int HLA_CONV_DMA_TAP = 1;
#ifndef HLA_CONV_DMA_EARLY1
#define HLA_CONV_DMA_EARLY1 0
#endif
#ifndef HLA_A0_ABL
#define HLA_A0_ABL 0
#endif
#ifndef HLA_UNPOOL_NT1_OCC
#define HLA_UNPOOL_NT1_OCC 3         // workgroups per CU of conv2's data gradient (the one launch of the UNPOOL, 32-channel-wave-tile kernel)
#endif
#ifndef HLA_CONV_SMALL_GRID
#define HLA_CONV_SMALL_GRID 320      // workgroups: below this a forward launch takes 4-row tiles (launch_conv)
#endif
// per-lane byte offsets of a wave's pixel fragments: [kx][kg] -> (x + kx) * PSTR + swizzled slot of (lane half g, k-group kg),
// relative to the wave's first halo row
struct FragOff { int o[3][2]; };
__device__ __forceinline__ FragOff frag_offsets(int lane, int row0) {
  const int x = lane & 31, g = lane >> 5;
  FragOff f;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) f.o[kx][kg] = halo_off(row0 * HWID + x + kx, x + kx, g | (kg << 1));
  return f;
}

// ---------------------------------------------------------------------------------------------
// shared epilogue: acc[i][j] holds, for lane (x = lane&31, g = lane>>5), output channels
// cb + j*32 + 8q + 4g + {0..3} (q = r>>2) of pixel (row i, column x).
// Writing that straight to NHWC memory is 8 B per lane into 32 different 128-B lines per store instruction
// (measured: 16-28 k cycles per wave, up to half of a block's lifetime).  Instead every wave transposes one
// pixel row at a time through a private LDS region (`stage`, >= 32*(NT*32*4+16) bytes) and stores it as whole
// pixel rows: 16 B per lane, consecutive lanes on consecutive addresses.
// The value of the neighbouring lane (lane ^ 1) as ONE VALU instruction (DPP quad_perm [1,0,3,2]).  `__shfl_xor(v, 1, 64)` compiles
// to ds_bpermute_b32 -- an LDS-crossbar round trip with a wait behind it: the 2x2 pooling epilogues issued 64-128 of them per wave
// tile (conv0 + conv2, conv7, conv14: 39 % of the inference step runs behind such an epilogue).
__device__ __forceinline__ float lane_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

template <typename E, int NT> struct RowStager {
  static constexpr int CW = NT * 32, PITCH = CW * (int)sizeof(E) + 16, CPP = CW * (int)sizeof(E) / 16;
  // lane-side write of 4 consecutive channels of pixel `px`
  static __device__ __forceinline__ void put(char* stage, int px, int ch, float v0, float v1, float v2, float v3) {
    store4((E*)(stage + px * PITCH) + ch, v0, v1, v2, v3);
  }
  // cooperative flush of `npx` pixels: dst points at channel cb of the first pixel; pixel stride = Cout elements
  // `mask` / `add` (optional) are indexed exactly like dst: v = (mask > 0 ? v : 0) + add, applied on the 16-B vectors
  // returns (not PLAIN, with a mask or an add) the max |v| of what was stored, i.e. AFTER mask and add: in split mode the scale
  // the consumer of this map will use
  template <bool PLAIN = false, int GG = 2>      // GG: vectors per lane whose mask / fan-in loads are in flight together
  static __device__ __forceinline__ float flush(const char* stage, E* dst, int npx, int npx_valid, int Cout, int lane,
                                                const E* mask = nullptr, const E* add = nullptr,
                                                int add_px_lo = 0, int add_px_hi = 1 << 30) {
    float vmax = 0.f;
    const int chunks = npx * CPP;
    constexpr int EPV = 16 / (int)sizeof(E), NIT = 32 * CPP / 64;
    if (PLAIN || !(mask || add)) {        // (kernel-uniform)
#pragma unroll
      for (int c0 = 0; c0 < 32 * CPP; c0 += 64) {
        const int c = c0 + lane;
        if (c0 < chunks && c < chunks) {
          const int px = c / CPP, part = c % CPP;
          if (px < npx_valid) {
            const unsigned off = (unsigned)px * (unsigned)Cout * (unsigned)sizeof(E) + part * 16;   // < 2^20: one row segment
            *(uint4*)((char*)dst + off) = *(const uint4*)(stage + px * PITCH + part * 16);
          }
        }
      }
      return vmax;
    }
    // With a mask or a fan-in every 16-byte vector needs one or two global loads before it can be stored.  They are issued
    // UNCONDITIONALLY (a lane without a vector reads the row segment's first one: valid memory, value unused) and GG
    // vectors per lane ahead of the first use: as `if (lane has a vector) { load; use; store }` the compiler put each load in
    // its own exec-masked block with s_waitcnt vmcnt(0) behind it -- 16 dependent HBM round trips per wave tile, a third of a
    // data-gradient workgroup's life (round 5; the same pathology as the weight-gradient loaders').
    constexpr int G = NIT < GG ? NIT : GG;
#pragma unroll
    for (int g0 = 0; g0 < NIT; g0 += G) {
      bool ok[G], ap[G];
      unsigned off[G];
      int pxs[G], parts[G];
      uint4 mk[G], ad[G];
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int c0 = (g0 + k) * 64, c = c0 + lane;
        const int px = c / CPP, part = c % CPP;
        ok[k] = c0 < chunks && c < chunks && px < npx_valid;
        pxs[k] = px; parts[k] = part;
        off[k] = ok[k] ? (unsigned)px * (unsigned)Cout * (unsigned)sizeof(E) + part * 16 : 0u;   // < 2^20: one row segment
        ap[k] = add && ok[k] && px >= add_px_lo && px < add_px_hi;
      }
      if (mask) {
#pragma unroll
        for (int k = 0; k < G; ++k) mk[k] = *(const uint4*)((const char*)mask + off[k]);
      }
      if (add) {
#pragma unroll
        for (int k = 0; k < G; ++k) ad[k] = *(const uint4*)((const char*)add + (ap[k] ? off[k] : 0u));
      }
#pragma unroll
      for (int k = 0; k < G; ++k) {
        if (ok[k]) {
          uint4 v = *(const uint4*)(stage + pxs[k] * PITCH + parts[k] * 16);
          E e[EPV], m[EPV], a2[EPV];
          __builtin_memcpy(e, &v, 16);
          if (mask) __builtin_memcpy(m, &mk[k], 16);
          if (add) __builtin_memcpy(a2, &ad[k], 16);
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            float f = (float)e[j];
            if (mask && !((float)m[j] > 0.f)) f = 0.f;
            if (ap[k]) f += (float)a2[j];
            e[j] = (E)f;
            vmax = fmaxf(vmax, fabsf(f));
          }
          __builtin_memcpy(&v, e, 16);
          *(uint4*)((char*)dst + off[k]) = v;
        }
      }
    }
    return vmax;
  }
};

// dsc (split mode): accumulators hold (s_x s_w) * result; 1 otherwise.  red: 8 floats of LDS.
// EPI selects what is COMPILED IN.  The generic epilogue evaluates every optional output at run time (raw copy, ReLU mask,
// gradient fan-in, pool argmax, sum-pool): 3200 instructions and 120 exec-mask branches per wave, 8.5 k cycles even with every
// store removed (per-wave cycle accounting) -- as long as two pipeline stages of MFMAs.  The forward pass only ever needs
// two shapes of it, so those are compiled separately and picked by one kernel-uniform branch (epilogue_mode):
enum { EPI_GENERIC = 0,   // everything, decided at run time (backward / training forward)
       EPI_ACT = 1,       // out_act = relu(acc + bias) only
       EPI_ACT_RAW = 2,   // + the raw fp32 copy and its per-sample sum of squares (the three feature layers)
       EPI_ACT_RAW_NOBIAS = 3,    // the same for a layer without bias (the decoder's): 32 registers less
       EPI_DGRAD = 4,     // backward data gradient: out_act = (mask > 0 ? acc : 0) [+ add], no bias, no ReLU of its own
       EPI_DGRAD_WG0 = 5 };   // conv2's data gradient, not stored: contracted with the image patch into conv0's weight gradient
__host__ __device__ __forceinline__ int epilogue_mode(const ConvArgs& a) {
  if (a.wg0_part) return EPI_DGRAD_WG0;
  if (a.mask_act && a.out_act && !a.relu_act && !a.bias && !a.out_raw && !a.sumsq && !a.idx_out && !a.pool_sum) return EPI_DGRAD;
  // (a 16-bit raw copy without the activation output: the last decoder layer when nothing consumes relu(x21) -- vgg.hip)
  if (!a.out_act && a.out_raw && a.raw16 && a.sumsq && !a.bias && !a.mask_act && !a.add_src && !a.idx_out && !a.pool_sum)
    return EPI_ACT_RAW_NOBIAS;
  if (a.mask_act || a.add_src || a.idx_out || a.pool_sum || !a.out_act || !a.relu_act) return EPI_GENERIC;
  if (!a.out_raw) return a.sumsq ? EPI_GENERIC : EPI_ACT;
  return a.sumsq ? (a.bias ? EPI_ACT_RAW : EPI_ACT_RAW_NOBIAS) : EPI_GENERIC;
}
// R16: the raw copy is written in fp16 (16-bit activation types only) instead of fp32; its sum of squares is that of the
// ROUNDED values, so that inv_norm normalises exactly the map the LM loop will read.
// STGW: bytes of the wave's row stager.  The activation-only form (EPI_ACT) stages as many output rows side by side as fit there
// and stores them together: the LDS write -> read turn-around is paid once per batch instead of once per row.
template <typename T, int MT, int NT, bool POOL, int EPI = EPI_GENERIC, bool R16 = false, int STGW = 0>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MT][NT], const ConvArgs& a, int b, int yrow0, int x0,
                                              int cb, float* red, char* stage, float dsc = 1.f,
                                              PixBox addb = PixBox{0, 1 << 30, 0, 1 << 30},
                                              const float4* prebias = nullptr) {      // [NT * 4] bias values the caller already holds
  using RawT = std::conditional_t<R16, f16, float>;      // 16-bit raw maps are fp16 also in bf16 mode: 11 significand bits for the LM loop
  constexpr bool GEN = EPI == EPI_GENERIC, RAW = EPI == EPI_ACT_RAW || EPI == EPI_ACT_RAW_NOBIAS, DG = EPI == EPI_DGRAD;
  // 16-bit types: the kernels START their accumulators at the bias (conv3x3_kernel / conv02_kernel), nothing to add here.
  // (The 4-byte types keep the add behind the sum: in exact-fp32 mode the other order moved max-pool near-ties of a small
  //  gradient test -- an equally valid rounding, but not worth re-baselining the fp32-class parity numbers for.)
  constexpr bool NOBIAS = EPI == EPI_ACT_RAW_NOBIAS || DG || sizeof(T) == 2;
  const bool has_raw = GEN ? a.out_raw != nullptr : RAW;
  const bool has_act = GEN ? a.out_act != nullptr : true;
  const bool relu = GEN ? a.relu_act != 0 : !DG;
  const bool pool_sum = GEN && a.pool_sum;
  const bool has_sumsq = GEN ? a.sumsq != nullptr : RAW;
  // The thread index is made opaque here so that nothing derived from it (stager addresses, lane offsets, masks) can be
  // hoisted above the main loop: with several epilogue forms compiled into one kernel that hoisting cost 18-50 spilled
  // registers in the 243-register main loop.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), x = lane & 31, g = lane >> 5;
  const int Ho = POOL ? a.H >> 1 : a.H, Wo = POOL ? a.W >> 1 : a.W;
  constexpr int NPX = POOL ? 16 : 32;
  float4 bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bias[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!NOBIAS && prebias) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bias[j][q] = prebias[j * 4 + q];
  } else if (!NOBIAS && a.bias) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bias[j][q] = *(const float4*)(a.bias + cb + j * 32 + q * 8 + g * 4);
  }
  const int xo0 = POOL ? x0 >> 1 : x0;
  const int nvalid = min(NPX, Wo - xo0);              // pixels of this row segment inside the image
  float ss = 0.f, mx = 0.f;
  constexpr int ROWB = 32 * RowStager<T, NT>::PITCH;
  constexpr bool BATCH = EPI == EPI_ACT && STGW >= 2 * ROWB;
  if constexpr (BATCH) {
    constexpr int STEP = POOL ? 2 : 1, NR = MT / STEP;
    constexpr int RB = STGW / ROWB >= NR ? NR : 2;       // output rows per batch
    static_assert(NR % RB == 0, "row batches");
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += RB) {
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int i = (r0 + rr) * STEP;
        const bool lane_ok = (yrow0 + i < a.H) && (x0 + x < a.W) && (!POOL || !(x & 1));
        const int px = POOL ? x >> 1 : x;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = acc[i][j][q * 4 + e];
              if (POOL) {
                t = fmaxf(t, acc[i + 1][j][q * 4 + e]);
                t = fmaxf(t, lane_xor1(t));
              }
              if (Prec<T>::SPLIT) t *= dsc;
              w[e] = t;
            }
            if (!NOBIAS) { w[0] += bias[j][q].x; w[1] += bias[j][q].y; w[2] += bias[j][q].z; w[3] += bias[j][q].w; }
            w[0] = fmaxf(w[0], 0.f); w[1] = fmaxf(w[1], 0.f); w[2] = fmaxf(w[2], 0.f); w[3] = fmaxf(w[3], 0.f);
            if (Prec<T>::SPLIT && lane_ok) mx = fmaxf(fmaxf(mx, fmaxf(w[0], w[1])), fmaxf(w[2], w[3]));
            if (!POOL || !(x & 1)) RowStager<T, NT>::put(stage + rr * ROWB, px, j * 32 + q * 8 + g * 4, w[0], w[1], w[2], w[3]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int y = yrow0 + (r0 + rr) * STEP;
        const size_t pix0 = ((size_t)b * Ho + (POOL ? y >> 1 : y)) * Wo + xo0;
        if (y < a.H) RowStager<T, NT>::template flush<true>(stage + rr * ROWB, (T*)a.out_act + pix0 * a.Cout + cb, NPX, nvalid, a.Cout, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < (BATCH ? 0 : MT); i += (POOL ? 2 : 1)) {
    const int y = yrow0 + i;
    const int yo = POOL ? y >> 1 : y;
    const bool row_ok = y < a.H;                      // wave-uniform
    const bool lane_ok = row_ok && (x0 + x < a.W) && (!POOL || !(x & 1));
    const int px = POOL ? x >> 1 : x;
    float v[NT][4][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[i][j][q * 4 + e];
          if (POOL) {
            const float u = acc[i + 1][j][q * 4 + e];
            if (pool_sum) {
              t += u;
              t += lane_xor1(t);
            } else {
              t = fmaxf(t, u);
              t = fmaxf(t, lane_xor1(t));
            }
          }
          if (Prec<T>::SPLIT) t *= dsc;           // exact: a power of two (and > 0, so it commutes with the pooling)
          v[j][q][e] = t;
        }
        if (!NOBIAS) { v[j][q][0] += bias[j][q].x; v[j][q][1] += bias[j][q].y; v[j][q][2] += bias[j][q].z; v[j][q][3] += bias[j][q].w; }
      }
    const size_t pix0 = ((size_t)b * Ho + yo) * Wo + xo0;
    // 16-bit activations: the raw fp32 row and the activation row fit side by side in the wave's stager, so one pass over the
    // accumulators feeds both (no 32-value array kept live between two passes)
    constexpr bool ONE_PASS = RAW && sizeof(T) == 2;
    if constexpr (ONE_PASS) {
      char* stage2 = stage + 32 * RowStager<RawT, NT>::PITCH;
      if (!POOL || !(x & 1)) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float w0 = v[j][q][0], w1 = v[j][q][1], w2 = v[j][q][2], w3 = v[j][q][3];
            if (R16) {      // saturate instead of overflowing to inf, then take the value the LM loop will read
              w0 = (float)(f16)fminf(fmaxf(w0, -65504.f), 65504.f); w1 = (float)(f16)fminf(fmaxf(w1, -65504.f), 65504.f);
              w2 = (float)(f16)fminf(fmaxf(w2, -65504.f), 65504.f); w3 = (float)(f16)fminf(fmaxf(w3, -65504.f), 65504.f);
            }
            RowStager<RawT, NT>::put(stage, px, j * 32 + q * 8 + g * 4, w0, w1, w2, w3);
            if (lane_ok) ss += w0 * w0 + w1 * w1 + w2 * w2 + w3 * w3;
            if (a.out_act)      // (kernel-uniform)
              RowStager<T, NT>::put(stage2, px, j * 32 + q * 8 + g * 4, fmaxf(w0, 0.f), fmaxf(w1, 0.f), fmaxf(w2, 0.f), fmaxf(w3, 0.f));
          }
      }
      // (the row's sum of squares is complete HERE: left to itself the compiler sank the 32 multiply-adds below the flush and kept
      //  their inputs alive across it -- 16 spilled registers and a scratch allocation for every wave of the kernel)
      asm volatile("" : "+v"(ss));
      __builtin_amdgcn_sched_barrier(0);
      if (row_ok) {
        RowStager<RawT, NT>::template flush<true>(stage, (RawT*)a.out_raw + pix0 * a.Cout + cb, NPX, nvalid, a.Cout, lane);
        if (a.out_act) RowStager<T, NT>::template flush<true>(stage2, (T*)a.out_act + pix0 * a.Cout + cb, NPX, nvalid, a.Cout, lane);
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the rows apart: hoisting the next rows' arithmetic up here spills
      continue;
    }
    if (has_raw) {
      if (!POOL || !(x & 1)) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            RowStager<float, NT>::put(stage, px, j * 32 + q * 8 + g * 4, v[j][q][0], v[j][q][1], v[j][q][2], v[j][q][3]);
            if (lane_ok) ss += v[j][q][0] * v[j][q][0] + v[j][q][1] * v[j][q][1] + v[j][q][2] * v[j][q][2] + v[j][q][3] * v[j][q][3];
          }
      }
      if (row_ok) RowStager<float, NT>::template flush<true>(stage, a.out_raw + pix0 * a.Cout + cb, NPX, nvalid, a.Cout, lane);
    }
    if (has_act) {
      if (!POOL || !(x & 1)) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float w0 = v[j][q][0], w1 = v[j][q][1], w2 = v[j][q][2], w3 = v[j][q][3];
            if (relu) { w0 = fmaxf(w0, 0.f); w1 = fmaxf(w1, 0.f); w2 = fmaxf(w2, 0.f); w3 = fmaxf(w3, 0.f); }
            // (with a mask or a fan-in the stored values are formed in the flush: the maximum is taken there)
            if (Prec<T>::SPLIT && lane_ok && !((GEN || DG) && (a.mask_act || a.add_src)))
              mx = fmaxf(fmaxf(mx, fmaxf(fabsf(w0), fabsf(w1))), fmaxf(fabsf(w2), fabsf(w3)));
            RowStager<T, NT>::put(stage, px, j * 32 + q * 8 + g * 4, w0, w1, w2, w3);
          }
      }
      const size_t o = pix0 * a.Cout + cb;
      if (GEN || DG) {
        if (row_ok) {
          // (8-row tiles: a whole row's four vectors per lane at once; the 4-row tiles of small launches keep three workgroups per CU with two)
          const float fm = RowStager<T, NT>::template flush<false, (MT >= 4 ? 4 : 2)>(stage, (T*)a.out_act + o, NPX, nvalid, a.Cout, lane,
                                                   a.mask_act ? (const T*)a.mask_act + o : nullptr,
                                                   (a.add_src && yo >= max(a.add_row_lo, addb.y0) && yo < addb.y1) ? (const T*)a.add_src + o : nullptr,
                                                   addb.x0 - xo0, addb.x1 - xo0);
          if (Prec<T>::SPLIT) mx = fmaxf(mx, fm);
        }
      } else {
        if (row_ok) RowStager<T, NT>::template flush<true>(stage, (T*)a.out_act + o, NPX, nvalid, a.Cout, lane);
      }
    }
    if (GEN && POOL && a.idx_out) {            // training only: recompute which of the 4 window positions won (2*row+col)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float am[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t0 = acc[i][j][q * 4 + e], u = acc[i + 1][j][q * 4 + e];
            const float rown = u > t0 ? 1.f : 0.f;             // first maximum wins, like F.max_pool2d
            const float t = fmaxf(t0, u);
            const float t2 = lane_xor1(t), r2 = lane_xor1(rown);
            am[e] = t2 > t ? 2.f * r2 + 1.f : 2.f * rown;
          }
          if (!(x & 1)) RowStager<unsigned char, NT>::put(stage, px, j * 32 + q * 8 + g * 4, am[0], am[1], am[2], am[3]);
        }
      if (row_ok) RowStager<unsigned char, NT>::flush(stage, a.idx_out + pix0 * a.Cout + cb, NPX, nvalid, a.Cout, lane);
    }
  }
  const bool want_max = Prec<T>::SPLIT && a.amax_out;          // kernel-uniform
  if (has_sumsq || want_max) {
    ss = wave_sum_f32(ss);
    if (want_max) mx = wave_max_f32(mx);
    if (lane == 0) { red[wv] = ss; red[4 + wv] = mx; }
    __syncthreads();
    if (tid == 0) {
      if (has_sumsq) {
        const int np = a.tiles_x * a.tiles_y * gridDim.y;
        const int tile = (xcd_contiguous(blockIdx.x, gridDim.x) % (a.tiles_x * a.tiles_y)) * gridDim.y + blockIdx.y;
        a.sumsq[(size_t)b * np + tile] = ((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]);
      }
      if (want_max) {    // non-negative floats order like their bit patterns; the plain read only filters (a stale value costs one
        const unsigned mb = __float_as_uint(fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));      // redundant atomic, never a missed one)
        if (mb > __hip_atomic_load(a.amax_out + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.amax_out + b, mb);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// conv0's weight gradient fused into the epilogue of conv2's (un-pooling) data gradient -- VGG.py:123-128 backwards.
// The data gradient of conv2 is d(loss)/d(relu(conv0)) [B,H,W,64] at full resolution: 1.07 GB (16-bit) / 2.1 GB (fp32 storage)
// per branch at B = 32 that only conv0's weight gradient ever read back (wgrad0_kernel).  Here every wave masks its 4 x 32-pixel x
// 32-channel piece of the tile (ReLU mask = relu(conv0) > 0, the vectors prefetched at the top), leaves it in its LDS stager in
// [pixel][channel] order and contracts it over the PIXELS with the im2col of the 3-channel image patch (K = pixels; the A
// fragments by the transposing LDS read of the weight-gradient kernels, frag_kmajor):
//   dW0[co][k = c*9+tap] += sum_px g[px][co] * x[c][py+ky-1][px+kx-1],    column 27 multiplies ones: the bias gradient.
// One [64][32] fp32 partial per workgroup (8 KB instead of the tile's 32 / 64 KB of gradient map) goes to wg0_part[blockIdx.x];
// reduce_rows_kernel + reduce_partials_kernel add them in a fixed order.  Arithmetic per mode: 16-bit -- the tile rounded to T
// exactly as the stored map was; exact fp32 -- v_mfma_f32_32x32x2_f32; split -- g and x as (hi, lo) fp16 pairs (g with a
// power-of-two scale per wave row from the un-masked maximum, x with one per patch), three MFMAs per product into a zeroed
// accumulator that is descaled into the running sum.
template <typename T, int MT>
__device__ __forceinline__ void conv_epilogue_wg0(f32x16 (&acc)[MT][1], const ConvArgs& a, int b, int y0, int x0, char* lds, float dsc) {
  constexpr bool SPLIT = Prec<T>::SPLIT;
  using ET = std::conditional_t<SPLIT, float, T>;            // element type of the stored maps (the mask)
  using FT = std::conditional_t<SPLIT, f16, ET>;             // element type of the MFMA fragments
  constexpr int TH = 2 * MT, IH = TH + 2, IW = 48;           // image patch [3][IH][IW]: 34 columns + the K-step's overrun
  constexpr int PITCH = RowStager<ET, 1>::PITCH, CPP = RowStager<ET, 1>::CPP, NIT = 32 * CPP / 64, EPV = 16 / (int)sizeof(ET);
  constexpr int HP = RowStager<f16, 1>::PITCH, HTILE = 32 * HP;      // split mode: the hi / lo fp16 tiles of a row
  constexpr int KPX = KStep<FT>::PX, STG = 5120;
  static_assert(32 * PITCH <= STG && 2 * HTILE <= STG, "wave stager");
  static_assert(4 * STG + 3 * IH * IW * 4 + 2 * 16 * 64 * 4 + 16 <= 2 * (TH + 2) * HWID * PSTR, "fits the halo buffers");
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));      // (see conv_epilogue: nothing of this is to be hoisted above the main loop)
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv >> 1, wn = wv & 1, x = lane & 31, g = lane >> 5;
  char* stage = lds + wv * STG;
  float* in = (float*)(lds + 4 * STG);
  float* xr = in + 3 * IH * IW;                              // [wn][16][64]: the lower row half's sums for the upper one
  float* redm = xr + 2 * 16 * 64;                            // [4]
  const int nvalid = min(32, a.W - x0), cb = wn * 32;

  // the ReLU mask's vectors: addressed like RowStager::flush's (a lane without a vector reads the row segment's first one)
  int pxs[NIT], parts[NIT];
  unsigned voff[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int c = k * 64 + lane;
    pxs[k] = c / CPP; parts[k] = c % CPP;
    voff[k] = pxs[k] < nvalid ? (unsigned)pxs[k] * 64u * (unsigned)sizeof(ET) + parts[k] * 16 : 0u;
  }
  constexpr int PRE = sizeof(ET) == 2 ? MT : 2;             // rows whose mask vectors are in flight ahead
  uint4 mk[MT][NIT];
  auto load_mask = [&](int i) __attribute__((always_inline)) {
    const int y = min(y0 + wm * MT + i, a.H - 1);
    const char* mrow = (const char*)((const ET*)a.mask_act + (((size_t)b * a.H + y) * a.W + x0) * 64 + cb);
#pragma unroll
    for (int k = 0; k < NIT; ++k) mk[i][k] = *(const uint4*)(mrow + voff[k]);
  };
#pragma unroll
  for (int i = 0; i < PRE; ++i) load_mask(i);

  // image patch (1-pixel border) -> LDS; split mode: its maximum -> one power-of-two scale
  constexpr int NIN = (3 * IH * IW + 255) / 256;
  float vin[NIN], amx = 0.f;
#pragma unroll
  for (int it = 0; it < NIN; ++it) {
    const int e = tid + it * 256;
    const int c = e / (IH * IW), r = e % (IH * IW), iy = r / IW, ix = r % IW;
    const int y = y0 - 1 + iy, xx = x0 - 1 + ix;
    vin[it] = 0.f;
    if (e < 3 * IH * IW && ix < HWID && y >= 0 && y < a.H && xx >= 0 && xx < a.W)
      vin[it] = a.wg0_x[((size_t)b * 3 + c) * a.wg0_x_plane + (size_t)y * a.W + xx];
  }
#pragma unroll
  for (int it = 0; it < NIN; ++it) {
    const int e = tid + it * 256;
    if (e < 3 * IH * IW) in[e] = vin[it];
    if (SPLIT) amx = fmaxf(amx, fabsf(vin[it]));
  }
  if (SPLIT) {
    amx = wave_max_f32(amx);
    if (lane == 0) redm[wv] = amx;
  }
  __syncthreads();
  float s_in = 1.f;
  if (SPLIT) s_in = split_scale(__float_as_uint(fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]))));

  f32x16 accw;
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[r] = 0.f;
  const int j = lane & 31, g5 = lane >> 5;                   // B operand: column j = k index (c, ky, kx); 27 = ones (bias)
  const int jc = j < 27 ? j / 9 : 0, jky = (j % 9) / 3, jkx = j % 3;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (i + PRE < MT) load_mask(i + PRE);
    const int rt = wm * MT + i;                              // row inside the tile
    const bool row_ok = y0 + rt < a.H;                       // wave-uniform
    float sG = 1.f;
    if (SPLIT) {
      float m = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[i][0][r]));
      sG = split_scale(__float_as_uint(wave_max_f32(m) * dsc));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float w0 = acc[i][0][q * 4 + 0], w1 = acc[i][0][q * 4 + 1], w2 = acc[i][0][q * 4 + 2], w3 = acc[i][0][q * 4 + 3];
      if (SPLIT) { w0 *= dsc; w1 *= dsc; w2 *= dsc; w3 *= dsc; }
      RowStager<ET, 1>::put(stage, x, q * 8 + g * 4, w0, w1, w2, w3);
    }
    uint4 v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      v[k] = *(const uint4*)(stage + pxs[k] * PITCH + parts[k] * 16);
      ET e[EPV], m[EPV];
      __builtin_memcpy(e, &v[k], 16);
      __builtin_memcpy(m, &mk[i][k], 16);
      const bool ok = row_ok && pxs[k] < nvalid;
#pragma unroll
      for (int t = 0; t < EPV; ++t)
        if (!ok || !((float)m[t] > 0.f)) e[t] = (ET)0.f;
      __builtin_memcpy(&v[k], e, 16);
    }
    if constexpr (SPLIT) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {      // (all of the row's fp32 vectors are in registers: the fp16 tiles overwrite them)
        uint2 hi, lo;
        split4(__uint_as_float(v[k].x), __uint_as_float(v[k].y), __uint_as_float(v[k].z), __uint_as_float(v[k].w), sG, hi, lo);
        *(uint2*)(stage + pxs[k] * HP + parts[k] * 8) = hi;
        *(uint2*)(stage + HTILE + pxs[k] * HP + parts[k] * 8) = lo;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NIT; ++k) *(uint4*)(stage + pxs[k] * PITCH + parts[k] * 16) = v[k];
    }
    if (row_ok) {
      f32x16 tmp;
      if (SPLIT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[r] = 0.f;
      }
#pragma unroll
      for (int kk = 0; kk < 32 / KPX; ++kk) {
        const float* row = in + (jc * IH + rt + jky) * IW + jkx + kk * KPX;
        if constexpr (SPLIT) {
          float xs[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) xs[jj] = j < 27 ? row[8 * g5 + jj] : 0.f;
          uint2 h0, l0, h1, l1;
          split4(xs[0], xs[1], xs[2], xs[3], s_in, h0, l0);
          split4(xs[4], xs[5], xs[6], xs[7], s_in, h1, l1);
          uint4 bhi = make_uint4(h0.x, h0.y, h1.x, h1.y);
          const uint4 blo = make_uint4(l0.x, l0.y, l1.x, l1.y);
          if (j == 27) bhi = frag_ones<f16>();
          const uint4 ahi = frag_kmajor<f16>(stage, HP, kk * KPX, 0, lane), alo = frag_kmajor<f16>(stage + HTILE, HP, kk * KPX, 0, lane);
          mma16<f16>(tmp, ahi, bhi);
          mma16<f16>(tmp, alo, bhi);
          mma16<f16>(tmp, ahi, blo);
        } else {
          const uint4 A = frag_kmajor<ET>(stage, PITCH, kk * KPX, 0, lane);
          ET e[EPV];
#pragma unroll
          for (int jj = 0; jj < EPV; ++jj) {
            const int k = sizeof(ET) == 2 ? 8 * g5 + jj : 2 * jj + g5;      // pixel offset inside the K-step (KStep)
            e[jj] = (ET)(j < 27 ? row[k] : (j == 27 ? 1.f : 0.f));
          }
          mma16<ET>(accw, A, __builtin_bit_cast(uint4, e));
        }
      }
      if (SPLIT) {
        const float ds = j == 27 ? 1.f / sG : 1.f / (sG * s_in);
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[r] += tmp[r] * ds;
      }
    }
  }
  // the two row halves of a channel half: lower + upper, in that order
  if (wm == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xr[(wn * 16 + r) * 64 + lane] = accw[r];
  }
  __syncthreads();
  if (wm == 0) {
    float* dst = a.wg0_part + (size_t)blockIdx.x * 64 * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb + (r & 3) + 8 * (r >> 2) + 4 * g5;
      dst[co * 32 + j] = accw[r] + xr[(wn * 16 + r) * 64 + lane];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// One pipeline stage of MFMAs: 9 taps x 2 k-groups against the halo tile at `cur` (the stage buffer; `fo` holds this lane's
// offsets into it: the wave's first row, the lane's pixel and the swizzled slot of its k-half).  Weight fragments come from global memory through a ring of
// WD+1 register sets filled WD taps ahead; `mid(tap)` runs right after the weight loads of each tap (used to
// issue the next stage's halo loads BEHIND them: VM loads of a wave retire in order).
template <typename T, int MT, int NT, int WD, int PF = (NT == 1 ? 4 : 3)>
struct WeightRing {
  static constexpr int RS = WD + 1;
  uint4 wb[RS][2][NT];
  // pixel fragments in flight ahead of the MFMAs (one MFMA per fragment at NT = 1: four reads = 128 matrix-pipe cycles of cover;
  // six, round 2's choice at two workgroups per CU, cost the three-workgroup kernel 27 spilled registers: 804 against 1100 TF)
  static constexpr int PFD = PF;     // default: 4 at NT = 1, 3 at NT = 2 (NT = 2: 2 and 4 measure the same as 3)
  uint4 pf[PFD + 1];     // pixel-fragment pipeline of the non-upfront MFMA loop (lives across taps)
  const uint4* wq[NT];   // per-lane pointer to this stage's fragments of output tile j: [tap][kg][lane]

  __device__ __forceinline__ void prime() {
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int j = 0; j < NT; ++j) wb[d][kg][j] = wq[j][(d * 2 + kg) * 64];
  }
  // after a stage the ring holds taps 0..WD-1 of the next stage in slots (9+d) % RS; rotate them to slot d
  __device__ __forceinline__ void next_stage() {
#pragma unroll
    for (int j = 0; j < NT; ++j) wq[j] += 18 * 64;
    if (9 % RS != 0) {
      uint4 tmp[WD][2][NT];
#pragma unroll
      for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
          for (int j = 0; j < NT; ++j) tmp[d][kg][j] = wb[(9 + d) % RS][kg][j];
#pragma unroll
      for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
          for (int j = 0; j < NT; ++j) wb[d][kg][j] = tmp[d][kg][j];
    }
  }
};

// The two waves that share a SIMD (one from each co-resident workgroup) run the same instruction stream and
// start together, so left alone they contend for the matrix pipe during their MFMA bursts and then both sit in
// their LDS / VM waits at the same time (a convoy: measured MFMA-busy 46 %).  Giving the wave in the odd
// hardware wave slot a higher static priority lets it run ahead, which staggers the two streams: one computes
// while the other waits.  HW_REG_HW_ID (id 4) bits [3:0] = wave slot within the SIMD.
__device__ __forceinline__ void stagger_priority() {
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
  if (slot & 1) __builtin_amdgcn_s_setprio(1);
}

template <typename T, int MT, int NT, int WD, int PF, typename Mid>
__device__ __forceinline__ void stage_mma(f32x16 (&acc)[MT][NT], const char* cur, const FragOff& fo,
                                          WeightRing<T, MT, NT, WD, PF>& ring, Mid&& mid) {
  constexpr int RS = WD + 1;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int j = 0; j < NT; ++j) ring.wb[(tap + WD) % RS][kg][j] = ring.wq[j][((tap + WD) * 2 + kg) * 64];
    mid(tap);
    // pixel fragments software-pipelined DEPTH reads ahead of the MFMAs that consume them, through tap boundaries.  (The
    // straightforward "read the fragments of a row, multiply" order leaves every ds_read_b128 one LDS latency -- about 100
    // cycles -- ahead of its first MFMA with only 64 cycles of matrix work queued behind it: the pipe idled ~4 x 50 cycles per
    // tap.  Requesting all fragments of a tap up front measured 906 against 960 TF on the 32-channel wave tile.)
    constexpr int FPT = MT * 2, DEPTH = PF;   // fragments per tap: (row i, k-group kg), kg fastest
    auto frag_ptr = [&](int f) {                          // f counts fragments within THIS tap; f >= FPT spills into the next tap
      const int tp = tap + f / FPT, r = f % FPT;
      return cur + fo.o[tp % 3][r & 1] + (tp / 3 + (r >> 1)) * HWID * PSTR;
    };
    if (tap == 0) {
#pragma unroll
      for (int f = 0; f < DEPTH; ++f) ring.pf[f] = *(const uint4*)frag_ptr(f);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      const int slot = (tap * FPT + f) % (DEPTH + 1), nslot = (tap * FPT + f + DEPTH) % (DEPTH + 1);
      if (tap * FPT + f + DEPTH < 9 * FPT) ring.pf[nslot] = *(const uint4*)frag_ptr(f + DEPTH);
      const int i = f >> 1, kg = f & 1;
      if constexpr (Prec<T>::SPLIT) {
        // fragment 0 = hi, fragment 1 = lo of both operands: hi hi + lo_w hi + hi lo_x (lo lo is 2^-24 relative: dropped).
        // kg 0: the hi pixel fragment against the hi and the lo weights; kg 1: the lo pixel fragment against the hi weights
#pragma unroll
        for (int c = 0; c < (kg == 0 ? 2 : 1); ++c)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma16<T>(acc[i][j], ring.wb[tap % RS][c][j], ring.pf[slot]);
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) mma16<T>(acc[i][j], ring.wb[tap % RS][kg][j], ring.pf[slot]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  ring.next_stage();
}

// Occupancy: the 64-channel wave tile (NT = 2) holds 128 accumulator registers: two workgroups per CU.  The 32-channel one
// (NT = 1, the Cout = 64 layers) fits three -- 43.5 KB of LDS each since the halo pixels lost their pad, <= 168 registers since
// the halo staging went through half as many -- which is what hides its per-tile prologue and epilogue.
// UNPOOL: the source is the gradient of a max-pooled map, expanded on the fly with the forward argmax bytes (ConvArgs::unpool_idx;
// the data-gradient convolutions of conv2 / conv7 / conv14).  A separate instantiation: its halo loader carries the argmax bytes
// of a piece next to the piece and applies them when the piece is WRITTEN to LDS -- masking right behind the load put an
// s_waitcnt vmcnt(0) after every single piece, six full memory round trips per stage in the middle of the MFMA stream.
template <typename T, int MT, int NT, int WM, int WN, bool POOL, int WD, bool UNPOOL = false>
__global__ __launch_bounds__(256, NT == 1 ? (UNPOOL ? HLA_UNPOOL_NT1_OCC : 3) : 2) void conv3x3_kernel(ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int EPL = 16 / sizeof(T), KC = SB / sizeof(T);     // channels per stage: 32 (bf16) / 16 (fp32)
  constexpr int TH = WM * MT, HPIX = (TH + 2) * HWID;
  constexpr int NPIECE = (HPIX * 4 + 255) / 256;               // 16-B pieces per thread per stage ...
  // DMA (HLA_CONV_HALO_DMA, 16-bit types, plain loader): the halo tile of a stage goes HBM -> LDS directly (buffer_load ... lds),
  // no staging registers, no ds_write.  The LDS image of such a load is lane-linear (lane L's 16 B at M0 + 16 L), so the XOR
  // swizzle is applied to WHICH 16-B slot of its pixel a lane fetches; a pixel outside the image / the source's written part
  // gets an out-of-range offset and the buffer descriptor's range check writes zeros (tools/probes/lds_dma_probe.hip pins both).
  // A stage buffer is padded to whole 64-pixel pieces: the lanes past the tile's last pixel write zeros there.
  constexpr bool DMA = HLA_CONV_HALO_DMA && sizeof(T) == 2 && !UNPOOL && !Prec<T>::SPLIT;
  constexpr int BUF = (DMA ? NPIECE * 64 : HPIX) * PSTR;
  constexpr int NHALF = (NPIECE + 1) / 2;                      // ... fetched in two halves through NHALF staging registers
  // the epilogue reuses the halo buffers as four wave-private row stagers; the widest form stages a raw fp32 row and a 16-bit
  // activation row side by side (EPI_ACT_RAW on 16-bit types), everything else one row of at most NT*32 fp32
  constexpr int STG = 32 * (RowStager<float, NT>::PITCH + (sizeof(T) == 2 ? RowStager<T, NT>::PITCH : 0));
  constexpr int LDSB = 2 * BUF > 4 * STG ? 2 * BUF : 4 * STG;
  __shared__ __attribute__((aligned(16))) char lds[LDSB];
  __shared__ float red[8];

  HLA_STAMP(0);
  HLA_STAMP_HWID();
  // (readfirstlane: the wave index is uniform, which lets every address that depends on it live in scalar registers)
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), wm = wv / WN, wn = wv % WN;
  // source / fan-in bounds: the image (and the caller's static first rows), narrowed by the device-side boxes if there are any
  const int sy0 = a.src_row_lo, sy1 = a.H;
  int tx, ty, b;
  // column interval of the source per halo row class: top halo row / the tile's own rows / bottom halo row
  // (six scalars, not arrays: once the halo row of a piece is a run-time value the compiler turns a select over array elements
  // into an indexed load from a stack copy of the array)
  struct { int lo0, lo1, lo2, hi0, hi1, hi2; } sb = {0, 0, 0, a.W, a.W, a.W};
  if (a.dyn) {                              // kernel-uniform; everything below is scalar loads
    // Only the live tiles are enumerated, by the FIRST workgroups of the grid; the rest exit.  (Launching every tile and
    // returning from the dead ones is not enough: workgroup ids go round-robin to the shader engines, so a dead / live
    // pattern with the period of a tile row -- 2 dead + 2 live columns at H/4 -- leaves whole engines idle: measured
    // 160 -> 145 us for half the tiles.)
    const ConvDyn& d = *(const ConvDyn*)(a.dyn + a.dyn_desc);
    const int nl = d.n_live, nlive = nl * a.B;
    if ((int)blockIdx.x >= nlive) return;
    int bid = xcd_contiguous(blockIdx.x, nlive);
    b = bid / nl;
    const int e = a.dyn[d.list + bid % nl];
    ty = e >> 16; tx = e & 0xffff;
    if (d.src_bands >= 0) {
      const int sh = (a.up1 || a.unpool_idx) ? 1 : 0, nb = ((a.H >> sh) + 7) >> 3, yy = a.row_begin + ty * TH;
      const int b0 = min(max(yy - 1, 0) >> sh >> 3, nb - 1), b1 = min(yy >> sh >> 3, nb - 1), b2 = min((yy + TH) >> sh >> 3, nb - 1);
      sb.lo0 = a.dyn[d.src_bands + 2 * b0] << sh; sb.hi0 = min(a.dyn[d.src_bands + 2 * b0 + 1] << sh, a.W);
      sb.lo1 = a.dyn[d.src_bands + 2 * b1] << sh; sb.hi1 = min(a.dyn[d.src_bands + 2 * b1 + 1] << sh, a.W);
      sb.lo2 = a.dyn[d.src_bands + 2 * b2] << sh; sb.hi2 = min(a.dyn[d.src_bands + 2 * b2 + 1] << sh, a.W);
    }
  } else {
    int bid = xcd_contiguous(blockIdx.x, gridDim.x);
    tx = bid % a.tiles_x; bid /= a.tiles_x;
    ty = bid % a.tiles_y;
    b = bid / a.tiles_y;
  }
  const int y0 = a.row_begin + ty * TH, x0 = tx * 32;
  const int nstage = (a.C1 + a.C2) / KC;
  const int part = t & 3, pbase = t >> 2;   // 256 % 4 == 0: a thread always moves the same 16-B part of a pixel
  float sx = 1.f, dsc = 1.f;                // split mode: activation scale of this sample, 1 / (activation scale * weight scale)
  if constexpr (Prec<T>::SPLIT) {
    unsigned m = a.amax1[b];
    if (a.amax2) m = max(m, a.amax2[b]);
    sx = split_scale(m);
    dsc = 1.f / (sx * *a.wscale);
  }

  // Everything about a thread's halo pieces except the channel stage is fixed for the workgroup's life, and is kept in the most
  // compact form that makes a stage's loads cheap to issue: per piece ONE signed 32-bit byte offset into the sample per source
  // (< 0: the pixel reads as zero -- outside the image, above the source's first row, outside the columns its producer wrote;
  // bits 0-1: the pixel's (y&1, x&1) position for the virtual unpool) and one LDS offset.  A stage then costs, per piece, a
  // compare, a mask and one saddr + 32-bit-offset load.  (Round 2 left this to the compiler, which hoisted ~20 registers of
  // 64-bit addresses and coordinates out of the stage loop; recomputing them per stage instead costs ~6 VALU per MFMA.)
  constexpr int ES = (int)sizeof(T);
  const int sh1 = (a.up1 || a.unpool_idx) ? 1 : 0, Hs1 = a.H >> sh1, Ws1 = a.W >> sh1;
  const char* s1b = (const char*)a.src1 + (size_t)b * Hs1 * Ws1 * a.C1 * ES;
  const char* s2b = (const char*)a.src2 + (size_t)b * a.H * a.W * a.C2 * ES;
  const unsigned char* idxb = a.unpool_idx ? a.unpool_idx + (size_t)b * Hs1 * Ws1 * a.C1 : nullptr;
  int off1[NPIECE], off2[NPIECE], wo[NPIECE];
  {
    // (the six bounds as uniform scalars: left as loads from the struct, the select over them below is rewritten into ONE load
    // from a selected address, which pins the struct in scratch memory)
    const int lo0 = __builtin_amdgcn_readfirstlane(sb.lo0), lo1 = __builtin_amdgcn_readfirstlane(sb.lo1),
              lo2 = __builtin_amdgcn_readfirstlane(sb.lo2), hi0 = __builtin_amdgcn_readfirstlane(sb.hi0),
              hi1 = __builtin_amdgcn_readfirstlane(sb.hi1), hi2 = __builtin_amdgcn_readfirstlane(sb.hi2);
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int pix = pbase + 64 * i;
      const int hy = pix / HWID, hx = pix - hy * HWID;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const int xlo = hy == 0 ? lo0 : (hy == TH + 1 ? lo2 : lo1), xhi = hy == 0 ? hi0 : (hy == TH + 1 ? hi2 : hi1);
      const bool ok = pix < HPIX && y >= sy0 && y < sy1 && x >= xlo && x < xhi;
      const int psrc = DMA ? (part ^ halo_key(hx)) : part;      // DMA: the lane's LDS slot is `part`; it holds channel slot part ^ key
      constexpr int BAD = DMA ? (int)0x80000000 : -1;           // DMA: an offset beyond any buffer's range -> the load writes zeros
      off1[i] = ok ? ((((y >> sh1) * Ws1 + (x >> sh1)) * a.C1 * ES + psrc * 16) | (DMA ? 0 : (((y & 1) << 1) | (x & 1)))) : BAD;
      off2[i] = ok ? ((y * a.W + x) * a.C2 * ES + psrc * 16) : BAD;
      // split mode: this thread's 4 channels are 8 B of slot part>>1 (hi); the lo half sits 2 slots further = offset ^ 32
      wo[i] = pix >= HPIX ? -1 : (Prec<T>::SPLIT ? halo_off(pix, hx, part >> 1) + (part & 1) * 8 : halo_off(pix, hx, part));
    }
  }
  // UNPOOL: the EPL argmax bytes of a piece (one per element) travel with it from load_stage to write_stage
  constexpr int IDW = UNPOOL ? (EPL == 8 ? 2 : 1) : 1;
  struct StageIds { unsigned w[NHALF][IDW]; };
  auto load_stage = [&](int sg, int half, uint4 (&st)[NHALF], StageIds& ids) __attribute__((always_inline)) {
    const int c0 = sg * KC;
    const bool first = c0 < a.C1;       // wave-uniform
    const char* base = first ? s1b + (size_t)c0 * ES : s2b + (size_t)(c0 - a.C1) * ES;
#pragma unroll
    for (int ii = 0; ii < NHALF; ++ii) {
      const int i = half * NHALF + ii;
      uint4 v = make_uint4(0, 0, 0, 0);
      const int o = i < NPIECE ? (first ? off1[i] : off2[i]) : -1;
      if constexpr (UNPOOL) {
#pragma unroll
        for (int k = 0; k < IDW; ++k) ids.w[ii][k] = 0xffffffffu;     // (matches no position: the piece stays zero)
      }
      if (o >= 0) {
        const unsigned ob = (unsigned)o & ~15u;
        v = *(const uint4*)(base + ob);
        if constexpr (UNPOOL) {
          const unsigned char* idp = idxb + ob / ES + c0;      // one argmax byte per element
#pragma unroll
          for (int k = 0; k < IDW; ++k) ids.w[ii][k] = ((const unsigned*)idp)[k];
        }
      }
      st[ii] = v;
    }
  };
  auto write_stage = [&](char* buf, int sg, int half, uint4 (&st)[NHALF], const StageIds& ids) __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < NHALF; ++ii) {
      const int i = half * NHALF + ii;
      const int w = i < NPIECE ? wo[i] : -1;
      if constexpr (UNPOOL) {      // keep only the elements whose forward argmax is this (y&1, x&1) position
        const bool first = sg * KC < a.C1;
        const unsigned pos = (unsigned)(i < NPIECE ? (first ? off1[i] : off2[i]) : 0) & 3u;
        uint4 v = st[ii];
        if constexpr (sizeof(T) == 2) {
          // 8 argmax bytes -> eight 16-bit keep masks with packed 16-bit math: (id ^ pos) - 1 is negative only for a match
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          const unsigned m0 = ids.w[ii][0] ^ (pos * 0x01010101u), m1 = ids.w[ii][1] ^ (pos * 0x01010101u);
          auto keep = [](unsigned m, unsigned sel) {
            s16x2 w2 = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, m, sel));   // two id bytes, zero-extended
            w2 = (w2 - (short)1) >> 15;
            return __builtin_bit_cast(unsigned, w2);
          };
          v.x &= keep(m0, 0x0c010c00u); v.y &= keep(m0, 0x0c030c02u);
          v.z &= keep(m1, 0x0c010c00u); v.w &= keep(m1, 0x0c030c02u);
        } else {
          T e[EPL];
          __builtin_memcpy(e, &v, 16);
#pragma unroll
          for (int k = 0; k < EPL; ++k) if (((ids.w[ii][0] >> (8 * k)) & 0xffu) != pos) e[k] = (T)0.f;
          __builtin_memcpy(&v, e, 16);
        }
        st[ii] = v;
      }
      if constexpr (Prec<T>::SPLIT) {
        // a stage = 16 channels: [pixel][hi: 16 x fp16 | lo: 16 x fp16]; lane half g of the MFMA reads channels 8g..8g+7
        uint2 hi, lo;
        split4(__uint_as_float(st[ii].x), __uint_as_float(st[ii].y), __uint_as_float(st[ii].z), __uint_as_float(st[ii].w), sx, hi, lo);
        if (w >= 0) { *(uint2*)(buf + w) = hi; *(uint2*)(buf + (w ^ 32)) = lo; }
      } else {
        if (w >= 0) *(uint4*)(buf + w) = st[ii];
      }
    }
  };

  // DMA loader: two raw buffer descriptors (one per source: base = this sample's map, range = its bytes) and, per stage, ONE asm
  // statement of NPIECE loads -- hipcc neither counts them (their completion is awaited by hand before the stage's barrier) nor
  // sees an LDS write it would have to order every following ds_read behind.
  typedef int rsrc_t __attribute__((ext_vector_type(4)));
  rsrc_t rs1 = {0, 0, 0, 0x00020000}, rs2 = {0, 0, 0, 0x00020000};
  unsigned lds_wave = 0;
  if constexpr (DMA) {
    const unsigned long long p1 = (unsigned long long)s1b, p2 = (unsigned long long)s2b;
    rs1[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p1); rs1[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p1 >> 32));
    rs1[2] = __builtin_amdgcn_readfirstlane(Hs1 * Ws1 * a.C1 * ES);
    rs2[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p2); rs2[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p2 >> 32));
    rs2[2] = __builtin_amdgcn_readfirstlane(a.src2 ? a.H * a.W * a.C2 * ES : 0);
    lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds) + wv * (16 * PSTR);
  }
  auto dma_stage = [&](int sg, int bufsel) __attribute__((always_inline)) {
    static_assert(!DMA || NPIECE == 6 || NPIECE == 4, "six pieces per thread (8 x 32 tile) or four (4 x 32)");
    const int c0 = sg * KC;
    const bool first = c0 < a.C1;       // wave-uniform
    const rsrc_t rs = first ? rs1 : rs2;
    const int soff = (first ? c0 : c0 - a.C1) * ES;
    const unsigned dst = lds_wave + bufsel * BUF;
    unsigned keep;
#define HLA_DMA_HEAD(NDST) "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %" #NDST "\n\ts_nop 0\n\t"
#define HLA_DMA_LD(K, NRS, NSO) "buffer_load_dwordx4 %" #K ", %" #NRS ", %" #NSO " offen lds\n\t"
#define HLA_DMA_STEP "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
    if constexpr (NPIECE == 6) {
#define HLA_DMA6(O)                                                                                                                    \
      asm volatile(HLA_DMA_HEAD(9) HLA_DMA_LD(1, 7, 8) HLA_DMA_STEP HLA_DMA_LD(2, 7, 8) HLA_DMA_STEP HLA_DMA_LD(3, 7, 8) HLA_DMA_STEP    \
                   HLA_DMA_LD(4, 7, 8) HLA_DMA_STEP HLA_DMA_LD(5, 7, 8) HLA_DMA_STEP HLA_DMA_LD(6, 7, 8) "s_mov_b32 m0, %0"             \
                   : "=&s"(keep) : "v"(O[0]), "v"(O[1]), "v"(O[2]), "v"(O[3]), "v"(O[4]), "v"(O[5]), "s"(rs), "s"(soff), "s"(dst)      \
                   : "memory", "scc")
      if (first) HLA_DMA6(off1); else HLA_DMA6(off2);
#undef HLA_DMA6
    } else if constexpr (NPIECE == 4) {
#define HLA_DMA4(O)                                                                                                                    \
      asm volatile(HLA_DMA_HEAD(7) HLA_DMA_LD(1, 5, 6) HLA_DMA_STEP HLA_DMA_LD(2, 5, 6) HLA_DMA_STEP HLA_DMA_LD(3, 5, 6) HLA_DMA_STEP    \
                   HLA_DMA_LD(4, 5, 6) "s_mov_b32 m0, %0"                                                                              \
                   : "=&s"(keep) : "v"(O[0]), "v"(O[1]), "v"(O[2]), "v"(O[3]), "s"(rs), "s"(soff), "s"(dst) : "memory", "scc")
      if (first) HLA_DMA4(off1); else HLA_DMA4(off2);
#undef HLA_DMA4
    }
#undef HLA_DMA_HEAD
#undef HLA_DMA_LD
#undef HLA_DMA_STEP
  };

  const int ntg0 = (blockIdx.y * WN + wn) * NT;     // first global 32-channel output tile of this wave
  // The accumulators start at the bias (16-bit types; it commutes with the max-pool, and the epilogues then have neither
  // the 8 x NT bias registers nor the adds); the 4-byte types add it in the epilogue (split mode: after its power-of-two descale).
  f32x16 acc[MT][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sizeof(T) == 2 && a.bias) b4 = *(const float4*)(a.bias + (ntg0 + j) * 32 + q * 8 + (lane >> 5) * 4);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][j][q * 4 + 0] = b4.x; acc[i][j][q * 4 + 1] = b4.y; acc[i][j][q * 4 + 2] = b4.z; acc[i][j][q * 4 + 3] = b4.w;
      }
    }

  const FragOff fo = frag_offsets(lane, wm * MT);
  // packed weights: [ntile][stage][tap][kg(2)][lane] 16-B fragments
  // (the un-pooling loader carries a piece's argmax bytes next to it: at the three-workgroup register cap that costs the 32-channel
  //  wave tile one fragment of read-ahead -- with four, 7 registers spilled and 3 scratch accesses sat inside the MFMA stream)
  constexpr int PF = NT == 1 ? (UNPOOL && sizeof(T) == 2 ? 3 : 4) : 3;
  WeightRing<T, MT, NT, WD, PF> ring;
#pragma unroll
  for (int j = 0; j < NT; ++j) ring.wq[j] = a.wpk + (size_t)(ntg0 + j) * nstage * 18 * 64 + lane;

  uint4 st[NHALF];
  StageIds ids;
  ring.prime();              // the first weight fragments do not depend on the halo tile: request them ahead of it
  if constexpr (DMA) {
    dma_stage(0, 0);
#if HLA_CONV_DMA_EARLY1
    // the second stage's tile is requested right behind the first: both round trips overlap, and the first stage no longer
    // ends on a tile that was requested one prologue later (per-wave cycle stamps, profiles/r06_conv_cycle_table_*.json)
    if (nstage > 1) {
      dma_stage(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
    } else
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    uint4 st1[NHALF];        // the prologue has registers to spare: both halves are requested back to back
    StageIds ids1;
    load_stage(0, 0, st, ids);
    load_stage(0, 1, st1, ids1);
    write_stage(lds, 0, 0, st, ids);
    write_stage(lds, 0, 1, st1, ids1);
  }
  __syncthreads();
  HLA_STAMP(1);
  stagger_priority();

  for (int sg = 0; sg < nstage; ++sg) {
    const bool more = sg + 1 < nstage;
    char* nxt = lds + ((sg + 1) & 1) * BUF;
    stage_mma<T, MT, NT, WD, PF>(acc, lds + (sg & 1) * BUF, fo, ring, [&](int tap) __attribute__((always_inline)) {
      if constexpr (DMA) {
        if (tap == HLA_CONV_DMA_TAP && more && !(HLA_CONV_DMA_EARLY1 && sg == 0)) dma_stage(sg + 1, (sg + 1) & 1);
      } else {
        if (tap == HALO_TAP0 && more) load_stage(sg + 1, 0, st, ids);
        if (tap == HALO_TAP1 && more) { write_stage(nxt, sg + 1, 0, st, ids); load_stage(sg + 1, 1, st, ids); }
      }
    });
    if constexpr (DMA) {
      // the next stage's tile has landed when at most the youngest 2 * NT * WD loads -- the weight fragments primed for the next
      // stage, all requested after the tile -- are still in flight
      // (only true while the tile is requested BEFORE the first of those weight loads, which go out at taps 9-WD..8)
      static_assert(!DMA || HLA_CONV_DMA_TAP <= 8 - WD, "the halo DMA must be issued before the next stage's weight fragments");
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NT * WD) : "memory");
    } else {
      if (more) write_stage(nxt, sg + 1, 1, st, ids);
    }
    if (sg == 0) HLA_STAMP(2);      // the first stage's MFMAs issued (+ the next tile awaited / written), before its barrier
    __syncthreads();
    if (sg == 0) HLA_STAMP(3);
  }
  HLA_STAMP(4);
  // the loop's last barrier guarantees nobody still reads the halo buffers: reuse them as 4 wave-private stagers
  static_assert((LDSB / 4) % 16 == 0, "stager alignment");
  {
    char* stager = lds + wv * (LDSB / 4);
    const int mode = epilogue_mode(a);      // kernel-uniform
    PixBox addb{a.add_row_lo, 1 << 30, 0, 1 << 30};
    if (a.dyn) {
      const ConvDyn& d = *(const ConvDyn*)(a.dyn + a.dyn_desc);
      if (d.add_bands >= 0) {               // the tile's output rows lie in ONE band of the fan-in map
        const int band = (POOL ? y0 >> 1 : y0) >> 3;
        addb.x0 = a.dyn[d.add_bands + 2 * band];
        addb.x1 = a.dyn[d.add_bands + 2 * band + 1];
      }
    }
    // (in the un-pooled 64-channel-wave-tile kernel the raw-copy epilogue WITH bias spills ~30 registers -- 31 k cycles against
    // the generic path's 21 k -- so only its bias-free form is compiled there; that is the one the model uses: dec1.3)
    constexpr bool RAW_SPECIAL = POOL || NT == 1;
    constexpr bool T16 = sizeof(T) == 2;
    if constexpr (UNPOOL && NT == 1 && MT == 4 && WN == 2) {      // conv2's data gradient: the only launch of this instantiation
      if (mode == EPI_DGRAD_WG0) {
        conv_epilogue_wg0<T, MT>(acc, a, b, y0, x0, lds, dsc);
        HLA_STAMP(5);
        return;
      }
    }
    if (mode == EPI_ACT) conv_epilogue<T, MT, NT, POOL, EPI_ACT, false, LDSB / 4>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, stager, dsc, addb);
    // (16-bit raw copy, HLA_VGG_FEAT16: only the three feature layers ask for it -- conv14: pooled + bias; dec1.3 / dec2.3:
    // no bias -- and exactly those forms are compiled)
    else if (T16 && RAW_SPECIAL && a.raw16 && mode == EPI_ACT_RAW)
      conv_epilogue<T, MT, NT, POOL, (T16 && RAW_SPECIAL) ? EPI_ACT_RAW : EPI_GENERIC, T16 && RAW_SPECIAL>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, stager, dsc, addb);
    else if (T16 && a.raw16 && mode == EPI_ACT_RAW_NOBIAS)
      conv_epilogue<T, MT, NT, POOL, T16 ? EPI_ACT_RAW_NOBIAS : EPI_GENERIC, T16>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, stager, dsc, addb);
    else if (RAW_SPECIAL && mode == EPI_ACT_RAW) conv_epilogue<T, MT, NT, POOL, RAW_SPECIAL ? EPI_ACT_RAW : EPI_GENERIC>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, stager, dsc, addb);
    else if (mode == EPI_DGRAD)
      conv_epilogue<T, MT, NT, POOL, EPI_DGRAD>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, stager, dsc, addb);
    else if ((RAW_SPECIAL || sizeof(T) == 2) && mode == EPI_ACT_RAW_NOBIAS)      // (4-byte storage: two passes, spills as well)
      conv_epilogue<T, MT, NT, POOL, (RAW_SPECIAL || sizeof(T) == 2) ? EPI_ACT_RAW_NOBIAS : EPI_GENERIC>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, stager, dsc, addb);
    else conv_epilogue<T, MT, NT, POOL, EPI_GENERIC>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, stager, dsc, addb);
  }
  HLA_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// conv0 + ReLU + conv2 + bias + 2x2 max-pool + ReLU in one kernel (VGG.py:123-128).
struct Conv02Args {
  const float* x;      // [B,3,H,W] NCHW fp32; channel planes x_plane elements apart (>= H*W: the image may be a row window
  size_t x_plane;      //   of a taller one), rows W apart
  const uint4* w0;     // conv0 fragments [2 ntiles][NFRAG][64 lanes], k = cin*9 + tap (27 padded to 32)
  const float* b0;     // [64]
  const uint4* w2;     // conv2 fragments, generic layout
  const float* b2;     // [64]
  void* out_act;       // NHWC T [B,H/2,W/2,64] = relu(pool(conv2))
  void* a0_out;        // training: NHWC T [B,H,W,64] = relu(conv0), or null
  unsigned char* idx_out;  // training: pool argmax [B,H/2,W/2,64], or null
  void* a2_out;        // level 4: NHWC T [B,H,W,64] = relu(conv2) before the pool (the skip input of conv_dec3), or null
  int B, H, W, tiles_x, tiles_y;
  int row_begin;       // first conv2 output row (full resolution, even) this launch computes; see ConvArgs::row_begin
  // --- split-fp16 mode ---
  const float* wtail;      // the packer's tail: [l] = power-of-two scale of layer l's packed weights, [16] = max_cout sum_k |w0|
  unsigned* amax_out;      // [B] atomicMax target: max of out_act per sample
  unsigned* amax_a2_out;   // likewise for a2_out (level 4), or null
  unsigned* amax_a0_out;   // likewise for a0_out (training: the split-mode weight gradient of conv2 scales its input by it), or null
#if HLA_CONV_STAMPS
  unsigned long long* stamps;
#endif
};

// conv0's 64 output channels are conv2's K.  With 2-byte activations they are two 32-channel stages = 54 KB of halo tile and
// two workgroups fit a CU; with 4-byte ones (exact fp32, split fp16) they would be four 16-channel stages = 109 KB, one
// workgroup per CU with nothing to overlap its phases -- so those types produce and consume them in TWO ROUNDS of 32 channels
// through the same two buffers (conv0 is computed per 32-channel half anyway: the same MFMA work, one more barrier).
template <typename T> constexpr int conv02_rounds() { return sizeof(T) == 4 ? 2 : 1; }
template <typename T> constexpr int conv02_lds_bytes() {
  return (64 * (int)sizeof(T) / SB) / conv02_rounds<T>() * (10 * HWID * PSTR) + 3 * 12 * 36 * 4;
}

template <typename T, int WD>
// (three workgroups per CU, except in split mode: there the kernel is matrix-bound already -- three MFMAs per product -- and
// the 168-register cap costs it 31 spills: 1430 us per launch either way, same-box A/B)
__global__ __launch_bounds__(256, Prec<T>::SPLIT ? 2 : 3) void conv02_kernel(Conv02Args a0) {
  constexpr bool SPLIT = Prec<T>::SPLIT;
  // 16-bit types: conv0's bias and the zeroing of halo pixels outside the image ride in the MFMA (k slots 27 / 28 = bias hi / lo
  // against an input of 1.0; an all-zero input column for a pixel outside) instead of 64 adds + 64 selects per 32 pixels and lane
  constexpr bool FOLD = sizeof(T) == 2 && !SPLIT;      // (513 / 517 -> 508 / 504 us per launch, 132 -> 128 registers; same-box A/B x2)
  constexpr int EPL = Prec<T>::CEPL, KC = SB / sizeof(T), NSG = 64 / KC, NFRAG = 32 / (2 * EPL), NH = SPLIT ? 2 : 1;
  constexpr int MT = 4, NT = 1, WN = 2, TH = 8, HPIX = (TH + 2) * HWID, BUF = HPIX * PSTR, IW = 36, IH = 12;
  constexpr int ROUNDS = conv02_rounds<T>(), SPR = NSG / ROUNDS, JPR = 2 / ROUNDS;   // stages / 32-channel halves per round
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds = smem;                                  // SPR halo buffers (conv0's channels of the current round)
  float* in = (float*)(smem + SPR * BUF);            // [3][12][36] input patch
  __shared__ float red[8];
  __shared__ float red2[4];
  float a0mx = 0.f;                                  // split mode, training: max of the relu(conv0) copy this thread wrote
#if HLA_CONV_STAMPS
  struct { unsigned long long* stamps; } a = {a0.stamps};
#endif
  HLA_STAMP(0);
  HLA_STAMP_HWID();

  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), wm = wv / WN, wn = wv % WN;
  int bid = blockIdx.x;                    // (the XCD-contiguous order measured 2-3 % slower for this kernel)
  const int tx = bid % a0.tiles_x; bid /= a0.tiles_x;
  const int ty = bid % a0.tiles_y;
  const int b = bid / a0.tiles_y;
  const int y0 = a0.row_begin + ty * TH, x0 = tx * 32;
  const int x = lane & 31, g = lane >> 5;

  // start the first conv2 weight loads before anything else (they do not depend on the input)
  WeightRing<T, MT, NT, WD> ring;
  ring.wq[0] = a0.w2 + (size_t)wn * NSG * 18 * 64 + lane;
  ring.prime();

  // phase A: input patch (2-pixel border) -> LDS
  // (all of a thread's loads are requested before the first one is awaited: as a rolled loop this was six dependent
  //  load -> wait -> LDS write round trips at the head of every workgroup's 11 us life)
  float amx = 0.f;
  constexpr int NIN = (3 * IH * IW + 255) / 256;
  float vin[NIN];
#pragma unroll
  for (int it = 0; it < NIN; ++it) {
    const int e = t + it * 256;
    const int c = e / (IH * IW), r = e % (IH * IW), iy = r / IW, ix = r % IW;
    const int y = y0 - 2 + iy, xx = x0 - 2 + ix;
    vin[it] = 0.f;
    if (e < 3 * IH * IW && y >= 0 && y < a0.H && xx >= 0 && xx < a0.W)
      vin[it] = a0.x[((size_t)b * 3 + c) * a0.x_plane + (size_t)y * a0.W + xx];
  }
#pragma unroll
  for (int it = 0; it < NIN; ++it) {
    const int e = t + it * 256;
    if (e < 3 * IH * IW) in[e] = vin[it];
    if (SPLIT) amx = fmaxf(amx, fabsf(vin[it]));
  }
  // conv0's weight fragments and biases of the 32-channel halves ONE round produces ([..][0] = hi, [..][1] = lo in split mode).
  // With two rounds (4-byte activation types) they are fetched per round: holding both halves' cost 32 registers, which kept
  // the exact-fp32 kernel (176) above the 168 of three workgroups per CU (3983 -> 3884 us per launch).
  uint4 wf0[JPR][NFRAG][NH];
  auto load_w0 = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < JPR; ++jj)
#pragma unroll
      for (int f = 0; f < NFRAG; ++f)
#pragma unroll
        for (int h = 0; h < NH; ++h) wf0[jj][f][h] = a0.w0[(((j0 + jj) * NFRAG + f) * NH + h) * 64 + lane];
  };
  load_w0(0);
  if (SPLIT) {
    amx = wave_max_f32(amx);
    if (lane == 0) red[4 + wv] = amx;
  }
  __syncthreads();
  HLA_STAMP(1);

  // phase B: conv0 on the 10x34 halo pixels, 32 pixels per MFMA tile, straight into the conv2 halo buffers
  float4 bias0[JPR][4];
  auto load_b0 = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < JPR; ++jj)
#pragma unroll
      for (int q = 0; q < 4; ++q) bias0[jj][q] = *(const float4*)(a0.b0 + (j0 + jj) * 32 + q * 8 + g * 4);
  };
  // split mode: every scale is local to this block (a power-of-two scale never changes a result).  Input patch: from its own
  // maximum.  relu(conv0): from the bound  max_co sum_k |w0[co][k]| * max |x| + max |b0|  (it is split into hi / lo while it is
  // produced, before its true maximum could be known; the bound is a few binades loose, the window is 17 binades wide).
  float s_in = 1.f, s_a0 = 1.f, d0 = 1.f, dsc2 = 1.f;
  if constexpr (SPLIT) {
    const float m = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    float bm = fabsf(a0.b0[lane]);                   // max |b0| over its 64 channels
    bm = wave_max_f32(bm);
    s_in = split_scale(__float_as_uint(m));
    s_a0 = split_scale(__float_as_uint(a0.wtail[16] * m + bm));
    d0 = 1.f / (s_in * a0.wtail[0]);
    dsc2 = 1.f / (s_a0 * a0.wtail[1]);
  }
  // conv2's accumulators and weight stream live across the rounds
  f32x16 acc[MT][NT];
  float4 bias2[4];                                   // 4-byte types: conv2's bias of this wave's channels, for the epilogue
#pragma unroll
  for (int q = 0; q < 4; ++q) bias2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  const FragOff fo = frag_offsets(lane, wm * MT);
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
  const int j0 = rd * JPR;                           // first 32-channel half of conv0's output this round produces
  if (rd == 0 && !FOLD) load_b0(0);                  // (later rounds: requested at the end of the round before, ahead of its a0 stores)
  if (rd > 0) __syncthreads();                       // the previous round's MFMAs (and its a0 copy) are done with the buffers
  for (int m = wv; m * 32 < HPIX; m += 4) {
    const int p = m * 32 + x, pc = p < HPIX ? p : HPIX - 1;
    const int hy = pc / HWID, hx = pc - hy * HWID;
    const float* ib = in + hy * IW + hx;
    // conv2 zero-pads conv0's OUTPUT map: halo pixels outside the image are 0, not conv0 of padded input
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool inside = yy >= 0 && yy < a0.H && xx >= 0 && xx < a0.W;
    f32x16 c0[2];
#pragma unroll
    for (int j = j0; j < j0 + JPR; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c0[j][r] = 0.f;
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) {
      float ev[EPL];
#pragma unroll
      for (int jj = 0; jj < EPL; ++jj) {
        // this lane's k is klo (lanes 0-31) or khi (lanes 32-63): select the OFFSET, then read once
        const int kl = f * 2 * EPL + jj, kh = kl + EPL;      // compile-time
        const int ol = kl < 27 ? (kl / 9) * IH * IW + ((kl % 9) / 3) * IW + (kl % 9) % 3 : 0;
        const int oh = kh < 27 ? (kh / 9) * IH * IW + ((kh % 9) / 3) * IW + (kh % 9) % 3 : 0;
        float v = ib[g ? oh : ol];
        if (kh >= 27 && g) v = (FOLD && kh <= 28) ? 1.f : 0.f;      // padded k (27..31) only occurs in the upper half;
        if (kl >= 27 && !g) v = (FOLD && kl <= 28) ? 1.f : 0.f;     //   FOLD: 27 / 28 multiply the bias (hi, lo)
        ev[jj] = v;
      }
      if constexpr (SPLIT) {
        uint2 h0, l0, h1, l1;
        split4(ev[0], ev[1], ev[2], ev[3], s_in, h0, l0);
        split4(ev[4], ev[5], ev[6], ev[7], s_in, h1, l1);
        const uint4 phi = make_uint4(h0.x, h0.y, h1.x, h1.y), plo = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
        for (int j = j0; j < j0 + JPR; ++j) {
          mma16<T>(c0[j], wf0[j - j0][f][0], phi);
          mma16<T>(c0[j], wf0[j - j0][f][NH - 1], phi);
          mma16<T>(c0[j], wf0[j - j0][f][0], plo);
        }
      } else {
        T e[EPL];
#pragma unroll
        for (int jj = 0; jj < EPL; ++jj) e[jj] = (T)ev[jj];
        uint4 pf = __builtin_bit_cast(uint4, e);
        if (FOLD && !inside) pf = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = j0; j < j0 + JPR; ++j) mma16<T>(c0[j], wf0[j - j0][f][0], pf);
      }
    }
    if (p < HPIX) {
#pragma unroll
      for (int j = j0; j < j0 + JPR; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = j * 32 + q * 8 + g * 4, sl = co / KC - rd * SPR;      // channel, its stage buffer within this round
          const int byte = (co % KC) * (SPLIT ? 2 : (int)sizeof(T));           // where the 4 channels start inside the pixel's stage
          float v0, v1, v2, v3;
          if constexpr (FOLD) {
            v0 = fmaxf(c0[j][q * 4 + 0], 0.f); v1 = fmaxf(c0[j][q * 4 + 1], 0.f);
            v2 = fmaxf(c0[j][q * 4 + 2], 0.f); v3 = fmaxf(c0[j][q * 4 + 3], 0.f);
          } else {
            const float4 bb = bias0[j - j0][q];
            v0 = fmaxf(c0[j][q * 4 + 0] * d0 + bb.x, 0.f); v1 = fmaxf(c0[j][q * 4 + 1] * d0 + bb.y, 0.f);
            v2 = fmaxf(c0[j][q * 4 + 2] * d0 + bb.z, 0.f); v3 = fmaxf(c0[j][q * 4 + 3] * d0 + bb.w, 0.f);
            if (!inside) v0 = v1 = v2 = v3 = 0.f;
          }
          if constexpr (SPLIT) {
            uint2 hi, lo;
            split4(v0, v1, v2, v3, s_a0, hi, lo);
            *(uint2*)(lds + sl * BUF + halo_off(p, hx, byte >> 4) + (byte & 15)) = hi;
            *(uint2*)(lds + sl * BUF + halo_off(p, hx, 2 + (byte >> 4)) + (byte & 15)) = lo;
          } else {
            store4((T*)(lds + sl * BUF + halo_off(p, hx, byte >> 4) + (byte & 15)), v0, v1, v2, v3);
          }
        }
    }
  }
  HLA_STAMP(2);
  __syncthreads();
  HLA_STAMP(3);

  // phase C: conv2 over this round's resident stages (no further loads, no barriers)
  if (rd == 0) {                // (16-bit types: the accumulators start at conv2's bias, like conv3x3_kernel's)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sizeof(T) == 2) b4 = *(const float4*)(a0.b2 + wn * 32 + q * 8 + g * 4);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][0][q * 4 + 0] = b4.x; acc[i][0][q * 4 + 1] = b4.y; acc[i][0][q * 4 + 2] = b4.z; acc[i][0][q * 4 + 3] = b4.w;
      }
    }
  }
  if (rd == 0) stagger_priority();
  // What the code BEHIND this round's a0 copy loads first -- the next round's conv0 fragments / biases, or conv2's bias for the
  // epilogue -- is requested here, ahead of the round's MFMAs, and pinned below: vmcnt retires in issue order, so a load queued
  // behind the copy's stores makes its consumer wait for their write acknowledgements.
  if (rd + 1 < ROUNDS) { load_w0(j0 + JPR); if (!FOLD) load_b0(j0 + JPR); }
  else if (sizeof(T) == 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bias2[q] = *(const float4*)(a0.b2 + wn * 32 + q * 8 + g * 4);
  }
#pragma unroll 1
  for (int sg = 0; sg < SPR; ++sg)
    stage_mma<T, MT, NT, WD>(acc, lds + sg * BUF, fo, ring, [](int) {});
  HLA_STAMP(4);
  if (rd + 1 < ROUNDS) {
#pragma unroll
    for (int jj = 0; jj < JPR; ++jj) {
#pragma unroll
      for (int f = 0; f < NFRAG; ++f)
#pragma unroll
        for (int h = 0; h < NH; ++h) asm volatile("" : "+v"(wf0[jj][f][h].x), "+v"(wf0[jj][f][h].y), "+v"(wf0[jj][f][h].z), "+v"(wf0[jj][f][h].w));
      if (!FOLD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bias0[jj][q].x), "+v"(bias0[jj][q].y), "+v"(bias0[jj][q].z), "+v"(bias0[jj][q].w));
      }
    }
  } else if (sizeof(T) == 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bias2[q].x), "+v"(bias2[q].y), "+v"(bias2[q].z), "+v"(bias2[q].w));
  }
  // (last round: the weight ring's look-ahead past the last tap is still in flight; the epilogue re-uses those registers, and at
  //  the join behind the branchy copy below the compiler can no longer count what is pending -- it would wait with vmcnt(0) for
  //  the copy's stores.  Drained here, before the stores, that wait is gone.)
  if (rd + 1 == ROUNDS) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched (both paths: they join below)
  if (a0.a0_out) {     // training: the backward pass needs relu(conv0) (conv2's wgrad input and ReLU mask)
    // Copied out of the halo buffers AFTER the round's MFMAs (nothing overwrites them before the next barrier), branch-free and
    // unrolled: raw buffer stores, a pixel outside the image gets an offset beyond the descriptor's range and is dropped.  Round 4
    // issued this copy between phase B's barrier and phase C from a rolled loop with a branch around each store: the compiler
    // could not count the stores in flight, so the first weight fragments of phase C waited with vmcnt(0) -- for the write
    // acknowledgements of the copy, twice per workgroup (training variant 3.04 ms per launch against 1.83 without the copy).
    constexpr int PPP = 64 * (int)sizeof(T) / 16 / ROUNDS;          // 16-B pieces per pixel over this round's stages
    static_assert(TH * 32 * PPP % 256 == 0, "a0 pieces per thread");
    const size_t a0s = (size_t)a0.H * a0.W * 64 * sizeof(T);
    const unsigned long long pb = (unsigned long long)a0.a0_out + (size_t)b * a0s;
    const void* pu = (const void*)(((unsigned long long)__builtin_amdgcn_readfirstlane((int)(unsigned)(pb >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pb));
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, __builtin_amdgcn_readfirstlane((int)a0s), 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int it = 0; it < TH * 32 * PPP / 256; ++it) {
      const int e = t + it * 256;
      const int pxl = e / PPP, piece = e % PPP, sgi = piece / 4, part = piece % 4;
      const int r = pxl / 32, c = pxl % 32, yy = y0 + r, xx = x0 + c;
      const bool ok = yy < a0.H && xx < a0.W;
      const int off = ok ? (yy * a0.W + xx) * 64 * (int)sizeof(T) + (rd * PPP + piece) * 16 : (int)0x80000000;
      const char* sb = lds + sgi * BUF;
      const int hpix = (r + 1) * HWID + c + 1;
      if constexpr (SPLIT) {      // what conv2 actually consumes: (hi + lo) / s, 23 of relu(conv0)'s 24 significand bits
        const f16x4 h = __builtin_bit_cast(f16x4, *(const uint2*)(sb + halo_off(hpix, c + 1, part >> 1) + (part & 1) * 8));
        const f16x4 l = __builtin_bit_cast(f16x4, *(const uint2*)(sb + halo_off(hpix, c + 1, 2 + (part >> 1)) + (part & 1) * 8));
        const float is = 1.f / s_a0;
        const float4 o4 = make_float4(((float)h[0] + (float)l[0]) * is, ((float)h[1] + (float)l[1]) * is,
                                      ((float)h[2] + (float)l[2]) * is, ((float)h[3] + (float)l[3]) * is);
#if HLA_A0_ABL == 1       // timing-only ablations of the copy (tools/ab_libs.py): 1 = no store instruction, 2 = nt stores
        if (o4.x == 12345.678f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), ra, off, 0, 0);
#elif HLA_A0_ABL == 2
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), ra, off, 0, 2);
#else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), ra, off, 0, 0);
#endif
        if (ok) a0mx = fmaxf(fmaxf(a0mx, fmaxf(o4.x, o4.y)), fmaxf(o4.z, o4.w));        // (post-ReLU: non-negative)
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *(const uint4*)(sb + halo_off(hpix, c + 1, part))), ra, off, 0, 0);
      }
    }
  }
  }     // rounds
  // the copy's per-sample maximum: ONE filtered atomic per workgroup.  (Round 4 issued one per wave and round -- 262 144 atomics per
  // launch on 32 addresses, which the L2 executes one after the other per address: 1.2 ms of the training variant's 3.0, taken for
  // the cost of the copy's stores until an ablation without any store instruction measured the same 3.0 ms.)
  if (SPLIT && a0.a0_out && a0.amax_a0_out) {       // kernel-uniform
    a0mx = wave_max_f32(a0mx);
    if (lane == 0) red2[wv] = a0mx;
  }

#if HLA_CONV_STAMPS
  ConvArgs ae{};
#define a ae
#else
  ConvArgs a{};
#endif
  a.bias = a0.b2; a.out_act = a0.out_act; a.B = a0.B; a.H = a0.H; a.W = a0.W; a.Cout = 64; a.relu_act = 1;
  a.tiles_x = a0.tiles_x; a.tiles_y = a0.tiles_y; a.idx_out = a0.idx_out; a.amax_out = a0.amax_out;
  __syncthreads();   // all waves are done with the halo buffers; reuse them as wave-private stagers
  if (SPLIT && a0.a0_out && a0.amax_a0_out && t == 0) {
    const unsigned mb = __float_as_uint(fmaxf(fmaxf(red2[0], red2[1]), fmaxf(red2[2], red2[3])));
    // (the plain read only filters: a stale value costs one redundant atomic, never a missed one)
    if (mb > __hip_atomic_load(a0.amax_a0_out + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a0.amax_a0_out + b, mb);
  }
  const float4* pb2 = sizeof(T) == 4 ? bias2 : nullptr;
  if (a0.a2_out) {   // the epilogue only reads the accumulators: run it twice, un-pooled first
    ConvArgs f = a;
    f.out_act = a0.a2_out; f.idx_out = nullptr; f.amax_out = a0.amax_a2_out;
    conv_epilogue<T, MT, NT, false, EPI_ACT, false, SPR * BUF / 4>(acc, f, b, y0 + wm * MT, x0, wn * 32, red, lds + wv * (SPR * BUF / 4), dsc2, PixBox{0, 1 << 30, 0, 1 << 30}, pb2);
    if (SPLIT) __syncthreads();     // `red` is reused by the second epilogue's maximum
  }
  if (a.idx_out) conv_epilogue<T, MT, NT, true, EPI_GENERIC>(acc, a, b, y0 + wm * MT, x0, wn * 32, red, lds + wv * (SPR * BUF / 4), dsc2, PixBox{0, 1 << 30, 0, 1 << 30}, pb2);
  else conv_epilogue<T, MT, NT, true, EPI_ACT, false, SPR * BUF / 4>(acc, a, b, y0 + wm * MT, x0, wn * 32, red, lds + wv * (SPR * BUF / 4), dsc2, PixBox{0, 1 << 30, 0, 1 << 30}, pb2);
#if HLA_CONV_STAMPS
#undef a
  { struct { unsigned long long* stamps; } a = {a0.stamps}; HLA_STAMP(5); }
#endif
}

// ---------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 -> MFMA fragment order, T elements.
//   generic: idx = ((((nt*nstage + sg)*9 + tap)*2 + kg)*64 + lane)*EPL + j
//            cout = nt*32 + (lane&31), cin = sg*KC + kg*2*EPL + (lane>>5)*EPL + j      (KC = 64 B of channels)
//   conv0:   idx = ((nt*NFRAG + f)*64 + lane)*EPL + j,  k = f*2*EPL + (lane>>5)*EPL + j  (k = cin*9+tap, <27)
//   Every layer's buffer is padded by two taps of fragments (the kernels prefetch up to 2 taps ahead).
//   mode 2 (dgrad): the packed conv is the transpose: Cout' = Cin_orig, Cin' = Cout_orig, taps flipped:
//            v = w_orig[cin'][cout'][8 - tap]  with w_orig laid out [Cout_orig = Cin'][Cin_orig = Cout'][9]
// conv0 (first == 1) with `b0`: k = 27 / 28 hold the bias as (hi, lo) of the packed type -- the fused kernel feeds 1.0 there
// for the 16-bit types, so the bias rides in the MFMA (its rounding error 2^-17 relative; for T = float lo is exactly 0)
template <typename T>
__device__ __forceinline__ void pack_weights_body(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int first,
                                                  size_t e0, size_t stride, const float* __restrict__ b0 = nullptr) {
  constexpr int EPL = 16 / sizeof(T), KC = SB / sizeof(T), NFRAG = 32 / (2 * EPL);
  const size_t total = first == 1 ? (size_t)(Cout / 32) * NFRAG * 64 * EPL : (size_t)Cout * Cin * 9;
  for (size_t e = e0; e < total; e += stride) {
    size_t r = e;
    const int j = r % EPL; r /= EPL;
    const int lane = r % 64; r /= 64;
    float v = 0.f;
    if (first == 1) {
      const int f = r % NFRAG; r /= NFRAG;
      const int nt = (int)r;
      const int k = f * 2 * EPL + (lane >> 5) * EPL + j, cout = nt * 32 + (lane & 31);
      if (k < 27) v = w[(size_t)cout * 27 + k];
      else if (b0 && k == 27) v = (float)(T)b0[cout];
      else if (b0 && k == 28) v = b0[cout] - (float)(T)b0[cout];
    } else {
      const int kg = r % 2; r /= 2;
      const int tap = r % 9; r /= 9;
      const int nsg = Cin / KC;
      const int sg = r % nsg; r /= nsg;
      const int nt = (int)r;
      const int cout = nt * 32 + (lane & 31), cin = sg * KC + kg * 2 * EPL + (lane >> 5) * EPL + j;
      v = first == 2 ? w[((size_t)cin * Cout + cout) * 9 + (8 - tap)] : w[((size_t)cout * Cin + cin) * 9 + tap];
    }
    out[e] = (T)v;
  }
}
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int first) {
  pack_weights_body<T>(w, out, Cout, Cin, first, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
// every layer of a network in ONE launch (blockIdx.y = table row): a training step repacks both directions of both networks,
// 44 launches of a few microseconds each when done layer by layer
struct PackTable {
  const float* w[16]; size_t off[16];      // source weights, byte offset of the packed layer
  int cout[16], cin[16], first[16];
  const float* b0;                         // conv0's bias (rides in its padded k slots, see pack_weights_body), or null
};
template <typename T>
__global__ void pack_weights_multi_kernel(PackTable tb, char* __restrict__ packed) {
  const int l = blockIdx.y;
  pack_weights_body<T>(tb.w[l], (T*)(packed + tb.off[l]), tb.cout[l], tb.cin[l], tb.first[l],
                       (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, tb.first[l] == 1 ? tb.b0 : nullptr);
}

// Split mode, every layer of a network in ONE launch each (blockIdx.y = table row), like pack_weights_multi_kernel: a training
// step repacks both directions of both networks after the optimizer step -- 84 launches of ~10 us when done layer by layer.
struct SplitPackTable {
  const float* w[16]; size_t off[16];      // source weights, byte offset of the packed layer
  int cout[16], cin[16], first[16], slot[16];      // slot: the layer's index into the |w| maxima / scale words behind the packed layers
};
// Split-fp16 packing: the same fragment order with the two k-groups of a stage replaced by (hi, lo) of ONE 16-channel
// k-group, elements fp16 of s_w * w:
//   generic: idx = ((((nt*nstage + sg)*9 + tap)*2 + hl)*64 + lane)*8 + j,  cout = nt*32 + (lane&31), cin = sg*16 + (lane>>5)*8 + j
//   conv0:   idx = (((nt*2 + f)*2 + hl)*64 + lane)*8 + j,                   k = f*16 + (lane>>5)*8 + j  (k = cin*9+tap, <27)
// wmax_bits = fp32 bit pattern of max |w| (absmax_kernel); thread 0 of block 0 publishes the scale for the conv kernels.
static __device__ __forceinline__ void pack_weights_split_body(const float* __restrict__ w, f16* __restrict__ out, int Cout, int Cin, int first,
                                                               const unsigned* __restrict__ wmax_bits, float* __restrict__ scale_out) {
  const float sw = split_scale(*wmax_bits);
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = sw;
  const size_t total = first == 1 ? (size_t)(Cout / 32) * 2 * 2 * 64 * 8 : (size_t)Cout * Cin * 9 * 2;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    size_t r = e;
    const int j = r % 8; r /= 8;
    const int lane = r % 64; r /= 64;
    const int hl = r % 2; r /= 2;
    float v = 0.f;
    if (first == 1) {
      const int f = r % 2; r /= 2;
      const int nt = (int)r;
      const int k = f * 16 + (lane >> 5) * 8 + j, cout = nt * 32 + (lane & 31);
      if (k < 27) v = w[(size_t)cout * 27 + k];
    } else {
      const int tap = r % 9; r /= 9;
      const int nsg = Cin / 16;
      const int sg = r % nsg; r /= nsg;
      const int nt = (int)r;
      const int cout = nt * 32 + (lane & 31), cin = sg * 16 + (lane >> 5) * 8 + j;
      // first == 2 (data gradient): the transposed, tap-flipped convolution, as pack_weights_body's mode 2
      v = first == 2 ? w[((size_t)cin * Cout + cout) * 9 + (8 - tap)] : w[((size_t)cout * Cin + cin) * 9 + tap];
    }
    v *= sw;
    const f16 h = (f16)v;
    out[e] = hl ? (f16)(v - (float)h) : h;
  }
}
static __global__ void pack_weights_split_kernel(const float* __restrict__ w, f16* __restrict__ out, int Cout, int Cin, int first,
                                          const unsigned* __restrict__ wmax_bits, float* __restrict__ scale_out) {
  pack_weights_split_body(w, out, Cout, Cin, first, wmax_bits, scale_out);
}
static __global__ void pack_weights_split_multi_kernel(SplitPackTable tb, char* __restrict__ packed, const unsigned* __restrict__ amax,
                                                       float* __restrict__ scales) {
  const int l = blockIdx.y;
  pack_weights_split_body(tb.w[l], (f16*)(packed + tb.off[l]), tb.cout[l], tb.cin[l], tb.first[l], amax + tb.slot[l], scales + tb.slot[l]);
}
static __global__ void absmax_multi_kernel(SplitPackTable tb, unsigned* __restrict__ amax, unsigned* __restrict__ l1max0) {
  const int l = blockIdx.y;
  const float* w = tb.w[l];
  const size_t n = (size_t)tb.cout[l] * tb.cin[l] * 9;
  float m = 0.f;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[e]));
  m = wave_max_f32(m);
  if ((threadIdx.x & 63) == 0) atomicMax(amax + tb.slot[l], __float_as_uint(m));
  if (tb.first[l] == 1 && l1max0 && blockIdx.x == 0)      // conv0: max over output channels of sum_k |w[co][k]| (the bound its fused kernel scales by)
    for (size_t r0 = threadIdx.x; r0 * 27 < n; r0 += blockDim.x) {
      float sum = 0.f;
      for (int k = 0; k < 27; ++k) sum += fabsf(w[r0 * 27 + k]);
      atomicMax(l1max0, __float_as_uint(sum));
    }
}

// max |w| (as fp32 bits) and, for conv0, max over output channels of sum_k |w[co][k]| (row = 27): both atomicMax'ed into
// zero-initialised words
static __global__ void absmax_kernel(const float* __restrict__ w, size_t n, int row, unsigned* __restrict__ amax, unsigned* __restrict__ l1max) {
  float m = 0.f;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[e]));
  m = wave_max_f32(m);
  if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(m));
  if (l1max && blockIdx.x == 0)
    for (size_t r0 = threadIdx.x; r0 * row < n; r0 += blockDim.x) {
      float s = 0.f;
      for (int k = 0; k < row; ++k) s += fabsf(w[r0 * row + k]);
      atomicMax(l1max, __float_as_uint(s));
    }
}

// ---------------------------------------------------------------------------------------------
// confidence head: sigmoid(-sigmoid(conv3x3(relu(x), C->1)))  VGG.py:62-81,160-163.  `act` is already ReLU'd.
// One block per 8x32 output tile.  Every pixel of the 10x34 halo is read ONCE and gives its nine per-tap dot products
// td[pixel][tap] = sum_c act[pixel][c] * w[c][tap] to LDS; an output pixel then adds the nine values its neighbours hold for
// it.  The dot products are a [9 (padded to 32) x C] x [C x 32 pixels] matrix product per wave: the weights sit in registers as
// MFMA row fragments (hi + lo halves for the 16-bit types: the product is as exact as an fp32 FMA chain on the stored
// activations) and a pixel's channels go from global memory straight into the column fragment (NHWC: 16 contiguous bytes per
// lane and K-step).  (History: gathering the nine neighbours per output pixel took 560 us for the three maps of a training
// step -- nine passes over the activations through the caches; a scalar version of the present scheme was bound by its
// 9 log2(C/8) cross-lane reduction steps per pixel and no faster.)
constexpr int CONF_TH = 8, CONF_TW = 32, CONF_HW = CONF_TW + 2, CONF_HPIX = (CONF_TH + 2) * CONF_HW;
constexpr int CONF_NT = (CONF_HPIX + 31) / 32;             // 32-pixel MFMA tiles per block
template <typename T, int C>
__global__ __launch_bounds__(256) void conf_kernel(const T* __restrict__ act, const float* __restrict__ w,
                                                   float* __restrict__ out, int B, int H, int W) {
  constexpr int EPL = 16 / sizeof(T), KS = 2 * EPL, NS = C / KS;      // channels per lane / per K-step, K-steps
  constexpr bool SPLITW = sizeof(T) == 2;
  __shared__ float td[CONF_NT * 32 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 31, g = lane >> 5;
  const int tiles_x = (W + CONF_TW - 1) / CONF_TW, tiles_y = (H + CONF_TH - 1) / CONF_TH;
  int bid = blockIdx.x;
  const int x0 = (bid % tiles_x) * CONF_TW; bid /= tiles_x;
  const int y0 = (bid % tiles_y) * CONF_TH;
  const size_t b = bid / tiles_y;
  // row fragments of the weights: row n = tap (zero for n >= 9), this lane's channels of K-step s: s*KS + g*EPL + [0, EPL)
  uint4 wh[NS], wl[SPLITW ? NS : 1];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    T hi[EPL], lo[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const float v = n < 9 ? w[(s * KS + g * EPL + k) * 9 + n] : 0.f;         // OIHW (O=1): w[c*9+tap]
      hi[k] = (T)v;
      lo[k] = (T)(v - to_f32(hi[k]));
    }
    __builtin_memcpy(&wh[s], hi, 16);
    if (SPLITW) __builtin_memcpy(&wl[s], lo, 16);
  }
  for (int t = wv; t < CONF_NT; t += 4) {
    const int hp = t * 32 + n, hy = hp / CONF_HW, hx = hp - hy * CONF_HW, y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = hp < CONF_HPIX && y >= 0 && y < H && x >= 0 && x < W;
    const T* src = act + ((b * H + (ok ? y : 0)) * W + (ok ? x : 0)) * C + g * EPL;
    uint4 a[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = ok ? *(const uint4*)(src + s * KS) : make_uint4(0, 0, 0, 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      mma16<T>(acc, wh[s], a[s]);
      if (SPLITW) mma16<T>(acc, wl[s], a[s]);
    }
    // D[tap][pixel]: lane -> pixel n; register r -> tap (r&3) + 8(r>>2) + 4g
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int tap = (r & 3) + 8 * (r >> 2) + 4 * g;
      if (tap < 9) td[hp * 9 + tap] = acc[r];
    }
  }
  __syncthreads();
  const int py = threadIdx.x / CONF_TW, px = threadIdx.x % CONF_TW, y = y0 + py, x = x0 + px;   // 256 threads = the 8x32 tile
  if (y < H && x < W) {
    float s = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) s += td[((py + tap / 3) * CONF_HW + px + tap % 3) * 9 + tap];
    const float sg = 1.f / (1.f + __expf(-s));
    out[(b * H + y) * W + x] = 1.f / (1.f + __expf(sg));
  }
}

// ---------------------------------------------------------------------------------------------
// L2_norm (VGG.py:511-514): x / max(||x||, 1e-12) per sample.
// inv_norm_kernel: fixed-order fp64 sum of the epilogue partials -> 1/max(||x||,1e-12) per sample.
static __global__ __launch_bounds__(256) void inv_norm_kernel(const double* __restrict__ sumsq, int np, double* __restrict__ inv) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += sumsq[(size_t)b * np + i];
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) inv[b] = 1.0 / fmax(sqrt((sh[0] + sh[1]) + (sh[2] + sh[3])), 1e-12);
}
// all levels of a network in one launch: blockIdx.y = level
struct InvNormArgs { const double* ss[4]; int np[4]; double* inv; int B; };
static __global__ __launch_bounds__(256) void inv_norm_multi_kernel(InvNormArgs a) {
  __shared__ double sh[4];
  const int b = blockIdx.x, l = blockIdx.y, np = a.np[l];
  const double* sumsq = a.ss[l];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += sumsq[(size_t)b * np + i];
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) a.inv[(size_t)l * a.B + b] = 1.0 / fmax(sqrt((sh[0] + sh[1]) + (sh[2] + sh[3])), 1e-12);
}
// scale_kernel: in-place x *= inv[b] (fp64 multiply, one rounding)
static __global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, const double* __restrict__ inv, size_t per_sample,
                                                    int blocks_per_sample) {
  const int b = blockIdx.x / blocks_per_sample, k = blockIdx.x % blocks_per_sample;
  const double scale = inv[b];
  float4* p = (float4*)(x + (size_t)b * per_sample);
  const size_t n4 = per_sample / 4;
  for (size_t i = (size_t)k * 256 + threadIdx.x; i < n4; i += (size_t)blocks_per_sample * 256) {
    float4 v = p[i];
    v.x = (float)((double)v.x * scale); v.y = (float)((double)v.y * scale);
    v.z = (float)((double)v.z * scale); v.w = (float)((double)v.w * scale);
    p[i] = v;
  }
}


// ---------------------------------------------------------------------------------------------
// host: pick the tile configuration for a 3x3 conv launch (forward convs and the dgrad convs of the backward)
// BWD: the caller is the backward pass, whose conv2 / conv7 / conv14 data gradients read a virtually un-pooled source
// (ConvArgs::unpool_idx): only then are the UNPOOL instantiations compiled into the translation unit.
// Returns false (with the library's error string set, nothing launched) for a combination no kernel was compiled for.
template <typename T, bool BWD = false>
static bool launch_conv(hipStream_t st, ConvArgs a, bool pool) {
  if (a.unpool_idx && !(BWD && !pool)) {      // (cannot happen from this library's callers; the plain kernels ignore the field)
    hla_set_error("launch_conv: an un-pooled source needs the UNPOOL kernel (backward, no pooling epilogue)");
    return false;
  }
  a.tiles_x = (a.W + 31) / 32;
  a.tiles_y = (a.H - a.row_begin + 7) / 8;
#if HLA_CONV_STAMPS
  a.stamps = nullptr;
  if (g_hla_stamp.buf && g_hla_stamp.counter++ == g_hla_stamp.want) {
    a.stamps = g_hla_stamp.buf;
    g_hla_stamp.grid_x = a.tiles_x * a.tiles_y * a.B; g_hla_stamp.grid_y = a.Cout >= 128 ? a.Cout / 128 : 1;
  }
#endif
  // (the 64-channel block at three workgroups per CU for the Cout >= 128 layers as well: 259 against 244 us per launch, same-box A/B)
  const bool big = a.Cout >= 128;
  const dim3 grid(a.tiles_x * a.tiles_y * a.B, big ? a.Cout / 128 : 1);
  const size_t es = sizeof(T), P = (size_t)a.B * (a.H - a.row_begin) * a.W, Po = pool ? P / 4 : P;
  const double flops = 2.0 * 9.0 * (a.C1 + a.C2) * a.Cout * (double)P + (a.wg0_part ? 2.0 * 27 * 64 * (double)P : 0.0);
  const double bytes = (double)P * ((a.up1 ? a.C1 / 4.0 : a.C1) + a.C2) * es + (double)Po * a.Cout * ((a.out_act ? es : 0) + (a.out_raw ? 4 : 0));
  // (a data-dependent launch visits n_live of its tiles_x * tiles_y tiles per sample: the record carries the executed share)
  hla_prof_begin_dyn(a.Cout >= 128 ? (pool ? K_CONV_NT2_POOL : K_CONV_NT2) : (pool ? K_CONV_NT1_POOL : K_CONV_NT1), flops, bytes, st,
                     a.dyn ? a.dyn + a.dyn_desc : nullptr, a.tiles_x * a.tiles_y);
  // A forward launch that cannot fill the chip (a single pair's H/4 layers are 64-128 workgroups on 256 CUs, and the forward
  // is then a chain of such launches) takes 4-row tiles (MT = 2): twice the workgroups, each with half the work -- every output
  // element's sum is formed in the same order, so the result is bit-identical.  Not for the three feature layers: their
  // sum-of-squares partials are per tile, and a sample's L2 norm must not depend on its batch mates to the last bit.
  if constexpr (!BWD) {
    const int gy = big ? a.Cout / 128 : 1;
    if (!a.dyn && !a.sumsq && !a.unpool_idx && a.tiles_x * a.tiles_y * a.B * gy < HLA_CONV_SMALL_GRID) {
      a.tiles_y = (a.H - a.row_begin + 3) / 4;
      const dim3 g4(a.tiles_x * a.tiles_y * a.B, gy);
      if (big) {
        if (pool) hipLaunchKernelGGL((conv3x3_kernel<T, 2, 2, 2, 2, true, 1>), g4, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3x3_kernel<T, 2, 2, 2, 2, false, 1>), g4, dim3(256), 0, st, a);
      } else {
        if (pool) hipLaunchKernelGGL((conv3x3_kernel<T, 2, 1, 2, 2, true, 2>), g4, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3x3_kernel<T, 2, 1, 2, 2, false, 2>), g4, dim3(256), 0, st, a);
      }
      hla_prof_end(st);
      return true;
    }
  }
  // Cout >= 128: block = 8x32 pixels x 128 channels, waves 2(M) x 2(N), wave tile 128 px x 64 ch, weights 1 tap ahead
  // Cout == 64 : block = 8x32 pixels x  64 channels, waves 2 x 2,       wave tile 128 px x 32 ch, weights 2 taps ahead
  if (big) {
    // (weights two taps ahead for this tile as well: spills with the register-staged loader, -1 %; with the LDS-DMA loader it
    //  fits -- 232 registers -- and measures the same: 4876 / 4871 against 4870 / 4841 pairs/s)
    if (pool) hipLaunchKernelGGL((conv3x3_kernel<T, 4, 2, 2, 2, true, 1>), grid, dim3(256), 0, st, a);
    else if (BWD && a.unpool_idx) hipLaunchKernelGGL((conv3x3_kernel<T, 4, 2, 2, 2, false, 1, BWD>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_kernel<T, 4, 2, 2, 2, false, 1>), grid, dim3(256), 0, st, a);
  } else {
    if (pool) hipLaunchKernelGGL((conv3x3_kernel<T, 4, 1, 2, 2, true, 2>), grid, dim3(256), 0, st, a);
    else if (BWD && a.unpool_idx) hipLaunchKernelGGL((conv3x3_kernel<T, 4, 1, 2, 2, false, 2, BWD>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_kernel<T, 4, 1, 2, 2, false, 2>), grid, dim3(256), 0, st, a);
  }
  hla_prof_end(st);
  return true;
}
