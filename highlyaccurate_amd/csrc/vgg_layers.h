// Host-side layer table, packed-weight layout and workspace plan shared by vgg.hip and vgg_backward.hip.
#pragma once
#include "common.h"

struct LayerDef { int cin, cout, has_bias; };
// state-dict order (VGG.py:23-56): conv0,2,5,7,10,12,14, dec1.1, dec1.3, dec2.1, dec2.3, dec3.1, dec3.3
static const LayerDef kLayers[13] = {
    {3, 64, 1}, {64, 64, 1}, {64, 128, 1}, {128, 128, 1}, {128, 256, 1}, {256, 256, 1}, {256, 256, 1},
    {384, 128, 0}, {128, 128, 0}, {192, 64, 0}, {64, 64, 0},
    // conv_dec3.1 (128 -> 32) and conv_dec3.3 (32 -> 16) are run on ZERO-PADDED weights (the host pads them to 64 output /
    // 64 input channels), so every kernel keeps its 64-channel granularity; the padded channels are exactly zero
    {128, 64, 0}, {64, 64, 0}};
constexpr int kPackedLayers = 11;   // conv0..dec2.3: what level 3 and the backward pass use
constexpr int kAllLayers = 13;      // + conv_dec3.1/3 (level 4, forward only)

// bytes per stored activation element / per packed weight (split mode: fp32 activations, (hi, lo) fp16 weight pairs)
static inline size_t hla_elem_bytes(int dtype) { return (dtype == HLA_F32 || dtype == HLA_F16X3) ? 4 : 2; }
constexpr size_t kPackTailBytes = 256;   // HLA_F16X3: float[16] weight scales, float[16] = conv0 L1 bound, 32 scratch words

static inline size_t packed_bytes(int l, int dtype) {
  const size_t es = hla_elem_bytes(dtype);
  if (l == 0) return (size_t)2 * 32 * 32 * es;   // 2 ntiles x 32 (padded K) x 32 couts
  return (size_t)kLayers[l].cin * kLayers[l].cout * 9 * es + 4096;   // + two taps of fragments: prefetch overrun
}
static inline size_t packed_offset(int l, int dtype) {
  size_t o = 0;
  for (int i = 0; i < l; ++i) o += hla_align_up(packed_bytes(i, dtype), 256);
  return o;
}

// Forward workspace: every activation a later layer (or the backward pass) reads.  All maps NHWC, T elements.
// split mode: one per-sample maximum per activation map that a convolution reads
enum { AM_X3 = 0, AM_A5, AM_X8, AM_A10, AM_A12, AM_X15, AM_D1A, AM_X18, AM_D2A, AM_X21, AM_X2, AM_D3A, AM_X24, AM_A0, kAmaxSlots = 16 };

struct VggPlan {
  size_t x3, a5, x8, a10, a12, x15r, d1a, x18r, d2a, x21r;   // post-ReLU activations
  size_t x2r, d3a, x24r;                                     // level 4 only: relu(conv2) at full resolution, dec3 maps
  size_t ss[4], inv;                                         // sum-of-squares partials, 1/norm
  size_t a0, idx3, idx8, idx15;                              // training only: conv0 output, pool argmax (u8)
  size_t amax;                                               // split mode: [kAmaxSlots][B] per-sample max |activation| (fp32 bits)
  int np[4];
  size_t total;
};

static inline void vgg_plan(int B, int H, int W, int dtype, bool train, VggPlan* p, bool level4 = false) {
  const size_t es = hla_elem_bytes(dtype);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += hla_align_up(bytes, 256); return r; };
  const size_t P = (size_t)B * H * W;
  p->x3 = take(P / 4 * 64 * es);
  p->a5 = take(P / 4 * 128 * es);
  p->x8 = take(P / 16 * 128 * es);
  p->a10 = take(P / 16 * 256 * es);
  p->a12 = take(P / 16 * 256 * es);
  p->x15r = take(P / 64 * 256 * es);
  p->d1a = take(P / 16 * 128 * es);
  p->x18r = take(P / 16 * 128 * es);
  p->d2a = take(P / 4 * 64 * es);
  p->x21r = take(P / 4 * 64 * es);
  // sum-of-squares partials: one per (image tile, cout block) of the producing layer
  auto tiles = [](int h, int w) { return ((h + 7) / 8) * ((w + 31) / 32); };
  p->np[0] = tiles(H / 4, W / 4) * 2;   // conv14: Cout 256 in blocks of 128
  p->np[1] = tiles(H / 4, W / 4) * 1;   // dec1.3: Cout 128
  p->np[2] = tiles(H / 2, W / 2) * 1;   // dec2.3: Cout 64
  p->np[3] = tiles(H, W) * 1;           // dec3.3: Cout 16 (padded to 64)
  for (int i = 0; i < 4; ++i) p->ss[i] = take((size_t)B * p->np[i] * sizeof(double));
  p->inv = take((size_t)4 * B * sizeof(double));
  p->amax = take((size_t)kAmaxSlots * B * sizeof(unsigned));
  p->x2r = p->d3a = p->x24r = 0;
  if (level4) {
    p->x2r = take(P * 64 * es);
    p->d3a = take(P * 64 * es);
    p->x24r = take(P * 64 * es);
  }
  p->a0 = p->idx3 = p->idx8 = p->idx15 = 0;
  if (train) {
    p->a0 = take(P * 64 * es);
    p->idx3 = take(P / 4 * 64);
    p->idx8 = take(P / 16 * 128);
    p->idx15 = take(P / 64 * 256);
  }
  p->total = o;
}
