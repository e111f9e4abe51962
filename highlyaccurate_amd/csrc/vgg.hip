// VGG16-U-Net forward: host orchestration (kernels live in conv_kernels.h).  VGG.py:13-203, 511-514.
#include "conv_kernels.h"
#include "vgg_layers.h"
#include <type_traits>
#include <stdlib.h>
#ifndef HLA_VGG_CHUNK_DEFAULT
#define HLA_VGG_CHUNK_DEFAULT 0      // 0: whole batch per launch
#endif
// Timing-only switches that are read from the environment (HLA_VGG_CHUNK, HLA_ABL_C02) exist in TOOLING builds only
// (`python -m highlyaccurate_amd.build --out=libhla_abl.so -DHLA_ABLATIONS=1`, selected with HLA_LIB): the product library reads no
// environment variable on its compute path (HLA_ABL_C02 leaves the backward's relu(conv0) / argmax buffers unwritten).
#ifndef HLA_ABLATIONS
#define HLA_ABLATIONS 0
#endif

template <typename T>
void vgg_pack_all(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st) {
  // split mode: the tail behind the last layer holds float[16] per-layer weight scales, float[16..31] (slot 16 = conv0's
  // L1 bound) and 32 scratch words for the two reductions
  float* tail = (float*)(packed + packed_offset(kAllLayers, dtype));
  unsigned* scratch = (unsigned*)tail + 32;
  if (Prec<T>::SPLIT) (void)hipMemsetAsync(tail, 0, kPackTailBytes, st);
  if constexpr (!Prec<T>::SPLIT) {
    PackTable tb{};
    tb.b0 = prm->b[0];
    int n = 0;
    for (int l = 0; l < kAllLayers; ++l) {
      if (l >= kPackedLayers && !prm->w[l]) continue;      // conv_dec3.* only when the caller supplies its padded weights
      tb.w[n] = prm->w[l]; tb.off[n] = packed_offset(l, dtype); tb.cout[n] = kLayers[l].cout; tb.cin[n] = kLayers[l].cin;
      tb.first[n] = l == 0 ? 1 : 0;
      ++n;
    }
    hla_prof_begin(K_PACK, 0, (double)packed_offset(kAllLayers, dtype) * (1.0 + 4.0 / sizeof(T)), st);
    hipLaunchKernelGGL((pack_weights_multi_kernel<T>), dim3(256, n), dim3(256), 0, st, tb, packed);
    hla_prof_end(st);
    return;
  }
  if constexpr (Prec<T>::SPLIT) {      // two launches for the whole network: the |w| maxima, then the (hi, lo) fragments scaled by them
    SplitPackTable tb{};
    int n = 0;
    for (int l = 0; l < kAllLayers; ++l) {
      if (l >= kPackedLayers && !prm->w[l]) continue;        // conv_dec3.* only when the caller supplies its padded weights
      tb.w[n] = prm->w[l]; tb.off[n] = packed_offset(l, dtype); tb.cout[n] = kLayers[l].cout; tb.cin[n] = kLayers[l].cin;
      tb.first[n] = l == 0 ? 1 : 0; tb.slot[n] = l;
      ++n;
    }
    hla_prof_begin(K_PACK, 0, (double)packed_offset(kAllLayers, dtype) * 3.0, st);
    hipLaunchKernelGGL(absmax_multi_kernel, dim3(64, n), dim3(256), 0, st, tb, scratch, (unsigned*)tail + 16);
    hipLaunchKernelGGL(pack_weights_split_multi_kernel, dim3(256, n), dim3(256), 0, st, tb, packed, (const unsigned*)scratch, tail);
    hla_prof_end(st);
  }
}

// Samples per launch of the full- / half-resolution layers (see vgg_forward_t).  Tooling builds (HLA_ABLATIONS): HLA_VGG_CHUNK=n overrides.
static int vgg_chunk(int B, int H, int W, size_t es) {
#if HLA_ABLATIONS
  static const int env = [] { const char* e = getenv("HLA_VGG_CHUNK"); return e ? atoi(e) : -1; }();
  int c = env >= 0 ? env : HLA_VGG_CHUNK_DEFAULT;
#else
  int c = HLA_VGG_CHUNK_DEFAULT;
#endif
  (void)H; (void)W; (void)es;
  if (c <= 0 || c > B) c = B;
  return c;
}

template <typename T>
int vgg_forward_t(const float* x, size_t x_plane, const hla_vgg_params* prm, const char* packed, int dtype, void* const feat[4],
                         float* const conf[4], double* inv_norm, char* ws, const VggPlan& pl, int B, int H, int W,
                         int flags, int first_row8, hipStream_t st) {
  const bool level4 = pl.x2r != 0;
  const int NL = level4 ? 4 : 3;
  auto W_ = [&](int l) { return (const uint4*)(packed + packed_offset(l, dtype)); };
  char* w = ws;
  constexpr bool SPLIT = Prec<T>::SPLIT;
  const float* wtail = (const float*)(packed + packed_offset(kAllLayers, dtype));       // split mode: weight scales
  unsigned* amax = (unsigned*)(w + pl.amax);                                           // split mode: [slot][B]
  auto AM = [&](int slot) { return (SPLIT && slot >= 0) ? amax + (size_t)slot * B : (unsigned*)nullptr; };
  if (SPLIT) HLA_CHECK_HIP(hipMemsetAsync(amax, 0, (size_t)kAmaxSlots * B * sizeof(unsigned), st));
  // conv0 + conv2 + pool fused (VGG.py:123-128): relu(x3)
  auto conv02 = [&](int b0, int nb) -> int {
    Conv02Args a{};
    const size_t es = sizeof(T), px = (size_t)H * W;      // (sample range [b0, b0 + nb): every pointer moves by b0 samples)
    a.x_plane = x_plane ? x_plane : px;
    a.x = x + (size_t)b0 * 3 * a.x_plane; a.w0 = W_(0); a.b0 = prm->b[0]; a.w2 = W_(1); a.b2 = prm->b[1];
    a.out_act = w + pl.x3 + (size_t)b0 * (px / 4) * 64 * es;
    if (flags & HLA_VGG_SAVE_FOR_BACKWARD) {
      a.a0_out = w + pl.a0 + (size_t)b0 * px * 64 * es;
      a.idx_out = (unsigned char*)(w + pl.idx3) + (size_t)b0 * (px / 4) * 64;
    }
    if (level4) a.a2_out = w + pl.x2r + (size_t)b0 * px * 64 * es;
    auto AMb = [&](int slot) { unsigned* p = AM(slot); return p ? p + b0 : p; };
#if HLA_ABLATIONS
    {      // timing-only ablations (tools/probes/conv02_probe.py): HLA_ABL_C02 bit 0 = no relu(conv0) copy, bit 1 = no pool argmax
      static const int abl = [] { const char* e = getenv("HLA_ABL_C02"); return e ? atoi(e) : 0; }();
      if (abl & 1) a.a0_out = nullptr;
      if (abl & 2) a.idx_out = nullptr;
    }
#endif
    a.wtail = wtail; a.amax_out = AMb(AM_X3); a.amax_a2_out = level4 ? AMb(AM_X2) : nullptr;
    a.amax_a0_out = (flags & HLA_VGG_SAVE_FOR_BACKWARD) ? AMb(AM_A0) : nullptr;
    // (first_row8, see below: x3 is needed from row 4f-16 on = conv2 row 8f-32)
    const int f0 = (level4 || (flags & HLA_VGG_SAVE_FOR_BACKWARD)) ? 0 : first_row8;
    a.row_begin = f0 ? 8 * f0 - 32 : 0;
    a.B = nb; a.H = H; a.W = W; a.tiles_x = (W + 31) / 32; a.tiles_y = (H - a.row_begin + 7) / 8;
    const double P = (double)nb * (H - a.row_begin) * W;
    constexpr int lds_bytes = conv02_lds_bytes<T>();
    static HlaPerDeviceOnce attr_once;
    HLA_CHECK_HIP(attr_once.run([] {
      return hipFuncSetAttribute((const void*)conv02_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    }));
#if HLA_CONV_STAMPS
    a.stamps = nullptr;      // (tooling build: want = -10 - k arms the k-th conv02 launch since the call)
    if (g_hla_stamp.buf && g_hla_stamp.want <= -10 && g_hla_stamp.want++ == -10) {
      a.stamps = g_hla_stamp.buf; g_hla_stamp.want = -1000;
      g_hla_stamp.grid_x = a.tiles_x * a.tiles_y * nb; g_hla_stamp.grid_y = 1;
    }
#endif
    hla_prof_begin(K_CONV02, 2.0 * 9 * (3 + 64) * 64 * P, P * (3 * 4 + 16 * sizeof(T)), st);
    hipLaunchKernelGGL((conv02_kernel<T, 2>), dim3(a.tiles_x * a.tiles_y * nb), dim3(256), lds_bytes, st, a);
    hla_prof_end(st);
    return HLA_OK;
  };
  const bool train = flags & HLA_VGG_SAVE_FOR_BACKWARD;
  int np_used[4] = {pl.np[0], pl.np[1], pl.np[2], pl.np[3]};
  bool launch_ok = true;
  auto conv = [&](int l, const void* s1, int C1, int H_, int W_h, void* act, int relu, bool pool, const void* s2 = nullptr,
                  int C2 = 0, int up1 = 0, void* raw = nullptr, double* ss = nullptr, unsigned char* idx = nullptr,
                  int row_begin = 0, int norm_level = -1, int b0 = 0, int nb = -1) {
    ConvArgs a{};
    if (nb < 0) nb = B;
    // sample range [b0, b0 + nb): every per-sample pointer moves by b0 samples (the chunked high-resolution chain below)
    const size_t es = sizeof(T), Cout = kLayers[l].cout;
    const size_t in1 = (size_t)(H_ >> up1) * (W_h >> up1) * C1 * es, in2 = (size_t)H_ * W_h * C2 * es;
    const size_t opx = pool ? (size_t)(H_ / 2) * (W_h / 2) : (size_t)H_ * W_h;
    a.raw16 = (raw && (flags & HLA_VGG_FEAT16)) ? 1 : 0;
    a.row_begin = row_begin < 0 ? 0 : row_begin;
    const size_t npart = (size_t)((W_h + 31) / 32) * ((H_ - a.row_begin + 7) / 8) * (Cout >= 128 ? Cout / 128 : 1);
    a.idx_out = (train && idx) ? idx + (size_t)b0 * opx * Cout : nullptr;
    a.src1 = (const char*)s1 + (size_t)b0 * in1; a.src2 = s2 ? (const char*)s2 + (size_t)b0 * in2 : nullptr;
    a.C1 = C1; a.C2 = C2; a.up1 = up1; a.wpk = W_(l);
    a.bias = kLayers[l].has_bias ? prm->b[l] : nullptr;
    a.out_act = act ? (char*)act + (size_t)b0 * opx * Cout * es : nullptr;
    a.out_raw = raw ? (float*)((char*)raw + (size_t)b0 * opx * Cout * (a.raw16 ? 2 : 4)) : nullptr;
    a.sumsq = ss ? ss + (size_t)b0 * npart : nullptr;
    a.B = nb; a.H = H_; a.W = W_h; a.Cout = (int)Cout;
    a.relu_act = relu;
    if (SPLIT) {      // which per-sample maxima the layer reads (its one or two sources) and writes (its activation output)
      static const signed char kAm[13][3] = {{-1, -1, -1}, {-1, -1, -1}, {AM_X3, -1, AM_A5}, {AM_A5, -1, AM_X8}, {AM_X8, -1, AM_A10},
                                             {AM_A10, -1, AM_A12}, {AM_A12, -1, AM_X15}, {AM_X15, AM_X8, AM_D1A}, {AM_D1A, -1, AM_X18},
                                             {AM_X18, AM_X3, AM_D2A}, {AM_D2A, -1, AM_X21}, {AM_X21, AM_X2, AM_D3A}, {AM_D3A, -1, AM_X24}};
      auto AMb = [&](int slot) { unsigned* p = AM(slot); return p ? p + b0 : p; };
      a.amax1 = AMb(kAm[l][0]); a.amax2 = AMb(kAm[l][1]); a.amax_out = AMb(kAm[l][2]); a.wscale = wtail + l;
    }
    if (!launch_conv<T>(st, a, pool)) launch_ok = false;
    if (norm_level >= 0)      // sum-of-squares partials actually written by this launch: one per (tile, 128-cout block)
      np_used[norm_level] = ((W_h + 31) / 32) * ((H_ - a.row_begin + 7) / 8) * (a.Cout >= 128 ? a.Cout / 128 : 1);
  };
  // first_row8 = f > 0: the caller reads the returned maps only from rows f (x15), 2f (x18), 4f (x21) on, so every layer only
  // has to produce the rows those depend on -- a 3x3 conv needs one more input row, a 2x upsample halves, a 2x2 pool doubles:
  //   x21 <- dec2.3 [4f..] <- dec2.1 [4f-1..] <- {up(x18) [2f-1..], x3 [4f-2..]};  x18 <- dec1.3 [2f-1..] <- dec1.1 [2f-2..] <-
  //   {up(x15) [f-2..], x8 [2f-3..]};  x15 [f-2..] <- pool(conv14 [2f-4..]) <- conv12 [2f-5..] <- conv10 [2f-6..] <- x8 [2f-7..]
  //   <- pool(conv7 [4f-14..]) <- conv5 [4f-15..] <- x3 [4f-16..]  (<- image row 8f-34: `dead_ground_rows`).
  // Each launch starts exactly at its first needed row, so the one halo row above it is the first row its producer wrote.
  // The 3x3 confidence heads read one row above the first confidence row that is used, so with them the decoder starts one
  // row earlier (the encoder's needs do not change: x15 [f-2..] and x8 [2f-4..] are covered).
  const int f = (level4 || train) ? 0 : first_row8;
  const int wc = ((flags & HLA_VGG_WANT_CONF) && conf) ? 1 : 0;
  const int r_c5 = f ? 4 * f - 15 : 0, r_c7 = f ? 4 * f - 14 : 0, r_c10 = f ? 2 * f - 6 : 0, r_c12 = f ? 2 * f - 5 : 0,
            r_c14 = f ? 2 * f - 4 : 0, r_d11 = f ? 2 * f - 2 - wc : 0, r_d13 = f ? 2 * f - 1 - wc : 0,
            r_d21 = f ? 4 * f - 1 - wc : 0, r_d23 = f ? 4 * f - wc : 0;
  // encoder (VGG.py:129-141).  ReLU commutes with max-pool, so pooled maps are stored post-ReLU.
  // The full- and half-resolution chain conv0+2 -> conv5 -> conv7+pool runs in CHUNKS of `chunk` samples, so that a chunk's x3 and a5
  // (8.4 + 16.8 MB per sample in 16-bit storage) are still in the 256 MB Infinity Cache when the next layer reads them; the H/4
  // layers keep the whole batch per launch.  Samples are independent and every kernel's per-sample arithmetic does not depend on
  // the launch's batch, so the result is bit-identical to the unchunked walk.
  const int chunk = vgg_chunk(B, H, W, sizeof(T));
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    if (const int rc = conv02(b0, nb)) return rc;
    conv(2, w + pl.x3, 64, H / 2, W / 2, w + pl.a5, 1, false, nullptr, 0, 0, nullptr, nullptr, nullptr, r_c5, -1, b0, nb);      // conv5
    conv(3, w + pl.a5, 128, H / 2, W / 2, w + pl.x8, 1, true, nullptr, 0, 0, nullptr, nullptr,
         (unsigned char*)(w + pl.idx8), r_c7, -1, b0, nb);                              // conv7 + pool -> relu(x8)
  }
  conv(4, w + pl.x8, 128, H / 4, W / 4, w + pl.a10, 1, false, nullptr, 0, 0, nullptr, nullptr, nullptr, r_c10);   // conv10
  conv(5, w + pl.a10, 256, H / 4, W / 4, w + pl.a12, 1, false, nullptr, 0, 0, nullptr, nullptr, nullptr, r_c12);  // conv12
  conv(6, w + pl.a12, 256, H / 4, W / 4, w + pl.x15r, 1, true, nullptr, 0, 0, feat[0],
       (double*)(w + pl.ss[0]), (unsigned char*)(w + pl.idx15), r_c14, 0);            // conv14 + pool -> x15
  // decoder (VGG.py:144-151): conv(relu(cat(up(a), skip))) with both inputs stored post-ReLU
  conv(7, w + pl.x15r, 256, H / 4, W / 4, w + pl.d1a, 1, false, w + pl.x8, 128, 1, nullptr, nullptr, nullptr, r_d11);   // dec1.1
  conv(8, w + pl.d1a, 128, H / 4, W / 4, w + pl.x18r, 1, false, nullptr, 0, 0, feat[1],
       (double*)(w + pl.ss[1]), nullptr, r_d13, 1);                                   // dec1.3 -> x18
  // relu(x21) feeds conv_dec3 (level 4), the conf2 head and the training backward only: without them the 16-bit feature path
  // stores just the raw map
  const bool x21r_dead = !level4 && !wc && !train && (flags & HLA_VGG_FEAT16) && sizeof(T) == 2;
  for (int b0 = 0; b0 < B; b0 += chunk) {      // (the half-resolution decoder pair in the same chunks: d2a is read once, right away)
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    conv(9, w + pl.x18r, 128, H / 2, W / 2, w + pl.d2a, 1, false, w + pl.x3, 64, 1, nullptr, nullptr, nullptr, r_d21, -1, b0, nb);    // dec2.1
    conv(10, w + pl.d2a, 64, H / 2, W / 2, x21r_dead ? (char*)nullptr : w + pl.x21r, 1, false, nullptr, 0, 0, feat[2],
         (double*)(w + pl.ss[2]), nullptr, r_d23, 2, b0, nb);                           // dec2.3 -> x21
  }
  if (level4) {      // VGG.py:153-155: conv_dec3 on cat(up(x21), x2), zero-padded to 64 channels (vgg_layers.h)
    conv(11, w + pl.x21r, 64, H, W, w + pl.d3a, 1, false, w + pl.x2r, 64, 1);          // dec3.1
    conv(12, w + pl.d3a, 64, H, W, w + pl.x24r, 1, false, nullptr, 0, 0, feat[3],
         (double*)(w + pl.ss[3]), nullptr, 0, 3);                                      // dec3.3 -> x24 (16 real channels)
  }
  // confidence heads on the ReLU'd maps
  if ((flags & HLA_VGG_WANT_CONF) && conf) {
    using CT = std::conditional_t<SPLIT, float, T>;      // split mode stores fp32 activations: the heads run in plain fp32
    const CT* acts[4] = {(const CT*)(w + pl.x15r), (const CT*)(w + pl.x18r), (const CT*)(w + pl.x21r), (const CT*)(w + pl.x24r)};
    const int Cs[4] = {256, 128, 64, 64}, hs[4] = {H / 8, H / 4, H / 2, H}, wsz[4] = {W / 8, W / 4, W / 2, W};
    for (int l = 0; l < NL; ++l) {
      if (!conf[l]) continue;
      const size_t npix = (size_t)B * hs[l] * wsz[l];
      const int grid = B * ((hs[l] + CONF_TH - 1) / CONF_TH) * ((wsz[l] + CONF_TW - 1) / CONF_TW);
      hla_prof_begin(K_CONF, 2.0 * 9 * Cs[l] * (double)npix, (double)npix * (Cs[l] * sizeof(CT) + 4), st);
      if (Cs[l] == 256) hipLaunchKernelGGL((conf_kernel<CT, 256>), dim3(grid), dim3(256), 0, st, acts[l], prm->w[13 + l], conf[l], B, hs[l], wsz[l]);
      else if (Cs[l] == 128) hipLaunchKernelGGL((conf_kernel<CT, 128>), dim3(grid), dim3(256), 0, st, acts[l], prm->w[13 + l], conf[l], B, hs[l], wsz[l]);
      else hipLaunchKernelGGL((conf_kernel<CT, 64>), dim3(grid), dim3(256), 0, st, acts[l], prm->w[13 + l], conf[l], B, hs[l], wsz[l]);
      hla_prof_end(st);
    }
  }
  // L2 normalisation: 1/||x|| per sample (always), in-place scaling unless the caller folds it downstream
  {
    const size_t per[4] = {(size_t)(H / 8) * (W / 8) * 256, (size_t)(H / 4) * (W / 4) * 128, (size_t)(H / 2) * (W / 2) * 64,
                           (size_t)H * W * 64};
    double* inv = inv_norm ? inv_norm : (double*)(w + pl.inv);
    {
      InvNormArgs ia{};
      double bytes = 0;
      for (int l = 0; l < NL; ++l) { ia.ss[l] = (const double*)(w + pl.ss[l]); ia.np[l] = np_used[l]; bytes += (double)B * pl.np[l] * 8; }
      ia.inv = inv; ia.B = B;
      hla_prof_begin(K_L2NORM, 0, bytes, st);
      hipLaunchKernelGGL(inv_norm_multi_kernel, dim3(B, NL), dim3(256), 0, st, ia);
      hla_prof_end(st);
    }
    for (int l = 0; l < NL; ++l) {
      if (flags & HLA_VGG_DEFER_NORM) continue;
      int bps = (int)(per[l] / 4 / 256 / 4);
      bps = bps < 1 ? 1 : (bps > 64 ? 64 : bps);
      hla_prof_begin(K_L2NORM, 0, (double)B * per[l] * 8, st);
      hipLaunchKernelGGL(scale_kernel, dim3(B * bps), dim3(256), 0, st, (float*)feat[l], inv + (size_t)l * B, per[l], bps);
      hla_prof_end(st);
    }
  }
  HLA_CHECK_HIP(hipGetLastError());
  return launch_ok ? HLA_OK : HLA_ERR_ARG;
}

#if HLA_TU_DTYPE >= 0
template void vgg_pack_all<TuT>(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st);
template int vgg_forward_t<TuT>(const float* x, size_t x_plane, const hla_vgg_params* prm, const char* packed, int dtype, void* const feat[4],
                              float* const conf[4], double* inv_norm, char* ws, const VggPlan& pl, int B, int H, int W,
                              int flags, int first_row8, hipStream_t st);
#else
#define HLA_EXTERN_T(T) \
  extern template void vgg_pack_all<T>(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st); \
  extern template int vgg_forward_t<T>(const float* x, size_t x_plane, const hla_vgg_params* prm, const char* packed, int dtype, void* const feat[4],                               float* const conf[4], double* inv_norm, char* ws, const VggPlan& pl, int B, int H, int W,                               int flags, int first_row8, hipStream_t st);
HLA_EXTERN_T(float) HLA_EXTERN_T(bf16) HLA_EXTERN_T(f16) HLA_EXTERN_T(split32)

extern "C" size_t hla_vgg_packed_weight_bytes(int dtype) {
  return packed_offset(kAllLayers, dtype) + (dtype == HLA_F16X3 ? kPackTailBytes : 0);
}

extern "C" int hla_vgg_pack_weights(const hla_vgg_params* params, void* packed, int dtype, hla_stream_t stream) {
  HLA_REQUIRE(params && packed, "hla_vgg_pack_weights: null argument");
  HLA_REQUIRE(hla_dtype_ok(dtype), "hla_vgg_pack_weights: bad dtype %d", dtype);
  HLA_REQUIRE(params->w[0] && params->b[0], "hla_vgg_pack_weights: conv0's weight and bias are required (the bias is packed with it)");
  if (dtype == HLA_BF16) vgg_pack_all<bf16>(params, (char*)packed, dtype, (hipStream_t)stream);
  else if (dtype == HLA_F16) vgg_pack_all<f16>(params, (char*)packed, dtype, (hipStream_t)stream);
  else if (dtype == HLA_F16X3) vgg_pack_all<split32>(params, (char*)packed, dtype, (hipStream_t)stream);
  else vgg_pack_all<float>(params, (char*)packed, dtype, (hipStream_t)stream);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}

extern "C" size_t hla_vgg_workspace_bytes(int B, int H, int W, int level, int dtype) {
  VggPlan p;
  vgg_plan(B, H, W, dtype, /*train=*/true, &p, level == 4);   // sized for training so one buffer serves both modes
  return p.total;
}

extern "C" int hla_vgg_forward(const float* x, size_t x_plane, const hla_vgg_params* params, const void* packed_weights,
                               void* const feat[4], float* const conf[4], double* inv_norm, void* workspace,
                               size_t workspace_bytes, int B, int H, int W, int level, int dtype, int flags,
                               int first_row8, hla_stream_t stream) {
  HLA_REQUIRE(x && params && packed_weights && feat && workspace, "hla_vgg_forward: null argument");
  HLA_REQUIRE(hla_dtype_ok(dtype), "hla_vgg_forward: dtype must be HLA_F32, HLA_BF16, HLA_F16 or HLA_F16X3 (got %d)", dtype);
  HLA_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, "hla_vgg_forward: H and W must be multiples of 8");
  HLA_REQUIRE(x_plane == 0 || x_plane >= (size_t)H * W, "hla_vgg_forward: x_plane (%zu) must be 0 or >= H*W", x_plane);
  // the conv kernels address a sample's activation map with signed 32-bit byte offsets (largest map: H x W x 64 fp32)
  HLA_REQUIRE((size_t)H * W * 64 * 4 < ((size_t)1 << 31), "hla_vgg_forward: image too large (H*W must be below 2^23 pixels)");
  HLA_REQUIRE(level == 3 || level == 4, "hla_vgg_forward: level must be 3 (x15,x18,x21) or 4 (+x24), got %d", level);
  HLA_REQUIRE(feat[0] && feat[1] && feat[2], "hla_vgg_forward: feat[0..2] are required");
  HLA_REQUIRE(level == 3 || (feat[3] && params->w[11] && params->w[12]),
              "hla_vgg_forward: level 4 needs feat[3] ([B,H,W,64], 16 real channels) and the zero-padded conv_dec3 weights in w[11], w[12]");
  HLA_REQUIRE(level == 3 || !(flags & HLA_VGG_WANT_CONF) || !conf || !conf[3] || params->w[16], "hla_vgg_forward: conf[3] needs w[16]");
  HLA_REQUIRE(!(flags & HLA_VGG_DEFER_NORM) || inv_norm, "hla_vgg_forward: HLA_VGG_DEFER_NORM needs inv_norm");
  HLA_REQUIRE(!(flags & HLA_VGG_FEAT16) || ((dtype == HLA_BF16 || dtype == HLA_F16) && (flags & HLA_VGG_DEFER_NORM) &&
                                            !(flags & HLA_VGG_SAVE_FOR_BACKWARD) && level == 3),
              "hla_vgg_forward: HLA_VGG_FEAT16 needs dtype HLA_BF16 / HLA_F16, HLA_VGG_DEFER_NORM, level 3 and no HLA_VGG_SAVE_FOR_BACKWARD");
  HLA_REQUIRE(first_row8 == 0 || (first_row8 >= 4 && first_row8 < H / 8), "hla_vgg_forward: first_row8 must be 0 or in [4, H/8)");
  VggPlan pl;
  vgg_plan(B, H, W, dtype, (flags & HLA_VGG_SAVE_FOR_BACKWARD) != 0, &pl, level == 4);
  if (workspace_bytes < pl.total) {
    hla_set_error("hla_vgg_forward: workspace %zu < %zu", workspace_bytes, pl.total);
    return HLA_ERR_WORKSPACE;
  }
  if (dtype == HLA_BF16)
    return vgg_forward_t<bf16>(x, x_plane, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                               B, H, W, flags, first_row8, (hipStream_t)stream);
  if (dtype == HLA_F16)
    return vgg_forward_t<f16>(x, x_plane, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                              B, H, W, flags, first_row8, (hipStream_t)stream);
  if (dtype == HLA_F16X3)
    return vgg_forward_t<split32>(x, x_plane, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                                  B, H, W, flags, first_row8, (hipStream_t)stream);
  return vgg_forward_t<float>(x, x_plane, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                              B, H, W, flags, first_row8, (hipStream_t)stream);
}
#endif
