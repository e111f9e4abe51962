// VGG16-U-Net forward: host orchestration (kernels live in conv_kernels.h).  VGG.py:13-203, 511-514.
#include "conv_kernels.h"
#include "vgg_layers.h"

template <typename T>
void vgg_pack_all(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st) {
  for (int l = 0; l < kAllLayers; ++l) {
    if (l >= kPackedLayers && !prm->w[l]) continue;        // conv_dec3.* only when the caller supplies its padded weights
    const size_t n = l == 0 ? (size_t)2 * 32 * 32 : (size_t)kLayers[l].cin * kLayers[l].cout * 9;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hla_prof_begin(K_PACK, 0, (double)n * (4 + sizeof(T)), st);
    hipLaunchKernelGGL((pack_weights_kernel<T>), dim3(grid), dim3(256), 0, st, prm->w[l],
                       (T*)(packed + packed_offset(l, dtype)), kLayers[l].cout, kLayers[l].cin, l == 0 ? 1 : 0);
    hla_prof_end(st);
  }
}

template <typename T>
int vgg_forward_t(const float* x, const hla_vgg_params* prm, const char* packed, int dtype, float* const feat[4],
                         float* const conf[4], double* inv_norm, char* ws, const VggPlan& pl, int B, int H, int W,
                         int flags, hipStream_t st) {
  const bool level4 = pl.x2r != 0;
  const int NL = level4 ? 4 : 3;
  auto W_ = [&](int l) { return (const uint4*)(packed + packed_offset(l, dtype)); };
  char* w = ws;
  // conv0 + conv2 + pool fused (VGG.py:123-128): relu(x3)
  {
    Conv02Args a{};
    a.x = x; a.w0 = W_(0); a.b0 = prm->b[0]; a.w2 = W_(1); a.b2 = prm->b[1]; a.out_act = w + pl.x3;
    if (flags & HLA_VGG_SAVE_FOR_BACKWARD) { a.a0_out = w + pl.a0; a.idx_out = (unsigned char*)(w + pl.idx3); }
    if (level4) a.a2_out = w + pl.x2r;
    a.B = B; a.H = H; a.W = W; a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 7) / 8;
    const double P = (double)B * H * W;
    constexpr int lds_bytes = conv02_lds_bytes<T>();
    static bool attr_set = false;
    if (!attr_set) {
      HLA_CHECK_HIP(hipFuncSetAttribute((const void*)conv02_kernel<T, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
      attr_set = true;
    }
    hla_prof_begin(K_CONV02, 2.0 * 9 * (3 + 64) * 64 * P, P * (3 * 4 + 16 * sizeof(T)), st);
    hipLaunchKernelGGL((conv02_kernel<T, 2, true>), dim3(a.tiles_x * a.tiles_y * B), dim3(256), lds_bytes, st, a);
    hla_prof_end(st);
  }
  const bool train = flags & HLA_VGG_SAVE_FOR_BACKWARD;
  auto conv = [&](int l, const void* s1, int C1, int H_, int W_h, void* act, int relu, bool pool, const void* s2 = nullptr,
                  int C2 = 0, int up1 = 0, float* raw = nullptr, double* ss = nullptr, unsigned char* idx = nullptr) {
    ConvArgs a{};
    a.idx_out = train ? idx : nullptr;
    a.src1 = s1; a.src2 = s2; a.C1 = C1; a.C2 = C2; a.up1 = up1; a.wpk = W_(l);
    a.bias = kLayers[l].has_bias ? prm->b[l] : nullptr;
    a.out_act = act; a.out_raw = raw; a.sumsq = ss; a.B = B; a.H = H_; a.W = W_h; a.Cout = kLayers[l].cout;
    a.relu_act = relu;
    launch_conv<T>(st, a, pool);
  };
  // encoder (VGG.py:129-141).  ReLU commutes with max-pool, so pooled maps are stored post-ReLU.
  conv(2, w + pl.x3, 64, H / 2, W / 2, w + pl.a5, 1, false);                          // conv5
  conv(3, w + pl.a5, 128, H / 2, W / 2, w + pl.x8, 1, true, nullptr, 0, 0, nullptr, nullptr,
       (unsigned char*)(w + pl.idx8));                                                // conv7 + pool -> relu(x8)
  conv(4, w + pl.x8, 128, H / 4, W / 4, w + pl.a10, 1, false);                        // conv10
  conv(5, w + pl.a10, 256, H / 4, W / 4, w + pl.a12, 1, false);                       // conv12
  conv(6, w + pl.a12, 256, H / 4, W / 4, w + pl.x15r, 1, true, nullptr, 0, 0, feat[0],
       (double*)(w + pl.ss[0]), (unsigned char*)(w + pl.idx15));                      // conv14 + pool -> x15
  // decoder (VGG.py:144-151): conv(relu(cat(up(a), skip))) with both inputs stored post-ReLU
  conv(7, w + pl.x15r, 256, H / 4, W / 4, w + pl.d1a, 1, false, w + pl.x8, 128, 1);   // dec1.1
  conv(8, w + pl.d1a, 128, H / 4, W / 4, w + pl.x18r, 1, false, nullptr, 0, 0, feat[1],
       (double*)(w + pl.ss[1]));                                                      // dec1.3 -> x18
  conv(9, w + pl.x18r, 128, H / 2, W / 2, w + pl.d2a, 1, false, w + pl.x3, 64, 1);    // dec2.1
  conv(10, w + pl.d2a, 64, H / 2, W / 2, w + pl.x21r, 1, false, nullptr, 0, 0, feat[2],
       (double*)(w + pl.ss[2]));                                                      // dec2.3 -> x21
  if (level4) {      // VGG.py:153-155: conv_dec3 on cat(up(x21), x2), zero-padded to 64 channels (vgg_layers.h)
    conv(11, w + pl.x21r, 64, H, W, w + pl.d3a, 1, false, w + pl.x2r, 64, 1);          // dec3.1
    conv(12, w + pl.d3a, 64, H, W, w + pl.x24r, 1, false, nullptr, 0, 0, feat[3],
         (double*)(w + pl.ss[3]));                                                     // dec3.3 -> x24 (16 real channels)
  }
  // confidence heads on the ReLU'd maps
  if ((flags & HLA_VGG_WANT_CONF) && conf) {
    const T* acts[4] = {(const T*)(w + pl.x15r), (const T*)(w + pl.x18r), (const T*)(w + pl.x21r), (const T*)(w + pl.x24r)};
    const int Cs[4] = {256, 128, 64, 64}, hs[4] = {H / 8, H / 4, H / 2, H}, wsz[4] = {W / 8, W / 4, W / 2, W};
    for (int l = 0; l < NL; ++l) {
      if (!conf[l]) continue;
      constexpr int EPL = 16 / sizeof(T);
      const int ppb = 256 / (Cs[l] / EPL);
      const size_t npix = (size_t)B * hs[l] * wsz[l];
      const int grid = (int)((npix + ppb - 1) / ppb < 4096 ? (npix + ppb - 1) / ppb : 4096);
      hla_prof_begin(K_CONF, 2.0 * 9 * Cs[l] * (double)npix, (double)npix * (Cs[l] * sizeof(T) + 4), st);
      hipLaunchKernelGGL((conf_kernel<T>), dim3(grid), dim3(256), 9 * Cs[l] * sizeof(float), st, acts[l],
                         prm->w[13 + l], conf[l], B, hs[l], wsz[l], Cs[l]);
      hla_prof_end(st);
    }
  }
  // L2 normalisation: 1/||x|| per sample (always), in-place scaling unless the caller folds it downstream
  {
    const size_t per[4] = {(size_t)(H / 8) * (W / 8) * 256, (size_t)(H / 4) * (W / 4) * 128, (size_t)(H / 2) * (W / 2) * 64,
                           (size_t)H * W * 64};
    double* inv = inv_norm ? inv_norm : (double*)(w + pl.inv);
    for (int l = 0; l < NL; ++l) {
      hla_prof_begin(K_L2NORM, 0, (double)B * pl.np[l] * 8, st);
      hipLaunchKernelGGL(inv_norm_kernel, dim3(B), dim3(256), 0, st, (const double*)(w + pl.ss[l]), pl.np[l], inv + (size_t)l * B);
      hla_prof_end(st);
      if (flags & HLA_VGG_DEFER_NORM) continue;
      int bps = (int)(per[l] / 4 / 256 / 4);
      bps = bps < 1 ? 1 : (bps > 64 ? 64 : bps);
      hla_prof_begin(K_L2NORM, 0, (double)B * per[l] * 8, st);
      hipLaunchKernelGGL(scale_kernel, dim3(B * bps), dim3(256), 0, st, feat[l], inv + (size_t)l * B, per[l], bps);
      hla_prof_end(st);
    }
  }
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}

#if HLA_TU_DTYPE >= 0
template void vgg_pack_all<TuT>(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st);
template int vgg_forward_t<TuT>(const float* x, const hla_vgg_params* prm, const char* packed, int dtype, float* const feat[4],
                              float* const conf[4], double* inv_norm, char* ws, const VggPlan& pl, int B, int H, int W,
                              int flags, hipStream_t st);
#else
#define HLA_EXTERN_T(T) \
  extern template void vgg_pack_all<T>(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st); \
  extern template int vgg_forward_t<T>(const float* x, const hla_vgg_params* prm, const char* packed, int dtype, float* const feat[4],                               float* const conf[4], double* inv_norm, char* ws, const VggPlan& pl, int B, int H, int W,                               int flags, hipStream_t st);
HLA_EXTERN_T(float) HLA_EXTERN_T(bf16) HLA_EXTERN_T(f16)

extern "C" size_t hla_vgg_packed_weight_bytes(int dtype) { return packed_offset(kAllLayers, dtype); }

extern "C" int hla_vgg_pack_weights(const hla_vgg_params* params, void* packed, int dtype, hla_stream_t stream) {
  HLA_REQUIRE(params && packed, "hla_vgg_pack_weights: null argument");
  HLA_REQUIRE(hla_dtype_ok(dtype), "hla_vgg_pack_weights: bad dtype %d", dtype);
  if (dtype == HLA_BF16) vgg_pack_all<bf16>(params, (char*)packed, dtype, (hipStream_t)stream);
  else if (dtype == HLA_F16) vgg_pack_all<f16>(params, (char*)packed, dtype, (hipStream_t)stream);
  else vgg_pack_all<float>(params, (char*)packed, dtype, (hipStream_t)stream);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}

extern "C" size_t hla_vgg_workspace_bytes(int B, int H, int W, int level, int dtype) {
  VggPlan p;
  vgg_plan(B, H, W, dtype, /*train=*/true, &p, level == 4);   // sized for training so one buffer serves both modes
  return p.total;
}

extern "C" int hla_vgg_forward(const float* x, const hla_vgg_params* params, const void* packed_weights,
                               float* const feat[4], float* const conf[4], double* inv_norm, void* workspace,
                               size_t workspace_bytes, int B, int H, int W, int level, int dtype, int flags,
                               hla_stream_t stream) {
  HLA_REQUIRE(x && params && packed_weights && feat && workspace, "hla_vgg_forward: null argument");
  HLA_REQUIRE(hla_dtype_ok(dtype), "hla_vgg_forward: dtype must be HLA_F32, HLA_BF16 or HLA_F16 (got %d)", dtype);
  HLA_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, "hla_vgg_forward: H and W must be multiples of 8");
  HLA_REQUIRE(level == 3 || level == 4, "hla_vgg_forward: level must be 3 (x15,x18,x21) or 4 (+x24), got %d", level);
  HLA_REQUIRE(feat[0] && feat[1] && feat[2], "hla_vgg_forward: feat[0..2] are required");
  HLA_REQUIRE(level == 3 || (feat[3] && params->w[11] && params->w[12]),
              "hla_vgg_forward: level 4 needs feat[3] ([B,H,W,64], 16 real channels) and the zero-padded conv_dec3 weights in w[11], w[12]");
  HLA_REQUIRE(level == 3 || !(flags & HLA_VGG_WANT_CONF) || !conf || !conf[3] || params->w[16], "hla_vgg_forward: conf[3] needs w[16]");
  HLA_REQUIRE(!(flags & HLA_VGG_DEFER_NORM) || inv_norm, "hla_vgg_forward: HLA_VGG_DEFER_NORM needs inv_norm");
  VggPlan pl;
  vgg_plan(B, H, W, dtype, (flags & HLA_VGG_SAVE_FOR_BACKWARD) != 0, &pl, level == 4);
  if (workspace_bytes < pl.total) {
    hla_set_error("hla_vgg_forward: workspace %zu < %zu", workspace_bytes, pl.total);
    return HLA_ERR_WORKSPACE;
  }
  if (dtype == HLA_BF16)
    return vgg_forward_t<bf16>(x, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                               B, H, W, flags, (hipStream_t)stream);
  if (dtype == HLA_F16)
    return vgg_forward_t<f16>(x, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                              B, H, W, flags, (hipStream_t)stream);
  return vgg_forward_t<float>(x, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                              B, H, W, flags, (hipStream_t)stream);
}
#endif
