/* hla.h -- C ABI of libhla.so: the MI355X (gfx950) implementation of the
 * HighlyAccurate per-pair localisation hot path.
 *
 * This is the drop-in boundary.  The reference is pure Python on PyTorch, so the
 * "FFI" a maintainer binds is ctypes; every entry point below
 * names the reference function it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless marked [host]
 *   - the caller allocates every buffer (PyTorch's caching allocator in practice);
 *     the library never allocates or frees device memory and keeps no global state
 *     except a thread-local last-error string, the opt-in timing log (hla_prof_*) and
 *     per-device "kernel attribute set / refused" bits (dynamic-LDS opt-in of the large-LDS kernels; a refused request
 *     makes hla_vgg_backward fall back to its smaller-footprint weight-gradient kernels)
 *   - kernels are launched on the CURRENT device: the caller makes the device that owns the
 *     buffers and the stream current before the call (highlyaccurate_amd/_lib.py on_device)
 *   - all work is enqueued on the given hipStream_t (passed as void*); no implicit
 *     synchronisation, no host<->device copies except the small [host] structs
 *     passed by value
 *   - return 0 on success, negative hla_status on error; hla_last_error() explains
 *   - activations are NHWC ("channels last"): element (b,y,x,c) at ((b*H+y)*W+x)*C+c
 */
#ifndef HLA_H
#define HLA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hla_stream_t; /* hipStream_t */

typedef enum hla_status {
  HLA_OK = 0,
  HLA_ERR_ARG = -1,       /* bad argument / unsupported shape */
  HLA_ERR_HIP = -2,       /* a HIP runtime call failed */
  HLA_ERR_WORKSPACE = -3  /* workspace too small */
} hla_status;

typedef enum hla_dtype {
  HLA_F32 = 0,  /* exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the parity mode      */
  HLA_BF16 = 1, /* bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 accumulate: perf mode */
  HLA_F16 = 2,  /* fp16 MFMA (v_mfma_f32_32x32x16_f16), fp32 accumulate: same speed, 3 more mantissa bits,
                   narrower range -- fine for inference on [0,1] images, not recommended for the backward pass */
  HLA_F16X3 = 3 /* split-fp16: fp32 activations and weights in memory; every operand enters the matrix cores as
                   hi + lo = fp16(s x) + fp16(s x - hi) (power-of-two scale s per tensor and sample), one product =
                   three fp16 MFMAs (hi hi + hi lo + lo hi), fp32 accumulate.  fp32-class results (meets the fp32
                   parity gates) at 1/3 of the fp16 MFMA rate instead of the 1/16 of HLA_F32: the matched-accuracy
                   throughput mode, for training too: hla_vgg_backward with this dtype runs split-fp16 data- and
                   weight-gradient kernels on the fp32 activations the split forward saved (gradient maps stay fp32; a
                   per-sample maximum of every gradient map is recorded by the kernel that writes it, as the forward does
                   for activations).  Workspaces have the HLA_F32 layout; packed weights the HLA_F32 size + 256 B of scales. */
} hla_dtype;

const char* hla_last_error(void);

/* Bumped whenever a struct or signature in this file changes.  A binding must refuse a library whose
 * hla_abi_version() differs from the HLA_ABI_VERSION it was written against, and should compare hla_sizeof_struct()
 * with its own struct sizes (ctypes structs are positional: a mismatch corrupts silently).  highlyaccurate_amd/_lib.py
 * does both at load time, and rebuilds or refuses a binary whose hla_source_hash() is not the hash of the sources
 * next to it (the library is git-ignored but shipped prebuilt). */
#define HLA_ABI_VERSION 21
int hla_abi_version(void);
const char* hla_source_hash(void); /* sha256 (hex) of the csrc sources, this header and the compiler flags at build time */
typedef enum hla_struct_id {
  HLA_STRUCT_VGG_PARAMS = 0, HLA_STRUCT_VGG_GRADS = 1, HLA_STRUCT_S2G_LEVEL = 2, HLA_STRUCT_S2G_CONFIG = 3,
  HLA_STRUCT_S2G_LEVEL_GRAD = 4, HLA_STRUCT_PROF_RECORD = 5, HLA_STRUCT_POSE_LOSS_ARGS = 6, HLA_STRUCT_FILL_REGION = 7
} hla_struct_id;
size_t hla_sizeof_struct(int id);  /* sizeof of the struct with that hla_struct_id, 0 for an unknown id */

/* ------------------------------------------------------------------------- *
 * VGGUnet.forward  (VGG.py:121-203; L2_norm VGG.py:511-514)
 * ------------------------------------------------------------------------- */

/* Device pointers to the 17 weight and 7 bias tensors exactly as PyTorch holds
 * them (OIHW fp32, contiguous), in state-dict order:
 *   w[0..6]  conv0,conv2,conv5,conv7,conv10,conv12,conv14   (b[0..6] their biases)
 *   w[7..12] conv_dec1.1, conv_dec1.3, conv_dec2.1, conv_dec2.3, conv_dec3.1, conv_dec3.3
 *   w[13..16] conf0.1, conf1.1, conf2.1, conf3.1                                    */
typedef struct hla_vgg_params {
  const float* w[17];
  const float* b[7];
} hla_vgg_params;

enum {
  HLA_VGG_WANT_CONF = 1,   /* compute the confidence maps (VGG.py:160-163)                              */
  HLA_VGG_DEFER_NORM = 2,  /* leave feat[] un-normalised and only report inv_norm (the LM loop folds the
                              scale into its normal equations: one full read+write pass less per map)  */
  HLA_VGG_SAVE_FOR_BACKWARD = 4, /* training: also keep relu(conv0) and the three max-pool argmax maps in the
                              workspace; the caller keeps the workspace alive until hla_vgg_backward   */
  HLA_VGG_FEAT16 = 8       /* dtype HLA_BF16 / HLA_F16 only, needs HLA_VGG_DEFER_NORM: feat[] are written as fp16 (saturating; also
                              in bf16 mode) instead of fp32 (inv_norm is that of the rounded maps); not with
                              HLA_VGG_SAVE_FOR_BACKWARD.  For hla_s2g_lm_solve with hla_s2g_level.feat_dtype set */
};

/* Weights are re-laid-out once into MFMA fragment order (bf16 or fp32) and reused until they change.
 * hla_vgg_pack_weights reads params->w[0..10] (conv0..conv_dec2.3) AND params->b[0] (conv0's bias rides in the padded k slots
 * of its fragments: the fused conv0 + conv2 kernel adds it inside the matrix product for the 16-bit types) and fills `packed`
 * (hla_vgg_packed_weight_bytes(dtype) bytes).  Call it again after an optimizer step -- also when only conv0's bias changed. */
size_t hla_vgg_packed_weight_bytes(int dtype);
int hla_vgg_pack_weights(const hla_vgg_params* params, void* packed, int dtype, hla_stream_t stream);

/* Bytes of scratch hla_vgg_forward needs for this shape. */
size_t hla_vgg_workspace_bytes(int B, int H, int W, int level, int dtype);

/* B, H, W  H and W multiples of 8, >= 8, and H*W < 2^23 pixels (8 388 608, e.g. below 2896 x 2896): the convolution kernels
 *          address a sample's activation map with signed 32-bit BYTE offsets and the largest map is H x W x 64 fp32.
 *          hla_vgg_forward / hla_vgg_backward return HLA_ERR_ARG above that.  B is unbounded (samples use 64-bit bases).
 * x        [B,3,H,W] NCHW fp32 (what the reference's DataLoader hands over).  With HLA_VGG_SAVE_FOR_BACKWARD the kernels of
 *          hla_vgg_backward read x AGAIN (conv0's weight gradient): the caller keeps it alive AND UNMODIFIED until then
 * x_plane  elements between consecutive channel planes of x: 0 = H*W (a dense tensor); larger when x is a window of H rows
 *          inside a taller image (rows stay W apart, samples 3*x_plane apart) -- mode='test' runs the ground extractor on the
 *          image rows that can reach the LM loop without first copying them out
 * params   biases b[0..6] and the confidence-head weights w[13..15] are read from here
 * packed_weights  output of hla_vgg_pack_weights for the same dtype
 * feat[l]  [B,H/2^(3-l),W/2^(3-l),C_l] NHWC fp32, C = 256,128,64 for l = 0..2 (x15,x18,x21):
 *          L2-normalised per sample, or raw when HLA_VGG_DEFER_NORM
 * conf[l]  [B,h_l,w_l] fp32 = sigmoid(-sigmoid(conv(relu(.)))), or NULL
 * inv_norm [3][B] fp64 out: 1/max(||map_l of sample b||_2, 1e-12) (VGG.py:511-514), or NULL
 * level    the reference's VGGUnet(level); 4 additionally computes x24 / conf3: feat[3] is [B,H,W,64] with the
 *          16 real channels first and zeros behind them, and params->w[11], w[12], w[16] must point at conv_dec3.1 /
 *          conv_dec3.3 / conf3.1 weights ZERO-PADDED to [64,128,3,3], [64,64,3,3], [1,64,3,3]; inv_norm is [4,B].
 *          3 -> maps 0..2 (the dead dec3/conf3 work of VGG.py:153-155,163 is skipped).
 * first_row8  0, or f in [4, H/8): a promise that the caller reads feat[0] / feat[1] / feat[2] only from rows f / 2f / 4f
 *          on (the LM loop reads rows h_l/2.. only, models_kitti.py:1194-1199).  Every layer then computes just the rows
 *          those depend on; rows above are left unwritten, and inv_norm covers the computed rows only (usable only where
 *          the per-sample scale cancels, as in LM_update).  conf[l], if requested, is likewise only valid from rows
 *          f / 2f / 4f on.  Ignored (treated as 0) at level 4 and with HLA_VGG_SAVE_FOR_BACKWARD.                */
int hla_vgg_forward(const float* x, size_t x_plane, const hla_vgg_params* params, const void* packed_weights, void* const feat[4],
                    float* const conf[4], double* inv_norm, void* workspace, size_t workspace_bytes, int B, int H,
                    int W, int level, int dtype, int flags, int first_row8, hla_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Backward of VGGUnet (autograd through VGG.py:121-203 in the reference).
 * ------------------------------------------------------------------------- */
/* fp32 gradient buffers with PyTorch's layouts (OIHW weights), overwritten: dw[0..10] = conv0..conv_dec2.3
 * (required), db[0..6] (NULL to skip), dw[13..16] = conf0.1..conf3.1 (required iff the matching d_conf is given).
 * Level 4: dw[11], dw[12] (required) and dw[16] have the ZERO-PADDED shapes [64,128,3,3], [64,64,3,3], [1,64,3,3]; the real
 * gradients are their leading [32,128], [16,32], [1,16] blocks.  At level 3 these three get no gradient. */
typedef struct hla_vgg_grads {
  float* dw[17];
  float* db[7];
} hla_vgg_grads;

size_t hla_vgg_packed_weight_T_bytes(int dtype);
/* transposed + tap-flipped fragment packing used by the data-gradient convolutions */
int hla_vgg_pack_weights_T(const hla_vgg_params* params, void* packed_T, int dtype, hla_stream_t stream);
size_t hla_vgg_bwd_workspace_bytes(int B, int H, int W, int level, int dtype);

/* x, x_plane, params  as in the forward call
 * fwd_workspace    the workspace of the forward call made with HLA_VGG_SAVE_FOR_BACKWARD | HLA_VGG_DEFER_NORM and the SAME dtype
 *                  (HLA_F16X3: it also holds the per-sample activation maxima the split weight-gradient kernels scale by)
 * feat[l], inv_norm  its outputs (raw maps + 1/norm)
 * d_feat[l]        d(loss)/d(L2-normalised map l), NHWC fp32 (what hla_s2g_lm_solve_bwd produces)
 * conf[l], d_conf[l]  the forward's confidence maps and d(loss)/d(conf map l) [B,h_l,w_l] fp32 (what
 *                  hla_s2g_lm_solve_bwd produces with using_weight=1, models_kitti.py:994-996); both arrays or
 *                  single entries may be NULL: then the heads get no gradient, as in the reference's default run.
 * flags            HLA_VGG_BWD_SCALE_INVARIANT: the consumer of the normalised maps does not depend on their per-sample
 *                  scale (LM_update renormalises both maps, models_kitti.py:982-990), so d_feat[l] is orthogonal to feat[l]
 *                  and the L2_norm Jacobian  dx = a*dy - a^3 (x.dy) x  (a = 1/||x||) reduces to a*dy: the (x.dy) pass and the
 *                  re-read of feat are skipped.  (In the reference that dot product is fp32 rounding noise, ~1e-7 |x||dy|.)
 *                  With this flag (level 3, no d_conf, first_row8 == 0, H <= 1024) the call also finds, per map row, the
 *                  column interval in which d_feat[l] is not exactly zero and skips every tile of every data- and
 *                  weight-gradient launch whose gradient is zero as a consequence (the satellite maps' gradient covers
 *                  ~10 % of the texels: the fan the camera sees).  Values do not change (weight gradients: the order of
 *                  the partial sums does); HLA_VGG_BWD_DENSE switches it off.
 * first_row8       0, or f in [4, H/8): a promise that d_feat[0] / d_feat[1] / d_feat[2] (and d_conf) are zero above rows
 *                  f / 2f / 4f -- the LM loop only reads rows h_l/2.. of the ground maps, so that is where its gradient
 *                  lives.  The rows of d_feat[l] above f * 2^l - 2 are then not even read (they may be uninitialised).  Every activation's gradient is then exactly zero above a first row that follows from the layer
 *                  graph, and the data- and weight-gradient launches skip those rows.  Needs HLA_VGG_BWD_SCALE_INVARIANT;
 *                  ignored (0) otherwise and at level 4. */
#define HLA_VGG_BWD_SCALE_INVARIANT 1
#define HLA_VGG_BWD_DENSE 2           /* visit every tile even where the incoming gradient is exactly zero (A/B and tests) */
#define HLA_VGG_BWD_WGRAD_TWO_PHASE 4 /* weight gradients on the two-phase kernels (512 workgroups, what a device that refuses the
                                         wave-specialised kernels' 96-115 KB LDS request runs) instead of the wave-specialised
                                         ones: the same products in another split-K grouping (A/B and tests); implies the next */
#define HLA_VGG_BWD_WGRAD0_UNFUSED 8  /* conv2's data gradient stored as a map and conv0's weight gradient computed from it by a
                                         kernel of its own (rounds 1-5), instead of inside that data gradient's epilogue, where the
                                         map is never written (level 3; A/B and tests) */
int hla_vgg_backward(const float* x, size_t x_plane, const hla_vgg_params* params, const void* packed_weights_T,
                     const void* fwd_workspace, const float* const feat[4], const double* inv_norm,
                     const float* const d_feat[4], const float* const conf[4], const float* const d_conf[4],
                     const hla_vgg_grads* grads, void* workspace, size_t workspace_bytes, int B, int H, int W, int level,
                     int dtype, int flags, int first_row8, hla_stream_t stream);

/* Diagnostics for the data-dependent trimming of the last hla_vgg_backward call that used `workspace` (its stream must have
 * completed): live and total tiles per sample of every data- and weight-gradient launch, summed.  Both are 0 when that call
 * took the dense walk.  Synchronous (one small device-to-host copy). */
int hla_vgg_backward_live_tiles(const void* workspace, int B, int H, int W, int level, int dtype, long long* live,
                                long long* total);

/* ------------------------------------------------------------------------- *
 * Dataset-side satellite tile: KITTI_dataset.py:128-157, Ford_dataset.py:185-209
 *   four Pillow resampling stages (NEAREST rotations in 16.16 fixed point, BILINEAR shifts in double with uint8
 *   truncation), TF.center_crop and ToTensor, evaluated lazily per output pixel; bit-identical to Pillow 12.2.
 * ------------------------------------------------------------------------- */
/* src     [B,S,S,3] uint8 (HWC) satellite images
 * stages  [B,4,8] fp64: per stage {kind (0 nearest / 1 bilinear), c0..c5, pad}.  For a nearest stage c[] are the
 *         16.16 fixed-point integers FIX(m0), FIX(m1), FIX(m2 + m0/2 + m1/2), FIX(m3), FIX(m4), FIX(m5 + m3/2 + m4/2),
 *         FIX(v) = floor(v*65536 + 0.5), of Image.rotate's matrix; for a bilinear stage the six AFFINE coefficients
 *         (highlyaccurate_amd/input_pipeline.py builds both)
 * out     [B,3,crop,crop] fp32 in [0,1] */
int hla_sat_tile(const unsigned char* src, const double* stages, float* out, int B, int S, int crop, hla_stream_t stream);

/* Ground image: torchvision Resize([out_h,out_w]) + ToTensor (KITTI_dataset.py:300-311, Ford_dataset.py:141-155) =
 * Pillow's antialiased BILINEAR resample: horizontal pass, uint8 intermediate, vertical pass, integer taps scaled by 2^22.
 * src [B,H,W,3] uint8; {h,v}bounds [n_out][2] = first input index, tap count; {h,v}taps [n_out][ksize] int32
 * (highlyaccurate_amd/input_pipeline.py builds them); mid [B,H,out_w,3] uint8 scratch; out [B,3,out_h,out_w] fp32 in [0,1] */
int hla_resize_bilinear(const unsigned char* src, const int* hbounds, const int* htaps, int hksize, const int* vbounds,
                        const int* vtaps, int vksize, unsigned char* mid, float* out, int B, int H, int W, int out_h,
                        int out_w, hla_stream_t stream);

/* ------------------------------------------------------------------------- *
 * jacobian.grid_sample  (jacobian.py:138-205) -- the stand-alone operator
 * ------------------------------------------------------------------------- */
/* image   [N,IH,IW,C] NHWC fp32;  optical [N,H,W,2] pixel coords (x,y)
 * jac     [M,N,H,W,2] or NULL;    out [N,H,W,C] NHWC;  jac_out [M,N,H,W,C] NHWC or NULL */
int hla_grid_sample(const float* image, const float* optical, const float* jac, float* out, float* jac_out,
                    int N, int C, int IH, int IW, int H, int W, int M, hla_stream_t stream);

/* ------------------------------------------------------------------------- *
 * The LM pose loop: project_map_to_grd + LM_update, N_iters x levels steps
 *   KITTI  models_kitti.py:700-1041, loop 1176-1283 (level-first 1352-1459)
 *   Ford   models_ford.py:173-466,  loop 682-835
 * ------------------------------------------------------------------------- */
typedef struct hla_s2g_level {
  const void* sat_feat;  /* [B,A,A,C]  NHWC, elements of feat_dtype */
  const void* grd_feat;  /* [B,h-grd_row_skip,w,C]  NHWC, same element type (rows grd_row_skip..h-1 of the level's map) */
  const float* grd_conf; /* [B,h-grd_row_skip,w] fp32 or NULL (needed iff using_weight) */
  const float* xyz;      /* [h,w,3] fp32 ground-plane points in the camera frame
                            (models_kitti.py:655-682 / models_ford.py:110-155) */
  const double* sat_inv_norm; /* [B] or NULL: sat_feat is raw, multiply by this (HLA_VGG_DEFER_NORM) */
  const double* grd_inv_norm; /* [B] or NULL: same for grd_feat */
  int A, h, w, C;
  int row0;              /* first ground-image row that takes part (h/2 for proj=='geo') */
  int grd_row_skip;      /* rows [0,grd_row_skip) of the ground map are not stored (<= row0): only the bottom half is
                            ever read (models_kitti.py:1194-1199), so a caller may extract features for the rows whose
                            receptive field reaches it and nothing above -- see DESIGN.md "dead rows"; 0 = full map */
  double meter_per_pixel;/* metres per satellite-feature pixel at this level */
  double centre;         /* A/2 (KITTI, float) or A//2 (Ford, integer) */
  int feat_dtype;        /* element type of sat_feat / grd_feat: HLA_F32 (always for the backward and for hla_g2s_*), HLA_F16 or
                            HLA_BF16.  The 16-bit maps hla_vgg_forward writes with HLA_VGG_FEAT16 are ALWAYS fp16 -- also when
                            the convolutions ran with dtype HLA_BF16 -- so they must be passed as HLA_F16 (HLA_BF16 here would
                            reinterpret fp16 bits; it is only for bf16 maps a caller made itself).  Inference in the
                            reduced-precision modes: half the bytes through the HBM-bound LM loop; all LM arithmetic stays
                            fp32 / fp64 */
} hla_s2g_level;

typedef struct hla_s2g_config {
  int ford;               /* 0: KITTI camera chain, 1: Ford cam->body->world chain */
  int n_levels, n_iters;
  int level_first;        /* 0: iter-outer/level-inner, 1: level-outer/iter-inner */
  int using_weight;       /* weight = grd_conf (models_kitti.py:994-996) */
  int use_hessian;        /* damping * diag(H) instead of damping * I */
  int dof;                /* 3: (u,v,theta); 2: rotation_range==0; 1: shift ranges == 0 (KITTI only) */
  double shift_range_lat, shift_range_lon, rotation_range; /* metres, metres, degrees */
  double damping[3];      /* lambda per pose component (already 10^(-6+11*sigmoid) if trained) */
  const unsigned char* keep;  /* args.dropout (models_kitti.py:968-974): [steps][keep_stride] bytes in execution order, 1 = the
                                 pixel (index within the rows row0..h-1 of that step's level) takes part; NULL = no dropout */
  size_t keep_stride;
  int optimizer;          /* 0: LM_update; 1: SGD_update (models_kitti.py:1056-1084: pose -= 0.01 * 2 J'(s - g), raw features,
                             no weights, no re-initialisation); 2: ADAM_update (1086-1125, beta1/beta2 below).  1 and 2 are
                             the reference's ablation optimisers; they ignore grd_conf and keep.  3: GN_update (Ford only,
                             models_ford.py:534-598: LM_update without damping and without renormalising the ground map;
                             reads grd_conf, ignores keep).  1-3 exist in the iteration-first loop only, as in the
                             reference; hla_s2g_lm_solve_bwd differentiates all four */
  double beta1, beta2;
  int count_in_view;      /* != 0 (and normal_eq given): slot 14 of normal_eq = the number of pixels of the WHOLE level map whose
                             satellite coordinates fall inside the map in that step -- what `assert mask.sum() > 0`
                             (jacobian.py:172) tests, summed over the batch.  One small extra launch per step. */
  int grd_grad_overwrite; /* hla_s2g_lm_solve_bwd only.  != 0: rows row0..h-1 of every level's d_grd_feat are WRITTEN, not added to
                             (the level's first visit of the reversed loop stores, the later ones add): the caller need not
                             zero-fill those rows (rows above row0 are never touched either way).  0: accumulate into the buffer */
  int deterministic;      /* hla_s2g_lm_solve_bwd only.  0: d(loss)/d(sat map) is scattered with fp32 atomics (their order, and with it
                             the last bits of every gradient behind it, varies from run to run).  != 0: it is accumulated in 64-bit
                             fixed point in the workspace (integer atomics commute) and written -- every element, no zero-fill needed --
                             to d_sat_feat by a closing pass: the same inputs give bitwise the same gradients.  The quantum is a
                             power of two per (level, sample): 2^-P of a bound of that sample's contributions at the level's first
                             visit (P = 40 for the value 1, or the value itself, 16..46); d_damping[3] counts the (step, sample)
                             pairs whose bound outgrew the first one's by more than 2^(50 - P), i.e. put the 63-bit range at
                             risk (must be 0: repeat the call with a smaller P).  Costs 8 B per satellite-map element of
                             workspace, its memset and the closing pass */
} hla_s2g_config;

size_t hla_s2g_workspace_bytes(const hla_s2g_config* cfg, const hla_s2g_level* levels, int B);

/* R_FL [B,3,3], T_FL [B,3] fp32 (Ford only, else NULL)
 * pose0    [B,3] fp32 (shift_u, shift_v, theta) normalised units, or NULL for zeros
 * rand_uv  [n_steps,2,B] fp32: the (rand_u, rand_v) re-initialisation draws of every step
 *          (models_kitti.py:1028-1033), drawn by the caller from torch's CPU generator
 * trace    [B,n_iters,n_levels,3] fp32 out: (shift_u, shift_v, theta) after every step
 * normal_eq[n_steps,B,16] fp64 out or NULL: ||s||^2, ||g||^2, H(6), J^T s (3), J^T g (3), pixels in view (count_in_view), pad */
int hla_s2g_lm_solve(const hla_s2g_config* cfg, const hla_s2g_level* levels, const float* R_FL,
                     const float* T_FL, const float* pose0, const float* rand_uv, float* trace,
                     double* normal_eq, void* workspace, size_t workspace_bytes, int B, hla_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Backward of the LM pose loop (what autograd does in the reference through models_kitti.py:1176-1283,
 * SURVEY Appendix C): d(loss)/d(trace) -> d(loss)/d(feature maps).  Gradients are taken w.r.t. the
 * L2-NORMALISED maps (inv_norm * stored map when the level carries inv norms); buffers are ACCUMULATED into
 * (the caller zero-fills them -- hla_zero_fill does it in one launch; d_grd_feat rows row0.. need not be with
 * cfg->grd_grad_overwrite, d_sat_feat not at all with cfg->deterministic), d_sat_feat with fp32 atomics unless cfg->deterministic.
 * ------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------- *
 * The same loop in the ground -> satellite direction: LM_G2SP (models_kitti.py:22-499, proj == 'geo')
 *   get_warp_sat2real 53-84, seq_warp_real2camera 86-161, project_grd_to_map 163-303, LM_update 333-379,
 *   loop 415-468.  Every satellite-map pixel samples the ground map where it projects to in the camera.
 * cfg        ford = 0, dof = 3, level_first = 0, use_hessian = 0 (the reference has no such variants here);
 *            damping[3] = the `damping` parameter (train_damping) or args.damping
 * levels[l]  sat_feat [B,A,A,C], grd_feat [B,h,w,C] (whole map: grd_row_skip = 0), grd_conf [B,h,w] iff using_weight,
 *            meter_per_pixel; xyz / row0 / centre are not read (the satellite grid is implicit, centre = A/2 integer)
 * camera_k   [B,3,3] fp32 intrinsics of the ori_h x ori_w ground IMAGE (left_camera_k, train_kitti.py:47)
 * trace      [B,N_iters,L,3] = (shift_u, shift_v, heading) after every step; no re-initialisation rule here.
 * normal_eq  [steps,B,16] or NULL: the 12 sums of every step in slots 2..13 (slots 0,1 = 1), needed by the backward */
size_t hla_g2s_workspace_bytes(const hla_s2g_config* cfg, const hla_s2g_level* levels, int B);
int hla_g2s_lm_solve(const hla_s2g_config* cfg, const hla_s2g_level* levels, const float* camera_k, int ori_h, int ori_w,
                     const float* pose0, float* trace, double* normal_eq, void* workspace, size_t workspace_bytes,
                     int B, hla_stream_t stream);

typedef struct hla_s2g_level_grad {
  float* d_sat_feat; /* [B,A,A,C] fp32 */
  float* d_grd_feat; /* [B,h,w,C] fp32 */
  float* d_grd_conf; /* [B,h,w] fp32 or NULL (only written when using_weight) */
} hla_s2g_level_grad;

size_t hla_s2g_bwd_workspace_bytes(const hla_s2g_config* cfg, const hla_s2g_level* levels, int B);

/* trace, normal_eq: outputs of the forward call (normal_eq is required here);  d_trace [B,n_iters,n_levels,3] fp32
 * d_damping [4] fp64 out (overwritten): [0..2] d(loss)/d(lambda_i) summed over samples and steps (for train_damping; summed per
 *           sample in step order, then over the samples in sample order: reproducible), [3] cfg->deterministic's range counter */
int hla_s2g_lm_solve_bwd(const hla_s2g_config* cfg, const hla_s2g_level* levels, const hla_s2g_level_grad* grads,
                         const float* R_FL, const float* T_FL, const float* pose0, const float* trace,
                         const double* normal_eq, const float* d_trace, double* d_damping, void* workspace,
                         size_t workspace_bytes, int B, hla_stream_t stream);

/* Backward of hla_g2s_lm_solve (autograd through the same reference lines).  grads[l]: d_sat_feat [B,A,A,C] and
 * d_grd_feat [B,h,w,C] are ACCUMULATED into (zero them first), likewise d_grd_conf [B,h,w] iff using_weight;
 * d_damping[3] (fp64) is overwritten with d(loss)/d(lambda) -- in this direction lambda IS the `damping` parameter
 * (models_kitti.py:41,358).  Gradients are w.r.t. the L2-normalised maps, as hla_vgg_backward expects. */
size_t hla_g2s_bwd_workspace_bytes(const hla_s2g_config* cfg, const hla_s2g_level* levels, int B);
int hla_g2s_lm_solve_bwd(const hla_s2g_config* cfg, const hla_s2g_level* levels, const hla_s2g_level_grad* grads,
                         const float* camera_k, int ori_h, int ori_w, const float* pose0, const float* trace,
                         const double* normal_eq, const float* d_trace, double* d_damping, void* workspace,
                         size_t workspace_bytes, int B, hla_stream_t stream);


/* Zero-fill up to 16 (strided) regions of device memory in ONE launch: what a caller of hla_s2g_lm_solve_bwd has to clear in front
 * of it (three d_sat_feat maps, the rows of the d_grd_feat maps its consumer reads above row0, d_grd_conf) -- as separate
 * fill launches these were ~10 of the ~25 small launches between the LM loop and its backward (the reference: autograd's own
 * zeros_like / accumulate nodes).  Region i = n_chunks pieces of chunk_bytes, stride_bytes apart, starting at ptr; ptr,
 * chunk_bytes and stride_bytes must be multiples of 4 (a region whose three are multiples of 16 is cleared with 16-byte stores). */
typedef struct hla_fill_region {
  void* ptr;
  size_t chunk_bytes, stride_bytes;
  int n_chunks;
} hla_fill_region;
int hla_zero_fill(const hla_fill_region* regions, int n_regions, int max_blocks, hla_stream_t stream);
/* max_blocks: 0 = as fast as the memory system takes it (up to 2048 workgroups per region); > 0 caps the workgroups per region --
 * a background fill on a side stream that leaves the chip to the kernels it runs under (training clears the LM backward's
 * 1.4 GB of gradient buffers that way while the forward's convolutions run) */

/* ------------------------------------------------------------------------- *
 * loss_func, loss_method 0  (models_ford.py:1041-1093; models_kitti.py imports the same function)
 * ------------------------------------------------------------------------- */
/* The pose loss of one training step and its gradient, one launch each (the reference: ~25 tensor ops + ~40 autograd nodes).
 *   d_k[n,l] = mean_b |x_k[b,n,l] - gt_k[b]|  (k = 0 lat, 1 lon, 2 theta; models_ford.py:1073-1079)
 *   losses   = coe[0] d_0 + coe[1] d_1 + coe[2] d_2                                   (models_ford.py:1086)
 * out[1 + 8 L], in the order of the reference's return tuple (models_ford.py:1091-1092):
 *   [0] mean(losses); then L values each of: losses[0]-losses[-1], d_0[0]-d_0[-1], d_1[0]-d_1[-1], d_2[0]-d_2[-1],
 *   losses[-1], d_0[-1], d_1[-1], d_2[-1].
 * gt_dtype selects the result type like torch's promotion does: fp32 inputs with fp32 ground truth give fp32 results
 * (KITTI), with fp64 ground truth fp64 results (the Ford loader hands float64).  `out` and `g_out` hold that type. */
enum { HLA_POSE_LOSS_F32 = 0, HLA_POSE_LOSS_F64 = 1 };
typedef struct hla_pose_loss_args {
  const float* x[3];          /* shift_lats, shift_lons, thetas [B,N,L] fp32: element (b,n,l) at x[k][b s0 + n s1 + l s2] */
  long long x_stride[3][3];   /* (s0, s1, s2) per input, in elements (three columns of one [B,N,L,3] trace: (3NL, 3L, 3)) */
  const void* gt[3];          /* gt_shift_lat, gt_shift_lon, gt_theta [B], fp32 or fp64 per gt_dtype */
  long long gt_stride[3];     /* in elements */
  double coe[3];              /* coe_shift_lat, coe_shift_lon, coe_theta */
  int B, N, L;                /* N * L <= 512 */
  int gt_dtype;
} hla_pose_loss_args;
int hla_pose_loss(const hla_pose_loss_args* args, void* out, hla_stream_t stream);
/* g_out[j]: d(scalar)/d(out piece j) (j = 0: one value; j = 1..8: L values, contiguous) or NULL (= zero).
 * dx[k]: gradient w.r.t. x[k], every element (b,n,l) written at dx[k][b t0 + n t1 + l t2], (t0,t1,t2) = dx_stride[k]. */
int hla_pose_loss_bwd(const hla_pose_loss_args* args, const void* const g_out[9], float* const dx[3],
                      const long long dx_stride[3][3], hla_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Measurement hooks (no reference counterpart; used by bench.py for the roofline numbers).
 * When enabled, every kernel launch of the entry points above is bracketed by two hipEvents
 * on its own stream; hla_prof_fetch synchronises them and returns one record per launch.
 * ------------------------------------------------------------------------- */
#define HLA_PROF_NKERNELS 17
typedef struct hla_prof_record {
  int kernel_id;  /* index for hla_prof_kernel_name */
  float ms;       /* event-to-event duration on the launch stream */
  double flops;   /* algorithmic FLOPs of the launch (2*9*Cin*Cout*rows*W*B for a conv, the rows it computes), else 0.  A
                     data-dependent launch of hla_vgg_backward (the satellite branch: only the tiles whose gradient is not exactly
                     zero) reports the EXECUTED share: dense FLOPs x live tiles / tiles, the live count read back through a pinned
                     host slot behind the kernel that wrote it */
  double bytes;   /* algorithmic HBM bytes of the launch (maps read once + outputs), scaled likewise, else 0 */
} hla_prof_record;
int hla_prof_enable(int on);
const char* hla_prof_kernel_name(int kernel_id);
int hla_prof_fetch(hla_prof_record* out, int max_records, int* n_out); /* also clears the log */
/* What the matrix pipe of this device SUSTAINS: back-to-back v_mfma_f32_32x32x16_{bf16,f16} on register-resident operands (no LDS,
 * no memory), two waves per SIMD, every CU busy for about ms_target milliseconds (0 < ms_target <= 200); synchronises the stream.
 * dtype HLA_BF16 / HLA_F16; data 0 = zero operands (reaches the nominal peak), 1 = random values, 2 = random with half the
 * elements zero (post-ReLU-like activations): on real data the package power limit, not the pipe, sets the rate.
 * tflops_out: achieved TFLOP/s.  bench.py reports it as roofline.mfma_sustained next to the nominal roofline.peak. */
int hla_prof_mfma_peak(int dtype, int data, float ms_target, float* tflops_out, hla_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HLA_H */
