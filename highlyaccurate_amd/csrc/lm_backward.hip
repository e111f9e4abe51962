// Backward of the LM pose loop (hla_s2g_lm_solve): given d(loss)/d(trace) for every step, walk the
// N_iters x levels steps in reverse and accumulate d(loss)/d(sat map), d(loss)/d(grd map) [, d/d(grd conf),
// d/d(damping)].  This is what autograd does in the reference through models_kitti.py:1176-1283 (gather values,
// bilinear weights, both norms, J^T W J, torch.inverse, the pose->uv chain of every later step; SURVEY
// Appendix C), restated as two pieces per step (ONE launch: the accumulate kernel's last-arriving tile of a sample runs the
// solve of the step before; only the very first solve is a launch of its own):
//   lm_bwd_solve   one wave per sample: closes the previous (later) step -- reduces its 12 projection-
//                  coefficient adjoints and pulls them back to the pose --, re-solves this step's damped
//                  system from the saved sums, and turns d(loss)/d(pose_out) into adjoints of the 14 sums.
//   lm_bwd_accum   same tiling as lm_accum: recomputes the gather, forms the per-element adjoints, scatters
//                  d/d(sat) to the 4 taps with fp32 atomics, adds d/d(grd) in place, and reduces the adjoints
//                  of the pixel coordinates / uv-Jacobians to 12 sums per tile.
// Gradients are w.r.t. the L2-NORMALISED maps (inv_norm * stored map); hla_vgg_backward applies the
// normalisation Jacobian.  d/d(sat) accumulation order is not deterministic (fp32 atomics) unless cfg->deterministic, which
// accumulates it in 64-bit fixed point (integer atomics commute: any order gives the same bits), see DetScale below.
#include "lm_common.h"
#include <math.h>

int hla_s2g_validate(const char* who, const hla_s2g_config* cfg, const hla_s2g_level* lv, const float* R_FL,
                     const float* T_FL, int B);

// ---------------------------------------------------------------------------------------------
struct BwdSolveArgs {
  // the later step (k+1) whose accum has just run, or part_next == null
  const double* part_next; int nt_next; LmGeom geom_next;
  // this step k
  const double* normal_eq;    // [B,16] forward sums of step k
  const float* pose_in;       // pose before step k: &trace[0][..][..][0] of step k-1, or pose0, or null (zeros)
  int pose_in_stride;
  const float* pose_out;      // pose after step k (= pose_in of step k+1), stride = trace_stride
  const float* d_trace;       // d(loss)/d(pose after step k), stride = trace_stride
  int trace_stride;
  double* gid;                // [B,3] identity-path adjoint carried between launches
  double* adj;                // [B,16] out
  double* coef;               // [B,COEF_N] out: forward coefficients of step k
  double* dlam;               // [B,4] per-sample sums of d(loss)/d(lambda_i) over the steps (one wave per sample and step adds, in
                              // step order: fixed order, no atomics) + [3]: deterministic mode's range-check counter; the
                              // LAST accumulate launch adds them up over the samples in sample order (BwdAccumArgs::dlam_final)
  const float* R_FL; const float* T_FL;
  int B, reinit, first;       // first: this is the last forward step (nothing to close, gid is not read)
  // the ablation updaters (models_kitti.py:1056-1116): 1 SGD, 2 ADAM.  ADAM re-runs its moment recurrence from the saved
  // sums of steps 0..t (neq_all) and carries the moment adjoints backwards in adam_adj [B,6]
  int optimizer, t;
  double beta1, beta2;
  const double* neq_all;
  double* adam_adj;
  LmSolveCfg cfg; LmGeom geom;
  // cfg->deterministic: the fixed-point exponent of this step's level and sample is chosen (q_first: the level's first visit of
  // the reversed loop) or checked against this step's bound (DetScale)
  int* qexp;                  // [B] of this step's level, or null
  int q_first, A, det_p;      // A: side of this step's satellite map (bounds |d(uv)/d(theta)|); det_p: precision bits below the bound
  // stand-alone launch only: zero this sample's tickets / ADAM adjoints / d_lambda sums (no memsets in front of the loop)
  unsigned* zero_ticket; int zero_steps;
};

// cfg->deterministic.  Every tap contribution c to d(loss)/d(sat map) is added as the integer rint(c / q) with one power-of-two
// quantum q per (level, sample), so the sum does not depend on the order of the atomics.  q comes from a rigorous bound of |c| in
// the step that visits the level first: with unit-norm maps (|feature| <= 1, |feature difference| <= 2), the step's 14 sum
// adjoints and jm_i = |d u/d pose_i| + |d v/d pose_i|,
//   |J_i| <= 2 jm_i,  |gs| <= 2|gS| + 2 sum_i |gU_i| jm_i,  |gJ_i| <= 2 sum_j |A_ij| jm_j + |gU_i| + |gV_i|,
//   |c| <= |gs| + 2 sum_i |gJ_i| jm_i  =: bound < 2^e,     q = 2^(e - DET_P).
// A texel collects at most 2^13 contributions over all visits of its level (KITTI: <= 400 ground pixels per texel cell x 4 cells x
// 5 visits), so |sum / q| < 2^(P + 13) * (largest bound / first bound): a later visit's bound may exceed the first one's by
// 2^(50 - P) before the 63-bit range is at risk -- checked per step and sample (the same bound, so the check itself is reproducible)
// and counted in d_damping[3], which the caller must find zero.  P = cfg->deterministic (1: 40; 16..46: that many bits).  Typical
// contributions are ~2^-11 of the bound (|feature| ~ 1 / sqrt(A A C)): at P = 40 they keep ~29 significant bits, more than the fp32
// atomics' 24 (deterministic against atomics gradients 1e-6 apart = the atomics' own run-to-run noise); at P = 32, 21 bits (1.6e-5).
// The reversed loop visits the LAST LM step first, whose adjoints are the smallest of the chain: at bench shapes with random
// weights an earlier step's bound outgrew it by more than the 2^10 that P = 40 leaves after a few Adam steps -- the host side
// (_s2gp.lm_backward) then repeats the call with P = 30 (2^20), which for given inputs is as reproducible as the first attempt.
#define DET_P_DEFAULT 40
#define DET_P_MIN 16
#define DET_P_MAX 46
__device__ static inline int det_bound_exp(const double* ad, const double* cf, int A) {
  const double jm[3] = {fabs(cf[8]) + fabs(cf[9]), fabs(cf[10]) + fabs(cf[11]), 2.0 * fabs(cf[12]) * (double)A};
  const double Am[3][3] = {{fabs(ad[2]), fabs(ad[3]), fabs(ad[4])}, {fabs(ad[3]), fabs(ad[5]), fabs(ad[6])}, {fabs(ad[4]), fabs(ad[6]), fabs(ad[7])}};
  double gs = 2.0 * fabs(ad[0]), tail = 0.0;
  for (int i = 0; i < 3; ++i) {
    gs += 2.0 * fabs(ad[8 + i]) * jm[i];
    double gJ = fabs(ad[8 + i]) + fabs(ad[11 + i]);
    for (int j = 0; j < 3; ++j) gJ += 2.0 * Am[i][j] * jm[j];
    tail += gJ * jm[i];
  }
  const double bound = gs + 2.0 * tail;
  if (!(bound > 0.0) || !(bound < 1e300)) return bound > 0.0 ? 1000 : -100;       // zero adjoints / a diverged solve
  const int e = ilogb(bound) + 1;
  return e < -100 ? -100 : e;
}

__device__ static inline void det_scale(const BwdSolveArgs& a, int b, const double* ad, const double* cf) {
  if (!a.qexp) return;
  const int e = det_bound_exp(ad, cf, a.A);
  if (a.q_first) {
    int q = e - a.det_p;
    a.qexp[b] = q < -126 ? -126 : (q > 100 ? 100 : q);
  } else if (e - a.det_p > a.qexp[b] + (50 - a.det_p)) {
    a.dlam[(size_t)b * 4 + 3] += 1.0;           // this step's contributions could leave the 63-bit range: reported, never silent
  }
}

// One wave per sample.  COHERENT: the tile sums of the later step were written by OTHER workgroups of the same launch (the
// accumulate kernel whose last-arriving tile of the sample closes with this solve): read them with agent-scope loads.
template <bool COHERENT>
__device__ __forceinline__ void lm_bwd_solve_body(const BwdSolveArgs& a, int b, int lane) {
  const float* R = a.R_FL ? a.R_FL + (size_t)b * 9 : nullptr;
  const float* T = a.T_FL ? a.T_FL + (size_t)b * 3 : nullptr;
  double c12[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) c12[k] = 0.0;
  if (a.part_next) {
    for (int i = lane; i < a.nt_next; i += 64) {
      const double* p = a.part_next + ((size_t)b * a.nt_next + i) * PART_N;
#pragma unroll
      for (int k = 0; k < 12; ++k) c12[k] += COHERENT ? __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[k];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) c12[k] = wave_sum_f64(c12[k]);
  }
  if (lane != 0) return;

  const float* po = a.pose_out + (size_t)b * a.trace_stride;
  const float* dt = a.d_trace + (size_t)b * a.trace_stride;
  double gout[3] = {dt[0], dt[1], dt[2]};
  if (!a.first) {
    double g3[3] = {0, 0, 0};
    if (a.part_next) lm_coefficients_bwd(a.geom_next, po[0], po[1], po[2], R, T, c12, g3);
    for (int p = 0; p < 3; ++p) gout[p] += g3[p] + a.gid[(size_t)b * 3 + p];
  }
  float pin[3] = {0.f, 0.f, 0.f};
  if (a.pose_in) { const float* pi = a.pose_in + (size_t)b * a.pose_in_stride; pin[0] = pi[0]; pin[1] = pi[1]; pin[2] = pi[2]; }

  const double* s = a.normal_eq + (size_t)b * 16;
  if (a.optimizer == 1 || a.optimizer == 2) {
    // pose_out = pose_in - 0.01 * f(g),  g = 2 (J^T s - J^T g_grd) on the whole-map-normalised features; no re-initialisation
    double gg[3];                    // adjoint of the raw gradient g
    for (int p = 0; p < 3; ++p) a.gid[(size_t)b * 3 + p] = gout[p];
    if (a.optimizer == 1) {
      for (int p = 0; p < 3; ++p) gg[p] = -0.01 * gout[p];
    } else {
      double* am = a.adam_adj + (size_t)b * 6;
      const double c1 = 1.0 - pow(a.beta1, a.t + 1), c2 = 1.0 - pow(a.beta2, a.t + 1);
      for (int p = 0; p < 3; ++p) {
        double m = 0.0, v = 0.0, g_t = 0.0;
        for (int j = 0; j <= a.t; ++j) {           // moments after step t
          const double* sj = a.neq_all + ((size_t)j * a.B + b) * 16;
          g_t = 2.0 * (sj[8 + p] - sj[11 + p]);
          m = a.beta1 * m + (1.0 - a.beta1) * g_t;
          v = a.beta2 * v + (1.0 - a.beta2) * g_t * g_t;
        }
        const double mh = m / c1, vh = v / c2, rt = sqrt(vh), den = rt + 1e-8;
        const double gd = -0.01 * gout[p];         // adjoint of delta_final
        const double g_m = am[p] + gd / (c1 * den);
        const double g_v = am[3 + p] + (rt > 0.0 ? -gd * mh / (den * den) * 0.5 / (rt * c2) : 0.0);
        gg[p] = (1.0 - a.beta1) * g_m + 2.0 * (1.0 - a.beta2) * g_t * g_v;
        am[p] = a.beta1 * g_m; am[3 + p] = a.beta2 * g_v;
      }
    }
    double* ad = a.adj + (size_t)b * 16;
    for (int k = 0; k < 16; ++k) ad[k] = 0.0;
    for (int p = 0; p < 3; ++p) { ad[8 + p] = 2.0 * gg[p]; ad[11 + p] = -2.0 * gg[p]; }
    lm_coefficients(a.geom, pin[0], pin[1], pin[2], R, T, a.coef + (size_t)b * COEF_N);
    det_scale(a, b, ad, a.coef + (size_t)b * COEF_N);
    return;
  }
  double H[3][3], g[3], Mi[3][3], d[3], ns, ng;
  lm_solve_step(a.cfg, s, H, g, Mi, d, ns, ng);
  // which components survived the re-initialisation rule (models_kitti.py:1032-1033)?
  double keep[3] = {1.0, 1.0, 1.0};
  if (a.reinit) {
    const float nu = (float)((double)pin[0] - d[0]), nv = (float)((double)pin[1] - d[1]);
    if (!(nu > -2.5f && nu < 2.5f)) keep[0] = 0.0;
    if (!(nv > -2.5f && nv < 2.5f)) keep[1] = 0.0;
  }
  double gnew[3], gd[3], y[3];
  for (int p = 0; p < 3; ++p) { gnew[p] = gout[p] * keep[p]; a.gid[(size_t)b * 3 + p] = gnew[p]; gd[p] = -gnew[p]; }
  for (int p = 0; p < 3; ++p) y[p] = Mi[p][0] * gd[0] + Mi[p][1] * gd[1] + Mi[p][2] * gd[2];
  // d = M^-1 g  =>  g_bar = y,  M_bar = -y d^T
  double gH[3][3];
  for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) gH[p][q] = -y[p] * d[q];
  const int nl = a.cfg.dof == 3 ? 3 : (a.cfg.dof == 2 ? 2 : 1);
  for (int i = 0; i < nl; ++i) {
    const int p = a.cfg.dof == 1 ? 2 : i;
    const double gM = gH[p][p];
    if (!a.cfg.gn) a.dlam[(size_t)b * 4 + i] += gM * (a.cfg.use_hessian ? H[p][p] : 1.0);     // GN_update has no damping
    if (a.cfg.use_hessian) gH[p][p] += a.cfg.lam[i] * gM;
  }
  const double is2 = 1.0 / (ns * ns), isg = 1.0 / (ns * ng);
  const double Hs[3][3] = {{s[2], s[3], s[4]}, {s[3], s[5], s[6]}, {s[4], s[6], s[7]}};
  double g_is2 = 0.0, g_isg = 0.0;
  for (int p = 0; p < 3; ++p) {
    for (int q = 0; q < 3; ++q) g_is2 += gH[p][q] * Hs[p][q];
    g_is2 += y[p] * s[8 + p];
    g_isg -= y[p] * s[11 + p];
  }
  const double g_ns = -2.0 * g_is2 / (ns * ns * ns) - g_isg / (ns * ns * ng);
  const double g_ng = -g_isg / (ns * ng * ng);
  double* ad = a.adj + (size_t)b * 16;
  ad[0] = sqrt(s[0]) > 1e-6 ? g_ns / (2.0 * ns) : 0.0;
  ad[1] = (!a.cfg.gn && sqrt(s[1]) > 1e-6) ? g_ng / (2.0 * ng) : 0.0;      // GN: ng is the constant 1
  ad[2] = 2.0 * is2 * gH[0][0]; ad[3] = is2 * (gH[0][1] + gH[1][0]); ad[4] = is2 * (gH[0][2] + gH[2][0]);
  ad[5] = 2.0 * is2 * gH[1][1]; ad[6] = is2 * (gH[1][2] + gH[2][1]); ad[7] = 2.0 * is2 * gH[2][2];
  for (int p = 0; p < 3; ++p) { ad[8 + p] = is2 * y[p]; ad[11 + p] = -isg * y[p]; }
  ad[14] = ad[15] = 0.0;
  lm_coefficients(a.geom, pin[0], pin[1], pin[2], R, T, a.coef + (size_t)b * COEF_N);
  det_scale(a, b, ad, a.coef + (size_t)b * COEF_N);
}

// stand-alone launch: the last forward step (nothing to close before it)
__global__ __launch_bounds__(64) void lm_bwd_solve(BwdSolveArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  // the per-sample state of the whole reversed loop starts here: arrival tickets of every step, ADAM's moment adjoints, the
  // d_lambda sums (three hipMemsetAsync launches in front of the loop until round 6)
  for (int k = lane; k < a.zero_steps; k += 64) a.zero_ticket[(size_t)k * a.B + b] = 0u;
  if (lane < 6 && a.adam_adj) a.adam_adj[(size_t)b * 6 + lane] = 0.0;
  if (lane < 4) a.dlam[(size_t)b * 4 + lane] = 0.0;
  __builtin_amdgcn_s_waitcnt(0);          // (lane 0 adds to its own zeroes below: same lane, program order)
  __syncthreads();
  lm_bwd_solve_body<false>(a, b, lane);
}


#ifndef BWD_MAX_TP
#define BWD_MAX_TP MAX_TP
#endif
struct BwdAccumArgs {
  const float* sat; const float* grd; const float* conf; const float* xyz;
  const double* coef;      // [B,COEF_N] forward coefficients of this step
  const double* adj;       // [B,16]: gS gG A00 A01 A02 A11 A12 A22 gU0 gU1 gU2 gV0 gV1 gV2
  const double* sat_inv;   // [B] or null
  const double* grd_inv;
  float* d_sat; float* d_grd; float* d_conf;
  double* part;            // [B,nt,PART_N]: 12 coefficient-adjoint sums
  int A, h, w, row0, npix, TP, nt, B, xcd_affine;
  int hs, rskip;      // stored rows of grd/conf (h - grd_row_skip) and the skip itself
  const unsigned char* keep;   // dropout: [npix] of this step, 1 = pixel takes part; or null
  unsigned* ticket;   // [B] arrival counters of this step (zeroed before the loop): the LAST tile of a sample closes with the
                      // solve of the step before (sa), or null: no closing
  int grd_assign;     // 1: d_grd rows row0.. are OVERWRITTEN by this launch (the first visit of the level: no zero-fill, no read), 0: added to
  // cfg->deterministic (DET instantiation): d(loss)/d(sat) goes to `d_sat_fix` as integers of the per-sample quantum 2^qexp[b]
  long long* d_sat_fix; const int* qexp;
  // the LAST launch of the reversed loop (step 0) only: add the per-sample d_lambda sums up in sample order -> d_damping[4]
  const double* dlam_final; double* d_damping_out;
};

template <int C, bool USE_W, bool DET = false>
__global__ __launch_bounds__(256, USE_W ? 3 : 4) void lm_bwd_accum(BwdAccumArgs a, BwdSolveArgs sa) {
  __shared__ PixParam pp[BWD_MAX_TP];
  __shared__ float pxyz[BWD_MAX_TP][3];   // the pixel's ground-plane point (the coefficient adjoints weight by it)
  __shared__ double red[4][12];
  __shared__ double c12s[12][256];
  int b, tile;
  if (!lm_block_map(a.xcd_affine, a.nt, a.B, b, tile)) return;
  const int t = threadIdx.x;
  if (a.dlam_final && b == 0 && tile == 0 && t < 4) {      // (step 0's launch: every sample's last solve ran in the launch before)
    double sum = 0.0;
    for (int bb = 0; bb < a.B; ++bb) sum += a.dlam_final[(size_t)bb * 4 + t];
    a.d_damping_out[t] = sum;
  }
  const int p0 = tile * a.TP;
  const int np = min(a.TP, a.npix - p0);
  const double* cf = a.coef + (size_t)b * COEF_N;

  for (int tt = t; tt < np; tt += 256) {
    const int p = p0 + tt;
    const int r = a.row0 + p / a.w, c = p % a.w;
    const float cw = USE_W ? a.conf[((size_t)b * a.hs + (r - a.rskip)) * a.w + c] : 1.f;
    const float* qx = a.xyz + ((size_t)r * a.w + c) * 3;
    PixParam P = lm_pixel<C>(cf, qx, a.A, cw);
    if (a.keep && !a.keep[p]) {          // dropped by args.dropout: the pixel leaves every sum (models_kitti.py:968-974)
      P.wx0 = P.wx1 = P.wy0 = P.wy1 = 0.f; P.off = P.dxo = P.dyo = 0; P.j2u = P.j2v = 0.f; P.gm = P.wt = P.m = 0.f;
    }
    pp[tt] = P;
    pxyz[tt][0] = qx[0]; pxyz[tt][1] = qx[1]; pxyz[tt][2] = qx[2];
  }
  __syncthreads();

  // The 22 per-sample scalars (adjoints of the 14 sums, both inverse norms, d(uv)/d(shift)) are uniform across the block: kept in
  // SGPRs (readfirstlane) they cost no vector registers.  Together with accumulating the tap gradients straight into the
  // texel-cell registers the kernel went from 184 VGPRs (2 waves per SIMD) to <= 128 (4): this gather / scatter loop waits on
  // memory, so resident waves are what it runs on.
  auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
  const double* ad = a.adj + (size_t)b * 16;
  const float gS = uni((float)ad[0]), gG = uni((float)ad[1]);
  const float A00 = uni((float)ad[2]), A01 = uni((float)ad[3]), A02 = uni((float)ad[4]), A11 = uni((float)ad[5]), A12 = uni((float)ad[6]),
              A22 = uni((float)ad[7]);
  const float gU0 = uni((float)ad[8]), gU1 = uni((float)ad[9]), gU2 = uni((float)ad[10]);
  const float gV0 = uni((float)ad[11]), gV1 = uni((float)ad[12]), gV2 = uni((float)ad[13]);
  const float as = uni(a.sat_inv ? (float)a.sat_inv[b] : 1.f), ag = uni(a.grd_inv ? (float)a.grd_inv[b] : 1.f);
  const float j0u = uni((float)cf[8]), j0v = uni((float)cf[9]), j1u = uni((float)cf[10]), j1v = uni((float)cf[11]);
  const float kkf = uni((float)cf[12]);
  constexpr int LPP = C / 4, PPW = 64 / LPP;
  const int lane = t & 63, wave = t >> 6;
  // lane j of a pixel's LPP lanes owns channels j, j+LPP, j+2LPP, j+3LPP: every load / atomic instruction then covers
  // LPP consecutive floats per pixel (whole cache lines) -- atomics are processed per touched line in L2
  const int sub = lane / LPP, cl = lane % LPP;
  // block-uniform bases (scalar registers) + 32-bit per-lane element offsets: a sample's map is < 2^31 elements
  const float* sat_u = a.sat + (size_t)b * a.A * a.A * C;
  float* dsat_u = a.d_sat + (size_t)b * a.A * a.A * C;
  const size_t grd_first = ((size_t)b * a.hs * a.w + (size_t)(a.row0 - a.rskip) * a.w + p0) * C;
  const float* grd_u = a.grd + grd_first;
  float* dgrd_u = a.d_grd + grd_first;

  // Every lane group walks a CONTIGUOUS run of ground pixels.  Neighbouring ground pixels oversample the satellite
  // map (2-20x laterally at KITTI geometry), so consecutive pixels mostly fall into the same texel cell: their tap
  // gradients are merged in registers and flushed with one set of atomics when the cell changes.
  const int RUN = (np + 4 * PPW - 1) / (4 * PPW);
  const int grp = wave * PPW + sub;
  int cur_off = -1, cur_dxo = 0, cur_dyo = 0;
  // Channel PAIRS (2h, 2h+1) as two-element vectors throughout: the loop compiles to v_pk_{mul,add,fma}_f32 on adjacent
  // registers.  (Written per channel the SLP vectoriser paired values that lived in unrelated registers: 35 of the loop's 447
  // instructions were v_mov_b32 shuffles.)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 c00[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}}, c01[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}}, c10[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}},
     c11[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};
  // DET: 1 / quantum of this sample (a power of two: the product below is exact, one rounding to the nearest integer)
  float qinv = 1.f;
  unsigned long long* dfix_u = nullptr;
  if constexpr (DET) {
    qinv = uni(ldexpf(1.f, -a.qexp[b]));
    dfix_u = (unsigned long long*)a.d_sat_fix + (size_t)b * a.A * a.A * C;
  }
  auto add1 = [&](int off, float v) __attribute__((always_inline)) {
    if constexpr (DET) atomicAdd(dfix_u + off, (unsigned long long)__float2ll_rn(v * qinv));      // two's complement: order-free
    else atomicAdd(dsat_u + off, v);
  };
  auto flush_cell = [&]() {
    if (cur_off >= 0) {
      const int o = cur_off + cl;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        add1(o + (2 * h) * LPP, c00[h].x); add1(o + (2 * h + 1) * LPP, c00[h].y);
        add1(o + cur_dxo + (2 * h) * LPP, c01[h].x); add1(o + cur_dxo + (2 * h + 1) * LPP, c01[h].y);
        add1(o + cur_dyo + (2 * h) * LPP, c10[h].x); add1(o + cur_dyo + (2 * h + 1) * LPP, c10[h].y);
        add1(o + cur_dyo + cur_dxo + (2 * h) * LPP, c11[h].x); add1(o + cur_dyo + cur_dxo + (2 * h + 1) * LPP, c11[h].y);
      }
    }
  };
  // the lane's 12 fp64 coefficient-adjoint sums live in LDS (one private column per thread, ds_add_f64): 24 registers less
#pragma unroll
  for (int k = 0; k < 12; ++k) c12s[k][t] = 0.0;
  for (int jr = 0; jr < RUN; ++jr) {
    const int i = grp * RUN + jr;
    if (i < np) {
      const PixParam P = pp[i];
      const float* sp = sat_u + (P.off + cl);
      const float* gq = grd_u + (i * C + cl);
      float* gp = dgrd_u + (i * C + cl);                               // this (pixel, channels) is owned by this lane
      // a pixel inside the map that falls into another texel cell than the run so far: flush the cell's sums, start a new one
      // (a pixel outside has all-zero weights: it adds exact zeros to whatever cell is open)
      if (P.m != 0.f && (P.off != cur_off || P.dxo != cur_dxo || P.dyo != cur_dyo)) {
        flush_cell();
        cur_off = P.off; cur_dxo = P.dxo; cur_dyo = P.dyo;
#pragma unroll
        for (int h = 0; h < 2; ++h) c00[h] = c01[h] = c10[h] = c11[h] = f2{0.f, 0.f};
      }
      float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f, q6 = 0.f, q7 = 0.f, q8 = 0.f;
      // two channels at a time: the loads of a pair (10 or 12 dwords per lane) are in flight together, the second pair's are
      // issued after the first pair's arithmetic -- half the load registers, and twice the waves to cover the latency instead
      const float w = USE_W ? P.wt : 1.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e0 = 2 * h * LPP, e1 = (2 * h + 1) * LPP;
        const f2 V00 = f2{sp[e0], sp[e1]} * as, V01 = f2{sp[P.dxo + e0], sp[P.dxo + e1]} * as;
        const f2 V10 = f2{sp[P.dyo + e0], sp[P.dyo + e1]} * as, V11 = f2{sp[P.dyo + P.dxo + e0], sp[P.dyo + P.dxo + e1]} * as;
        const f2 vg = f2{gq[e0], gq[e1]};
        const f2 og = a.grd_assign ? f2{0.f, 0.f} : f2{gp[e0], gp[e1]};
        const f2 top = P.wx0 * V00 + P.wx1 * V01, bot = P.wx0 * V10 + P.wx1 * V11;
        const f2 s = P.wy0 * top + P.wy1 * bot;
        const f2 dsy = bot - top;
        const f2 e01 = V01 - V00, e11 = V11 - V10;
        const f2 dsx = P.wy0 * e01 + P.wy1 * e11;
        const f2 dxy = (e11 - e01) * P.m;
        const f2 g = vg * (ag * P.gm);
        const f2 J0 = dsx * j0u + dsy * j0v, J1 = dsx * j1u + dsy * j1v, J2 = dsx * P.j2u + dsy * P.j2v;
        const f2 aj0 = A00 * J0 + A01 * J1 + A02 * J2, aj1 = A01 * J0 + A11 * J1 + A12 * J2, aj2 = A02 * J0 + A12 * J1 + A22 * J2;
        const f2 jU = J0 * gU0 + J1 * gU1 + J2 * gU2, jV = J0 * gV0 + J1 * gV1 + J2 * gV2;
        const f2 gs = (2.f * gS) * s + w * jU;
        const f2 ggr = (2.f * gG) * g + w * jV;
        const f2 gJ0 = w * (aj0 + s * gU0 + g * gV0), gJ1 = w * (aj1 + s * gU1 + g * gV1), gJ2 = w * (aj2 + s * gU2 + g * gV2);
        const f2 gdsx = gJ0 * j0u + gJ1 * j1u + gJ2 * P.j2u, gdsy = gJ0 * j0v + gJ1 * j1v + gJ2 * P.j2v;
        const f2 t0 = gs * dsx + gdsy * dxy, t1 = gs * dsy + gdsx * dxy;
        const f2 t2 = gJ0 * dsx, t3 = gJ0 * dsy, t4 = gJ1 * dsx, t5 = gJ1 * dsy, t6 = gJ2 * dsx, t7 = gJ2 * dsy;
        q0 += t0.x + t0.y; q1 += t1.x + t1.y; q2 += t2.x + t2.y; q3 += t3.x + t3.y;
        q4 += t4.x + t4.y; q5 += t5.x + t5.y; q6 += t6.x + t6.y; q7 += t7.x + t7.y;
        if (USE_W) {
          const f2 t8 = 0.5f * (J0 * aj0 + J1 * aj1 + J2 * aj2) + s * jU + g * jV;
          q8 += t8.x + t8.y;
        }
        const f2 gsy0 = gs * P.wy0, gsy1 = gs * P.wy1;
        c00[h] += gsy0 * P.wx0 - gdsx * P.wy0 - gdsy * P.wx0;
        c01[h] += gsy0 * P.wx1 + gdsx * P.wy0 - gdsy * P.wx1;
        c10[h] += gsy1 * P.wx0 - gdsx * P.wy1 + gdsy * P.wx0;
        c11[h] += gsy1 * P.wx1 + gdsx * P.wy1 + gdsy * P.wx1;
        const f2 ng = og + ggr * P.gm;
        gp[e0] = ng.x; gp[e1] = ng.y;
      }
      // pixel adjoints -> adjoints of the 12 projection coefficients.  The map is linear, so every lane applies it to its own
      // partial sums (its 4 channels of the pixel) and accumulates in fp64 across its run of pixels; ONE cross-lane reduction at
      // the end replaces round 2's per-pixel butterfly of eight values over the LPP lanes that share a pixel (72 of the loop's
      // ~400 instructions) and the pass over an LDS copy of the per-pixel sums.
      const double X = pxyz[i][0], Y = pxyz[i][1], Z = pxyz[i][2];
      const double gu = (double)q0 - (double)kkf * (double)q7;     // j2v = -k (u - ctr)
      const double gv = (double)q1 + (double)kkf * (double)q6;     // j2u =  k (v - ctr)
      atomicAdd(&c12s[0][t], gu * X); atomicAdd(&c12s[1][t], gu * Y); atomicAdd(&c12s[2][t], gu * Z); atomicAdd(&c12s[3][t], gu);
      atomicAdd(&c12s[4][t], gv * X); atomicAdd(&c12s[5][t], gv * Y); atomicAdd(&c12s[6][t], gv * Z); atomicAdd(&c12s[7][t], gv);
      atomicAdd(&c12s[8][t], (double)q2); atomicAdd(&c12s[9][t], (double)q3); atomicAdd(&c12s[10][t], (double)q4);
      atomicAdd(&c12s[11][t], (double)q5);
      if (USE_W) {       // d(loss)/d(confidence) is per pixel: that one value is still reduced over the pixel's lanes
        float qw = q8;   // (the LPP lanes of a pixel share `i`, so they are all inside this branch together)
#pragma unroll
        for (int o = LPP >> 1; o > 0; o >>= 1) qw += __shfl_xor(qw, o, 64);
        if (cl == 0 && a.d_conf) {
          const int p = p0 + i;
          const int r = a.row0 + p / a.w, c = p % a.w;
          a.d_conf[((size_t)b * a.hs + (r - a.rskip)) * a.w + c] += qw * P.gm;
        }
      }
    }
  }
  flush_cell();
  __syncthreads();
  double c12[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) c12[k] = wave_sum_f64(c12s[k][t]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) red[wave][k] = c12[k];
  }
  __syncthreads();
  if (wave != 0) return;
  if (t < PART_N) {
    double v = 0.0;
    if (t < 12) v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    __hip_atomic_store(a.part + ((size_t)b * a.nt + tile) * PART_N + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!a.ticket) return;
  // The last tile of the sample to get here closes with the NEXT launch's solve (the step before this one): one launch per
  // step instead of two.  Same publication recipe as the forward (lm_solve.hip): agent-scope stores of the sums -> vmcnt(0) ->
  // relaxed agent-scope ticket; the closing wave reads with agent-scope loads and overwrites this sample's adj / coef, which
  // every other tile of the sample has finished reading (they drew their tickets).  The sums are added in tile order whoever closes.
  const unsigned old = lm_draw_ticket(a.ticket + b, lane);      // (lm_common.h)
  if (old + 1 != (unsigned)a.nt) return;
  lm_bwd_solve_body<true>(sa, b, lane);
}

// pixels per block of lm_bwd_accum (level-dependent only)
// (round 5, at 4 waves per SIMD, same box x2, us per launch C64 / C128 / C256: 256/256/128 (these) 321 / 154 / 104; 256/128/64 326 /
//  167 / 110; 128/128/64 383 / 161 / 108; 128/64/32 385 / 227 / 120; 512/256/128 412 / 187 / 103; 512/512/256 414 / 181 / 143:
//  smaller tiles merge fewer taps per texel cell, larger ones leave two blocks per CU)
// (measured, same-box A/B twice: 256 / 128 / 64 -- the sizes LM_G2SP's kernels use -- 227.7 us per launch on average against
// 218.3 for these; 128 / 128 / 64: 245)
#ifndef HLA_LMB_TP0
#define HLA_LMB_TP0 256     // npix >= 16384 (KITTI: the 64-channel level)
#define HLA_LMB_TP1 256     // npix >= 4096  (the 128-channel level)
#define HLA_LMB_TP2 128     // below         (the 256-channel level)
#endif
static inline int lm_pick_tile_bwd(int npix) { return npix >= 16384 ? HLA_LMB_TP0 : (npix >= 4096 ? HLA_LMB_TP1 : HLA_LMB_TP2); }

// ---------------------------------------------------------------------------------------------
// off[0..5]: coef, adj, gid, part, ADAM adjoints, tickets; off[6]: dlam [B,4]; off[7]: qexp [L,B]; off[8 + l]: level l's 64-bit
// fixed-point d(loss)/d(sat) accumulators (cfg->deterministic only)
static size_t bwd_layout(const hla_s2g_config* cfg, const hla_s2g_level* lv, int B, size_t off[12]) {
  int max_nt = 1;
  for (int l = 0; l < cfg->n_levels; ++l) {
    const int npix = (lv[l].h - lv[l].row0) * lv[l].w;
    const int tp = lm_pick_tile_bwd(npix);
    max_nt = max(max_nt, (npix + tp - 1) / tp);
  }
  size_t o = 0;
  off[0] = o; o += hla_align_up((size_t)B * COEF_N * sizeof(double), 256);          // coef
  off[1] = o; o += hla_align_up((size_t)B * 16 * sizeof(double), 256);              // adj
  off[2] = o; o += hla_align_up((size_t)B * 3 * sizeof(double), 256);               // gid
  off[3] = o; o += hla_align_up((size_t)B * max_nt * PART_N * sizeof(double), 256); // part
  off[4] = o; o += hla_align_up((size_t)B * 6 * sizeof(double), 256);               // ADAM moment adjoints
  off[5] = o; o += hla_align_up((size_t)B * cfg->n_levels * cfg->n_iters * sizeof(unsigned), 256);   // arrival tickets [steps][B]
  off[6] = o; o += hla_align_up((size_t)B * 4 * sizeof(double), 256);               // per-sample d_lambda sums + range counter
  off[7] = o; o += hla_align_up((size_t)B * 4 * sizeof(int), 256);                  // fixed-point exponents [4 levels][B]
  for (int l = 0; l < 4; ++l) {
    off[8 + l] = o;
    if (cfg->deterministic && l < cfg->n_levels) o += hla_align_up((size_t)B * lv[l].A * lv[l].A * lv[l].C * sizeof(long long), 256);
  }
  return o;
}

extern "C" size_t hla_s2g_bwd_workspace_bytes(const hla_s2g_config* cfg, const hla_s2g_level* levels, int B) {
  size_t off[12];
  return bwd_layout(cfg, levels, B, off);
}

// cfg->deterministic: the fixed-point sums -> the caller's fp32 d_sat_feat buffers (every element written: no zero-fill needed there)
struct DetConvertArgs { const long long* fix[4]; float* out[4]; const int* qexp; size_t per[4]; int B; };
static __global__ __launch_bounds__(256) void det_convert_kernel(DetConvertArgs a) {
  const int l = blockIdx.z, b = blockIdx.y;
  const double q = ldexp(1.0, a.qexp[(size_t)l * a.B + b]);
  const long long* src = a.fix[l] + (size_t)b * a.per[l];
  float* dst = a.out[l] + (size_t)b * a.per[l];
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  const size_t n4 = a.per[l] / 4;                       // (C is a multiple of 16)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const ll2 v0 = ((const ll2*)src)[2 * i], v1 = ((const ll2*)src)[2 * i + 1];
    ((float4*)dst)[i] = make_float4((float)((double)v0.x * q), (float)((double)v0.y * q), (float)((double)v1.x * q), (float)((double)v1.y * q));
  }
}

template <bool W, bool DET>
static void launch_bwd_accum(int C, dim3 grid, hipStream_t st, const BwdAccumArgs& a, const BwdSolveArgs& sa) {
  switch (C) {
    case 256: hipLaunchKernelGGL((lm_bwd_accum<256, W, DET>), grid, dim3(256), 0, st, a, sa); break;
    case 128: hipLaunchKernelGGL((lm_bwd_accum<128, W, DET>), grid, dim3(256), 0, st, a, sa); break;
    case 64: hipLaunchKernelGGL((lm_bwd_accum<64, W, DET>), grid, dim3(256), 0, st, a, sa); break;
    case 16: hipLaunchKernelGGL((lm_bwd_accum<16, W, DET>), grid, dim3(256), 0, st, a, sa); break;
  }
}

extern "C" int hla_s2g_lm_solve_bwd(const hla_s2g_config* cfg, const hla_s2g_level* lv, const hla_s2g_level_grad* gr,
                                    const float* R_FL, const float* T_FL, const float* pose0, const float* trace,
                                    const double* normal_eq, const float* d_trace, double* d_damping, void* workspace,
                                    size_t workspace_bytes, int B, hla_stream_t stream) {
  HLA_REQUIRE(gr && trace && normal_eq && d_trace && d_damping && workspace, "hla_s2g_lm_solve_bwd: null argument");
  HLA_REQUIRE(cfg && cfg->optimizer >= 0 && cfg->optimizer <= 3, "hla_s2g_lm_solve_bwd: optimizer must be 0 (LM), 1 (SGD), 2 (ADAM) or 3 (GN)");
  const int rc = hla_s2g_validate("hla_s2g_lm_solve_bwd", cfg, lv, R_FL, T_FL, B);
  if (rc) return rc;
  for (int l = 0; l < cfg->n_levels; ++l) {
    HLA_REQUIRE(gr[l].d_sat_feat && gr[l].d_grd_feat, "hla_s2g_lm_solve_bwd: level %d gradient buffers missing", l);
    HLA_REQUIRE(lv[l].feat_dtype == HLA_F32, "hla_s2g_lm_solve_bwd: level %d: the backward needs fp32 feature maps", l);
  }
  size_t off[12];
  const size_t need = bwd_layout(cfg, lv, B, off);
  if (workspace_bytes < need) {
    hla_set_error("hla_s2g_lm_solve_bwd: workspace %zu < %zu", workspace_bytes, need);
    return HLA_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  double* coef = (double*)(ws + off[0]);
  double* adj = (double*)(ws + off[1]);
  double* gid = (double*)(ws + off[2]);
  double* part = (double*)(ws + off[3]);
  // (tickets, ADAM adjoints and the d_lambda sums are zeroed by the first launch, lm_bwd_solve: no memset launches here)
  double* adam_adj = (double*)(ws + off[4]);
  unsigned* tickets = (unsigned*)(ws + off[5]);
  double* dlam = (double*)(ws + off[6]);
  int* qexp = (int*)(ws + off[7]);
  const bool det = cfg->deterministic != 0;
  HLA_REQUIRE(cfg->n_levels <= 4, "hla_s2g_lm_solve_bwd: at most 4 levels");
  if (det) {      // the fixed-point accumulators of all levels are one contiguous region
    const size_t bytes = need - off[8];
    HLA_CHECK_HIP(hipMemsetAsync(ws + off[8], 0, bytes, st));
  }

  const bool newton = cfg->optimizer == 0 || cfg->optimizer == 3;
  const bool reinit = (cfg->ford || cfg->dof == 3) && newton;
  const int L = cfg->n_levels, N = cfg->n_iters, steps = L * N, tstride = N * L * 3;
  auto step_level = [&](int k) { return cfg->level_first ? k / N : k % L; };
  auto step_iter = [&](int k) { return cfg->level_first ? k % N : k / L; };
  auto slot = [&](int k) { return ((size_t)step_iter(k) * L + step_level(k)) * 3; };
  auto geom = [&](int l) {
    LmGeom g{};
    g.ford = cfg->ford; g.lat = cfg->shift_range_lat; g.lon = cfg->shift_range_lon; g.rot = cfg->rotation_range;
    g.mpp = lv[l].meter_per_pixel; g.ctr = lv[l].centre;
    return g;
  };

  // One launch per step: lm_bwd_accum(k), whose last-arriving tile of every sample closes with the solve of step k - 1 (the
  // pose adjoint pulled back through step k's coefficients, step k - 1's damped system re-solved -> its 14 sum adjoints).  Only
  // the solve of the LAST forward step, which has nothing to close, is a launch of its own.
  auto nt_of = [&](int l) {
    const int npix = (lv[l].h - lv[l].row0) * lv[l].w;
    const int tp = lm_pick_tile_bwd(npix);
    return (npix + tp - 1) / tp;
  };
  auto solve_args = [&](int k) {
    BwdSolveArgs sa{};
    const int l = step_level(k);
    sa.first = (k == steps - 1) ? 1 : 0;
    if (!sa.first) { sa.part_next = part; sa.nt_next = nt_of(step_level(k + 1)); sa.geom_next = geom(step_level(k + 1)); }
    sa.normal_eq = normal_eq + (size_t)k * B * 16;
    if (k > 0) { sa.pose_in = trace + slot(k - 1); sa.pose_in_stride = tstride; }
    else { sa.pose_in = pose0; sa.pose_in_stride = 3; }
    sa.pose_out = trace + slot(k); sa.d_trace = d_trace + slot(k); sa.trace_stride = tstride;
    sa.gid = gid; sa.adj = adj; sa.coef = coef; sa.dlam = dlam;
    if (det) {       // is step k the first visit of its level in the reversed loop?
      bool first_visit = true;
      for (int j = k + 1; j < steps; ++j) if (step_level(j) == l) first_visit = false;
      sa.qexp = qexp + (size_t)l * B; sa.q_first = first_visit ? 1 : 0; sa.A = lv[l].A;
      sa.det_p = cfg->deterministic == 1 ? DET_P_DEFAULT : (cfg->deterministic < DET_P_MIN ? DET_P_MIN : (cfg->deterministic > DET_P_MAX ? DET_P_MAX : cfg->deterministic));
    }
    sa.R_FL = R_FL; sa.T_FL = T_FL; sa.B = B; sa.reinit = reinit ? 1 : 0;
    sa.cfg.gn = cfg->optimizer == 3 ? 1 : 0;
    sa.cfg.dof = cfg->dof; sa.cfg.use_hessian = sa.cfg.gn ? 0 : cfg->use_hessian;
    for (int i = 0; i < 3; ++i) sa.cfg.lam[i] = sa.cfg.gn ? 0.0 : cfg->damping[i];
    sa.optimizer = cfg->optimizer; sa.t = k; sa.beta1 = cfg->beta1; sa.beta2 = cfg->beta2;
    sa.neq_all = normal_eq; sa.adam_adj = adam_adj;
    sa.geom = geom(l);
    return sa;
  };
  {
    BwdSolveArgs sa = solve_args(steps - 1);
    sa.zero_ticket = tickets; sa.zero_steps = steps;
    hla_prof_begin(K_LMSOLVE, 0, 0, st);
    hipLaunchKernelGGL(lm_bwd_solve, dim3(B), dim3(64), 0, st, sa);
    hla_prof_end(st);
  }
  unsigned visited = 0;       // levels whose d_grd rows this call has written already (cfg->grd_grad_overwrite)
  for (int k = steps - 1; k >= 0; --k) {
    const int l = step_level(k);
    const hla_s2g_level& v = lv[l];
    BwdAccumArgs aa{};
    aa.sat = (const float*)v.sat_feat; aa.grd = (const float*)v.grd_feat; aa.conf = v.grd_conf; aa.xyz = v.xyz; aa.coef = coef; aa.adj = adj;
    aa.sat_inv = v.sat_inv_norm; aa.grd_inv = v.grd_inv_norm;
    aa.d_sat = gr[l].d_sat_feat; aa.d_grd = gr[l].d_grd_feat; aa.d_conf = gr[l].d_grd_conf; aa.part = part;
    aa.A = v.A; aa.h = v.h; aa.w = v.w; aa.row0 = v.row0; aa.npix = (v.h - v.row0) * v.w;
    aa.hs = v.h - v.grd_row_skip; aa.rskip = v.grd_row_skip;
    aa.keep = cfg->keep ? cfg->keep + (size_t)k * cfg->keep_stride : nullptr;
    aa.grd_assign = (cfg->grd_grad_overwrite && !((visited >> l) & 1u)) ? 1 : 0;
    visited |= 1u << l;
    aa.TP = lm_pick_tile_bwd(aa.npix); aa.nt = nt_of(l); aa.B = B;
    aa.xcd_affine = (B >= 8) ? 1 : 0;
    // step 0 has no earlier step to solve for: no ticket, no closing
    aa.ticket = k > 0 ? tickets + (size_t)k * B : nullptr;
    const BwdSolveArgs sa = k > 0 ? solve_args(k - 1) : BwdSolveArgs{};
    if (det) { aa.d_sat_fix = (long long*)(ws + off[8 + l]); aa.qexp = qexp + (size_t)l * B; }
    if (k == 0) { aa.dlam_final = dlam; aa.d_damping_out = d_damping; }
    const int nblk = aa.xcd_affine ? 8 * ((B + 7) / 8) * aa.nt : B * aa.nt;
    hla_prof_begin(K_LMBWD, 0, (double)B * (5.0 * (double)v.A * v.A + 3.0 * (double)aa.npix) * v.C * 4.0, st);
    const bool w = cfg->using_weight && newton;
    if (det) { if (w) launch_bwd_accum<true, true>(v.C, dim3(nblk), st, aa, sa); else launch_bwd_accum<false, true>(v.C, dim3(nblk), st, aa, sa); }
    else { if (w) launch_bwd_accum<true, false>(v.C, dim3(nblk), st, aa, sa); else launch_bwd_accum<false, false>(v.C, dim3(nblk), st, aa, sa); }
    hla_prof_end(st);
  }
  if (det) {
    DetConvertArgs ca{};
    size_t maxper = 0;
    for (int l = 0; l < L; ++l) {
      ca.fix[l] = (const long long*)(ws + off[8 + l]); ca.out[l] = gr[l].d_sat_feat;
      ca.per[l] = (size_t)lv[l].A * lv[l].A * lv[l].C;
      maxper = ca.per[l] > maxper ? ca.per[l] : maxper;
    }
    ca.qexp = qexp; ca.B = B;
    int gx = (int)((maxper / 4 + 255) / 256);
    gx = gx > 256 ? 256 : (gx < 1 ? 1 : gx);
    hla_prof_begin(K_ELEMWISE, 0, 0, st);
    hipLaunchKernelGGL(det_convert_kernel, dim3(gx, B, L), dim3(256), 0, st, ca);
    hla_prof_end(st);
  }
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}
