// Fused satellite->ground projection + bilinear gather + feature Jacobian + normal equations,
// and the per-sample damped 3x3 solve: the LM pose loop of
//   models_kitti.py:700-1041 (grd2cam2world2sat, project_map_to_grd, LM_update), loop 1176-1283
//   models_ford.py:173-466, loop 682-835
//   jacobian.py:138-205 (grid_sample)
//
// HBM-bound.  One step = two launches on the caller's stream:
//   lm_accum<C>  grid (tiles x samples): every wave walks ground pixels with its lanes spread over
//                the channel axis (NHWC, 16 B per lane => each bilinear tap is one fully coalesced
//                C*4-byte read), and accumulates the 14 sums that define the normal equations.
//                The [3,B,C,h,w] Jacobian of the reference (>=369 MB/sample/step) is never formed.
//   lm_solve     one wave per sample: fixed-order fp64 reduction of the tile partials (bitwise
//                deterministic), 3x3/2x2/1x1 damped solve, pose update + re-initialisation rule,
//                and the projection coefficients of the NEXT step (so no extra launch for them).
#include "lm_common.h"

struct SolveArgs {
  const double* part;    // [B,nt,PART_N] of the step being closed, or null (init launch)
  const double* sat_inv; // [B] or null: feature maps are stored un-normalised, sums are rescaled here
  const double* grd_inv;
  int nt;
  float* pose;           // [B,3] running pose (shift_u, shift_v, theta), fp32 like the reference
  float* trace_out;      // &trace[0][iter][level][0] of this step (sample stride = trace_stride)
  int trace_stride;
  const float* rand_uv;  // [2,B] for this step, or null
  double* normal_eq;     // [B,16] for this step, or null
  double* coef;          // [B,COEF_N] out for the next step, or null
  const float* R_FL;     // [B,3,3]
  const float* T_FL;     // [B,3]
  int B, reinit;
  int optimizer, t;      // 0 LM; 1 SGD; 2 ADAM (t = step index in execution order); 3 GN (cfg.gn)
  double beta1, beta2;
  double* adam;          // [B,6] first / second moment of the three pose components (ADAM only)
  const int* in_view;    // [B] pixels in view in this step (count_in_view), or null
  LmSolveCfg cfg;
  LmGeom next;           // geometry of the level the NEXT step runs on
  // init launch only (part == null): what used to be two hipMemsetAsync launches in front of every forward
  unsigned* zero_ticket; // [steps][B] arrival counters of all steps, or null
  int zero_steps, zero_pose;   // zero_pose: no initial pose was given, start from 0
};

// One wave closes a step for sample b.  COHERENT: the tile partials were written by OTHER workgroups of the same launch
// (the fused accumulate + solve kernel): read them with agent-scope (sc1) loads, which bypass this CU's L1.
template <bool COHERENT>
__device__ __forceinline__ void lm_solve_body(const SolveArgs& a, int b, int lane) {
  float su = a.pose[b * 3 + 0], sv = a.pose[b * 3 + 1], th = a.pose[b * 3 + 2];
  if (!a.part) {          // the init launch also clears this sample's arrival counters and, without an initial pose, its pose
    if (a.zero_ticket)
      for (int k = lane; k < a.zero_steps; k += 64) a.zero_ticket[(size_t)k * a.B + b] = 0u;
    if (a.zero_pose) {
      su = sv = th = 0.f;
      if (lane < 3) a.pose[b * 3 + lane] = 0.f;
    }
  }

  if (a.part) {
    double s[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) s[k] = 0.0;
    for (int i = lane; i < a.nt; i += 64) {
      const double* p = a.part + ((size_t)b * a.nt + i) * PART_N;
#pragma unroll
      for (int k = 0; k < 14; ++k)
        s[k] += COHERENT ? __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[k];
    }
#pragma unroll
    for (int k = 0; k < 14; ++k) s[k] = wave_sum_f64(s[k]);

    if (lane == 0) {
      // deferred L2_norm (VGG.py:511-514): s -> as*s, J -> as*J, g -> ag*g
      const double as = a.sat_inv ? a.sat_inv[b] : 1.0, ag = a.grd_inv ? a.grd_inv[b] : 1.0;
      s[0] *= as * as; s[1] *= ag * ag;
      for (int k = 2; k < 11; ++k) s[k] *= as * as;
      for (int k = 11; k < 14; ++k) s[k] *= as * ag;
      if (a.normal_eq) {
        for (int k = 0; k < 14; ++k) a.normal_eq[(size_t)b * 16 + k] = s[k];
        a.normal_eq[(size_t)b * 16 + 14] = a.in_view ? (double)a.in_view[b] : 0.0;
        a.normal_eq[(size_t)b * 16 + 15] = 0.0;
      }
      double H[3][3], g[3], Mi[3][3], d[3], ns, ng;
      if (a.optimizer == 0 || a.optimizer == 3) {
        lm_solve_step(a.cfg, s, H, g, Mi, d, ns, ng);
      } else {                       // ablation optimisers on the raw residual: delta_pose = sum 2 r J (models_kitti.py:1075-1076)
        for (int p = 0; p < 3; ++p) d[p] = 2.0 * (s[8 + p] - s[11 + p]);
        if (a.optimizer == 2) {      // ADAM_update, 1110-1116
          double* mv = a.adam + (size_t)b * 6;
          for (int p = 0; p < 3; ++p) {
            const double m = a.beta1 * mv[p] + (1.0 - a.beta1) * d[p];
            const double v = a.beta2 * mv[3 + p] + (1.0 - a.beta2) * d[p] * d[p];
            mv[p] = m; mv[3 + p] = v;
            d[p] = (m / (1.0 - pow(a.beta1, a.t + 1))) / (sqrt(v / (1.0 - pow(a.beta2, a.t + 1))) + 1e-8);
          }
        }
        for (int p = 0; p < 3; ++p) d[p] *= 0.01;
      }
      su = (float)((double)su - d[0]);
      sv = (float)((double)sv - d[1]);
      th = (float)((double)th - d[2]);
      if (a.reinit) {                // models_kitti.py:1028-1033
        const float ru = a.rand_uv[b], rv = a.rand_uv[a.B + b];
        su = (su > -2.5f && su < 2.5f) ? su : ru;
        sv = (sv > -2.5f && sv < 2.5f) ? sv : rv;
      }
      a.pose[b * 3 + 0] = su; a.pose[b * 3 + 1] = sv; a.pose[b * 3 + 2] = th;
      float* tr = a.trace_out + (size_t)b * a.trace_stride;
      tr[0] = su; tr[1] = sv; tr[2] = th;
    }
  }
  if (a.coef && lane == 0)
    lm_coefficients(a.next, su, sv, th, a.R_FL ? a.R_FL + (size_t)b * 9 : nullptr, a.T_FL ? a.T_FL + (size_t)b * 3 : nullptr,
                    a.coef + (size_t)b * COEF_N);
}

// stand-alone launch: the coefficients of step 0 (part == null)
__global__ __launch_bounds__(64) void lm_solve(SolveArgs a) { lm_solve_body<false>(a, blockIdx.x, threadIdx.x); }

struct AccumArgs {
  const void* sat;    // [B,A,A,C]  F elements (fp32, or bf16 / fp16 in the reduced-precision inference modes)
  const void* grd;    // [B,h,w,C]
  const float* conf;  // [B,h,w] or null
  const float* xyz;   // [h,w,3]
  const double* coef; // [B,COEF_N]
  double* part;       // [B,nt,PART_N]
  int A, h, w, row0, npix, TP, nt, B, xcd_affine;
  int hs, rskip;      // stored rows of grd/conf (h - grd_row_skip) and the skip itself
  const unsigned char* keep;   // dropout: [npix] of this step, 1 = pixel takes part; or null
  unsigned* ticket;   // [B] arrival counters of this step (zeroed before the loop): the LAST tile of a sample closes the step
};

// channel pair ep (compile-time after unrolling) of a 16-byte vector of F, as two fp32 in ADJACENT registers: the gather loop is
// written on such pairs so that it compiles to v_pk_{mul,add,fma}_f32 without operand shuffling.  (Left to itself the SLP
// vectoriser paired unrelated scalars: a third of the loop was v_mov_b32, and the kernel is VALU-issue-bound -- 80 % VALU busy.)
typedef float lm_f2 __attribute__((ext_vector_type(2)));
template <typename F> __device__ __forceinline__ lm_f2 lm_pair(const uint4& v, int ep);
template <> __device__ __forceinline__ lm_f2 lm_pair<float>(const uint4& v, int ep) {
  lm_f2 r;
  r.x = __uint_as_float(ep == 0 ? v.x : v.z);
  r.y = __uint_as_float(ep == 0 ? v.y : v.w);
  return r;
}
template <> __device__ __forceinline__ lm_f2 lm_pair<__bf16>(const uint4& v, int ep) {      // bf16 -> fp32 is a 16-bit shift
  const unsigned w = ep == 0 ? v.x : ep == 1 ? v.y : ep == 2 ? v.z : v.w;
  lm_f2 r;
  r.x = __uint_as_float(w << 16);
  r.y = __uint_as_float(w & 0xffff0000u);
  return r;
}
template <> __device__ __forceinline__ lm_f2 lm_pair<_Float16>(const uint4& v, int ep) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const unsigned w = ep == 0 ? v.x : ep == 1 ? v.y : ep == 2 ? v.z : v.w;
  const h2 h = __builtin_bit_cast(h2, w);
  lm_f2 r;
  r.x = (float)h[0];
  r.y = (float)h[1];
  return r;
}

// The last tile of a sample to finish (an arrival ticket per sample) reduces the sample's tile partials in fixed order and
// closes the step: no separate solve launch.  Publication follows the write-through recipe of cdna_hip_programming.md
// Guideline 16: 8-byte agent-scope (sc1) stores of the partials -> s_waitcnt vmcnt(0) -> relaxed agent-scope ticket; the
// last arriver reads with agent-scope loads.  Which tile arrives last never changes a bit of the result.
// (6 waves per SIMD = 80 registers: what the gather loop needs; the closing solve, one wave per sample, may spill a little)
// (measured: 3, 4 or 8 waves per SIMD with 1-4 pixels in flight per lane change nothing -- the kernel waits on its dependent taps)
#define LM_OCC 6
#define LM_UNROLL(F) (sizeof(F) == 4 ? 2 : 1)
// Pixels per block of the forward accumulate kernels (level-dependent only, like lm_pick_tile: a sample's partial-sum grouping
// must not depend on its batch mates).  Larger than the backward's tiles: a block's fixed cost -- the fp64 pixel set-up, the
// reductions, the published partials, the ticket -- is what these kernels spend their time on.
#define FWD_MAX_TP 512
// (measured on one box against 256 / 128 / 64: lm_accum<64> 72.3 -> 70.8 us, <128> 45.3 -> 41.7, <256> 30.8 -> 29.5; with 512 / 512 / 256
// the two coarse levels lose 10-30 %)
static inline int lm_pick_tile_fwd(int npix) { return npix >= 16384 ? 512 : (npix >= 4096 ? 256 : 128); }
template <int C, bool USE_W, typename F>
__global__ __launch_bounds__(256, LM_OCC) void lm_accum(AccumArgs a, SolveArgs sa) {
  __shared__ PixParam pp[FWD_MAX_TP];
  __shared__ float red[4][14];
  __shared__ double redd[14];
  int b, tile;
  if (!lm_block_map(a.xcd_affine, a.nt, a.B, b, tile)) return;
  const int t = threadIdx.x;
  const int p0 = tile * a.TP;
  const int np = min(a.TP, a.npix - p0);
  const double* cf = a.coef + (size_t)b * COEF_N;

  for (int tt = t; tt < np; tt += 256) {
    const int p = p0 + tt;
    const int r = a.row0 + p / a.w, c = p % a.w;
    const float cw = USE_W ? a.conf[((size_t)b * a.hs + (r - a.rskip)) * a.w + c] : 1.f;
    PixParam P = lm_pixel<C>(cf, a.xyz + ((size_t)r * a.w + c) * 3, a.A, cw);
    if (a.keep && !a.keep[p]) {          // dropped by args.dropout: the pixel leaves every sum (models_kitti.py:968-974)
      P.wx0 = P.wx1 = P.wy0 = P.wy1 = 0.f; P.off = P.dxo = P.dyo = 0; P.j2u = P.j2v = 0.f; P.gm = P.wt = P.m = 0.f;
    }
    pp[tt] = P;
  }
  __syncthreads();

  const float j0u = (float)cf[8], j0v = (float)cf[9], j1u = (float)cf[10], j1v = (float)cf[11];
  constexpr int EPL = 16 / (int)sizeof(F);   // channels per lane (16 B)
  constexpr int LPP = C / EPL;        // lanes per pixel
  constexpr int PPW = 64 / LPP;       // pixels per wave-iteration
  const int lane = t & 63, wave = t >> 6;
  const int sub = lane / LPP, cl = (lane % LPP) * EPL;
  const F* satb = (const F*)a.sat + (size_t)b * a.A * a.A * C + cl;
  const F* grdb = (const F*)a.grd + ((size_t)b * a.hs * a.w + (size_t)(a.row0 - a.rskip) * a.w + p0) * C + cl;

  // With J_a = dsx*a_u + dsy*a_v (a = u, v, theta) and only the theta row's (a_u, a_v) = (j2u, j2v) varying per pixel, every
  // normal-equation sum is a combination of CHANNEL sums of a pixel -- Sxx = sum dsx^2, Sxy, Syy, Sxs = sum dsx*s, Sys, Sxg, Syg --
  // with per-pixel (j2u, j2v, weight) or per-sample (j0*, j1*) coefficients.  So the inner loop only forms those nine channel
  // sums (9 FMAs per element instead of 6 for the J's + 14), the per-pixel step folds in what varies per pixel, and the
  // per-sample coefficients are applied once per block, in fp64, after the reduction:
  //   T1..3 = sum w (Sxx, Sxy, Syy)        B1 = sum w (j2u Sxx + j2v Sxy)   B2 = sum w (j2u Sxy + j2v Syy)
  //   Q = sum (j2u b1 + j2v b2) = H22      U1,2 = sum w (Sxs, Sys)   U3 = sum w (j2u Sxs + j2v Sys)   V likewise with g
  float aS = 0, aG = 0, T1 = 0, T2 = 0, T3 = 0, B1 = 0, B2 = 0, Q = 0, U1 = 0, U2 = 0, U3 = 0, V1 = 0, V2 = 0, V3 = 0;

  constexpr int UNR = LM_UNROLL(F);
#pragma unroll UNR
  // (measured with the gather loop compiled out: a launch's FIXED cost -- the fp64 pixel set-up, the reductions, the published
  // partials, the ticket round trip, the closing solve -- is 31 / 20 / 13 us of the 73 / 43 / 30 us at C = 64 / 128 / 256)
  for (int i = wave * PPW + sub; i < np; i += 4 * PPW) {
    const PixParam P = pp[i];
    const uint4 t00 = *(const uint4*)(satb + P.off);
    const uint4 t01 = *(const uint4*)(satb + P.off + P.dxo);
    const uint4 t10 = *(const uint4*)(satb + P.off + P.dyo);
    const uint4 t11 = *(const uint4*)(satb + P.off + P.dyo + P.dxo);
    const uint4 gg = *(const uint4*)(grdb + (size_t)i * C);
    lm_f2 xx = {0.f, 0.f}, xy = xx, yy = xx, xs = xx, ys = xx, xg = xx, yg = xx, ss2 = xx, gg2 = xx;
#pragma unroll
    for (int ep = 0; ep < EPL / 2; ++ep) {
      const lm_f2 c00 = lm_pair<F>(t00, ep), c01 = lm_pair<F>(t01, ep), c10 = lm_pair<F>(t10, ep), c11 = lm_pair<F>(t11, ep);
      const lm_f2 top = P.wx0 * c00 + P.wx1 * c01;
      const lm_f2 bot = P.wx0 * c10 + P.wx1 * c11;
      const lm_f2 s = P.wy0 * top + P.wy1 * bot;
      const lm_f2 dsy = bot - top;
      const lm_f2 dsx = P.wy0 * (c01 - c00) + P.wy1 * (c11 - c10);
      const lm_f2 g = lm_pair<F>(gg, ep) * P.gm;
      xx += dsx * dsx; xy += dsx * dsy; yy += dsy * dsy;
      xs += dsx * s; ys += dsy * s; xg += dsx * g; yg += dsy * g;
      ss2 += s * s; gg2 += g * g;
    }
    float Sxx = xx.x + xx.y, Sxy = xy.x + xy.y, Syy = yy.x + yy.y, Sxs = xs.x + xs.y, Sys = ys.x + ys.y;
    float Sxg = xg.x + xg.y, Syg = yg.x + yg.y;
    const float Sss = ss2.x + ss2.y, Sgg = gg2.x + gg2.y;
    aS += Sss; aG += Sgg;
    if (USE_W) { Sxx *= P.wt; Sxy *= P.wt; Syy *= P.wt; Sxs *= P.wt; Sys *= P.wt; Sxg *= P.wt; Syg *= P.wt; }
    T1 += Sxx; T2 += Sxy; T3 += Syy;
    const float b1 = P.j2u * Sxx + P.j2v * Sxy, b2 = P.j2u * Sxy + P.j2v * Syy;
    B1 += b1; B2 += b2;
    Q += P.j2u * b1 + P.j2v * b2;
    U1 += Sxs; U2 += Sys; U3 += P.j2u * Sxs + P.j2v * Sys;
    V1 += Sxg; V2 += Syg; V3 += P.j2u * Sxg + P.j2v * Syg;
  }

  float acc[14] = {aS, aG, T1, T2, T3, B1, B2, Q, U1, U2, U3, V1, V2, V3};
#pragma unroll
  for (int k = 0; k < 14; ++k) acc[k] = wave_sum_f32(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 14; ++k) red[wave][k] = acc[k];
  }
  __syncthreads();
  if (t < 14) redd[t] = ((double)red[0][t] + (double)red[1][t]) + ((double)red[2][t] + (double)red[3][t]);
  __syncthreads();
  if (wave != 0) return;
  if (t < PART_N) {
    // the per-sample rows of d(uv)/d(pose): fp32 values (the gather used to apply them in fp32), combined in fp64
    const double au = (double)j0u, av = (double)j0v, bu = (double)j1u, bv = (double)j1v;
    const double t1 = redd[2], t2 = redd[3], t3 = redd[4], b1 = redd[5], b2 = redd[6];
    double v = 0.0;
    switch (t) {
      case 0: v = redd[0]; break;                                             // sum s^2
      case 1: v = redd[1]; break;                                             // sum g^2
      case 2: v = au * au * t1 + 2.0 * au * av * t2 + av * av * t3; break;    // H00
      case 3: v = au * bu * t1 + (au * bv + av * bu) * t2 + av * bv * t3; break;   // H01
      case 4: v = au * b1 + av * b2; break;                                   // H02
      case 5: v = bu * bu * t1 + 2.0 * bu * bv * t2 + bv * bv * t3; break;    // H11
      case 6: v = bu * b1 + bv * b2; break;                                   // H12
      case 7: v = redd[7]; break;                                             // H22
      case 8: v = au * redd[8] + av * redd[9]; break;                         // J^T W s
      case 9: v = bu * redd[8] + bv * redd[9]; break;
      case 10: v = redd[10]; break;
      case 11: v = au * redd[11] + av * redd[12]; break;                      // J^T W g
      case 12: v = bu * redd[11] + bv * redd[12]; break;
      case 13: v = redd[13]; break;
      default: break;
    }
    __hip_atomic_store(a.part + ((size_t)b * a.nt + tile) * PART_N + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const unsigned old = lm_draw_ticket(a.ticket + b, lane);      // (publication recipe: lm_common.h)
  if (old + 1 != (unsigned)a.nt) return;
  lm_solve_body<true>(sa, b, lane);
}

// ---------------------------------------------------------------------------------------------
// hla_s2g_config.count_in_view: the quantity jacobian.py:172 asserts on -- how many pixels of the WHOLE level map (all rows,
// whatever their z > 0 mask) have satellite coordinates inside the map.  Geometry only; one thread per pixel.
__global__ __launch_bounds__(256) void lm_inview_kernel(const double* __restrict__ coef, const float* __restrict__ xyz, int A,
                                                        int npix, int* __restrict__ count) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const double* cf = coef + (size_t)b * COEF_N;
  int in = 0;
  if (p < npix) {
    const double X = xyz[(size_t)p * 3], Y = xyz[(size_t)p * 3 + 1], Z = xyz[(size_t)p * 3 + 2];
    const double u = cf[0] * X + cf[1] * Y + cf[2] * Z + cf[3];
    const double v = cf[4] * X + cf[5] * Y + cf[6] * Z + cf[7];
    const double lim = (double)(A - 1);
    in = ((u >= 0.0) && (u <= lim) && (v >= 0.0) && (v <= lim)) ? 1 : 0;
  }
  const unsigned long long m = __ballot(in);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count + b, __popcll(m));
}

// ---------------------------------------------------------------------------------------------
static size_t ws_layout(const hla_s2g_config* cfg, const hla_s2g_level* lv, int B, size_t* off_coef, size_t* off_pose,
                        size_t* off_part) {
  int max_nt = 1;
  for (int l = 0; l < cfg->n_levels; ++l) {
    const int npix = (lv[l].h - lv[l].row0) * lv[l].w;
    const int tp = lm_pick_tile_fwd(npix);
    max_nt = max(max_nt, (npix + tp - 1) / tp);
  }
  size_t o = 0;
  *off_coef = o; o += hla_align_up((size_t)B * COEF_N * sizeof(double), 256);
  *off_pose = o; o += hla_align_up((size_t)B * 3 * sizeof(float), 256);
  *off_part = o; o += hla_align_up((size_t)B * max_nt * PART_N * sizeof(double), 256);
  o += hla_align_up((size_t)B * cfg->n_levels * cfg->n_iters * sizeof(unsigned), 256);   // arrival tickets, one per step and sample
  o += hla_align_up((size_t)B * cfg->n_levels * cfg->n_iters * sizeof(int), 256);   // in-view counts (count_in_view)
  o += hla_align_up((size_t)B * 6 * sizeof(double), 256);      // ADAM moments (last region)
  return o;
}

extern "C" size_t hla_s2g_workspace_bytes(const hla_s2g_config* cfg, const hla_s2g_level* levels, int B) {
  size_t a, b, c;
  return ws_layout(cfg, levels, B, &a, &b, &c);
}

template <bool W, typename F>
static void launch_accum_f(int C, dim3 grid, hipStream_t st, const AccumArgs& a, const SolveArgs& sa) {
  switch (C) {
    case 256: hipLaunchKernelGGL((lm_accum<256, W, F>), grid, dim3(256), 0, st, a, sa); break;
    case 128: hipLaunchKernelGGL((lm_accum<128, W, F>), grid, dim3(256), 0, st, a, sa); break;
    case 64: hipLaunchKernelGGL((lm_accum<64, W, F>), grid, dim3(256), 0, st, a, sa); break;
    case 16: hipLaunchKernelGGL((lm_accum<16, W, F>), grid, dim3(256), 0, st, a, sa); break;
  }
}
template <bool W>
static void launch_accum(int C, int feat_dtype, dim3 grid, hipStream_t st, const AccumArgs& a, const SolveArgs& sa) {
  if (feat_dtype == HLA_BF16) launch_accum_f<W, __bf16>(C, grid, st, a, sa);
  else if (feat_dtype == HLA_F16) launch_accum_f<W, _Float16>(C, grid, st, a, sa);
  else launch_accum_f<W, float>(C, grid, st, a, sa);
}

int hla_s2g_validate(const char* who, const hla_s2g_config* cfg, const hla_s2g_level* lv, const float* R_FL,
                     const float* T_FL, int B) {
  HLA_REQUIRE(cfg && lv, "%s: null argument", who);
  HLA_REQUIRE(B > 0 && cfg->n_levels >= 1 && cfg->n_levels <= 4 && cfg->n_iters >= 1, "%s: bad sizes", who);
  HLA_REQUIRE(cfg->dof >= 1 && cfg->dof <= 3, "%s: dof must be 1..3", who);
  HLA_REQUIRE(!cfg->ford || (R_FL && T_FL), "%s: Ford mode needs R_FL and T_FL", who);
  for (int l = 0; l < cfg->n_levels; ++l) {
    const int C = lv[l].C;
    HLA_REQUIRE(C == 256 || C == 128 || C == 64 || C == 16, "%s: unsupported channel count %d", who, C);
    HLA_REQUIRE(lv[l].feat_dtype == HLA_F32 || lv[l].feat_dtype == HLA_BF16 || lv[l].feat_dtype == HLA_F16,
                "%s: level %d feat_dtype must be HLA_F32, HLA_BF16 or HLA_F16", who, l);
    HLA_REQUIRE(lv[l].sat_feat && lv[l].grd_feat && lv[l].xyz, "%s: level %d has null maps", who, l);
    HLA_REQUIRE(!cfg->using_weight || lv[l].grd_conf, "%s: using_weight needs grd_conf", who);
    HLA_REQUIRE(lv[l].row0 >= 0 && lv[l].row0 < lv[l].h, "%s: bad row0", who);
    HLA_REQUIRE(lv[l].grd_row_skip >= 0 && lv[l].grd_row_skip <= lv[l].row0, "%s: grd_row_skip must be in [0,row0]", who);
    HLA_REQUIRE((size_t)lv[l].A * lv[l].A * C < (1u << 31), "%s: satellite map too large", who);
  }
  return HLA_OK;
}

extern "C" int hla_s2g_lm_solve(const hla_s2g_config* cfg, const hla_s2g_level* lv, const float* R_FL,
                                const float* T_FL, const float* pose0, const float* rand_uv, float* trace,
                                double* normal_eq, void* workspace, size_t workspace_bytes, int B,
                                hla_stream_t stream) {
  HLA_REQUIRE(trace && workspace, "hla_s2g_lm_solve: null argument");
  const int rc = hla_s2g_validate("hla_s2g_lm_solve", cfg, lv, R_FL, T_FL, B);
  if (rc) return rc;
  HLA_REQUIRE(cfg->optimizer >= 0 && cfg->optimizer <= 3, "hla_s2g_lm_solve: optimizer must be 0 (LM), 1 (SGD), 2 (ADAM) or 3 (GN)");
  HLA_REQUIRE(cfg->optimizer == 0 || !cfg->level_first, "hla_s2g_lm_solve: SGD / ADAM / GN exist for the iteration-first loop only");
  HLA_REQUIRE(cfg->optimizer != 3 || cfg->ford, "hla_s2g_lm_solve: GN_update exists in the Ford model only");
  const bool newton = cfg->optimizer == 0 || cfg->optimizer == 3;
  const bool reinit = (cfg->ford || cfg->dof == 3) && newton;     // SGD_update / ADAM_update never re-initialise
  HLA_REQUIRE(!reinit || rand_uv, "hla_s2g_lm_solve: rand_uv required");
  size_t oc, op, opart;
  const size_t need = ws_layout(cfg, lv, B, &oc, &op, &opart);
  if (workspace_bytes < need) {
    hla_set_error("hla_s2g_lm_solve: workspace %zu < %zu", workspace_bytes, need);
    return HLA_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  double* coef = (double*)(ws + oc);
  float* pose = (float*)(ws + op);
  double* part = (double*)(ws + opart);

  if (pose0) HLA_CHECK_HIP(hipMemcpyAsync(pose, pose0, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
  // (otherwise the init launch below starts the pose at zero; it also clears the arrival counters: two memset launches less)

  const int L = cfg->n_levels, N = cfg->n_iters, steps = L * N;
  auto step_level = [&](int k) { return cfg->level_first ? k / N : k % L; };
  auto step_iter = [&](int k) { return cfg->level_first ? k % N : k / L; };
  auto geom = [&](int l) {
    LmGeom g{};
    g.ford = cfg->ford; g.lat = cfg->shift_range_lat; g.lon = cfg->shift_range_lon; g.rot = cfg->rotation_range;
    g.mpp = lv[l].meter_per_pixel; g.ctr = lv[l].centre;
    return g;
  };

  double* adam = (double*)(ws + need - hla_align_up((size_t)B * 6 * sizeof(double), 256));
  int* in_view = (int*)((char*)adam - hla_align_up((size_t)B * steps * sizeof(int), 256));
  unsigned* ticket = (unsigned*)((char*)in_view - hla_align_up((size_t)B * steps * sizeof(unsigned), 256));
  const bool count = cfg->count_in_view && normal_eq;
  if (count) HLA_CHECK_HIP(hipMemsetAsync(in_view, 0, (size_t)B * steps * sizeof(int), st));
  if (cfg->optimizer == 2) HLA_CHECK_HIP(hipMemsetAsync(adam, 0, (size_t)B * 6 * sizeof(double), st));
  SolveArgs sa{};
  sa.optimizer = cfg->optimizer; sa.beta1 = cfg->beta1; sa.beta2 = cfg->beta2; sa.adam = adam;
  sa.pose = pose; sa.B = B; sa.reinit = reinit ? 1 : 0; sa.R_FL = R_FL; sa.T_FL = T_FL;
  sa.cfg.gn = cfg->optimizer == 3 ? 1 : 0;
  sa.cfg.dof = cfg->dof; sa.cfg.use_hessian = sa.cfg.gn ? 0 : cfg->use_hessian;
  for (int i = 0; i < 3; ++i) sa.cfg.lam[i] = sa.cfg.gn ? 0.0 : cfg->damping[i];     // GN_update: delta = -H^-1 J^T W r
  sa.coef = coef;

  // init launch: coefficients of step 0
  sa.part = nullptr; sa.next = geom(step_level(0));
  sa.zero_ticket = ticket; sa.zero_steps = steps; sa.zero_pose = pose0 ? 0 : 1;
  hipLaunchKernelGGL(lm_solve, dim3(B), dim3(64), 0, st, sa);
  sa.zero_ticket = nullptr; sa.zero_steps = 0; sa.zero_pose = 0;

  for (int k = 0; k < steps; ++k) {
    const int l = step_level(k), it = step_iter(k);
    const hla_s2g_level& v = lv[l];
    if (count)
      hipLaunchKernelGGL(lm_inview_kernel, dim3((v.h * v.w + 255) / 256, B), dim3(256), 0, st, coef, v.xyz, v.A, v.h * v.w,
                         in_view + (size_t)k * B);
    AccumArgs aa{};
    aa.sat = v.sat_feat; aa.grd = v.grd_feat; aa.conf = v.grd_conf; aa.xyz = v.xyz; aa.coef = coef; aa.part = part;
    aa.A = v.A; aa.h = v.h; aa.w = v.w; aa.row0 = v.row0; aa.npix = (v.h - v.row0) * v.w;
    aa.hs = v.h - v.grd_row_skip; aa.rskip = v.grd_row_skip;
    aa.keep = cfg->keep ? cfg->keep + (size_t)k * cfg->keep_stride : nullptr;
    aa.TP = lm_pick_tile_fwd(aa.npix); aa.nt = (aa.npix + aa.TP - 1) / aa.TP; aa.B = B;
    aa.xcd_affine = (B >= 8) ? 1 : 0;
    const int nblk = aa.xcd_affine ? 8 * ((B + 7) / 8) * aa.nt : B * aa.nt;
    aa.ticket = ticket + (size_t)k * B;
    // the step's closing solve runs inside the same launch (last tile of each sample): its arguments
    sa.t = k;
    sa.part = part; sa.nt = aa.nt; sa.sat_inv = v.sat_inv_norm; sa.grd_inv = v.grd_inv_norm;
    sa.trace_out = trace + ((size_t)it * L + l) * 3; sa.trace_stride = N * L * 3;
    sa.rand_uv = reinit ? rand_uv + (size_t)k * 2 * B : nullptr;
    sa.normal_eq = normal_eq ? normal_eq + (size_t)k * B * 16 : nullptr;
    sa.in_view = count ? in_view + (size_t)k * B : nullptr;
    if (k + 1 < steps) { sa.coef = coef; sa.next = geom(step_level(k + 1)); }
    else sa.coef = nullptr;
    const double esz = v.feat_dtype == HLA_F32 ? 4.0 : 2.0;
    hla_prof_begin(v.C == 256 ? K_LM256 : v.C == 128 ? K_LM128 : v.C == 64 ? K_LM64 : K_LM16, 0,
                   (double)B * ((double)v.A * v.A + (double)aa.npix) * v.C * esz, st);
    if (cfg->using_weight && newton) launch_accum<true>(v.C, v.feat_dtype, dim3(nblk), st, aa, sa);   // SGD / ADAM ignore the confidence
    else launch_accum<false>(v.C, v.feat_dtype, dim3(nblk), st, aa, sa);
    hla_prof_end(st);
  }
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}
