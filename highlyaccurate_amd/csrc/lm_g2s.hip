// The LM pose loop in the ground -> satellite direction (LM_G2SP, models_kitti.py:22-499, proj == 'geo';
// SURVEY 8(f).2).  Same fusion as lm_solve.hip, roles swapped: for every pixel of the A x A satellite map the ground
// FEATURE map is sampled where that pixel projects to in the camera (get_warp_sat2real 53-84, seq_warp_real2camera
// 86-161: a perspective projection with the per-sample intrinsics, so all three Jacobian columns vary per pixel),
// and LM_update (333-379) uses the residual and Jacobian as they are: no renormalisation, no re-initialisation,
// always 3-DoF, optional weight = the PROJECTED ground confidence.
//   g2s_accum<C>  grid (tiles x samples), lanes spread over channels: 12 sums  H(6), J'W f (3), J'W s (3)
//   g2s_solve     one wave per sample: fixed-order fp64 reduction, damped 3x3 solve, next step's 21 coefficients
#include "lm_common.h"

#define G2S_COEF_N 24   // a[3] b[3] c[3]  (uv1_k = a_k*v_c + b_k*u_c + c_k)   dx[3] dy[3]   dth_a[3] dth_b[3]   pad[3]

struct __attribute__((aligned(16))) G2sPix {
  int off, dxo, dyo;            // element offsets of the NW tap and the +x / +y neighbours in the ground map
  float wx0, wx1, wy0, wy1;     // clamped-corner bilinear weights x in-bounds mask (jacobian.py:146-177)
  float j0u, j0v, j1u, j1v, j2u, j2v;   // d(uv)/d(shift_u, shift_v, heading) at this pixel (0 where z <= 1e-6)
  float wt;                     // LM weight: projected confidence (or 1)
  float pad0, pad1;
};

struct G2sAccumArgs {
  const float* src;   // ground feature map [B,h,w,C] (gathered)
  const float* fix;   // satellite feature map [B,A,A,C]
  const float* conf;  // ground confidence [B,h,w] or null
  const double* coef; // [B,G2S_COEF_N]
  double* part;       // [B,nt,PART_N]
  int A, h, w, ctr, npix, TP, nt, B, xcd_affine;
};

template <int C, bool USE_W>
__device__ __forceinline__ G2sPix g2s_pixel(const double* cf, int row, int col, int ctr, int h, int w, const float* conf) {
  const double vc = (double)(row - ctr), uc = (double)(col - ctr);
  const double q0 = cf[0] * vc + cf[3] * uc + cf[6];
  const double q1 = cf[1] * vc + cf[4] * uc + cf[7];
  const double q2 = cf[2] * vc + cf[5] * uc + cf[8];
  const double z = fmax(q2, 1e-6);                                         // models_kitti.py:121-124
  const bool front = q2 > 1e-6;
  const double iz = 1.0 / z, u = q0 * iz, v = q1 * iz;
  const double limx = (double)(w - 1), limy = (double)(h - 1);
  const bool inb = (u >= 0.0) && (u <= limx) && (v >= 0.0) && (v <= limy);  // jacobian.py:168-170
  G2sPix o;
  o.pad0 = o.pad1 = 0.f;
  if (inb) {
    const double x0 = floor(u), y0 = floor(v);
    const double x1 = fmin(x0 + 1.0, limx), y1 = fmin(y0 + 1.0, limy);
    o.wx0 = (float)(x1 - u); o.wx1 = (float)(u - x0);
    o.wy0 = (float)(y1 - v); o.wy1 = (float)(v - y0);
    const int ix0 = (int)x0, iy0 = (int)y0;
    o.off = (iy0 * w + ix0) * C;
    o.dxo = ((int)x1 - ix0) * C;
    o.dyo = ((int)y1 - iy0) * w * C;
    o.wt = 1.f;
    if (USE_W) {                 // grd_conf_proj = grid_sample(grd_c, uv) (models_kitti.py:298-299)
      const float* cp = conf + (size_t)iy0 * w + ix0;
      const int dx = (int)x1 - ix0, dy = ((int)y1 - iy0) * w;
      o.wt = o.wy0 * (o.wx0 * cp[0] + o.wx1 * cp[dx]) + o.wy1 * (o.wx0 * cp[dy] + o.wx1 * cp[dy + dx]);
    }
    if (front) {                 // duv = d1[0:2]/z - uv1[0:2]*d1[2]/z^2, zero where z <= 1e-6 (143-149)
      const double iz2 = iz * iz;
      const double tx0 = cf[9], tx1 = cf[10], tx2 = cf[11], ty0 = cf[12], ty1 = cf[13], ty2 = cf[14];
      const double tt0 = cf[15] * vc + cf[18] * uc, tt1 = cf[16] * vc + cf[19] * uc, tt2 = cf[17] * vc + cf[20] * uc;
      o.j0u = (float)(tx0 * iz - q0 * tx2 * iz2); o.j0v = (float)(tx1 * iz - q1 * tx2 * iz2);
      o.j1u = (float)(ty0 * iz - q0 * ty2 * iz2); o.j1v = (float)(ty1 * iz - q1 * ty2 * iz2);
      o.j2u = (float)(tt0 * iz - q0 * tt2 * iz2); o.j2v = (float)(tt1 * iz - q1 * tt2 * iz2);
    } else {
      o.j0u = o.j0v = o.j1u = o.j1v = o.j2u = o.j2v = 0.f;
    }
  } else {
    o.wx0 = o.wx1 = o.wy0 = o.wy1 = 0.f;
    o.off = o.dxo = o.dyo = 0;
    o.j0u = o.j0v = o.j1u = o.j1v = o.j2u = o.j2v = 0.f;
    o.wt = USE_W ? 0.f : 1.f;
  }
  return o;
}

template <int C, bool USE_W>
__global__ __launch_bounds__(256) void g2s_accum(G2sAccumArgs a) {
  __shared__ G2sPix pp[MAX_TP];
  __shared__ float red[4][12];
  int b, tile;
  if (!lm_block_map(a.xcd_affine, a.nt, a.B, b, tile)) return;
  const int t = threadIdx.x;
  const int p0 = tile * a.TP;
  const int np = min(a.TP, a.npix - p0);
  const double* cf = a.coef + (size_t)b * G2S_COEF_N;
  if (t < np) {
    const int p = p0 + t;
    pp[t] = g2s_pixel<C, USE_W>(cf, p / a.A, p % a.A, a.ctr, a.h, a.w, USE_W ? a.conf + (size_t)b * a.h * a.w : nullptr);
  }
  __syncthreads();

  constexpr int LPP = C / 4, PPW = 64 / LPP;
  const int lane = t & 63, wave = t >> 6;
  const int sub = lane / LPP, cl = (lane % LPP) * 4;
  const float* srcb = a.src + (size_t)b * a.h * a.w * C + cl;
  const float* fixb = a.fix + ((size_t)b * a.npix + p0) * C + cl;
  float h00 = 0, h01 = 0, h02 = 0, h11 = 0, h12 = 0, h22 = 0, u0 = 0, u1 = 0, u2 = 0, v0 = 0, v1 = 0, v2 = 0;
#pragma unroll 2
  for (int i = wave * PPW + sub; i < np; i += 4 * PPW) {
    const G2sPix P = pp[i];
    const float4 t00 = *(const float4*)(srcb + P.off);
    const float4 t01 = *(const float4*)(srcb + P.off + P.dxo);
    const float4 t10 = *(const float4*)(srcb + P.off + P.dyo);
    const float4 t11 = *(const float4*)(srcb + P.off + P.dyo + P.dxo);
    const float4 gg = *(const float4*)(fixb + (size_t)i * C);
    const float a00[4] = {t00.x, t00.y, t00.z, t00.w}, a01[4] = {t01.x, t01.y, t01.z, t01.w};
    const float a10[4] = {t10.x, t10.y, t10.z, t10.w}, a11[4] = {t11.x, t11.y, t11.z, t11.w};
    const float ag[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float top = P.wx0 * a00[e] + P.wx1 * a01[e];
      const float bot = P.wx0 * a10[e] + P.wx1 * a11[e];
      const float f = P.wy0 * top + P.wy1 * bot;                       // projected ground feature
      const float dsy = bot - top;
      const float dsx = P.wy0 * (a01[e] - a00[e]) + P.wy1 * (a11[e] - a10[e]);
      const float J0 = dsx * P.j0u + dsy * P.j0v, J1 = dsx * P.j1u + dsy * P.j1v, J2 = dsx * P.j2u + dsy * P.j2v;
      const float W0 = USE_W ? J0 * P.wt : J0, W1 = USE_W ? J1 * P.wt : J1, W2 = USE_W ? J2 * P.wt : J2;
      h00 += W0 * J0; h01 += W0 * J1; h02 += W0 * J2; h11 += W1 * J1; h12 += W1 * J2; h22 += W2 * J2;
      u0 += W0 * f; u1 += W1 * f; u2 += W2 * f;
      v0 += W0 * ag[e]; v1 += W1 * ag[e]; v2 += W2 * ag[e];
    }
  }
  float acc[12] = {h00, h01, h02, h11, h12, h22, u0, u1, u2, v0, v1, v2};
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = wave_sum_f32(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) red[wave][k] = acc[k];
  }
  __syncthreads();
  if (t < PART_N) {
    double v = 0.0;
    if (t < 12) v = ((double)red[0][t] + (double)red[1][t]) + ((double)red[2][t] + (double)red[3][t]);
    a.part[((size_t)b * a.nt + tile) * PART_N + t] = v;
  }
}

// ---------------------------------------------------------------------------------------------
struct G2sGeom { double lat, lon, rot, mpp, sx, sy; };   // sx, sy: intrinsics scale to the level's ground map (111-113)

// pose -> the 21 coefficients of the level the next step runs on (models_kitti.py:91-141)
__device__ static inline void g2s_coefficients(const G2sGeom& G, double su, double sv, double th, const float* K9, double* cf) {
  const double k = G.rot / 180.0 * 3.14159265358979323846;
  const double ang = -th * k, c = cos(ang), s = sin(ang);
  double K[3][3];
  for (int j = 0; j < 3; ++j) { K[0][j] = (double)K9[j] * G.sx; K[1][j] = (double)K9[3 + j] * G.sy; K[2][j] = (double)K9[6 + j]; }
  const double T[3] = {sv * G.lat, 1.65, -su * G.lon};                 // utils.get_camera_height() = 1.65
  for (int r = 0; r < 3; ++r) {
    cf[r] = (K[r][0] * c + K[r][2] * s) * G.mpp;                       // P[:,0] = K (c,0,s)'   x X = mpp*v_c
    cf[3 + r] = (-K[r][0] * s + K[r][2] * c) * G.mpp;                  // P[:,2] = K (-s,0,c)'  x Z = mpp*u_c
    cf[6 + r] = K[r][0] * T[0] + K[r][1] * T[1] + K[r][2] * T[2];      // P[:,3] = K T
    cf[9 + r] = -G.lon * K[r][2];                                      // K dT/dx,  dT/dx = lon (0,0,-1)'
    cf[12 + r] = G.lat * K[r][0];                                      // K dT/dy,  dT/dy = lat (1,0,0)'
    cf[15 + r] = k * (K[r][0] * s - K[r][2] * c) * G.mpp;              // K dR[:,0], dR[:,0] = k (s,0,-c)'
    cf[18 + r] = k * (K[r][0] * c + K[r][2] * s) * G.mpp;              // K dR[:,2], dR[:,2] = k (c,0,s)'
  }
  cf[21] = cf[22] = cf[23] = 0.0;
}

struct G2sSolveArgs {
  const double* part; int nt;
  const double* src_inv;   // [B] or null: deferred L2_norm of the ground map (gathered)
  const double* fix_inv;   // [B] or null: ... of the satellite map
  float* pose; float* trace_out; int trace_stride;
  double* normal_eq; double* coef;
  const float* camera_k;   // [B,3,3]
  int B;
  LmSolveCfg cfg; G2sGeom next;
};

__global__ __launch_bounds__(64) void g2s_solve(G2sSolveArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float su = a.pose[b * 3 + 0], sv = a.pose[b * 3 + 1], th = a.pose[b * 3 + 2];
  if (a.part) {
    double s[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) s[k] = 0.0;
    for (int i = lane; i < a.nt; i += 64) {
      const double* p = a.part + ((size_t)b * a.nt + i) * PART_N;
#pragma unroll
      for (int k = 0; k < 12; ++k) s[2 + k] += p[k];
    }
#pragma unroll
    for (int k = 2; k < 14; ++k) s[k] = wave_sum_f64(s[k]);
    if (lane == 0) {
      const double af = a.src_inv ? a.src_inv[b] : 1.0, as = a.fix_inv ? a.fix_inv[b] : 1.0;
      for (int k = 2; k < 11; ++k) s[k] *= af * af;      // H and J'W f carry the ground map's scale twice
      for (int k = 11; k < 14; ++k) s[k] *= af * as;
      s[0] = 1.0; s[1] = 1.0;                             // no renormalisation (models_kitti.py:351-355): ||.|| := 1
      if (a.normal_eq) {
        for (int k = 0; k < 14; ++k) a.normal_eq[(size_t)b * 16 + k] = s[k];
        a.normal_eq[(size_t)b * 16 + 14] = 0.0; a.normal_eq[(size_t)b * 16 + 15] = 0.0;
      }
      double H[3][3], g[3], Mi[3][3], d[3], ns, ng;
      lm_solve_step(a.cfg, s, H, g, Mi, d, ns, ng);       // delta = -(H + lam I)^-1 J'W (f - s)
      su = (float)((double)su - d[0]);
      sv = (float)((double)sv - d[1]);
      th = (float)((double)th - d[2]);
      a.pose[b * 3 + 0] = su; a.pose[b * 3 + 1] = sv; a.pose[b * 3 + 2] = th;
      float* tr = a.trace_out + (size_t)b * a.trace_stride;
      tr[0] = su; tr[1] = sv; tr[2] = th;
    }
  }
  if (a.coef && lane == 0) g2s_coefficients(a.next, su, sv, th, a.camera_k + (size_t)b * 9, a.coef + (size_t)b * G2S_COEF_N);
}

// ---------------------------------------------------------------------------------------------
static size_t g2s_layout(const hla_s2g_config* cfg, const hla_s2g_level* lv, int B, size_t* oc, size_t* op, size_t* opart) {
  int max_nt = 1;
  for (int l = 0; l < cfg->n_levels; ++l) {
    const int npix = lv[l].A * lv[l].A, tp = lm_pick_tile(npix);
    max_nt = max(max_nt, (npix + tp - 1) / tp);
  }
  size_t o = 0;
  *oc = o; o += hla_align_up((size_t)B * G2S_COEF_N * sizeof(double), 256);
  *op = o; o += hla_align_up((size_t)B * 3 * sizeof(float), 256);
  *opart = o; o += hla_align_up((size_t)B * max_nt * PART_N * sizeof(double), 256);
  return o;
}

extern "C" size_t hla_g2s_workspace_bytes(const hla_s2g_config* cfg, const hla_s2g_level* levels, int B) {
  size_t a, b, c;
  return g2s_layout(cfg, levels, B, &a, &b, &c);
}

template <bool W>
static void launch_g2s(int C, dim3 grid, hipStream_t st, const G2sAccumArgs& a) {
  switch (C) {
    case 256: hipLaunchKernelGGL((g2s_accum<256, W>), grid, dim3(256), 0, st, a); break;
    case 128: hipLaunchKernelGGL((g2s_accum<128, W>), grid, dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL((g2s_accum<64, W>), grid, dim3(256), 0, st, a); break;
  }
}

extern "C" int hla_g2s_lm_solve(const hla_s2g_config* cfg, const hla_s2g_level* lv, const float* camera_k, int ori_h,
                                int ori_w, const float* pose0, float* trace, double* normal_eq, void* workspace,
                                size_t workspace_bytes, int B, hla_stream_t stream) {
  HLA_REQUIRE(cfg && lv && camera_k && trace && workspace, "hla_g2s_lm_solve: null argument");
  HLA_REQUIRE(B > 0 && cfg->n_levels >= 1 && cfg->n_levels <= 4 && cfg->n_iters >= 1 && ori_h > 0 && ori_w > 0,
              "hla_g2s_lm_solve: bad sizes");
  HLA_REQUIRE(!cfg->ford && cfg->dof == 3 && !cfg->level_first && !cfg->use_hessian,
              "hla_g2s_lm_solve: LM_G2SP is KITTI-only, 3-DoF, iteration-first, identity damping (models_kitti.py:333-379)");
  for (int l = 0; l < cfg->n_levels; ++l) {
    const int C = lv[l].C;
    HLA_REQUIRE(C == 256 || C == 128 || C == 64, "hla_g2s_lm_solve: unsupported channel count %d", C);
    HLA_REQUIRE(lv[l].sat_feat && lv[l].grd_feat, "hla_g2s_lm_solve: level %d has null maps", l);
    HLA_REQUIRE(lv[l].feat_dtype == HLA_F32, "hla_g2s_lm_solve: level %d: fp32 feature maps only", l);
    HLA_REQUIRE(!cfg->using_weight || lv[l].grd_conf, "hla_g2s_lm_solve: using_weight needs grd_conf");
    HLA_REQUIRE(lv[l].grd_row_skip == 0, "hla_g2s_lm_solve: the whole ground map is sampled (grd_row_skip must be 0)");
    HLA_REQUIRE((size_t)lv[l].h * lv[l].w * C < (1u << 31), "hla_g2s_lm_solve: ground map too large");
  }
  size_t oc, op, opart;
  const size_t need = g2s_layout(cfg, lv, B, &oc, &op, &opart);
  if (workspace_bytes < need) {
    hla_set_error("hla_g2s_lm_solve: workspace %zu < %zu", workspace_bytes, need);
    return HLA_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  double* coef = (double*)(ws + oc);
  float* pose = (float*)(ws + op);
  double* part = (double*)(ws + opart);
  if (pose0) HLA_CHECK_HIP(hipMemcpyAsync(pose, pose0, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
  else HLA_CHECK_HIP(hipMemsetAsync(pose, 0, (size_t)B * 3 * sizeof(float), st));

  const int L = cfg->n_levels, N = cfg->n_iters, steps = L * N;
  auto geom = [&](int l) {
    G2sGeom g{};
    g.lat = cfg->shift_range_lat; g.lon = cfg->shift_range_lon; g.rot = cfg->rotation_range;
    g.mpp = lv[l].meter_per_pixel; g.sx = (double)lv[l].w / ori_w; g.sy = (double)lv[l].h / ori_h;
    return g;
  };
  G2sSolveArgs sa{};
  sa.pose = pose; sa.B = B; sa.camera_k = camera_k; sa.cfg.dof = 3; sa.cfg.use_hessian = 0;
  for (int i = 0; i < 3; ++i) sa.cfg.lam[i] = cfg->damping[i];
  sa.coef = coef; sa.part = nullptr; sa.next = geom(0);
  hipLaunchKernelGGL(g2s_solve, dim3(B), dim3(64), 0, st, sa);
  for (int k = 0; k < steps; ++k) {
    const int l = k % L, it = k / L;
    const hla_s2g_level& v = lv[l];
    G2sAccumArgs aa{};
    aa.src = (const float*)v.grd_feat; aa.fix = (const float*)v.sat_feat; aa.conf = v.grd_conf; aa.coef = coef; aa.part = part;
    aa.A = v.A; aa.h = v.h; aa.w = v.w; aa.ctr = v.A / 2; aa.npix = v.A * v.A;
    aa.TP = lm_pick_tile(aa.npix); aa.nt = (aa.npix + aa.TP - 1) / aa.TP; aa.B = B;
    aa.xcd_affine = (B >= 8) ? 1 : 0;
    const int nblk = aa.xcd_affine ? 8 * ((B + 7) / 8) * aa.nt : B * aa.nt;
    hla_prof_begin(v.C == 256 ? K_LM256 : v.C == 128 ? K_LM128 : K_LM64, 0,
                   (double)B * ((double)v.h * v.w + (double)aa.npix) * v.C * 4.0, st);
    if (cfg->using_weight) launch_g2s<true>(v.C, dim3(nblk), st, aa);
    else launch_g2s<false>(v.C, dim3(nblk), st, aa);
    hla_prof_end(st);
    sa.part = part; sa.nt = aa.nt; sa.src_inv = v.grd_inv_norm; sa.fix_inv = v.sat_inv_norm;
    sa.trace_out = trace + ((size_t)it * L + l) * 3; sa.trace_stride = N * L * 3;
    sa.normal_eq = normal_eq ? normal_eq + (size_t)k * B * 16 : nullptr;
    if (k + 1 < steps) { sa.coef = coef; sa.next = geom((k + 1) % L); }
    else sa.coef = nullptr;
    hla_prof_begin(K_LMSOLVE, 0, (double)B * aa.nt * PART_N * 8.0, st);
    hipLaunchKernelGGL(g2s_solve, dim3(B), dim3(64), 0, st, sa);
    hla_prof_end(st);
  }
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}

// =============================================================================================
// Backward of the ground -> satellite loop (what autograd does through models_kitti.py:86-161, 163-303, 333-379 and
// jacobian.py:138-205 in this direction).  Same structure as lm_backward.hip: walk the steps in reverse; per step
//   g2s_bwd_solve  closes the later step (reduces its 21 coefficient adjoints and pulls them back through
//                  pose -> coefficients), converts d(loss)/d(pose_out) into adjoints of the 12 sums (+ d/d lambda),
//                  and rewrites this step's forward coefficients;
//   g2s_bwd_accum  recomputes the gather, forms the element adjoints, scatters d/d(ground map) (and d/d(ground conf))
//                  with merged fp32 atomics, adds d/d(satellite map) in place, and reduces the pixel adjoints through
//                  the perspective division into the 21 coefficient adjoints.
#define G2S_PART_N 24

struct G2sBwdAccumArgs {
  const float* src; const float* fix; const float* conf;
  const double* coef; const double* adj;
  const double* src_inv; const double* fix_inv;
  float* d_src; float* d_fix; float* d_conf;
  double* part;            // [B,nt,G2S_PART_N]
  int A, h, w, ctr, npix, TP, nt, B, xcd_affine;
};

template <int C, bool USE_W>
__global__ __launch_bounds__(256) void g2s_bwd_accum(G2sBwdAccumArgs a) {
  __shared__ G2sPix pp[MAX_TP];
  __shared__ float pixacc[MAX_TP][9];     // a_u a_v a_j0u a_j0v a_j1u a_j1v a_j2u a_j2v a_w
  __shared__ double red[4][21];
  int b, tile;
  if (!lm_block_map(a.xcd_affine, a.nt, a.B, b, tile)) return;
  const int t = threadIdx.x;
  const int p0 = tile * a.TP;
  const int np = min(a.TP, a.npix - p0);
  const double* cf = a.coef + (size_t)b * G2S_COEF_N;
  const float* confb = USE_W ? a.conf + (size_t)b * a.h * a.w : nullptr;
  if (t < np) {
    const int p = p0 + t;
    pp[t] = g2s_pixel<C, USE_W>(cf, p / a.A, p % a.A, a.ctr, a.h, a.w, confb);
  }
  __syncthreads();

  const double* ad = a.adj + (size_t)b * 16;
  const float A00 = (float)ad[2], A01 = (float)ad[3], A02 = (float)ad[4], A11 = (float)ad[5], A12 = (float)ad[6], A22 = (float)ad[7];
  const float gU0 = (float)ad[8], gU1 = (float)ad[9], gU2 = (float)ad[10];
  const float gV0 = (float)ad[11], gV1 = (float)ad[12], gV2 = (float)ad[13];
  const float af = a.src_inv ? (float)a.src_inv[b] : 1.f, as = a.fix_inv ? (float)a.fix_inv[b] : 1.f;
  constexpr int LPP = C / 4, PPW = 64 / LPP;
  const int lane = t & 63, wave = t >> 6;
  const int sub = lane / LPP, cl = lane % LPP;      // lane j owns channels j, j+LPP, j+2LPP, j+3LPP (line-coalesced atomics)
  const size_t src_base = (size_t)b * a.h * a.w * C + cl;
  const size_t fix_base = ((size_t)b * a.npix + p0) * C + cl;

  const int RUN = (np + 4 * PPW - 1) / (4 * PPW);
  const int grp = wave * PPW + sub;
  int cur_off = -1, cur_dxo = 0, cur_dyo = 0;
  float c00[4] = {0, 0, 0, 0}, c01[4] = {0, 0, 0, 0}, c10[4] = {0, 0, 0, 0}, c11[4] = {0, 0, 0, 0};
  auto flush_cell = [&]() {
    if (cur_off >= 0) {
      float* dp = a.d_src + src_base + cur_off;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        atomicAdd(dp + e * LPP, c00[e]);
        atomicAdd(dp + cur_dxo + e * LPP, c01[e]);
        atomicAdd(dp + cur_dyo + e * LPP, c10[e]);
        atomicAdd(dp + cur_dyo + cur_dxo + e * LPP, c11[e]);
      }
    }
  };
  for (int jr = 0; jr < RUN; ++jr) {
    const int i = grp * RUN + jr;
    const bool live = i < np;
    float q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
      const G2sPix P = pp[i];
      const float m = (P.wx0 + P.wx1 + P.wy0 + P.wy1) != 0.f ? 1.f : 0.f;     // in-bounds
      const float* sp = a.src + src_base + P.off;
      const float* gq = a.fix + fix_base + (size_t)i * C;
      float v00[4], v01[4], v10[4], v11[4], vg[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v00[e] = sp[e * LPP]; v01[e] = sp[P.dxo + e * LPP];
        v10[e] = sp[P.dyo + e * LPP]; v11[e] = sp[P.dyo + P.dxo + e * LPP];
        vg[e] = gq[e * LPP];
      }
      float d00[4], d01[4], d10[4], d11[4], dg[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float V00 = v00[e] * af, V01 = v01[e] * af, V10 = v10[e] * af, V11 = v11[e] * af;
        const float top = P.wx0 * V00 + P.wx1 * V01, bot = P.wx0 * V10 + P.wx1 * V11;
        const float f = P.wy0 * top + P.wy1 * bot;
        const float dsy = bot - top;
        const float e01 = V01 - V00, e11 = V11 - V10;
        const float dsx = P.wy0 * e01 + P.wy1 * e11;
        const float dxy = (e11 - e01) * m;
        const float g = vg[e] * as;
        const float J0 = dsx * P.j0u + dsy * P.j0v, J1 = dsx * P.j1u + dsy * P.j1v, J2 = dsx * P.j2u + dsy * P.j2v;
        const float w = USE_W ? P.wt : 1.f;
        const float aj0 = A00 * J0 + A01 * J1 + A02 * J2, aj1 = A01 * J0 + A11 * J1 + A12 * J2, aj2 = A02 * J0 + A12 * J1 + A22 * J2;
        const float jU = J0 * gU0 + J1 * gU1 + J2 * gU2, jV = J0 * gV0 + J1 * gV1 + J2 * gV2;
        const float gf = w * jU, ggr = w * jV;
        const float gJ0 = w * (aj0 + f * gU0 + g * gV0), gJ1 = w * (aj1 + f * gU1 + g * gV1), gJ2 = w * (aj2 + f * gU2 + g * gV2);
        const float gdsx = gJ0 * P.j0u + gJ1 * P.j1u + gJ2 * P.j2u, gdsy = gJ0 * P.j0v + gJ1 * P.j1v + gJ2 * P.j2v;
        q[0] += gf * dsx + gdsy * dxy; q[1] += gf * dsy + gdsx * dxy;
        q[2] += gJ0 * dsx; q[3] += gJ0 * dsy; q[4] += gJ1 * dsx; q[5] += gJ1 * dsy; q[6] += gJ2 * dsx; q[7] += gJ2 * dsy;
        if (USE_W) q[8] += 0.5f * (J0 * aj0 + J1 * aj1 + J2 * aj2) + f * jU + g * jV;
        d00[e] = gf * P.wy0 * P.wx0 - gdsx * P.wy0 - gdsy * P.wx0;
        d01[e] = gf * P.wy0 * P.wx1 + gdsx * P.wy0 - gdsy * P.wx1;
        d10[e] = gf * P.wy1 * P.wx0 - gdsx * P.wy1 + gdsy * P.wx0;
        d11[e] = gf * P.wy1 * P.wx1 + gdsx * P.wy1 + gdsy * P.wx1;
        dg[e] = ggr;
      }
      if (m != 0.f) {
        if (P.off != cur_off || P.dxo != cur_dxo || P.dyo != cur_dyo) {
          flush_cell();
          cur_off = P.off; cur_dxo = P.dxo; cur_dyo = P.dyo;
#pragma unroll
          for (int e = 0; e < 4; ++e) { c00[e] = d00[e]; c01[e] = d01[e]; c10[e] = d10[e]; c11[e] = d11[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { c00[e] += d00[e]; c01[e] += d01[e]; c10[e] += d10[e]; c11[e] += d11[e]; }
        }
      }
      float* gp = a.d_fix + fix_base + (size_t)i * C;                   // this (pixel, channels) is owned by this lane
#pragma unroll
      for (int e = 0; e < 4; ++e) gp[e * LPP] += dg[e];
    }
#pragma unroll
    for (int k = 0; k < (USE_W ? 9 : 8); ++k) {
#pragma unroll
      for (int o = LPP >> 1; o > 0; o >>= 1) q[k] += __shfl_xor(q[k], o, 64);
    }
    if (live && (lane % LPP) == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) pixacc[i][k] = q[k];
    }
  }
  flush_cell();
  __syncthreads();

  // pixel adjoints -> through the bilinear confidence sample and the perspective division -> 21 coefficient adjoints
  double c21[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) c21[k] = 0.0;
  if (t < np) {
    const int p = p0 + t;
    const G2sPix P = pp[t];
    const bool inb = (P.wx0 + P.wx1 + P.wy0 + P.wy1) != 0.f;
    if (inb) {
      const double vc = (double)(p / a.A - a.ctr), uc = (double)(p % a.A - a.ctr);
      const double q0 = cf[0] * vc + cf[3] * uc + cf[6], q1 = cf[1] * vc + cf[4] * uc + cf[7], q2 = cf[2] * vc + cf[5] * uc + cf[8];
      const bool front = q2 > 1e-6;
      const double iz = 1.0 / fmax(q2, 1e-6), iz2 = iz * iz, iz3 = iz2 * iz;
      double au = pixacc[t][0], av = pixacc[t][1];
      if (USE_W) {            // wt = bilinear(conf) depends on the map values and, through (u,v), on the pose
        const double aw = pixacc[t][8];
        const int ix = (P.off / C) % a.w, iy = (P.off / C) / a.w, dx = P.dxo / C, dy = P.dyo / C;    // dy in elements of w
        const float* cp = confb + (size_t)iy * a.w + ix;
        const float k00 = cp[0], k01 = cp[dx], k10 = cp[dy], k11 = cp[dy + dx];
        au += aw * (double)(P.wy0 * (k01 - k00) + P.wy1 * (k11 - k10));
        av += aw * (double)(P.wx0 * (k10 - k00) + P.wx1 * (k11 - k01));
        if (a.d_conf) {
          float* dc = a.d_conf + (size_t)b * a.h * a.w + (size_t)iy * a.w + ix;
          const float w_ = (float)aw;
          atomicAdd(dc, w_ * P.wy0 * P.wx0); atomicAdd(dc + dx, w_ * P.wy0 * P.wx1);
          atomicAdd(dc + dy, w_ * P.wy1 * P.wx0); atomicAdd(dc + dy + dx, w_ * P.wy1 * P.wx1);
        }
      }
      double aq0 = au * iz, aq1 = av * iz, aq2 = front ? -(au * q0 + av * q1) * iz2 : 0.0;
      if (front) {
        const double tx[3] = {cf[9], cf[10], cf[11]}, ty[3] = {cf[12], cf[13], cf[14]};
        const double tt[3] = {cf[15] * vc + cf[18] * uc, cf[16] * vc + cf[19] * uc, cf[17] * vc + cf[20] * uc};
        const double* tp[3] = {tx, ty, tt};
        double atp[3][3];
        for (int pi = 0; pi < 3; ++pi) {
          const double aju = pixacc[t][2 + 2 * pi], ajv = pixacc[t][3 + 2 * pi];
          const double* T3 = tp[pi];
          atp[pi][0] = aju * iz; atp[pi][1] = ajv * iz; atp[pi][2] = -(aju * q0 + ajv * q1) * iz2;
          aq0 -= aju * T3[2] * iz2; aq1 -= ajv * T3[2] * iz2;
          aq2 += aju * (-T3[0] * iz2 + 2.0 * q0 * T3[2] * iz3) + ajv * (-T3[1] * iz2 + 2.0 * q1 * T3[2] * iz3);
        }
        for (int k = 0; k < 3; ++k) {
          c21[9 + k] = atp[0][k]; c21[12 + k] = atp[1][k];
          c21[15 + k] = atp[2][k] * vc; c21[18 + k] = atp[2][k] * uc;
        }
      }
      const double aq[3] = {aq0, aq1, aq2};
      for (int k = 0; k < 3; ++k) { c21[k] = aq[k] * vc; c21[3 + k] = aq[k] * uc; c21[6 + k] = aq[k]; }
    }
  }
#pragma unroll
  for (int k = 0; k < 21; ++k) c21[k] = wave_sum_f64(c21[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 21; ++k) red[wave][k] = c21[k];
  }
  __syncthreads();
  if (t < G2S_PART_N) {
    double v = 0.0;
    if (t < 21) v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    a.part[((size_t)b * a.nt + tile) * G2S_PART_N + t] = v;
  }
}

// adjoint of g2s_coefficients: a[0..20] -> d(loss)/d(su, sv, th)
__device__ static inline void g2s_coefficients_bwd(const G2sGeom& G, double th, const float* K9, const double* a, double* g3) {
  const double k = G.rot / 180.0 * 3.14159265358979323846;
  const double ang = -th * k, c = cos(ang), s = sin(ang);
  double gc = 0.0, gs = 0.0, gT0 = 0.0, gT2 = 0.0;
  for (int r = 0; r < 3; ++r) {
    const double K0 = (double)K9[r * 3 + 0] * (r == 0 ? G.sx : (r == 1 ? G.sy : 1.0));
    const double K2 = (double)K9[r * 3 + 2] * (r == 0 ? G.sx : (r == 1 ? G.sy : 1.0));
    gc += (a[r] * K0 + a[3 + r] * K2 - a[15 + r] * k * K2 + a[18 + r] * k * K0) * G.mpp;
    gs += (a[r] * K2 - a[3 + r] * K0 + a[15 + r] * k * K0 + a[18 + r] * k * K2) * G.mpp;
    gT0 += a[6 + r] * K0; gT2 += a[6 + r] * K2;
  }
  g3[0] = -G.lon * gT2;                 // T2 = -su*lon
  g3[1] = G.lat * gT0;                  // T0 =  sv*lat
  g3[2] = k * (s * gc - c * gs);        // ang = -th*k
}

struct G2sBwdSolveArgs {
  const double* part_next; int nt_next; G2sGeom geom_next;
  const double* normal_eq;
  const float* pose_in; int pose_in_stride;
  const float* pose_out; const float* d_trace; int trace_stride;
  double* gid; double* adj; double* coef; double* d_lambda;
  const float* camera_k;
  int B, first;
  LmSolveCfg cfg; G2sGeom geom;
};

__global__ __launch_bounds__(64) void g2s_bwd_solve(G2sBwdSolveArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* K9 = a.camera_k + (size_t)b * 9;
  double c21[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) c21[k] = 0.0;
  if (a.part_next) {
    for (int i = lane; i < a.nt_next; i += 64) {
      const double* p = a.part_next + ((size_t)b * a.nt_next + i) * G2S_PART_N;
#pragma unroll
      for (int k = 0; k < 21; ++k) c21[k] += p[k];
    }
#pragma unroll
    for (int k = 0; k < 21; ++k) c21[k] = wave_sum_f64(c21[k]);
  }
  if (lane != 0) return;
  const float* po = a.pose_out + (size_t)b * a.trace_stride;
  const float* dt = a.d_trace + (size_t)b * a.trace_stride;
  double gout[3] = {dt[0], dt[1], dt[2]};
  if (!a.first) {
    double g3[3] = {0, 0, 0};
    if (a.part_next) g2s_coefficients_bwd(a.geom_next, po[2], K9, c21, g3);
    for (int p = 0; p < 3; ++p) gout[p] += g3[p] + a.gid[(size_t)b * 3 + p];
  }
  float pin[3] = {0.f, 0.f, 0.f};
  if (a.pose_in) { const float* pi = a.pose_in + (size_t)b * a.pose_in_stride; pin[0] = pi[0]; pin[1] = pi[1]; pin[2] = pi[2]; }
  const double* s = a.normal_eq + (size_t)b * 16;          // s[0] = s[1] = 1: no renormalisation in this direction
  double H[3][3], g[3], Mi[3][3], d[3], ns, ng;
  lm_solve_step(a.cfg, s, H, g, Mi, d, ns, ng);
  double gd[3], y[3];
  for (int p = 0; p < 3; ++p) { a.gid[(size_t)b * 3 + p] = gout[p]; gd[p] = -gout[p]; }
  for (int p = 0; p < 3; ++p) y[p] = Mi[p][0] * gd[0] + Mi[p][1] * gd[1] + Mi[p][2] * gd[2];
  double gH[3][3];
  for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) gH[p][q] = -y[p] * d[q];
  for (int p = 0; p < 3; ++p) atomicAdd(a.d_lambda + p, gH[p][p]);      // M = H + diag(lambda)
  double* ad = a.adj + (size_t)b * 16;
  ad[0] = ad[1] = 0.0;
  ad[2] = 2.0 * gH[0][0]; ad[3] = gH[0][1] + gH[1][0]; ad[4] = gH[0][2] + gH[2][0];
  ad[5] = 2.0 * gH[1][1]; ad[6] = gH[1][2] + gH[2][1]; ad[7] = 2.0 * gH[2][2];
  for (int p = 0; p < 3; ++p) { ad[8 + p] = y[p]; ad[11 + p] = -y[p]; }
  ad[14] = ad[15] = 0.0;
  g2s_coefficients(a.geom, pin[0], pin[1], pin[2], K9, a.coef + (size_t)b * G2S_COEF_N);
}

static size_t g2s_bwd_layout(const hla_s2g_config* cfg, const hla_s2g_level* lv, int B, size_t off[5]) {
  int max_nt = 1;
  for (int l = 0; l < cfg->n_levels; ++l) {
    const int npix = lv[l].A * lv[l].A, tp = lm_pick_tile(npix);
    max_nt = max(max_nt, (npix + tp - 1) / tp);
  }
  size_t o = 0;
  off[0] = o; o += hla_align_up((size_t)B * G2S_COEF_N * sizeof(double), 256);
  off[1] = o; o += hla_align_up((size_t)B * 16 * sizeof(double), 256);
  off[2] = o; o += hla_align_up((size_t)B * 3 * sizeof(double), 256);
  off[3] = o; o += hla_align_up((size_t)B * max_nt * G2S_PART_N * sizeof(double), 256);
  off[4] = o;
  return o;
}

extern "C" size_t hla_g2s_bwd_workspace_bytes(const hla_s2g_config* cfg, const hla_s2g_level* levels, int B) {
  size_t off[5];
  return g2s_bwd_layout(cfg, levels, B, off);
}

template <bool W>
static void launch_g2s_bwd(int C, dim3 grid, hipStream_t st, const G2sBwdAccumArgs& a) {
  switch (C) {
    case 256: hipLaunchKernelGGL((g2s_bwd_accum<256, W>), grid, dim3(256), 0, st, a); break;
    case 128: hipLaunchKernelGGL((g2s_bwd_accum<128, W>), grid, dim3(256), 0, st, a); break;
    case 64: hipLaunchKernelGGL((g2s_bwd_accum<64, W>), grid, dim3(256), 0, st, a); break;
  }
}

extern "C" int hla_g2s_lm_solve_bwd(const hla_s2g_config* cfg, const hla_s2g_level* lv, const hla_s2g_level_grad* gr,
                                    const float* camera_k, int ori_h, int ori_w, const float* pose0, const float* trace,
                                    const double* normal_eq, const float* d_trace, double* d_damping, void* workspace,
                                    size_t workspace_bytes, int B, hla_stream_t stream) {
  HLA_REQUIRE(cfg && lv && gr && camera_k && trace && normal_eq && d_trace && d_damping && workspace,
              "hla_g2s_lm_solve_bwd: null argument");
  HLA_REQUIRE(B > 0 && !cfg->ford && cfg->dof == 3 && !cfg->level_first && !cfg->use_hessian && ori_h > 0 && ori_w > 0,
              "hla_g2s_lm_solve_bwd: bad configuration");
  for (int l = 0; l < cfg->n_levels; ++l) {
    HLA_REQUIRE(lv[l].C == 256 || lv[l].C == 128 || lv[l].C == 64, "hla_g2s_lm_solve_bwd: unsupported channel count");
    HLA_REQUIRE(lv[l].sat_feat && lv[l].grd_feat && gr[l].d_sat_feat && gr[l].d_grd_feat, "hla_g2s_lm_solve_bwd: level %d buffers missing", l);
    HLA_REQUIRE(!cfg->using_weight || lv[l].grd_conf, "hla_g2s_lm_solve_bwd: using_weight needs grd_conf");
  }
  size_t off[5];
  const size_t need = g2s_bwd_layout(cfg, lv, B, off);
  if (workspace_bytes < need) {
    hla_set_error("hla_g2s_lm_solve_bwd: workspace %zu < %zu", workspace_bytes, need);
    return HLA_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  double* coef = (double*)(ws + off[0]);
  double* adj = (double*)(ws + off[1]);
  double* gid = (double*)(ws + off[2]);
  double* part = (double*)(ws + off[3]);
  HLA_CHECK_HIP(hipMemsetAsync(d_damping, 0, 3 * sizeof(double), st));
  const int L = cfg->n_levels, N = cfg->n_iters, steps = L * N, tstride = N * L * 3;
  auto slot = [&](int k) { return ((size_t)(k / L) * L + (k % L)) * 3; };
  auto geom = [&](int l) {
    G2sGeom g{};
    g.lat = cfg->shift_range_lat; g.lon = cfg->shift_range_lon; g.rot = cfg->rotation_range;
    g.mpp = lv[l].meter_per_pixel; g.sx = (double)lv[l].w / ori_w; g.sy = (double)lv[l].h / ori_h;
    return g;
  };
  int nt_prev = 0;
  for (int k = steps - 1; k >= 0; --k) {
    const int l = k % L;
    const hla_s2g_level& v = lv[l];
    G2sBwdSolveArgs sa{};
    sa.first = (k == steps - 1) ? 1 : 0;
    if (!sa.first) { sa.part_next = part; sa.nt_next = nt_prev; sa.geom_next = geom((k + 1) % L); }
    sa.normal_eq = normal_eq + (size_t)k * B * 16;
    if (k > 0) { sa.pose_in = trace + slot(k - 1); sa.pose_in_stride = tstride; }
    else { sa.pose_in = pose0; sa.pose_in_stride = 3; }
    sa.pose_out = trace + slot(k); sa.d_trace = d_trace + slot(k); sa.trace_stride = tstride;
    sa.gid = gid; sa.adj = adj; sa.coef = coef; sa.d_lambda = d_damping; sa.camera_k = camera_k; sa.B = B;
    sa.cfg.dof = 3; sa.cfg.use_hessian = 0;
    for (int i = 0; i < 3; ++i) sa.cfg.lam[i] = cfg->damping[i];
    sa.geom = geom(l);
    hla_prof_begin(K_LMSOLVE, 0, 0, st);
    hipLaunchKernelGGL(g2s_bwd_solve, dim3(B), dim3(64), 0, st, sa);
    hla_prof_end(st);

    G2sBwdAccumArgs aa{};
    aa.src = (const float*)v.grd_feat; aa.fix = (const float*)v.sat_feat; aa.conf = v.grd_conf; aa.coef = coef; aa.adj = adj;
    aa.src_inv = v.grd_inv_norm; aa.fix_inv = v.sat_inv_norm;
    aa.d_src = gr[l].d_grd_feat; aa.d_fix = gr[l].d_sat_feat; aa.d_conf = gr[l].d_grd_conf; aa.part = part;
    aa.A = v.A; aa.h = v.h; aa.w = v.w; aa.ctr = v.A / 2; aa.npix = v.A * v.A;
    aa.TP = lm_pick_tile(aa.npix); aa.nt = (aa.npix + aa.TP - 1) / aa.TP; aa.B = B;
    aa.xcd_affine = (B >= 8) ? 1 : 0;
    const int nblk = aa.xcd_affine ? 8 * ((B + 7) / 8) * aa.nt : B * aa.nt;
    hla_prof_begin(K_LMBWD, 0, (double)B * (5.0 * (double)v.h * v.w + 3.0 * (double)aa.npix) * v.C * 4.0, st);
    if (cfg->using_weight) launch_g2s_bwd<true>(v.C, dim3(nblk), st, aa);
    else launch_g2s_bwd<false>(v.C, dim3(nblk), st, aa);
    hla_prof_end(st);
    nt_prev = aa.nt;
  }
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}
