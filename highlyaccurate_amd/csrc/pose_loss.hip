// loss_func, method 0 (models_ford.py:1041-1093; models_kitti.py imports the same function): the pose loss of a training
// step and its gradient, one launch each.  The reference evaluates it as ~25 element-wise / reduction ops on [B,N,L] tensors
// and autograd replays ~40 more -- on the device 90 launches of 5-7 us, 0.55 ms between the LM loop and its backward.
//   d_k[n,l]   = mean_b |x_k[b,n,l] - gt_k[b]|          k = lat, lon, theta
//   losses     = coe_lat d_lat + coe_lon d_lon + coe_theta d_theta          [N,L]
//   out        = mean(losses) | losses[0]-losses[-1] | d_k[0]-d_k[-1] (3) | losses[-1] | d_k[-1] (3)      1 + 8 L values
// The per-(n,l) batch means are summed in fp64 in batch order (fixed: bitwise reproducible), then rounded to the
// reference's result type OT (fp32; fp64 when the ground truth is fp64, torch's type promotion -- the Ford loader) and
// combined in the reference's operation order.
#include "common.h"

#define POSE_LOSS_MAX_NL 512

struct PoseLossArgs {
  const float* x[3];        // shift_lats, shift_lons, thetas: element (b,n,l) at x[k][b sB + n sN + l sL]
  long long sB[3], sN[3], sL[3];
  const void* gt[3];        // [B] of OT, element stride sG
  long long sG[3];
  double coe[3];
  int B, N, L;
  void* out;                // [1 + 8 L] of OT
  const void* g[9];         // backward: d(loss)/d(out piece j), contiguous OT, or null (= zero)
  float* dx[3];             // backward: gradient w.r.t. x[k], element (b,n,l) at dx[k][b dB + n dN + l dL]
  long long dB[3], dN[3], dL[3];
};

template <typename OT>
__global__ __launch_bounds__(256) void pose_loss_kernel(PoseLossArgs a) {
  __shared__ OT d[3][POSE_LOSS_MAX_NL];
  __shared__ OT losses[POSE_LOSS_MAX_NL];
  const int NL = a.N * a.L, t = threadIdx.x;
  for (int i = t; i < 3 * NL; i += 256) {
    const int k = i / NL, nl = i % NL, n = nl / a.L, l = nl % a.L;
    const float* x = a.x[k] + (long long)n * a.sN[k] + (long long)l * a.sL[k];
    const OT* gt = (const OT*)a.gt[k];
    double s = 0.0;
    for (int b = 0; b < a.B; ++b) {
      const OT v = (OT)x[(long long)b * a.sB[k]] - gt[(long long)b * a.sG[k]];      // the subtraction in the promoted type
      s += (double)(v < (OT)0 ? -v : v);
    }
    d[k][nl] = (OT)(s / (double)a.B);
  }
  __syncthreads();
  const OT c0 = (OT)a.coe[0], c1 = (OT)a.coe[1], c2 = (OT)a.coe[2];
  for (int nl = t; nl < NL; nl += 256) {
    // (coe_lat * d_lat + coe_lon * d_lon) + coe_theta * d_theta, every product and sum rounded on its own like the tensor ops
    const OT p0 = c0 * d[0][nl], p1 = c1 * d[1][nl], p2 = c2 * d[2][nl];
    OT s01, s;
    if (sizeof(OT) == 4) { s01 = (OT)__fadd_rn((float)p0, (float)p1); s = (OT)__fadd_rn((float)s01, (float)p2); }
    else { s01 = (OT)__dadd_rn((double)p0, (double)p1); s = (OT)__dadd_rn((double)s01, (double)p2); }
    losses[nl] = s;
  }
  __syncthreads();
  OT* out = (OT*)a.out;
  const int L = a.L, last = (a.N - 1) * a.L;
  if (t == 0) {
    double s = 0.0;
    for (int nl = 0; nl < NL; ++nl) s += (double)losses[nl];
    out[0] = (OT)(s / (double)NL);
  }
  for (int l = t; l < L; l += 256) {
    out[1 + 0 * L + l] = losses[l] - losses[last + l];
    for (int k = 0; k < 3; ++k) {
      out[1 + (1 + k) * L + l] = d[k][l] - d[k][last + l];
      out[1 + (5 + k) * L + l] = d[k][last + l];
    }
    out[1 + 4 * L + l] = losses[last + l];
  }
}

template <typename OT>
__global__ __launch_bounds__(256) void pose_loss_bwd_kernel(PoseLossArgs a) {
  const int NL = a.N * a.L;
  const long long total = 3LL * a.B * NL;
  const OT* g0 = (const OT*)a.g[0];
  const OT gm = g0 ? g0[0] / (OT)NL : (OT)0;          // mean(losses) backward: grad / numel
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int k = (int)(e / ((long long)a.B * NL));
    const long long r = e % ((long long)a.B * NL);
    const int b = (int)(r / NL), nl = (int)(r % NL), n = nl / a.L, l = nl % a.L;
    const bool first = n == 0, lastn = n == a.N - 1;
    // adjoint of losses[n,l] and of d_k[n,l]
    OT gl = gm, gd = (OT)0;
    auto gv = [&](int j) { const OT* p = (const OT*)a.g[j]; return p ? p[l] : (OT)0; };
    if (first) { gl += gv(1); gd += gv(2 + k); }
    if (lastn) { gl += gv(5) - gv(1); gd += gv(6 + k) - gv(2 + k); }
    const OT gk = (gl * (OT)a.coe[k] + gd) / (OT)a.B;      // ... * coe, then mean(dim=0) backward: / B
    const float xv = a.x[k][(long long)b * a.sB[k] + (long long)n * a.sN[k] + (long long)l * a.sL[k]];
    const OT v = (OT)xv - ((const OT*)a.gt[k])[(long long)b * a.sG[k]];
    // abs backward: grad * sign(x), 0 at 0; a NaN pose (a diverged solve) gives a NaN gradient, like torch's sgn
    const OT sg = v != v ? v : (v > (OT)0 ? (OT)1 : (v < (OT)0 ? (OT)-1 : (OT)0));
    a.dx[k][(long long)b * a.dB[k] + (long long)n * a.dN[k] + (long long)l * a.dL[k]] = (float)(gk * sg);
  }
}

static int pose_loss_check(const char* who, const hla_pose_loss_args* p) {
  HLA_REQUIRE(p, "%s: null argument", who);
  HLA_REQUIRE(p->B > 0 && p->N > 0 && p->L > 0 && p->N * p->L <= POSE_LOSS_MAX_NL, "%s: need B, N, L > 0 and N * L <= %d", who,
              POSE_LOSS_MAX_NL);
  HLA_REQUIRE(p->gt_dtype == HLA_POSE_LOSS_F32 || p->gt_dtype == HLA_POSE_LOSS_F64, "%s: gt_dtype must be 0 (fp32) or 1 (fp64)", who);
  for (int k = 0; k < 3; ++k) HLA_REQUIRE(p->x[k] && p->gt[k], "%s: null input %d", who, k);
  return HLA_OK;
}

static PoseLossArgs pose_loss_pack(const hla_pose_loss_args* p) {
  PoseLossArgs a{};
  for (int k = 0; k < 3; ++k) {
    a.x[k] = p->x[k]; a.sB[k] = p->x_stride[k][0]; a.sN[k] = p->x_stride[k][1]; a.sL[k] = p->x_stride[k][2];
    a.gt[k] = p->gt[k]; a.sG[k] = p->gt_stride[k]; a.coe[k] = p->coe[k];
  }
  a.B = p->B; a.N = p->N; a.L = p->L;
  return a;
}

extern "C" int hla_pose_loss(const hla_pose_loss_args* p, void* out, hla_stream_t stream) {
  const int rc = pose_loss_check("hla_pose_loss", p);
  if (rc) return rc;
  HLA_REQUIRE(out, "hla_pose_loss: null output");
  PoseLossArgs a = pose_loss_pack(p);
  a.out = out;
  if (p->gt_dtype == HLA_POSE_LOSS_F64) hipLaunchKernelGGL(pose_loss_kernel<double>, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(pose_loss_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}

extern "C" int hla_pose_loss_bwd(const hla_pose_loss_args* p, const void* const g_out[9], float* const dx[3],
                                 const long long dx_stride[3][3], hla_stream_t stream) {
  const int rc = pose_loss_check("hla_pose_loss_bwd", p);
  if (rc) return rc;
  HLA_REQUIRE(g_out && dx && dx_stride, "hla_pose_loss_bwd: null argument");
  PoseLossArgs a = pose_loss_pack(p);
  for (int j = 0; j < 9; ++j) a.g[j] = g_out[j];
  for (int k = 0; k < 3; ++k) {
    HLA_REQUIRE(dx[k], "hla_pose_loss_bwd: null gradient buffer %d", k);
    a.dx[k] = dx[k]; a.dB[k] = dx_stride[k][0]; a.dN[k] = dx_stride[k][1]; a.dL[k] = dx_stride[k][2];
  }
  const long long total = 3LL * p->B * p->N * p->L;
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  if (p->gt_dtype == HLA_POSE_LOSS_F64) hipLaunchKernelGGL(pose_loss_bwd_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(pose_loss_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}
