// Shared device code of the LM pose loop (forward: lm_solve.hip, backward: lm_backward.hip).
#pragma once
#include "common.h"

#define COEF_N 16      // doubles per sample: au[3] tu av[3] tv jsu[2] jsv[2] k centre pad pad
#define PART_N 16      // doubles per (sample, tile)
#define MAX_TP 256

struct __attribute__((aligned(16))) PixParam {
  int off, dxo, dyo;            // element offsets of the NW tap and the +x / +y neighbours
  float wx0, wx1, wy0, wy1;     // clamped-corner bilinear weights x in-bounds x ground mask
  float j2u, j2v;               // d(uv)/d(theta) at this pixel
  float gm, wt;                 // ground mask (z>0), LM weight
  float m;                      // in-bounds x ground mask (0/1)
};

// Pixels per block.  Depends on the level only, never on the batch size, so that a sample's partial-sum
// grouping -- and therefore its pose, bit for bit -- does not depend on its batch mates.
static inline int lm_pick_tile(int npix) {
  if (npix >= 16384) return 256;
  if (npix >= 4096) return 128;
  return 64;
}

// Arrival ticket of the fused accumulate + closing-solve kernels: a tile publishes its sums (8-byte agent-scope `sc1` stores,
// which write through this XCD's L2) and draws a ticket; the LAST arriver of a sample reads every tile's sums with agent-scope
// loads (they bypass its CU's L1 and its XCD's L2) and closes the step.
//   HLA_LM_FORMAL_FENCE = 0 (shipped): s_waitcnt vmcnt(0) -- the sc1 stores have been acknowledged by memory -- then a RELAXED
//     agent-scope ticket; the asm's "memory" clobber keeps the compiler from moving anything across it.  This is the write-through
//     recipe of cdna_hip_programming.md (guideline 16); it relies on the gfx942 / gfx950 meaning of sc1 on stores and loads.
//   HLA_LM_FORMAL_FENCE = 1: the ticket is an ACQ_REL agent-scope atomic, i.e. what the language memory model asks for.  On gfx950
//     that is `buffer_wbl2 sc1` (write back EVERY dirty line of this XCD's L2 -- the d_sat atomics and d_grd stores of the whole
//     XCD, not just this tile's sums) before the ticket and `buffer_inv sc1` behind it, per tile.  Kept as a same-box A/B switch
//     (EXPERIMENTS.md round 6); tests and fuzz pass either way.
#ifndef HLA_LM_FORMAL_FENCE
#define HLA_LM_FORMAL_FENCE 0
#endif
__device__ __forceinline__ unsigned lm_draw_ticket(unsigned* ticket, int lane) {
  unsigned old = 0;
#if HLA_LM_FORMAL_FENCE
  if (lane == 0) old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partials have left this CU before the ticket is drawn
  if (lane == 0) old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  return (unsigned)__builtin_amdgcn_readfirstlane((int)old);
}

// block -> (sample, tile).  With >= 8 samples keep every tile of a sample on one XCD (blocks are dealt
// round-robin to the 8 XCDs) so its satellite map stays in that XCD's L2.  Returns false for idle blocks.
__device__ __forceinline__ bool lm_block_map(int xcd_affine, int nt, int B, int& b, int& tile) {
  if (xcd_affine) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    b = (j / nt) * 8 + xcd;
    tile = j % nt;
    return b < B;
  }
  b = blockIdx.x / nt;
  tile = blockIdx.x % nt;
  return true;
}

// One ground pixel: satellite coordinates in fp64 from the sample's projection coefficients and the fp32
// ground-plane table; clamped-corner bilinear weights with the hard in-bounds mask of jacobian.py:146-177.
template <int C>
__device__ __forceinline__ PixParam lm_pixel(const double* cf, const float* q, int A, float conf_w) {
  const double X = q[0], Y = q[1], Z = q[2];
  const double u = cf[0] * X + cf[1] * Y + cf[2] * Z + cf[3];
  const double v = cf[4] * X + cf[5] * Y + cf[6] * Z + cf[7];
  const double lim = (double)(A - 1);
  const bool gm = q[2] > 0.f;
  const bool inb = (u >= 0.0) && (u <= lim) && (v >= 0.0) && (v <= lim);   // jacobian.py:168-170
  PixParam o;
  o.gm = gm ? 1.f : 0.f;
  o.wt = conf_w * o.gm;                                                    // grd_conf * mask (or 1)
  if (inb && gm) {
    const double x0 = floor(u), y0 = floor(v);
    const double x1 = fmin(x0 + 1.0, lim), y1 = fmin(y0 + 1.0, lim);       // clamped corners, 146-166
    o.wx0 = (float)(x1 - u); o.wx1 = (float)(u - x0);
    o.wy0 = (float)(y1 - v); o.wy1 = (float)(v - y0);
    const int ix0 = (int)x0, iy0 = (int)y0;
    o.off = (iy0 * A + ix0) * C;
    o.dxo = ((int)x1 - ix0) * C;
    o.dyo = ((int)y1 - iy0) * A * C;
    const double k = cf[12], ctr = cf[13];
    o.j2u = (float)(k * (v - ctr));
    o.j2v = (float)(-k * (u - ctr));
    o.m = 1.f;
  } else {
    o.wx0 = o.wx1 = o.wy0 = o.wy1 = 0.f;
    o.off = o.dxo = o.dyo = 0;
    o.j2u = o.j2v = 0.f;
    o.m = 0.f;
  }
  return o;
}

struct LmGeom {           // per-launch scalars of the pose -> coefficient map
  int ford;
  double lat, lon, rot;   // shift_range_lat, shift_range_lon (m), rotation_range (deg)
  double mpp, ctr;        // metres per sat pixel and centre of the level the coefficients are for
};

// pose (su, sv, th normalised) -> cf[0..13]   (models_kitti.py:719-799 / models_ford.py:208-253)
__device__ __forceinline__ void lm_coefficients(const LmGeom& G, double su, double sv, double th, const float* R,
                                                const float* T, double* cf) {
  const double k = G.rot / 180.0 * 3.14159265358979323846;
  const double ang = th * k, c = cos(ang), s = sin(ang), im = 1.0 / G.mpp, ctr = G.ctr;
  if (!G.ford) {
    const double sum = su * G.lon, svm = sv * G.lat;
    cf[0] = s * im; cf[1] = 0.0; cf[2] = c * im; cf[3] = (c * sum - s * svm) * im + ctr;
    cf[4] = c * im; cf[5] = 0.0; cf[6] = -s * im; cf[7] = (-c * svm - s * sum) * im + ctr;
    cf[8] = c * G.lon * im; cf[9] = -s * G.lon * im;
    cf[10] = -s * G.lat * im; cf[11] = -c * G.lat * im;
  } else {
    const double sum = su * G.lat, svm = sv * G.lon;
    const double t0 = (double)T[0] + svm, t1 = (double)T[1] - sum;
    for (int i = 0; i < 3; ++i) {
      cf[i] = (-s * (double)R[i] + c * (double)R[3 + i]) * im;
      cf[4 + i] = -(c * (double)R[i] + s * (double)R[3 + i]) * im;
    }
    cf[3] = (-s * t0 + c * t1) * im + ctr;
    cf[7] = -(c * t0 + s * t1) * im + ctr;
    cf[8] = -c * G.lat * im; cf[9] = s * G.lat * im;
    cf[10] = -s * G.lon * im; cf[11] = -c * G.lon * im;
  }
  cf[12] = k; cf[13] = ctr; cf[14] = 0.0; cf[15] = 0.0;
}

// adjoint of lm_coefficients: a[0..11] = d(loss)/d(cf[0..11])  ->  d(loss)/d(su, sv, th)
__device__ __forceinline__ void lm_coefficients_bwd(const LmGeom& G, double su, double sv, double th, const float* R,
                                                    const float* T, const double* a, double* g3) {
  const double k = G.rot / 180.0 * 3.14159265358979323846;
  const double ang = th * k, c = cos(ang), s = sin(ang), im = 1.0 / G.mpp;
  double gc, gs;
  if (!G.ford) {
    const double sum = su * G.lon, svm = sv * G.lat;
    gc = (a[2] + a[3] * sum + a[4] - a[7] * svm + a[8] * G.lon - a[11] * G.lat) * im;
    gs = (a[0] - a[3] * svm - a[6] - a[7] * sum - a[9] * G.lon - a[10] * G.lat) * im;
    g3[0] = G.lon * (c * a[3] - s * a[7]) * im;
    g3[1] = G.lat * (-s * a[3] - c * a[7]) * im;
  } else {
    const double sum = su * G.lat, svm = sv * G.lon;
    const double t0 = (double)T[0] + svm, t1 = (double)T[1] - sum;
    gc = (a[3] * t1 - a[7] * t0 - a[8] * G.lat - a[11] * G.lon) * im;
    gs = (-a[3] * t0 - a[7] * t1 + a[9] * G.lat - a[10] * G.lon) * im;
    for (int i = 0; i < 3; ++i) {
      gc += (a[i] * (double)R[3 + i] - a[4 + i] * (double)R[i]) * im;
      gs += (-a[i] * (double)R[i] - a[4 + i] * (double)R[3 + i]) * im;
    }
    const double gt0 = (-s * a[3] - c * a[7]) * im, gt1 = (c * a[3] - s * a[7]) * im;
    g3[1] = G.lon * gt0;
    g3[0] = -G.lat * gt1;
  }
  g3[2] = k * (-s * gc + c * gs);
}

__device__ static inline void lm_inv3(const double M[3][3], double I[3][3]) {
  const double c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1];
  const double c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2];
  const double c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
  const double id = 1.0 / (M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02);
  I[0][0] = c00 * id; I[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) * id; I[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) * id;
  I[1][0] = c01 * id; I[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) * id; I[1][2] = (M[0][2] * M[1][0] - M[0][0] * M[1][2]) * id;
  I[2][0] = c02 * id; I[2][1] = (M[0][1] * M[2][0] - M[0][0] * M[2][1]) * id; I[2][2] = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) * id;
}

struct LmSolveCfg { int dof, use_hessian; double lam[3]; int gn; };   // gn: GN_update (models_ford.py:534-598)

// The damped normal-equation solve of one step from the 14 (already de-normalised) sums.
// Fills H (normalised), g, M^-1 restricted to the active DoFs (others zero) and d = M^-1 g.
__device__ static inline void lm_solve_step(const LmSolveCfg& S, const double* s, double H[3][3], double g[3],
                                            double Mi[3][3], double d[3], double& ns, double& ng) {
  // models_kitti.py:976-984: both norms clamped at 1e-6; J is divided by ||s|| as well
  ns = fmax(sqrt(s[0]), 1e-6); ng = fmax(sqrt(s[1]), 1e-6);
  if (S.gn) ng = 1.0;            // GN_update keeps the whole-map L2_norm of the ground features (the host zeroes lam)
  const double is2 = 1.0 / (ns * ns), isg = 1.0 / (ns * ng);
  H[0][0] = s[2] * is2; H[0][1] = H[1][0] = s[3] * is2; H[0][2] = H[2][0] = s[4] * is2;
  H[1][1] = s[5] * is2; H[1][2] = H[2][1] = s[6] * is2; H[2][2] = s[7] * is2;
  for (int p = 0; p < 3; ++p) g[p] = s[8 + p] * is2 - s[11 + p] * isg;
  for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) Mi[p][q] = 0.0;
  d[0] = d[1] = d[2] = 0.0;
  if (S.dof == 3) {
    double M[3][3];
    for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) M[p][q] = H[p][q];
    for (int p = 0; p < 3; ++p) M[p][p] += S.lam[p] * (S.use_hessian ? H[p][p] : 1.0);
    lm_inv3(M, Mi);
  } else if (S.dof == 2) {       // rotation_range == 0: (u,v) only, models_kitti.py:954-955,1015-1018
    const double m00 = H[0][0] + S.lam[0] * (S.use_hessian ? H[0][0] : 1.0);
    const double m11 = H[1][1] + S.lam[1] * (S.use_hessian ? H[1][1] : 1.0);
    const double m01 = H[0][1], id = 1.0 / (m00 * m11 - m01 * m01);
    Mi[0][0] = m11 * id; Mi[1][1] = m00 * id; Mi[0][1] = Mi[1][0] = -m01 * id;
  } else {                       // shift ranges == 0: theta only, models_kitti.py:956-957,1019-1022
    Mi[2][2] = 1.0 / (H[2][2] + S.lam[0] * (S.use_hessian ? H[2][2] : 1.0));
  }
  for (int p = 0; p < 3; ++p) d[p] = Mi[p][0] * g[0] + Mi[p][1] * g[1] + Mi[p][2] * g[2];
}
