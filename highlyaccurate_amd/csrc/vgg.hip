// VGG16-U-Net feature extractor on gfx950 matrix cores: VGG.py:13-203, L2_norm VGG.py:511-514.
//
// Layout: activations NHWC; T = bf16 (throughput mode, fp32 accumulate) or float (exact-fp32 MFMA, parity mode).
//
// conv3x3_kernel -- implicit GEMM computed as D^T = W * X^T, so a lane owns one output pixel and four
//   consecutive output channels per accumulator quad:
//     M (MFMA rows) = 32 output channels   A operand = weight fragment, straight from L2/L1: weights are
//                                          pre-packed in fragment order, so a fragment is one coalesced 1 KiB load
//     N (MFMA cols) = 32 consecutive pixels of one image row; B operand = pixel fragment from the LDS halo tile
//     K             = 9 taps x Cin, walked as  stage (64 B of channels) -> tap -> 2 k-groups of 16 B per lane
//   The (TH+2)x34 input halo tile of a stage is written to LDS once (80-B pixel stride: conflict-free
//   ds_read_b128) and reused by all 9 taps; LDS is double-buffered, one barrier per stage; the next stage's
//   tile is prefetched into registers mid-stage; weights are prefetched 1-2 taps ahead into a register ring.
//   The loader handles "virtual concat + nearest 2x upsample" (VGG.py:144-151) without materialising it.
//   The epilogue fuses bias, 2x2 max-pool, ReLU, the raw fp32 feature copy and its per-sample sum of squares.
// conv02_kernel -- conv0 (3->64 on the NCHW fp32 input, K = 27 padded to 32) computed by MFMA directly into the
//   LDS halo tile of conv2, then conv2 + pool: the 64-channel full-resolution map never touches HBM.
#include "common.h"

typedef __bf16 bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <typename T> __device__ __forceinline__ void mma16(f32x16& acc, const uint4& w, const uint4& p);
template <> __device__ __forceinline__ void mma16<bf16>(f32x16& acc, const uint4& w, const uint4& p) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, p), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x16& acc, const uint4& w, const uint4& p) {
  // element t of both fragments: channels {8q+t (lanes 0-31), 8q+4+t (lanes 32-63)}
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(p.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(p.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(p.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(p.w), acc, 0, 0, 0);
}

__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
  const bf16 h[4] = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
  *(uint2*)p = __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *(float4*)p = make_float4(a, b, c, d);
}
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(float v) { return v; }

struct ConvArgs {
  const void* src1;   // NHWC T, C1 channels; at half resolution when up1
  const void* src2;   // NHWC T, C2 channels (virtual concat after src1), or null
  const uint4* wpk;   // fragment-packed weights
  const float* bias;  // [Cout] or null
  void* out_act;      // NHWC T [B,Ho,Wo,Cout] (post-ReLU when relu_act) or null
  float* out_raw;     // NHWC fp32 (pre-ReLU) or null
  double* sumsq;      // [B, tiles_per_img * gridDim.y] sum of squares of out_raw, or null
  int C1, C2, up1;
  int B, H, W, Cout;
  int relu_act;
  int tiles_x, tiles_y;
  unsigned long long* dbg;   // CONV_VARIANT 40 only: per-wave cycle accounting
};

constexpr int HWID = 34;   // halo tile width in pixels
constexpr int SB = 64;     // bytes of channels per pixel per pipeline stage
constexpr int PSTR = 80;   // LDS bytes per halo pixel (64 B of channels + 16 B pad: conflict-free ds_read_b128)
constexpr int HALO_TAP = 3;  // tap at which the next stage's halo loads are issued
#ifndef CONV_VARIANT
#define CONV_VARIANT 0
#endif
// timing ablations for tools/variants.py (results are wrong on purpose): 10 no weight loads, 11 no LDS reads,
// 12 no halo staging, 13 = all three, 14 = 13 + no barrier
constexpr bool ABL_ALL = (CONV_VARIANT == 13 || CONV_VARIANT == 14);
constexpr bool ABL_NO_W = (CONV_VARIANT == 10 || ABL_ALL);
constexpr bool ABL_NO_LDS = (CONV_VARIANT == 11 || ABL_ALL);
constexpr bool ABL_NO_HALO = (CONV_VARIANT == 12 || ABL_ALL);
constexpr bool ABL_NO_BAR = (CONV_VARIANT == 14);

// ---------------------------------------------------------------------------------------------
// shared epilogue: acc[i][j] holds, for lane (x = lane&31, g = lane>>5), output channels
// cb + j*32 + 8q + 4g + {0..3} (q = r>>2) of pixel (row i, column x).
// Writing that straight to NHWC memory is 8 B per lane into 32 different 128-B lines per store instruction
// (measured: 16-28 k cycles per wave, up to half of a block's lifetime).  Instead every wave transposes one
// pixel row at a time through a private LDS region (`stage`, >= 32*(NT*32*4+16) bytes) and stores it as whole
// pixel rows: 16 B per lane, consecutive lanes on consecutive addresses.
template <typename E, int NT> struct RowStager {
  static constexpr int CW = NT * 32, PITCH = CW * (int)sizeof(E) + 16, CPP = CW * (int)sizeof(E) / 16;
  // lane-side write of 4 consecutive channels of pixel `px`
  static __device__ __forceinline__ void put(char* stage, int px, int ch, float v0, float v1, float v2, float v3) {
    store4((E*)(stage + px * PITCH) + ch, v0, v1, v2, v3);
  }
  // cooperative flush of `npx` pixels: dst points at channel cb of the first pixel; pixel stride = Cout elements
  static __device__ __forceinline__ void flush(const char* stage, E* dst, int npx, int npx_valid, int Cout, int lane) {
    const int chunks = npx * CPP;
#pragma unroll
    for (int c0 = 0; c0 < 32 * CPP; c0 += 64) {
      const int c = c0 + lane;
      if (c0 < chunks && c < chunks) {
        const int px = c / CPP, part = c % CPP;
        if (px < npx_valid)
          *(uint4*)((char*)(dst + (size_t)px * Cout) + part * 16) = *(const uint4*)(stage + px * PITCH + part * 16);
      }
    }
  }
};

template <typename T, int MT, int NT, bool POOL>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MT][NT], const ConvArgs& a, int b, int yrow0, int x0,
                                              int cb, float* red, char* stage) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, x = lane & 31, g = lane >> 5;
  const int Ho = POOL ? a.H >> 1 : a.H, Wo = POOL ? a.W >> 1 : a.W;
  constexpr int NPX = POOL ? 16 : 32;
  float4 bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bias[j][q] = a.bias ? *(const float4*)(a.bias + cb + j * 32 + q * 8 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int xo0 = POOL ? x0 >> 1 : x0;
  const int nvalid = min(NPX, Wo - xo0);              // pixels of this row segment inside the image
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MT; i += (POOL ? 2 : 1)) {
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
            t = fmaxf(t, acc[i + 1][j][q * 4 + e]);
            t = fmaxf(t, __shfl_xor(t, 1, 64));
          }
          v[j][q][e] = t;
        }
        v[j][q][0] += bias[j][q].x; v[j][q][1] += bias[j][q].y; v[j][q][2] += bias[j][q].z; v[j][q][3] += bias[j][q].w;
      }
    const size_t pix0 = ((size_t)b * Ho + yo) * Wo + xo0;
    if (a.out_raw) {
      if (!POOL || !(x & 1)) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            RowStager<float, NT>::put(stage, px, j * 32 + q * 8 + g * 4, v[j][q][0], v[j][q][1], v[j][q][2], v[j][q][3]);
            if (lane_ok) ss += v[j][q][0] * v[j][q][0] + v[j][q][1] * v[j][q][1] + v[j][q][2] * v[j][q][2] + v[j][q][3] * v[j][q][3];
          }
      }
      if (row_ok) RowStager<float, NT>::flush(stage, a.out_raw + pix0 * a.Cout + cb, NPX, nvalid, a.Cout, lane);
    }
    if (a.out_act) {
      if (!POOL || !(x & 1)) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float w0 = v[j][q][0], w1 = v[j][q][1], w2 = v[j][q][2], w3 = v[j][q][3];
            if (a.relu_act) { w0 = fmaxf(w0, 0.f); w1 = fmaxf(w1, 0.f); w2 = fmaxf(w2, 0.f); w3 = fmaxf(w3, 0.f); }
            RowStager<T, NT>::put(stage, px, j * 32 + q * 8 + g * 4, w0, w1, w2, w3);
          }
      }
      if (row_ok) RowStager<T, NT>::flush(stage, (T*)a.out_act + pix0 * a.Cout + cb, NPX, nvalid, a.Cout, lane);
    }
  }
  if (a.sumsq) {
    ss = wave_sum_f32(ss);
    if (lane == 0) red[wv] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int np = a.tiles_x * a.tiles_y * gridDim.y;
      const int tile = (blockIdx.x % (a.tiles_x * a.tiles_y)) * gridDim.y + blockIdx.y;
      a.sumsq[(size_t)b * np + tile] = ((double)red[0] + (double)red[1]) + ((double)red[2] + (double)red[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// One pipeline stage of MFMAs: 9 taps x 2 k-groups against the halo tile at `cur` (already offset to this
// wave's first row / this lane's pixel + k-half).  Weight fragments come from global memory through a ring of
// WD+1 register sets filled WD taps ahead; `mid(tap)` runs right after the weight loads of each tap (used to
// issue the next stage's halo loads BEHIND them: VM loads of a wave retire in order).
template <typename T, int MT, int NT, int WD>
struct WeightRing {
  static constexpr int RS = WD + 1;
  uint4 wb[RS][2][NT];
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
#if CONV_VARIANT != 31
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
  if (slot & 1) __builtin_amdgcn_s_setprio(CONV_VARIANT == 32 ? 3 : 1);
#endif
}

template <typename T, int MT, int NT, int WD, bool PF_UPFRONT, typename Mid>
__device__ __forceinline__ void stage_mma(f32x16 (&acc)[MT][NT], const char* cur, WeightRing<T, MT, NT, WD>& ring,
                                          Mid&& mid) {
  constexpr int RS = WD + 1;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap % 3;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (!ABL_NO_W) ring.wb[(tap + WD) % RS][kg][j] = ring.wq[j][((tap + WD) * 2 + kg) * 64];
    mid(tap);
    const char* ap = cur + (ky * HWID + kx) * PSTR;
    if (PF_UPFRONT) {
      // all pixel fragments of the tap are requested up front; the MFMAs then wait on counted lgkmcnt
      uint4 pf[2][MT];
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (!ABL_NO_LDS) pf[kg][i] = *(const uint4*)(ap + i * HWID * PSTR + kg * 32);
          else pf[kg][i] = ring.wb[0][kg][0];
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma16<T>(acc[i][j], ring.wb[tap % RS][kg][j], pf[kg][i]);
    } else {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        uint4 pf[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (!ABL_NO_LDS) pf[i] = *(const uint4*)(ap + i * HWID * PSTR + kg * 32);
          else pf[i] = ring.wb[0][kg][0];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma16<T>(acc[i][j], ring.wb[tap % RS][kg][j], pf[i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  ring.next_stage();
}

template <typename T, int MT, int NT, int WM, int WN, bool POOL, int WD, bool PF_UPFRONT>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int EPL = 16 / sizeof(T), KC = SB / sizeof(T);     // channels per stage: 32 (bf16) / 16 (fp32)
  constexpr int TH = WM * MT, HPIX = (TH + 2) * HWID, BUF = HPIX * PSTR;
  constexpr int NPIECE = (HPIX * 4 + 255) / 256;               // 16-B pieces per thread per stage
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
  __shared__ float red[4];

#if CONV_VARIANT == 40
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv / WN, wn = wv % WN;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * 32;
  const int nstage = (a.C1 + a.C2) / KC;
  const int part = t & 3, pbase = t >> 2;   // 256 % 4 == 0: a thread always moves the same 16-B part of a pixel

  auto load_stage = [&](int sg, uint4 (&st)[NPIECE]) {
    const int c0 = sg * KC;
    const bool first = c0 < a.C1;       // wave-uniform
    const T* src = first ? (const T*)a.src1 : (const T*)a.src2;
    const int Cs = first ? a.C1 : a.C2;
    const int coff = (first ? c0 : c0 - a.C1) + part * EPL;
    const int sh = (first && a.up1) ? 1 : 0;
    const int Hs = a.H >> sh, Ws = a.W >> sh;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int pix = pbase + 64 * i;
      const int hy = pix / HWID, hx = pix - hy * HWID;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (pix < HPIX && y >= 0 && y < a.H && x >= 0 && x < a.W)
        v = *(const uint4*)(src + (((size_t)b * Hs + (y >> sh)) * Ws + (x >> sh)) * Cs + coff);
      st[i] = v;
    }
  };
  auto write_stage = [&](char* buf, const uint4 (&st)[NPIECE]) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int pix = pbase + 64 * i;
      if (pix < HPIX) *(uint4*)(buf + pix * PSTR + part * 16) = st[i];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int x = lane & 31, g = lane >> 5;
  const int aoff = ((wm * MT) * HWID + x) * PSTR + g * 16;
  const int ntg0 = (blockIdx.y * WN + wn) * NT;     // first global 32-channel output tile of this wave
  // packed weights: [ntile][stage][tap][kg(2)][lane] 16-B fragments
  WeightRing<T, MT, NT, WD> ring;
#pragma unroll
  for (int j = 0; j < NT; ++j) ring.wq[j] = a.wpk + (size_t)(ntg0 + j) * nstage * 18 * 64 + lane;

  uint4 st[NPIECE];
  load_stage(0, st);
  write_stage(lds, st);
  __syncthreads();
  ring.prime();
  stagger_priority();

#if CONV_VARIANT == 40   // cycle accounting per wave: [prologue, mma, halo write, barrier wait, epilogue]
  unsigned long long tc[5] = {0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_readcyclecounter();
  tc[0] = t_prev - t_begin;
#define TICK(k) { const unsigned long long _n = __builtin_readcyclecounter(); tc[k] += _n - t_prev; t_prev = _n; }
#else
#define TICK(k)
#endif
  for (int sg = 0; sg < nstage; ++sg) {
    const bool more = sg + 1 < nstage;
    stage_mma<T, MT, NT, WD, PF_UPFRONT>(acc, lds + (sg & 1) * BUF + aoff, ring, [&](int tap) {
      if (tap == HALO_TAP && more && !ABL_NO_HALO) load_stage(sg + 1, st);
    });
    TICK(1)
    if (more && !ABL_NO_HALO) write_stage(lds + ((sg + 1) & 1) * BUF, st);
    TICK(2)
    if (!ABL_NO_BAR) __syncthreads();
    TICK(3)
  }
  // the loop's last barrier guarantees nobody still reads the halo buffers: reuse them as 4 wave-private stagers
  static_assert(2 * BUF / 4 >= 32 * (NT * 32 * 4 + 16) && (2 * BUF / 4) % 16 == 0, "stager does not fit");
  conv_epilogue<T, MT, NT, POOL>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, lds + wv * (2 * BUF / 4));
#if CONV_VARIANT == 40
  TICK(4)
  if (a.dbg && lane == 0) {
    unsigned long long* d = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 5;
    for (int k = 0; k < 5; ++k) d[k] = tc[k];
  }
#endif
#undef TICK
}

// ---------------------------------------------------------------------------------------------
// conv0 + ReLU + conv2 + bias + 2x2 max-pool + ReLU in one kernel (VGG.py:123-128).
struct Conv02Args {
  const float* x;      // [B,3,H,W] NCHW fp32
  const uint4* w0;     // conv0 fragments [2 ntiles][NFRAG][64 lanes], k = cin*9 + tap (27 padded to 32)
  const float* b0;     // [64]
  const uint4* w2;     // conv2 fragments, generic layout
  const float* b2;     // [64]
  void* out_act;       // NHWC T [B,H/2,W/2,64] = relu(pool(conv2))
  int B, H, W, tiles_x, tiles_y;
};

template <typename T> constexpr int conv02_lds_bytes() {
  return (64 * (int)sizeof(T) / SB) * (10 * HWID * PSTR) + 3 * 12 * 36 * 4;
}

template <typename T, int WD, bool PF_UPFRONT>
__global__ __launch_bounds__(256, 2) void conv02_kernel(Conv02Args a0) {
  constexpr int EPL = 16 / sizeof(T), KC = SB / sizeof(T), NSG = 64 / KC, NFRAG = 32 / (2 * EPL);
  constexpr int MT = 4, NT = 1, WN = 2, TH = 8, HPIX = (TH + 2) * HWID, BUF = HPIX * PSTR, IW = 36, IH = 12;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds = smem;                                  // NSG halo buffers (all of conv0's 64 channels)
  float* in = (float*)(smem + NSG * BUF);            // [3][12][36] input patch
  __shared__ float red[4];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv / WN, wn = wv % WN;
  int bid = blockIdx.x;
  const int tx = bid % a0.tiles_x; bid /= a0.tiles_x;
  const int ty = bid % a0.tiles_y;
  const int b = bid / a0.tiles_y;
  const int y0 = ty * TH, x0 = tx * 32;
  const int x = lane & 31, g = lane >> 5;

  // start the first conv2 weight loads before anything else (they do not depend on the input)
  WeightRing<T, MT, NT, WD> ring;
  ring.wq[0] = a0.w2 + (size_t)wn * NSG * 18 * 64 + lane;
  ring.prime();

  // phase A: input patch (2-pixel border) -> LDS
  for (int e = t; e < 3 * IH * IW; e += 256) {
    const int c = e / (IH * IW), r = e % (IH * IW), iy = r / IW, ix = r % IW;
    const int y = y0 - 2 + iy, xx = x0 - 2 + ix;
    float v = 0.f;
    if (y >= 0 && y < a0.H && xx >= 0 && xx < a0.W) v = a0.x[(((size_t)b * 3 + c) * a0.H + y) * a0.W + xx];
    in[e] = v;
  }
  uint4 wf0[2][NFRAG];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) wf0[j][f] = a0.w0[(j * NFRAG + f) * 64 + lane];
  __syncthreads();

  // phase B: conv0 on the 10x34 halo pixels, 32 pixels per MFMA tile, straight into the conv2 halo buffers
  for (int m = wv; m * 32 < HPIX; m += 4) {
    const int p = m * 32 + x, pc = p < HPIX ? p : HPIX - 1;
    const int hy = pc / HWID, hx = pc - hy * HWID;
    const float* ib = in + hy * IW + hx;
    f32x16 c0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c0[j][r] = 0.f;
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) {
      T e[EPL];
#pragma unroll
      for (int jj = 0; jj < EPL; ++jj) {
        const int klo = f * 2 * EPL + jj, khi = klo + EPL;   // compile-time
        float lo = 0.f, hi = 0.f;
        if (klo < 27) lo = ib[(klo / 9) * IH * IW + ((klo % 9) / 3) * IW + (klo % 9) % 3];
        if (khi < 27) hi = ib[(khi / 9) * IH * IW + ((khi % 9) / 3) * IW + (khi % 9) % 3];
        e[jj] = (T)(g ? hi : lo);
      }
      const uint4 pf = __builtin_bit_cast(uint4, e);
#pragma unroll
      for (int j = 0; j < 2; ++j) mma16<T>(c0[j], wf0[j][f], pf);
    }
    // conv2 zero-pads conv0's OUTPUT map: halo pixels outside the image are 0, not conv0 of padded input
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool inside = yy >= 0 && yy < a0.H && xx >= 0 && xx < a0.W;
    if (p < HPIX) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = j * 32 + q * 8 + g * 4;
          const float4 bb = *(const float4*)(a0.b0 + co);
          float v0 = fmaxf(c0[j][q * 4 + 0] + bb.x, 0.f), v1 = fmaxf(c0[j][q * 4 + 1] + bb.y, 0.f);
          float v2 = fmaxf(c0[j][q * 4 + 2] + bb.z, 0.f), v3 = fmaxf(c0[j][q * 4 + 3] + bb.w, 0.f);
          if (!inside) v0 = v1 = v2 = v3 = 0.f;
          store4((T*)(lds + (co / KC) * BUF + p * PSTR) + (co % KC), v0, v1, v2, v3);
        }
    }
  }
  __syncthreads();

  // phase C: conv2 over the NSG resident stages (no further loads, no barriers)
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  const int aoff = ((wm * MT) * HWID + x) * PSTR + g * 16;
  stagger_priority();
#pragma unroll 1
  for (int sg = 0; sg < NSG; ++sg)
    stage_mma<T, MT, NT, WD, PF_UPFRONT>(acc, lds + sg * BUF + aoff, ring, [](int) {});

  ConvArgs a{};
  a.bias = a0.b2; a.out_act = a0.out_act; a.B = a0.B; a.H = a0.H; a.W = a0.W; a.Cout = 64; a.relu_act = 1;
  a.tiles_x = a0.tiles_x; a.tiles_y = a0.tiles_y;
  __syncthreads();   // all waves are done with the halo buffers; reuse them as wave-private stagers
  conv_epilogue<T, MT, NT, true>(acc, a, b, y0 + wm * MT, x0, wn * 32, red, lds + wv * (NSG * BUF / 4));
}

// ---------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 -> MFMA fragment order, T elements.
//   generic: idx = ((((nt*nstage + sg)*9 + tap)*2 + kg)*64 + lane)*EPL + j
//            cout = nt*32 + (lane&31), cin = sg*KC + kg*2*EPL + (lane>>5)*EPL + j      (KC = 64 B of channels)
//   conv0:   idx = ((nt*NFRAG + f)*64 + lane)*EPL + j,  k = f*2*EPL + (lane>>5)*EPL + j  (k = cin*9+tap, <27)
//   Every layer's buffer is padded by two taps of fragments (the kernels prefetch up to 2 taps ahead).
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int first) {
  constexpr int EPL = 16 / sizeof(T), KC = SB / sizeof(T), NFRAG = 32 / (2 * EPL);
  const size_t total = first ? (size_t)(Cout / 32) * NFRAG * 64 * EPL : (size_t)Cout * Cin * 9;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    size_t r = e;
    const int j = r % EPL; r /= EPL;
    const int lane = r % 64; r /= 64;
    float v = 0.f;
    if (first) {
      const int f = r % NFRAG; r /= NFRAG;
      const int nt = (int)r;
      const int k = f * 2 * EPL + (lane >> 5) * EPL + j, cout = nt * 32 + (lane & 31);
      if (k < 27) v = w[(size_t)cout * 27 + k];
    } else {
      const int kg = r % 2; r /= 2;
      const int tap = r % 9; r /= 9;
      const int nsg = Cin / KC;
      const int sg = r % nsg; r /= nsg;
      const int nt = (int)r;
      const int cout = nt * 32 + (lane & 31), cin = sg * KC + kg * 2 * EPL + (lane >> 5) * EPL + j;
      v = w[((size_t)cout * Cin + cin) * 9 + tap];
    }
    out[e] = (T)v;
  }
}

// ---------------------------------------------------------------------------------------------
// confidence head: sigmoid(-sigmoid(conv3x3(relu(x), C->1)))  VGG.py:62-81,160-163.  `act` is already ReLU'd.
template <typename T>
__global__ __launch_bounds__(256) void conf_kernel(const T* __restrict__ act, const float* __restrict__ w,
                                                   float* __restrict__ out, int B, int H, int W, int C) {
  constexpr int EPL = 16 / sizeof(T);
  extern __shared__ float ws[];   // [9][C]
  for (int e = threadIdx.x; e < 9 * C; e += 256) ws[(e % 9) * C + e / 9] = w[e];   // OIHW (O=1): w[c*9+tap]
  __syncthreads();
  const int G = C / EPL;                     // threads per pixel (8 channels bf16 / 4 fp32 each), power of 2 <= 64
  const int ppb = 256 / G;
  const size_t npix = (size_t)B * H * W;
  const int gi = threadIdx.x % G;
  for (size_t pix = (size_t)blockIdx.x * ppb + threadIdx.x / G; pix < (npix + ppb - 1) / ppb * ppb;
       pix += (size_t)gridDim.x * ppb) {
    float s = 0.f;
    const bool live = pix < npix;
    if (live) {
      const int x = (int)(pix % W), y = (int)((pix / W) % H);
      const size_t b = pix / ((size_t)W * H);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const uint4 raw = *(const uint4*)(act + ((b * H + yy) * W + xx) * C + gi * EPL);
        T e[EPL];
        __builtin_memcpy(e, &raw, 16);
#pragma unroll
        for (int k = 0; k < EPL; ++k) s += to_f32(e[k]) * ws[tap * C + gi * EPL + k];
      }
    }
    for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (live && gi == 0) {
      const float sg = 1.f / (1.f + __expf(-s));
      out[pix] = 1.f / (1.f + __expf(sg));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// L2_norm (VGG.py:511-514): x / max(||x||, 1e-12) per sample.
// inv_norm_kernel: fixed-order fp64 sum of the epilogue partials -> 1/max(||x||,1e-12) per sample.
__global__ __launch_bounds__(256) void inv_norm_kernel(const double* __restrict__ sumsq, int np, double* __restrict__ inv) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += sumsq[(size_t)b * np + i];
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) inv[b] = 1.0 / fmax(sqrt((sh[0] + sh[1]) + (sh[2] + sh[3])), 1e-12);
}
// scale_kernel: in-place x *= inv[b] (fp64 multiply, one rounding)
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, const double* __restrict__ inv, size_t per_sample,
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
// host side
struct LayerDef { int cin, cout, has_bias; };
static const LayerDef kLayers[13] = {
    {3, 64, 1}, {64, 64, 1}, {64, 128, 1}, {128, 128, 1}, {128, 256, 1}, {256, 256, 1}, {256, 256, 1},
    {384, 128, 0}, {128, 128, 0}, {192, 64, 0}, {64, 64, 0}, {128, 32, 0}, {32, 16, 0}};
constexpr int kPackedLayers = 11;   // conv0..dec2.3 (dec3.* only feed the unused x24 at level 3)

static size_t packed_bytes(int l, int dtype) {
  const size_t es = dtype == HLA_BF16 ? 2 : 4;
  if (l == 0) return (size_t)2 * 32 * 32 * es;   // 2 ntiles x 32 (padded K) x 32 couts
  return (size_t)kLayers[l].cin * kLayers[l].cout * 9 * es + 4096;   // + two taps of fragments: prefetch overrun
}
static size_t packed_offset(int l, int dtype) {
  size_t o = 0;
  for (int i = 0; i < l; ++i) o += hla_align_up(packed_bytes(i, dtype), 256);
  return o;
}

extern "C" size_t hla_vgg_packed_weight_bytes(int dtype) { return packed_offset(kPackedLayers, dtype); }

template <typename T>
static void pack_all(const hla_vgg_params* prm, char* packed, int dtype, hipStream_t st) {
  for (int l = 0; l < kPackedLayers; ++l) {
    const size_t n = l == 0 ? (size_t)2 * 32 * 32 : (size_t)kLayers[l].cin * kLayers[l].cout * 9;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hla_prof_begin(K_PACK, 0, (double)n * (4 + sizeof(T)), st);
    hipLaunchKernelGGL((pack_weights_kernel<T>), dim3(grid), dim3(256), 0, st, prm->w[l],
                       (T*)(packed + packed_offset(l, dtype)), kLayers[l].cout, kLayers[l].cin, l == 0 ? 1 : 0);
    hla_prof_end(st);
  }
}

extern "C" int hla_vgg_pack_weights(const hla_vgg_params* params, void* packed, int dtype, hla_stream_t stream) {
  HLA_REQUIRE(params && packed, "hla_vgg_pack_weights: null argument");
  HLA_REQUIRE(dtype == HLA_F32 || dtype == HLA_BF16, "hla_vgg_pack_weights: bad dtype");
  if (dtype == HLA_BF16) pack_all<bf16>(params, (char*)packed, dtype, (hipStream_t)stream);
  else pack_all<float>(params, (char*)packed, dtype, (hipStream_t)stream);
  HLA_CHECK_HIP(hipGetLastError());
  return HLA_OK;
}

struct VggPlan {
  size_t x3, a5, x8, a10, a12, x15r, d1a, x18r, d2a, x21r;
  size_t ss[3], inv;
  int np[3];
  size_t total;
};

static void vgg_plan(int B, int H, int W, int dtype, VggPlan* p) {
  const size_t es = dtype == HLA_BF16 ? 2 : 4;
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
  for (int i = 0; i < 3; ++i) p->ss[i] = take((size_t)B * p->np[i] * sizeof(double));
  p->inv = take((size_t)3 * B * sizeof(double));
  p->total = o;
}

extern "C" size_t hla_vgg_workspace_bytes(int B, int H, int W, int level, int dtype) {
  (void)level;
  VggPlan p;
  vgg_plan(B, H, W, dtype, &p);
  return p.total;
}

template <typename T>
static void launch_conv(hipStream_t st, ConvArgs a, bool pool) {
  a.tiles_x = (a.W + 31) / 32;
  a.tiles_y = (a.H + 7) / 8;
  const dim3 grid(a.tiles_x * a.tiles_y * a.B, a.Cout >= 128 ? a.Cout / 128 : 1);
  const size_t es = sizeof(T), P = (size_t)a.B * a.H * a.W, Po = pool ? P / 4 : P;
  const double flops = 2.0 * 9.0 * (a.C1 + a.C2) * a.Cout * (double)P;
  const double bytes = (double)P * ((a.up1 ? a.C1 / 4.0 : a.C1) + a.C2) * es + (double)Po * a.Cout * ((a.out_act ? es : 0) + (a.out_raw ? 4 : 0));
#if CONV_VARIANT == 40
  static unsigned long long* dbg = nullptr;
  static int n_reported = 0;
  if (!dbg) (void)hipMallocManaged((void**)&dbg, (size_t)1 << 26);
  a.dbg = dbg;
#endif
  hla_prof_begin(a.Cout >= 128 ? (pool ? K_CONV_NT2_POOL : K_CONV_NT2) : (pool ? K_CONV_NT1_POOL : K_CONV_NT1), flops, bytes, st);
  // Cout >= 128: block = 8x32 pixels x 128 channels, waves 2(M) x 2(N), wave tile 128 px x 64 ch, weights 1 tap ahead
  // Cout == 64 : block = 8x32 pixels x  64 channels, waves 2 x 2,       wave tile 128 px x 32 ch, weights 2 taps ahead
  if (a.Cout >= 128) {
    if (pool) hipLaunchKernelGGL((conv3x3_kernel<T, 4, 2, 2, 2, true, 1, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_kernel<T, 4, 2, 2, 2, false, 1, false>), grid, dim3(256), 0, st, a);
  } else {
    if (pool) hipLaunchKernelGGL((conv3x3_kernel<T, 4, 1, 2, 2, true, 2, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_kernel<T, 4, 1, 2, 2, false, 2, true>), grid, dim3(256), 0, st, a);
  }
  hla_prof_end(st);
#if CONV_VARIANT == 40
  if (n_reported++ < 40) {
    (void)hipStreamSynchronize(st);
    const size_t nw = (size_t)grid.x * grid.y * 4;
    double m[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < nw; ++i) for (int k = 0; k < 5; ++k) m[k] += (double)dbg[i * 5 + k];
    const int nstage = (a.C1 + a.C2) / (int)(SB / sizeof(T));
    fprintf(stderr, "[dbg] Cin %d Cout %d H %d pool %d stages %d | per wave cycles: prologue %.0f, mma/stage %.0f (ideal 4608 alone), "
            "write/stage %.0f, barrier/stage %.0f, epilogue %.0f, total %.0f\n", a.C1 + a.C2, a.Cout, a.H, (int)pool, nstage,
            m[0] / nw, m[1] / nw / nstage, m[2] / nw / nstage, m[3] / nw / nstage, m[4] / nw,
            (m[0] + m[1] + m[2] + m[3] + m[4]) / nw);
  }
#endif
}

template <typename T>
static int vgg_forward_t(const float* x, const hla_vgg_params* prm, const char* packed, int dtype, float* const feat[4],
                         float* const conf[4], double* inv_norm, char* ws, const VggPlan& pl, int B, int H, int W,
                         int flags, hipStream_t st) {
  auto W_ = [&](int l) { return (const uint4*)(packed + packed_offset(l, dtype)); };
  char* w = ws;
  // conv0 + conv2 + pool fused (VGG.py:123-128): relu(x3)
  {
    Conv02Args a{};
    a.x = x; a.w0 = W_(0); a.b0 = prm->b[0]; a.w2 = W_(1); a.b2 = prm->b[1]; a.out_act = w + pl.x3;
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
  auto conv = [&](int l, const void* s1, int C1, int H_, int W_h, void* act, int relu, bool pool, const void* s2 = nullptr,
                  int C2 = 0, int up1 = 0, float* raw = nullptr, double* ss = nullptr) {
    ConvArgs a{};
    a.src1 = s1; a.src2 = s2; a.C1 = C1; a.C2 = C2; a.up1 = up1; a.wpk = W_(l);
    a.bias = kLayers[l].has_bias ? prm->b[l] : nullptr;
    a.out_act = act; a.out_raw = raw; a.sumsq = ss; a.B = B; a.H = H_; a.W = W_h; a.Cout = kLayers[l].cout;
    a.relu_act = relu;
    launch_conv<T>(st, a, pool);
  };
  // encoder (VGG.py:129-141).  ReLU commutes with max-pool, so pooled maps are stored post-ReLU.
  conv(2, w + pl.x3, 64, H / 2, W / 2, w + pl.a5, 1, false);                          // conv5
  conv(3, w + pl.a5, 128, H / 2, W / 2, w + pl.x8, 1, true);                          // conv7 + pool -> relu(x8)
  conv(4, w + pl.x8, 128, H / 4, W / 4, w + pl.a10, 1, false);                        // conv10
  conv(5, w + pl.a10, 256, H / 4, W / 4, w + pl.a12, 1, false);                       // conv12
  conv(6, w + pl.a12, 256, H / 4, W / 4, w + pl.x15r, 1, true, nullptr, 0, 0, feat[0],
       (double*)(w + pl.ss[0]));                                                      // conv14 + pool -> x15
  // decoder (VGG.py:144-151): conv(relu(cat(up(a), skip))) with both inputs stored post-ReLU
  conv(7, w + pl.x15r, 256, H / 4, W / 4, w + pl.d1a, 1, false, w + pl.x8, 128, 1);   // dec1.1
  conv(8, w + pl.d1a, 128, H / 4, W / 4, w + pl.x18r, 1, false, nullptr, 0, 0, feat[1],
       (double*)(w + pl.ss[1]));                                                      // dec1.3 -> x18
  conv(9, w + pl.x18r, 128, H / 2, W / 2, w + pl.d2a, 1, false, w + pl.x3, 64, 1);    // dec2.1
  conv(10, w + pl.d2a, 64, H / 2, W / 2, w + pl.x21r, 1, false, nullptr, 0, 0, feat[2],
       (double*)(w + pl.ss[2]));                                                      // dec2.3 -> x21
  // confidence heads on the ReLU'd maps
  if ((flags & HLA_VGG_WANT_CONF) && conf) {
    const T* acts[3] = {(const T*)(w + pl.x15r), (const T*)(w + pl.x18r), (const T*)(w + pl.x21r)};
    const int Cs[3] = {256, 128, 64}, hs[3] = {H / 8, H / 4, H / 2}, wsz[3] = {W / 8, W / 4, W / 2};
    for (int l = 0; l < 3; ++l) {
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
    const size_t per[3] = {(size_t)(H / 8) * (W / 8) * 256, (size_t)(H / 4) * (W / 4) * 128, (size_t)(H / 2) * (W / 2) * 64};
    double* inv = inv_norm ? inv_norm : (double*)(w + pl.inv);
    for (int l = 0; l < 3; ++l) {
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

extern "C" int hla_vgg_forward(const float* x, const hla_vgg_params* params, const void* packed_weights,
                               float* const feat[4], float* const conf[4], double* inv_norm, void* workspace,
                               size_t workspace_bytes, int B, int H, int W, int level, int dtype, int flags,
                               hla_stream_t stream) {
  HLA_REQUIRE(x && params && packed_weights && feat && workspace, "hla_vgg_forward: null argument");
  HLA_REQUIRE(dtype == HLA_F32 || dtype == HLA_BF16, "hla_vgg_forward: dtype must be HLA_F32 or HLA_BF16");
  HLA_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, "hla_vgg_forward: H and W must be multiples of 8");
  HLA_REQUIRE(level == 3, "hla_vgg_forward: only level 3 (x15,x18,x21) is built so far (got %d)", level);
  HLA_REQUIRE(feat[0] && feat[1] && feat[2], "hla_vgg_forward: level 3 needs feat[0..2]");
  HLA_REQUIRE(!(flags & HLA_VGG_DEFER_NORM) || inv_norm, "hla_vgg_forward: HLA_VGG_DEFER_NORM needs inv_norm");
  VggPlan pl;
  vgg_plan(B, H, W, dtype, &pl);
  if (workspace_bytes < pl.total) {
    hla_set_error("hla_vgg_forward: workspace %zu < %zu", workspace_bytes, pl.total);
    return HLA_ERR_WORKSPACE;
  }
  if (dtype == HLA_BF16)
    return vgg_forward_t<bf16>(x, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                               B, H, W, flags, (hipStream_t)stream);
  return vgg_forward_t<float>(x, params, (const char*)packed_weights, dtype, feat, conf, inv_norm, (char*)workspace, pl,
                              B, H, W, flags, (hipStream_t)stream);
}
