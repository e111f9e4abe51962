// EXPERIMENT (CONV_VARIANT 100-104), included by conv_kernels.h only for those builds.  Negative result, kept because it is
// correct, parity-tested and carries the cycle accounting that explains why.
#pragma once

// ---------------------------------------------------------------------------------------------
// conv3x3_lw_kernel -- same tile, loader and epilogue as conv3x3_kernel, but the weight fragments reach the MFMAs through
// LDS: per tap the block's WN*NT*2 fragments (1 KiB each, already in lane order) are copied global -> LDS by
// global_load_lds_dwordx4 (no registers, 2 wave-instructions per wave per tap) into a 3-slot ring two taps ahead, and every
// wave reads its NT*2 fragments with ds_read_b128 right before use.  Measured motivation (CONV_VARIANT 97/98): feeding
// the A operand from LDS instead of through L1 -> VGPR is worth 28 % on this kernel, one workgroup barrier per tap costs 3 %.
template <typename T, int MT, int NT, int WM, int WN, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_lw_kernel(ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int EPL = 16 / sizeof(T), KC = SB / sizeof(T);
  constexpr int TH = WM * MT, HPIX = (TH + 2) * HWID, BUF = HPIX * PSTR;
  constexpr int NPIECE = (HPIX * 4 + 255) / 256;
  constexpr int NFRAG = WN * NT * 2, FPW = NFRAG / 4, WSLOT = NFRAG * 1024, NSLOT = 3;
  static_assert(NFRAG % 4 == 0, "every wave copies the same number of fragments");
  extern __shared__ __attribute__((aligned(16))) char lds[];     // [2*BUF halo double buffer][NSLOT*WSLOT weight ring]
  __shared__ float red[4];
  char* wl = lds + 2 * BUF;

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv / WN, wn = wv % WN;
  int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * 32;
  const int nstage = (a.C1 + a.C2) / KC, ntap = nstage * 9;
  const int part = t & 3, pbase = t >> 2;

  // Exactly NPIECE load instructions per wave and stage, whatever the tile position, so that the hand-counted vmcnt below
  // is exact: a piece outside the image (or past the tile) is loaded from the nearest pixel inside and zeroed when it is
  // written to LDS, instead of being predicated off.  (Launches that need the unpool mask use conv3x3_kernel.)
  unsigned okmask = 0;                  // bit i: piece i of the stage in flight is a real pixel
  auto load_stage = [&](int sg, uint4 (&st)[NPIECE]) {
    const int c0 = sg * KC;
    const bool first = c0 < a.C1;
    const T* src = first ? (const T*)a.src1 : (const T*)a.src2;
    const int Cs = first ? a.C1 : a.C2;
    const int coff = (first ? c0 : c0 - a.C1) + part * EPL;
    const int sh = (first && a.up1) ? 1 : 0;
    const int Hs = a.H >> sh, Ws = a.W >> sh;
    okmask = 0;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int pix = pbase + 64 * i;
      const int hy = pix / HWID, hx = pix - hy * HWID;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      if (pix < HPIX && y >= 0 && y < a.H && x >= 0 && x < a.W) okmask |= 1u << i;
      const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
      st[i] = *(const uint4*)(src + (((size_t)b * Hs + (yc >> sh)) * Ws + (xc >> sh)) * Cs + coff);
    }
  };
  auto write_stage = [&](char* buf, const uint4 (&st)[NPIECE]) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int pix = pbase + 64 * i;
      if (pix < HPIX) *(uint4*)(buf + pix * PSTR + part * 16) = (okmask >> i) & 1 ? st[i] : make_uint4(0, 0, 0, 0);
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
  const int ntg0 = (blockIdx.y * WN + wn) * NT;
  // this wave's share of every tap's fragment slab: fragment fl = wv*FPW + q  <->  (block-local cout tile fl/2, k-group fl%2);
  // packed weights are [cout tile][stage][tap][kg][lane] x 16 B, i.e. linear in the global tap counter (stride 128 units)
  const uint4* wsrc[FPW];
#pragma unroll
  for (int q = 0; q < FPW; ++q) {
    const int fl = wv * FPW + q;
    wsrc[q] = a.wpk + ((size_t)(blockIdx.y * WN * NT + fl / 2) * ntap * 2 + (fl & 1)) * 64 + lane;
  }
  // The copies are issued from inline asm so that hipcc's waitcnt pass does not see them (it would otherwise drain vmcnt(0)
  // before the next ds_read, cdna_hip_programming.md "What hipcc does not do"); their completion is counted by hand below.
  const unsigned wl_addr = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) char*)(wl + wv * FPW * 1024));
  auto dma = [&](int gt, int slot) {
#pragma unroll
    for (int q = 0; q < FPW; ++q) {
      const uint4* src = wsrc[q] + (CONV_VARIANT == 103 ? 0 : (size_t)gt * 128);   // (ablation 103: always the same 1 KiB)
      const unsigned dst = wl_addr + slot * WSLOT + q * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  };

  uint4 st[NPIECE];
#if CONV_VARIANT == 104      // per-wave cycle accounting: [prologue, wait+barrier, issue (copies + halo loads), ds_read+mma, halo write, epilogue]
  unsigned long long tc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_readcyclecounter();
#define LTICK(k) { const unsigned long long _n = __builtin_readcyclecounter(); tc[k] += _n - t_prev; t_prev = _n; }
#else
#define LTICK(k)
#endif
  load_stage(0, st);
  write_stage(lds, st);
  dma(0, 0);
  dma(1, 1);                                   // ntap >= 9
  stagger_priority();
  LTICK(0)

  for (int sg = 0; sg < nstage; ++sg) {
    const bool more = sg + 1 < nstage;
    const char* cur = lds + (sg & 1) * BUF + aoff;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int gt = sg * 9 + tap;             // slot = gt % 3 = tap % 3
      // all waves: my copies for this tap have landed (the barrier's release waits vmcnt(0)) -> visible to everyone;
      // the halo tile written at the end of the previous stage is visible; slot (tap+2)%3 is no longer being read
      // vmcnt(FPW): everything but the FPW youngest VM operations of this wave has completed -- the copies for tap gt
      // (issued two taps ago) certainly have, whatever else (tap gt+1's copies, halo loads) is still in flight; the very
      // last tap has nothing younger.  lgkmcnt(0): this wave's halo writes of the previous stage end are in LDS.
      // Wait for this wave's copies of tap gt (issued two taps ago).  VM operations retire in order, so "all but the K
      // youngest have completed" is exact when K = the operations issued after them: tap gt+1's FPW copies, plus the
      // NPIECE halo loads when those were issued in one of the last two taps (they then stay in flight until the stage
      // end, five taps after their issue).  The very last tap has nothing younger.  lgkmcnt(0): this wave's halo writes
      // of the previous stage end are in LDS before anyone passes the barrier.
      const bool halo_young = (tap == HALO_TAP + 1 || tap == HALO_TAP + 2);
      if (gt + 1 >= ntap) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if (halo_young && more && CONV_VARIANT != 101) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FPW + NPIECE) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FPW) : "memory");
      __builtin_amdgcn_s_barrier();
      LTICK(1)
      if (gt + 2 < ntap && CONV_VARIANT != 102) dma(gt + 2, (tap + 2) % NSLOT);         // (ablation 102: no weight copies)
      if (tap == HALO_TAP && more && CONV_VARIANT != 101) load_stage(sg + 1, st);     // (ablation 101: no halo loads)
      LTICK(2)
      const char* ws = wl + (tap % NSLOT) * WSLOT + (wn * NT) * 2048 + lane * 16;
      const char* ap = cur + ((tap / 3) * HWID + tap % 3) * PSTR;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        uint4 wf[NT], pf[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *(const uint4*)(ws + j * 2048 + kg * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i) pf[i] = *(const uint4*)(ap + i * HWID * PSTR + kg * 32);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma16<T>(acc[i][j], wf[j], pf[i]);
      }
      __builtin_amdgcn_sched_barrier(0);
      LTICK(3)
    }
    if (more && CONV_VARIANT != 101) write_stage(lds + ((sg + 1) & 1) * BUF, st);
    LTICK(4)
  }
  __syncthreads();                             // nobody reads the halo buffers any more: reuse them as 4 wave-private stagers
  static_assert(2 * BUF / 4 >= 32 * (NT * 32 * 4 + 16) && (2 * BUF / 4) % 16 == 0, "stager does not fit");
  conv_epilogue<T, MT, NT, POOL>(acc, a, b, y0 + wm * MT, x0, ntg0 * 32, red, lds + wv * (2 * BUF / 4));
#if CONV_VARIANT == 104
  LTICK(5)
  if (a.dbg && lane == 0) {
    unsigned long long* d = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 6;
    for (int k = 0; k < 6; ++k) d[k] = tc[k];
  }
#endif
#undef LTICK
}

template <typename T, int MT, int NT, int WM, int WN>
constexpr int conv_lw_lds_bytes() { return 2 * ((WM * MT + 2) * HWID * PSTR) + 3 * (WN * NT * 2 * 1024); }

