// Optional per-launch timing (hipEvents on the launch stream) used by bench.py to price each kernel
// against its roofline.  Off by default; when off the hooks cost one predictable branch.
#include "common.h"
#include <mutex>
#include <vector>

namespace {
struct Rec { int id; double flops, bytes; hipEvent_t a, b; int live_slot, live_denom; };
std::vector<Rec> g_recs;
std::mutex g_mu;
bool g_on = false;
// data-dependent launches (the backward's trimmed walks): the live-tile count is only known on the device; it is copied into a
// pinned host slot on the launch stream, right behind the kernel that wrote it, and applied to the record's FLOPs at fetch time
constexpr int kLiveSlots = 4096;
int* g_live = nullptr;
int g_live_used = 0;
const char* kNames[HLA_PROF_NKERNELS] = {
    "pack_weights_kernel", "conv02_kernel", "conv3x3_kernel<MT4,NT2>", "conv3x3_kernel<MT4,NT2,pool>",
    "conv3x3_kernel<MT4,NT1>", "conv3x3_kernel<MT4,NT1,pool>", "conf_kernel", "inv_norm+scale_kernel",
    "lm_accum<256>", "lm_accum<128>", "lm_accum<64>", "lm_accum<16>", "lm_solve", "grid_sample_kernel", "lm_bwd_accum", "wgrad_kernel", "elementwise_bwd"};
}  // namespace

bool hla_prof_on() { return g_on; }

void hla_prof_begin(int id, double flops, double bytes, hipStream_t st) { hla_prof_begin_dyn(id, flops, bytes, st, nullptr, 0); }

// `dev_live` (device, or null): the number of tiles per sample this launch visits; `denom`: the tiles per sample of the dense walk.
// flops / bytes are the DENSE launch's; the record reports them scaled by live / denom (executed work).
void hla_prof_begin_dyn(int id, double flops, double bytes, hipStream_t st, const int* dev_live, int denom) {
  if (!g_on) return;
  Rec r{id, flops, bytes, nullptr, nullptr, -1, denom};
  std::lock_guard<std::mutex> lk(g_mu);
  if (dev_live && denom > 0) {
    if (!g_live && hipHostMalloc((void**)&g_live, kLiveSlots * sizeof(int), hipHostMallocDefault) != hipSuccess) g_live = nullptr;
    if (g_live && g_live_used < kLiveSlots) {
      r.live_slot = g_live_used++;
      g_live[r.live_slot] = denom;
      (void)hipMemcpyAsync(g_live + r.live_slot, dev_live, sizeof(int), hipMemcpyDeviceToHost, st);
    }
  }
  (void)hipEventCreate(&r.a);
  (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
}

void hla_prof_end(hipStream_t st) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, st);
}

extern "C" int hla_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return HLA_OK;
}

extern "C" const char* hla_prof_kernel_name(int id) { return (id >= 0 && id < HLA_PROF_NKERNELS) ? kNames[id] : "?"; }

extern "C" int hla_prof_fetch(hla_prof_record* out, int max_records, int* n_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess) (void)hipEventElapsedTime(&ms, r.a, r.b);
    double frac = 1.0;
    if (r.live_slot >= 0 && g_live) {
      const int live = g_live[r.live_slot];
      frac = live < 0 ? 0.0 : (live > r.live_denom ? 1.0 : (double)live / r.live_denom);
    }
    if (out && n < max_records) { out[n].kernel_id = r.id; out[n].flops = r.flops * frac; out[n].bytes = r.bytes * frac; out[n].ms = ms; ++n; }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  g_live_used = 0;
  if (n_out) *n_out = n;
  return HLA_OK;
}

#if HLA_CONV_STAMPS
HlaStampCfg g_hla_stamp = {nullptr, -1, 0, 0, 0};
// tooling builds only: arm the stamps for the `want`-th launch_conv call from now on (buf: device memory, HLA_STAMP_N * 8 bytes per
// wave of that launch); grid_out (2 ints, may be null) receives that launch's grid after it ran
extern "C" int hla_debug_conv_stamps(void* buf, int want) { g_hla_stamp.buf = (unsigned long long*)buf; g_hla_stamp.want = want; g_hla_stamp.counter = 0; return 0; }
extern "C" int hla_debug_conv_stamps_grid(unsigned* gx, unsigned* gy) { *gx = g_hla_stamp.grid_x; *gy = g_hla_stamp.grid_y; return 0; }
#endif

// ---------------------------------------------------------------------------------------------
// hla_prof_mfma_peak: what the matrix pipe of THIS chip sustains -- back-to-back v_mfma_f32_32x32x16_{bf16,f16} on
// register-resident operands, two waves per SIMD (the conv kernels' occupancy), every CU busy for `ms_target` milliseconds;
// no LDS, no memory traffic.  The nominal 2.5 PFLOP/s is reached on ZERO operands only: on random data the package power limit
// holds the clock near 1.65 GHz under this load alone (measured: 2490 / 1722 / 1915 TFLOP/s for zeros / random / random with
// half the elements zero, i.e. post-ReLU-like activations).  bench.py reports it next to roofline.peak, so that the fraction of
// the nominal peak can be read against what any MFMA-bound kernel could reach on this box, today.
typedef __attribute__((ext_vector_type(8))) __bf16 pk_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 pk_f16x8;
typedef __attribute__((ext_vector_type(16))) float pk_f32x16;
template <int F16>
static __global__ __launch_bounds__(256, 2) void mfma_burn_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = threadIdx.x;
  uint4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 4095]; b[i] = src[(t * 8 + 4 + i) & 4095]; }
  pk_f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (F16)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pk_f16x8, a[i & 3]), __builtin_bit_cast(pk_f16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
      else
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pk_bf16x8, a[i & 3]), __builtin_bit_cast(pk_bf16x8, b[(i >> 1) & 3]), acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[blockIdx.x * 256 + t] = s;      // (keeps the loop alive; never true for these operands)
}

extern "C" int hla_prof_mfma_peak(int dtype, int data, float ms_target, float* tflops_out, hla_stream_t stream) {
  HLA_REQUIRE(tflops_out && (dtype == HLA_BF16 || dtype == HLA_F16) && data >= 0 && data <= 2 && ms_target > 0.f && ms_target <= 200.f,
              "hla_prof_mfma_peak: dtype HLA_BF16 / HLA_F16, data 0 (zeros) / 1 (random) / 2 (random, half zeros), 0 < ms_target <= 200");
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, cus = 0;
  HLA_CHECK_HIP(hipGetDevice(&dev));
  HLA_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = cus * 2;
  uint4* src = nullptr;
  float* out = nullptr;
  HLA_CHECK_HIP(hipMalloc(&src, 4096 * sizeof(uint4)));
  {
    const hipError_t em = hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    if (em != hipSuccess) { (void)hipFree(src); HLA_CHECK_HIP(em); }
  }
  std::vector<unsigned> h(4096 * 4);
  unsigned long long x = 88172645463325252ull;      // xorshift: the same operands on every box
  for (auto& w : h) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    unsigned lo = (unsigned)x & 0xffffu, hi = (unsigned)(x >> 16) & 0xffffu;
    lo = (lo & 0x83ffu) | 0x3800u; hi = (hi & 0x83ffu) | 0x3800u;        // +-0.5..1 as fp16, small normal numbers as bf16
    w = data == 0 ? 0u : (lo | (hi << 16));
    if (data == 2 && ((x >> 40) & 1)) w = 0u;
  }
  hipError_t e = hipMemcpyAsync(src, h.data(), h.size() * 4, hipMemcpyHostToDevice, st);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float tf = 0.f;
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  // 8 MFMAs of 2 * 32 * 32 * 16 flops per wave and iteration at the nominal 2.5 PF: iterations for ms_target at that rate;
  // a first short launch warms the clocks up, the second is the measurement
  const double flops_it = (double)grid * 4 * 8 * 2.0 * 32 * 32 * 16;
  const int iters = (int)(2.5e15 * (ms_target * 1e-3) / flops_it) + 1;
  for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {
    (void)hipEventRecord(e0, st);
    if (dtype == HLA_F16) hipLaunchKernelGGL(mfma_burn_kernel<1>, dim3(grid), dim3(256), 0, st, src, out, rep ? iters : iters / 4 + 1);
    else hipLaunchKernelGGL(mfma_burn_kernel<0>, dim3(grid), dim3(256), 0, st, src, out, rep ? iters : iters / 4 + 1);
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess && rep == 1 && ms > 0.f) tf = (float)(flops_it * iters / (ms * 1e-3) / 1e12);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(out);
  HLA_CHECK_HIP(e);
  *tflops_out = tf;
  return HLA_OK;
}
