// Optional per-launch timing (hipEvents on the launch stream) used by bench.py to price each kernel
// against its roofline.  Off by default; when off the hooks cost one predictable branch.
#include "common.h"
#include <mutex>
#include <vector>

namespace {
struct Rec { int id; double flops, bytes; hipEvent_t a, b; };
std::vector<Rec> g_recs;
std::mutex g_mu;
bool g_on = false;
const char* kNames[HLA_PROF_NKERNELS] = {
    "pack_weights_kernel", "conv02_kernel", "conv3x3_kernel<MT4,NT2>", "conv3x3_kernel<MT4,NT2,pool>",
    "conv3x3_kernel<MT4,NT1>", "conv3x3_kernel<MT4,NT1,pool>", "conf_kernel", "inv_norm+scale_kernel",
    "lm_accum<256>", "lm_accum<128>", "lm_accum<64>", "lm_accum<16>", "lm_solve", "grid_sample_kernel", "lm_bwd_accum", "wgrad_kernel", "elementwise_bwd"};
}  // namespace

bool hla_prof_on() { return g_on; }

void hla_prof_begin(int id, double flops, double bytes, hipStream_t st) {
  if (!g_on) return;
  Rec r{id, flops, bytes, nullptr, nullptr};
  (void)hipEventCreate(&r.a);
  (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, st);
  std::lock_guard<std::mutex> lk(g_mu);
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
    if (out && n < max_records) { out[n].kernel_id = r.id; out[n].flops = r.flops; out[n].bytes = r.bytes; out[n].ms = ms; ++n; }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  if (n_out) *n_out = n;
  return HLA_OK;
}
