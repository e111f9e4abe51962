// libhla: error reporting, ABI version and the self-description a binding uses to detect a stale or mismatched binary.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void hla_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* hla_last_error(void) { return g_err; }
extern "C" int hla_abi_version(void) { return HLA_ABI_VERSION; }

// sha256 of the sources this binary was built from (highlyaccurate_amd/build.py source_hash()); the "HLA_SOURCE_HASH="
// prefix lets build.lib_hash() find it in the file without loading the library
#ifndef HLA_SOURCE_HASH_HEX
#define HLA_SOURCE_HASH_HEX "unknown"
#endif
static const char kSourceHash[] = "HLA_SOURCE_HASH=" HLA_SOURCE_HASH_HEX;
extern "C" const char* hla_source_hash(void) { return kSourceHash + 16; }

// sizeof of every struct that crosses the boundary, in the order of hla_struct_id
extern "C" size_t hla_sizeof_struct(int id) {
  switch (id) {
    case HLA_STRUCT_VGG_PARAMS: return sizeof(hla_vgg_params);
    case HLA_STRUCT_VGG_GRADS: return sizeof(hla_vgg_grads);
    case HLA_STRUCT_S2G_LEVEL: return sizeof(hla_s2g_level);
    case HLA_STRUCT_S2G_CONFIG: return sizeof(hla_s2g_config);
    case HLA_STRUCT_S2G_LEVEL_GRAD: return sizeof(hla_s2g_level_grad);
    case HLA_STRUCT_PROF_RECORD: return sizeof(hla_prof_record);
    case HLA_STRUCT_POSE_LOSS_ARGS: return sizeof(hla_pose_loss_args);
    case HLA_STRUCT_FILL_REGION: return sizeof(hla_fill_region);
    default: return 0;
  }
}
