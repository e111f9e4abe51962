// libhla: error reporting + ABI version.
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
extern "C" int hla_abi_version(void) { return 12; }
