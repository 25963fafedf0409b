// C-ABI plumbing: version, thread-local error text, launch check.  See include/aladin_hip.h.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/aladin_hip.h"
#include "common.hpp"

static thread_local char g_err[512] = "";

void aladin_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int aladin_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return ALADIN_OK;
  aladin_set_error("%s: %s", what, hipGetErrorString(e));
  return ALADIN_ERR_HIP;
}

extern "C" int aladin_version(void) { return ALADIN_ABI_VERSION; }
extern "C" const char* aladin_last_error(void) { return g_err; }
