// C-ABI plumbing: version, thread-local error text, launch check.  See include/aladin_hip.h.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

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

int aladin_reserve_lds(const void* kernel, int bytes, unsigned long long* done, const char* what) {
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;      // slot 63 is never marked: always re-set
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 63 && (*done >> dev) & 1ull) return ALADIN_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    aladin_set_error("%s: cannot reserve %d B of LDS on device %d", what, bytes, dev);
    return ALADIN_ERR_HIP;
  }
  if (dev < 63) *done |= 1ull << dev;
  return ALADIN_OK;
}
