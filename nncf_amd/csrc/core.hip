// Error state, version and device probing for libnncf_hip.so.
#include <cstring>

#include "common.h"

namespace nncf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return status;
}

}  // namespace nncf

extern "C" {

const char* nncf_last_error(void) { return nncf::g_err; }

const char* nncf_version(void) { return "nncf_hip 0.1 (gfx950)"; }

int nncf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
