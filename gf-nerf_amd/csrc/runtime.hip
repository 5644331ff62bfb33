// runtime.hip -- error reporting and device probing for the C ABI.
#include <stdarg.h>
#include <string.h>

#include "gfn_common.h"

namespace {
thread_local char g_err[512] = "";
}

void gfn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* gfn_last_error(void) { return g_err; }

int gfn_version(void) { return 100; /* 0.1.0 */ }

int gfn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

}  // extern "C"
