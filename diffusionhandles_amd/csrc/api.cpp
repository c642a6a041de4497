// Error state, version and device probes of the C ABI.
#include "common.h"

namespace dh {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
}  // namespace dh

extern "C" const char* dh_last_error(void) { return dh::g_err.c_str(); }
extern "C" int dh_version(void) { return 100; }
extern "C" int dh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
