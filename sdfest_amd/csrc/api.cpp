// api.cpp -- version + per-thread error string of libsdfr_hip.so.
#include <cstdarg>
#include <cstdio>

#include <hip/hip_runtime_api.h>

#include "common.hpp"

namespace sdfr {
namespace {
thread_local char g_error[512] = "";
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  snprintf(g_error, sizeof(g_error), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return (int)e;
}
}  // namespace sdfr

extern "C" int sdfr_version(void) { return SDFR_VERSION; }
extern "C" const char* sdfr_last_error(void) { return sdfr::g_error; }
