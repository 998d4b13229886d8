// Error convention of the nasseg C-ABI (include/nasseg.h): every entry point
// returns 0 or a negative code and leaves a per-thread message behind. The
// Python binding turns a non-zero return into RuntimeError, which is what the
// reference's try_except wrapper (src/helpers/utils.py:172-187) expects from a
// candidate that cannot be evaluated.
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>

#include "common.h"

static thread_local char g_err[512] = "";

int nasseg_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" {

const char* nasseg_last_error(void) { return g_err; }

int nasseg_abi_version(void) { return 1; }

// Number of HIP devices visible to the library; negative on runtime failure.
int nasseg_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return nasseg_fail(NASSEG_ERR_LAUNCH, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  return n;
}

}  // extern "C"
