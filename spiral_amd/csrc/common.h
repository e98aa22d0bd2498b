// Shared host-side helpers of libspmi: error reporting and launch checks.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/spmi.h"

namespace spmi {

// Thread-local message behind spmi_last_error().
char* error_buffer();
constexpr size_t kErrorBufferBytes = 512;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), kErrorBufferBytes, fmt, ap);
  va_end(ap);
  return code;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace spmi

#define SPMI_REQUIRE(cond, code, ...)            \
  do {                                           \
    if (!(cond)) return ::spmi::fail((code), __VA_ARGS__); \
  } while (0)

// Check the launch itself (bad configuration, missing code object); execution
// faults surface at the caller's next synchronisation, as with any HIP work.
#define SPMI_CHECK_LAUNCH(what)                                                        \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess)                                                              \
      return ::spmi::fail(SPMI_ERR_HIP, "%s: launch failed: %s", (what), hipGetErrorString(e_)); \
  } while (0)

#define SPMI_CHECK_HIP(expr)                                                           \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return ::spmi::fail(SPMI_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
