// Shared host-side helpers of libspmi: error reporting and launch checks.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
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

namespace spmi {

// A kernel's dynamic-LDS limit is a property of the kernel ON ONE DEVICE: a process that
// drives several GPUs has to raise it on each.  One LdsLimit per kernel (a function-local
// static at the launch site) remembers what has been set where.
constexpr int kLdsLimitDevices = 64;
struct LdsLimit {
  std::atomic<size_t> bytes[kLdsLimitDevices] = {};
};

inline int raise_lds_limit(const void* kernel, size_t bytes, LdsLimit& seen) {
  int dev = 0;
  SPMI_CHECK_HIP(hipGetDevice(&dev));
  SPMI_REQUIRE(dev >= 0 && dev < kLdsLimitDevices, SPMI_ERR_UNSUPPORTED,
               "device ordinal %d beyond %d", dev, kLdsLimitDevices);
  if (seen.bytes[dev].load(std::memory_order_acquire) >= bytes) return SPMI_OK;
  SPMI_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(bytes)));
  seen.bytes[dev].store(bytes, std::memory_order_release);
  return SPMI_OK;
}

}  // namespace spmi

#define SPMI_RAISE_LDS_LIMIT(kernel, bytes)                                              \
  do {                                                                                   \
    static ::spmi::LdsLimit seen_;                                                       \
    if (int rc_ = ::spmi::raise_lds_limit(reinterpret_cast<const void*>(kernel), (bytes), seen_)) \
      return rc_;                                                                        \
  } while (0)
