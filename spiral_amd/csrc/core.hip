// Library-wide state of libspmi: version and the thread-local error string.
#include "common.h"

namespace spmi {
char* error_buffer() {
  static thread_local char buf[kErrorBufferBytes] = {0};
  return buf;
}
}  // namespace spmi

extern "C" {
int spmi_version(void) { return 101; }  // 0.1.1
const char* spmi_last_error(void) { return spmi::error_buffer(); }
}
