// libdiqt_hip.so: version + thread-local error string.
#include "common.h"
#include <string.h>

namespace diqt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace diqt

extern "C" int diqt_version(void) { return 100; }   // 0.1.0
extern "C" const char* diqt_last_error(void) { return diqt::g_err; }
