// libhps.so bookkeeping entry points: version and thread-local error text.
#include <stdarg.h>

#include "hps_common.h"

namespace hps {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace hps

extern "C" int hps_version(void) { return 100; }  // 0.1.0
extern "C" const char* hps_last_error(void) { return hps::g_err; }
