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

extern "C" int hps_version(void) { return 201; }  // 0.2.1: hps_conv3x3_winograd takes splitk_ws

extern "C" int64_t hps_query_workspace(int what, int64_t d0, int64_t d1, int64_t d2) {
    if (d0 < 0 || d1 < 0 || d2 < 0) { hps::set_error("hps_query_workspace: negative dimension"); return -1; }
    const int64_t f = (int64_t)sizeof(float);
    auto up = [](int64_t x, int64_t m) { return (x + m - 1) / m * m; };
    switch (what) {
        case HPS_WS_CONV_SPLITK: return d0 <= 1 ? 0 : d0 * d1 * d2 * f;
        case HPS_WS_SMPL_MP: return up(d0, 128);
        case HPS_WS_SMPL_XT: return d1 * up(d0, 128) * f;
        case HPS_WS_SMPL_A: return d0 * d1 * 12 * f;
        case HPS_WS_SMPL_VPOSED: return d0 * up(3 * d1, 128) * f;
        case HPS_WS_HEAD_F: return d0 * d1 * 9 * f;
        case HPS_WS_HEAD_USV: return d0 * d1 * 21 * f;
        default: hps::set_error("hps_query_workspace: unknown item %d", what); return -1;
    }
}
extern "C" const char* hps_last_error(void) { return hps::g_err; }
