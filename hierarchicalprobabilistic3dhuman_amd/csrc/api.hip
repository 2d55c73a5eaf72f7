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

extern "C" int hps_version(void) { return 502; }  // 0.5.2: + hps_smpl_*_bf16x3; 0.5.1: + hps_stem_winograd_pooled (0.5.0: second-generation pose-prep / joint kernels)

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
#ifdef HPS_DEV_BUILD
        case HPS_DEV_WS_HEAD_SYNC: return ((d0 + 3) / 4) * (HPS_HEAD_MAX_LEVELS + 1) * (int64_t)sizeof(int32_t);
#endif
        default: hps::set_error("hps_query_workspace: unknown item %d", what); return -1;
    }
}
extern "C" const char* hps_last_error(void) { return hps::g_err; }

// A HIP stream whose kernels may only run on CUs [first_cu, first_cu + num_cus) OF EVERY XCD (hipExtStreamCreateWithCUMask).
// Mask bit i addresses CU i / n_xcd of XCD i % n_xcd (probed on MI355X: a mask that leaves an XCD without CUs is ignored by the
// runtime), so a partition is always "the same CU subset on each of the 8 XCDs": both sides keep the block -> XCD round robin and
// their own share of every L2.
extern "C" int hps_stream_create_cu_partition(int first_cu, int num_cus, hps_stream_t* stream_out) {
    if (!stream_out) return hps::bad_arg("hps_stream_create_cu_partition: null pointer");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { hps::set_error("hps_stream_create_cu_partition: %s", hipGetErrorString(e)); return (int)e; }
    const int total = prop.multiProcessorCount, n_xcd = 8;
    const int per_xcd = total / n_xcd;
    if (total % n_xcd != 0 || first_cu < 0 || num_cus < 1 || first_cu + num_cus > per_xcd)
        return hps::bad_arg("hps_stream_create_cu_partition: CU range outside [0, CUs per XCD)");
    uint32_t mask[16] = {0};
    for (int c = first_cu; c < first_cu + num_cus; ++c)
        for (int x = 0; x < n_xcd; ++x) {
            const int bit = c * n_xcd + x;
            mask[bit / 32] |= 1u << (bit % 32);
        }
    hipStream_t s = nullptr;
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
    if (e != hipSuccess) { hps::set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return (int)e; }
    *stream_out = (hps_stream_t)s;
    return HPS_OK;
}

extern "C" int hps_stream_destroy(hps_stream_t stream) {
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { hps::set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return (int)e; }
    return HPS_OK;
}
