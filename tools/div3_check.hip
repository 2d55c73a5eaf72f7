// Exhaustive check of the short correctly rounded division the Canny kernel uses (csrc/hps_common.h: div3_rn) against the
// IEEE quotient, for every fp32 bit pattern:   ./div3_check   ->   "... values checked N, mismatches M" (exit code 1 on any mismatch)
#include <hip/hip_runtime.h>
#include "hps_common.h"        // hps::div3_rn -- the very function the kernel uses
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void check(unsigned long long* bad, unsigned long long* n) {
    const uint32_t bits0 = (uint32_t)(blockIdx.x * 256u + threadIdx.x) << 8;
    unsigned long long b = 0, c = 0;
    for (uint32_t i = 0; i < 256; ++i) {
        const uint32_t u = bits0 | i;
        float v;
        memcpy(&v, &u, 4);
        if (!(fabsf(v) <= 3.4028234663852886e38f)) continue;          // NaN / Inf
        const float f = hps::div3_rn(v);
        const float w = __fdiv_rn(v, 3.0f);
        uint32_t a, d;
        memcpy(&a, &f, 4);
        memcpy(&d, &w, 4);
        b += a != d;
        ++c;
    }
    atomicAdd(bad, b);
    atomicAdd(n, c);
}

int main() {
    unsigned long long *d, h[2] = {0, 0};
    hipMalloc(&d, 16);
    hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1u << 16), dim3(256), 0, 0, d, d + 1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("div3_rn: finite values checked %llu, mismatches %llu\n", h[1], h[0]);
    return h[0] != 0;
}
// (A five-instruction sqrt -- v_sqrt_f32, one residual correction through v_rcp_f32 -- was tried the same way: 289 041 642 of the
//  2 139 095 040 non-negative finite values came out one ulp off; the kernel keeps the library's sqrtf.)
