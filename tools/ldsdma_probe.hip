// Dev probe: where does global_load_lds_dwordx3 (12 bytes per lane, gfx950) put a wave's data in LDS?
// Lane l loads the three floats {3l, 3l+1, 3l+2}; the kernel then dumps the first 256 dwords of LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* src, float* dump, int size) {
    __shared__ float lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -1.0f;
    __syncthreads();
    const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)(src + threadIdx.x * (size / 4));
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)lds;
    if (size == 12) __builtin_amdgcn_global_load_lds(g, l, 12, 0, 0);
    else if (size == 16) __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) dump[i] = lds[i];
}
int main() {
    float *src, *dump; hipMalloc(&src, 4096); hipMalloc(&dump, 2048);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    for (int size : {4, 12, 16}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, dump, size);
        float o[512]; hipMemcpy(o, dump, 2048, hipMemcpyDeviceToHost);
        printf("size %d:", size);
        for (int i = 0; i < 280; ++i) printf(" %g", o[i]);
        printf("\n");
    }
    return 0;
}
