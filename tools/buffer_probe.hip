#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k(float* dst, const float* src, int nbytes_src, int nbytes_dst) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes_src, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, nbytes_dst, 0x00020000);
    unsigned off = 16u * (threadIdx.x - 1);         // lane 0: wraps to 0xfffffff0
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    v.x += 0x00800000u;                               // x2 as a float bit trick: marks that the lane ran
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, 16u * threadIdx.x, 0, 0);
}
int main() {
    float *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i;
    hipMemcpy(s, h, 4096, hipMemcpyHostToDevice); hipMemset(d, 0xff, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s, 256, 512);
    hipDeviceSynchronize();
    float o[1024]; hipMemcpy(o, d, 4096, hipMemcpyDeviceToHost);
    for (int l = 0; l < 40; l += 1) printf("lane %2d: %g %g %g %g\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
    printf("err %s\n", hipGetErrorString(hipGetLastError()));
}
