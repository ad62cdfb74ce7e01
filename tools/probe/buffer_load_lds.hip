#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float4* __restrict__ src, float4* dst, int nvalid) {
    __shared__ float4 buf[256];
    buf[threadIdx.x] = float4{-1.f, -1.f, -1.f, -1.f};
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nvalid * 16, 0x00020000);
    // lanes beyond nvalid are out of range -> expect zeros in LDS
    int voff = threadIdx.x * 16;
    if (threadIdx.x % 7 == 3) voff = 0x7fffff00;       // forced out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(buf + (threadIdx.x & ~63)), 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    dst[threadIdx.x] = buf[threadIdx.x];
}
int main() {
    float4 *s, *d; float4 h[256];
    hipMalloc(&s, 256 * 16); hipMalloc(&d, 256 * 16);
    for (int i = 0; i < 256; i++) h[i] = float4{(float)i, (float)i + 0.25f, (float)i + 0.5f, (float)i + 0.75f};
    hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d, 200);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; i += 1) if (i < 12 || i > 195 && i < 206) printf("%d: %g %g %g %g\n", i, h[i].x, h[i].y, h[i].z, h[i].w);
    return 0;
}
