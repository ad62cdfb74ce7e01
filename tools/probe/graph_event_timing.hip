// Can plain HIP events recorded inside a stream capture be used for timing after the graph is replayed?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/graph_event_timing.hip -o /tmp/gev && /tmp/gev
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(float* p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }
int main() {
    float* d; CK(hipMalloc(&d, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, 200000);
    CK(hipEventRecord(e1, s));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, 400000);
    CK(hipEventRecord(e2, s));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; r++) {
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        float a = -1, b = -1;
        hipError_t ea = hipEventElapsedTime(&a, e0, e1), eb = hipEventElapsedTime(&b, e1, e2);
        printf("replay %d: e0->e1 %.3f ms (%s), e1->e2 %.3f ms (%s)\n", r, a, hipGetErrorString(ea), b, hipGetErrorString(eb));
    }
    return 0;
}
