// Does hipEventElapsedTime work on events recorded by event-record nodes of a captured hipGraph (ROCm 7, gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main() {
    float* d; hipMalloc(&d, 1024 * 4); hipMemset(d, 0, 4096);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, s, d, 10);
    hipError_t ra = hipEventRecord(a, s);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, s, d, 200000);
    hipError_t rb = hipEventRecord(b, s);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, s, d, 10);
    hipError_t rc = hipStreamEndCapture(s, &g);
    hipError_t ri = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
    printf("record in capture: %d %d, end %d, instantiate %d, nodes %zu\n", (int)ra, (int)rb, (int)rc, (int)ri, nn);
    for (int rep = 0; rep < 3; ++rep) {
        hipError_t rl = hipGraphLaunch(ge, s);
        hipError_t rs = hipStreamSynchronize(s);
        hipError_t qa = hipEventQuery(a), qb = hipEventQuery(b);
        hipError_t sy = hipEventSynchronize(b);
        float ms = -1.f;
        hipError_t re = hipEventElapsedTime(&ms, a, b);
        printf("rep %d: launch %d sync %d query %d %d evsync %d elapsed rc %d (%s) = %.3f ms\n", rep, (int)rl, (int)rs, (int)qa, (int)qb, (int)sy, (int)re,
               hipGetErrorName(re), ms);
    }
    return 0;
}
