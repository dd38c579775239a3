// What a same-address ticket counter sustains: `grid` workgroups claim `total` tickets (thread 0: atomicAdd, LDS broadcast, barrier), with
// `work` x 64 cycles of sleep per ticket standing in for the group's work.  hipcc --offload-arch=gfx950 -O3 ticket_rate.hip -o ticket_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_tickets(unsigned* ticket, unsigned total, int work, unsigned* sink) {
    __shared__ unsigned s_g;
    unsigned acc = 0;
    while (true) {
        if (threadIdx.x == 0) s_g = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned g = s_g;
        if (g >= total) break;
        acc += g;
        for (int i = 0; i < work; ++i) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
int main() {
    unsigned *ticket, *sink;
    hipMalloc(&ticket, 4); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned total = 524288;
    for (int grid : {256, 512, 1024, 1280, 2048})
        for (int work : {0, 100, 400}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemset(ticket, 0, 4);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k_tickets, dim3(grid), dim3(256), 0, 0, ticket, total, work, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("grid %4d work %3d: %.3f ms, %.1f ns per ticket\n", grid, work, best, best * 1e6f / total);
        }
    return 0;
}
