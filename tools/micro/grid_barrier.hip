// Micro-benchmark: cost of a device-wide barrier inside one kernel (atomic counter + agent-scope fences) against the cost of a kernel
// boundary, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o tools/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);     // agent scope by default for device code
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 22)) { *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_barriers(unsigned* counter, int n_barriers, float* data, int n, int* err) {
    const int nb = gridDim.x;
    for (int b = 0; b < n_barriers; ++b) {
        // a little dependent work between barriers: every block writes its slice, then reads its neighbour's after the barrier
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) data[i] = data[(i + 4099) % n] + 1.0f;
        grid_barrier(counter, (unsigned)(b + 1) * nb, err);
    }
}

__global__ void __launch_bounds__(256) k_step(float* data, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) data[i] = data[(i + 4099) % n] + 1.0f;
}

int main() {
    const int n = 1 << 18, nb_list[3] = {64, 256, 512}, reps = 20, steps = 50;
    float* data; unsigned* counter; int* err;
    hipMalloc(&data, n * 4); hipMalloc(&counter, 4); hipMalloc(&err, 4);
    hipMemset(data, 0, n * 4); hipMemset(err, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nb : nb_list) {
        float best_bar = 1e9f, best_launch = 1e9f;
        for (int r = 0; r < reps; ++r) {
            hipMemset(counter, 0, 4);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_barriers, dim3(nb), dim3(256), 0, 0, counter, steps, data, n, err);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best_bar) best_bar = ms;
            hipEventRecord(a);
            for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(k_step, dim3(nb), dim3(256), 0, 0, data, n);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b); if (ms < best_launch) best_launch = ms;
        }
        // the same 50 steps as a captured graph
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(k_step, dim3(nb), dim3(256), 0, st, data, n);
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float best_graph = 1e9f;
        for (int r = 0; r < reps; ++r) {
            hipEventRecord(a, st); hipGraphLaunch(ge, st); hipEventRecord(b, st); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best_graph) best_graph = ms;
        }
        int h_err = 0; hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
        printf("{\"workgroups\": %d, \"us_per_step_grid_barrier\": %.2f, \"us_per_step_kernel_launches\": %.2f, \"us_per_step_graph_nodes\": %.2f, \"err\": %d}\n",
               nb, best_bar * 1e3f / steps, best_launch * 1e3f / steps, best_graph * 1e3f / steps, h_err);
    }
    return 0;
}
