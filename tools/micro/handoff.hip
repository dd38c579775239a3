// Two questions about the runtime, answered on the box:
//  (A) what does it cost to hand a dependency from one HIP stream (hardware queue) to another — an event, a one-wave gate kernel polling a
//      word the producer writes, hipStreamWaitValue32 on that word — measured as producer-end -> consumer-start in wall-clock stamps the
//      kernels take themselves (s_memrealtime, 100 MHz), and as the time of a ping-pong chain;
//  (B) how does a device -> pinned-host copy of a frame's triangle rows (64 KB .. 2 MB) travel: hipMemcpyAsync (blit kernel or SDMA?
//      rocprofv3 --kernel-trace shows __amd_rocclr_copyBuffer for a blit), and the HSA copy call on the SDMA engines directly.
// hipcc --offload-arch=gfx950 -O3 handoff.hip -o handoff -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter() * 0 + __builtin_amdgcn_s_memrealtime(); }

// spins `ticks` x 10 ns, stamps (start, end) at stamps[2 idx], then (flag != null) publishes `value` behind an agent-scope release
__global__ void k_work(int ticks, unsigned long long* stamps, int idx, unsigned* flag, unsigned value) {
    const unsigned long long t0 = wall();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * idx] = t0;
    while (wall() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[2 * idx + 1] = wall();
        if (flag) { __threadfence(); __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

// one wave: polls until *flag >= value (gives up after ~20 ms)
__global__ void k_gate(const unsigned* flag, unsigned value, unsigned* err) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
        __builtin_amdgcn_s_sleep(1);
        if (wall() - t0 > 2000000ull) { *err = 1; return; }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void stats(const char* what, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    printf("  %-64s median %7.2f us   p10 %7.2f   p90 %7.2f   (n=%zu)\n", what, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], v.size());
}

int main(int argc, char** argv) {
    const int iters = 200;
    int dev = 0; CK(hipSetDevice(dev));
    int can_wait = 0; hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
    hipStream_t sA, sB;
    CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    unsigned long long* stamps; CK(hipHostMalloc(&stamps, sizeof(unsigned long long) * 4 * (iters + 8)));
    unsigned *flag, *err; CK(hipMalloc(&flag, 256)); CK(hipMalloc(&err, 4));
    unsigned* sigflag = nullptr;
    if (hipExtMallocWithFlags((void**)&sigflag, 8, hipMallocSignalMemory) != hipSuccess) { sigflag = nullptr; printf("signal memory: not available\n"); (void)hipGetLastError(); }
    hipEvent_t ev[8]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int W = 1000;     // 10 us of work per kernel

    // (A0) same stream: kernel boundary, and with an event record in between
    for (int with_ev = 0; with_ev < 2; ++with_ev) {
        std::vector<double> gap;
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sA, W, stamps, 0, (unsigned*)nullptr, 0u);
            if (with_ev) CK(hipEventRecord(ev[0], sA));
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sA, W, stamps, 1, (unsigned*)nullptr, 0u);
            CK(hipStreamSynchronize(sA));
            gap.push_back((double)(stamps[2] - stamps[1]) * 0.01);
        }
        stats(with_ev ? "same stream, event record between two kernels: gap" : "same stream, back to back: gap", gap);
    }

    // (A1..) cross-stream hand-off: P on sA, Q on sB behind P; producer-end -> consumer-start
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 2 && !can_wait) continue;
        if (mode == 3 && (!can_wait || !sigflag)) continue;
        unsigned* f = mode == 3 ? sigflag : flag;
        CK(hipMemset(f, 0, 4)); CK(hipMemset(err, 0, 4)); CK(hipDeviceSynchronize());
        std::vector<double> gap, host;
        for (int i = 0; i < iters; ++i) {
            const unsigned v = (unsigned)i + 1;
            const double h0 = now_us();
            // the consumer side is enqueued FIRST where the mechanism allows (a host that runs ahead)
            if (mode == 1) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sB, (const unsigned*)f, v, err);
            if (mode >= 2) { hipError_t e = hipStreamWaitValue32(sB, f, v, hipStreamWaitValueGte, 0xFFFFFFFFu); if (e != hipSuccess) { printf("  hipStreamWaitValue32: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); break; } }
            if (mode >= 1) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sB, W, stamps, 1, (unsigned*)nullptr, 0u);
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sA, W, stamps, 0, mode ? f : (unsigned*)nullptr, v);
            if (mode == 0) {
                CK(hipEventRecord(ev[0], sA)); CK(hipStreamWaitEvent(sB, ev[0], 0));
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sB, W, stamps, 1, (unsigned*)nullptr, 0u);
            }
            host.push_back(now_us() - h0);
            CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB));
            gap.push_back((double)(stamps[2] - stamps[1]) * 0.01);
        }
        const char* names[] = {"cross stream, hipEventRecord + hipStreamWaitEvent: P.end -> Q.start", "cross stream, one-wave gate kernel polling a device word: P.end -> Q.start",
                               "cross stream, hipStreamWaitValue32 on a device word: P.end -> Q.start", "cross stream, hipStreamWaitValue32 on signal memory: P.end -> Q.start"};
        if (!gap.empty()) { stats(names[mode], gap); stats("     host time to enqueue the pair", host); }
        unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); if (e) printf("  gate kernel gave up!\n");
    }

    // (A5) a two-queue pipeline like the frame's: A: F(i) [fuse], FE(i+1) 50 us;  B: behind F(i): E(i) 90 us;  F(i+1) behind E(i).  Time per frame
    // against the single-queue chain F, E, FE (150 us).
    for (int mode = -1; mode < 3; ++mode) {
        if (mode == 2 && !can_wait) continue;
        CK(hipMemset(flag, 0, 256)); CK(hipDeviceSynchronize());
        unsigned* fF = flag; unsigned* fE = flag + 32;
        const int n = 100;
        const double t0 = now_us();
        for (int i = 1; i <= n; ++i) {
            if (mode == -1) {
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sA, 800, stamps, 2, (unsigned*)nullptr, 0u);      // F
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sA, 9000, stamps, 2, (unsigned*)nullptr, 0u);     // E
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sA, 5000, stamps, 2, (unsigned*)nullptr, 0u);     // FE
                continue;
            }
            // stream A: [wait E(i-1)] F(i) -> FE(i+1)
            if (i > 1) {
                if (mode == 0) CK(hipStreamWaitEvent(sA, ev[2 + (i & 1)], 0));
                else if (mode == 1) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sA, (const unsigned*)fE, (unsigned)(i - 1), err);
                else CK(hipStreamWaitValue32(sA, fE, (unsigned)(i - 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
            }
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sA, 800, stamps, 2, mode ? fF : (unsigned*)nullptr, (unsigned)i);
            if (mode == 0) CK(hipEventRecord(ev[4 + (i & 1)], sA));
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sA, 5000, stamps, 2, (unsigned*)nullptr, 0u);
            // stream B: [wait F(i)] E(i)
            if (mode == 0) CK(hipStreamWaitEvent(sB, ev[4 + (i & 1)], 0));
            else if (mode == 1) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sB, (const unsigned*)fF, (unsigned)i, err);
            else CK(hipStreamWaitValue32(sB, fF, (unsigned)i, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, sB, 9000, stamps, 3, mode ? fE : (unsigned*)nullptr, (unsigned)i);
            if (mode == 0) CK(hipEventRecord(ev[2 + ((i + 1) & 1)], sB));
        }
        CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB));
        const char* names[] = {"single queue F,E,FE (8+90+50 us of work)", "two queues, events", "two queues, gate kernels", "two queues, hipStreamWaitValue32"};
        printf("  pipeline %-44s %7.2f us per frame (ideal: 148 single, 98 overlapped)\n", names[mode + 1], (now_us() - t0) / n);
    }

    // (B) device -> pinned host copies
    printf("copies (device -> pinned host), per copy, stream-ordered, timed by the host around enqueue + synchronize:\n");
    const size_t sizes[] = {64 << 10, 300 << 10, 1 << 20, 4 << 20};
    char* dsrc; CK(hipMalloc(&dsrc, 8 << 20)); CK(hipMemset(dsrc, 1, 8 << 20));
    char* hdst; CK(hipHostMalloc(&hdst, 8 << 20));
    char* hreg = (char*)aligned_alloc(4096, 8 << 20); memset(hreg, 0, 8 << 20); CK(hipHostRegister(hreg, 8 << 20, hipHostRegisterDefault));
    for (size_t sz : sizes) {
        for (int which = 0; which < 3; ++which) {
            std::vector<double> t;
            for (int i = 0; i < 50; ++i) {
                const double h0 = now_us();
                if (which == 0) CK(hipMemcpyAsync(hdst, dsrc, sz, hipMemcpyDeviceToHost, sB));
                else if (which == 1) CK(hipMemcpyDtoHAsync(hdst, (hipDeviceptr_t)dsrc, sz, sB));
                else CK(hipMemcpyAsync(hreg, dsrc, sz, hipMemcpyDeviceToHost, sB));
                CK(hipStreamSynchronize(sB));
                t.push_back(now_us() - h0);
            }
            char name[128]; snprintf(name, sizeof name, "%s %zu KB", which == 0 ? "hipMemcpyAsync -> hipHostMalloc" : which == 1 ? "hipMemcpyDtoHAsync -> hipHostMalloc" : "hipMemcpyAsync -> hipHostRegister", sz >> 10);
            stats(name, t);
        }
    }
    // the HSA copy on the SDMA engines, directly
    {
        struct Agents { hsa_agent_t gpu{}, cpu{}; bool g = false, c = false; } ag;
        if (hsa_init() != HSA_STATUS_SUCCESS) { printf("hsa_init failed\n"); return 0; }
        hsa_iterate_agents([](hsa_agent_t a, void* d) {
            auto* ag = (Agents*)d; hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
            if (t == HSA_DEVICE_TYPE_GPU && !ag->g) { ag->gpu = a; ag->g = true; }
            if (t == HSA_DEVICE_TYPE_CPU && !ag->c) { ag->cpu = a; ag->c = true; }
            return HSA_STATUS_SUCCESS; }, &ag);
        uint32_t engines = 0; hsa_status_t st = hsa_amd_memory_copy_engine_status(ag.cpu, ag.gpu, &engines);
        printf("hsa_amd_memory_copy_engine_status(dst cpu, src gpu): status %d, free engine mask 0x%x\n", (int)st, engines);
        hsa_signal_t sig; hsa_signal_create(1, 0, nullptr, &sig);
        for (size_t sz : sizes) {
            std::vector<double> t;
            for (int i = 0; i < 50; ++i) {
                hsa_signal_store_relaxed(sig, 1);
                const double h0 = now_us();
                hsa_status_t s2 = hsa_amd_memory_async_copy(hdst, ag.cpu, dsrc, ag.gpu, sz, 0, nullptr, sig);
                if (s2 != HSA_STATUS_SUCCESS) { printf("  hsa_amd_memory_async_copy failed: %d\n", (int)s2); break; }
                while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
                t.push_back(now_us() - h0);
            }
            char name[128]; snprintf(name, sizeof name, "hsa_amd_memory_async_copy (SDMA) -> hipHostMalloc %zu KB", sz >> 10);
            if (!t.empty()) stats(name, t);
        }
        // three copies of a frame's rows (300 + 64 + 100 KB) in one go, one signal each vs one packed copy
        {
            hsa_signal_t s3[3]; for (auto& s : s3) hsa_signal_create(1, 0, nullptr, &s);
            std::vector<double> t3, t1;
            const size_t parts[3] = {300 << 10, 64 << 10, 100 << 10};
            for (int i = 0; i < 50; ++i) {
                for (auto& s : s3) hsa_signal_store_relaxed(s, 1);
                double h0 = now_us();
                size_t off = 0;
                for (int k = 0; k < 3; ++k) { hsa_amd_memory_async_copy(hdst + off, ag.cpu, dsrc + off, ag.gpu, parts[k], 0, nullptr, s3[k]); off += parts[k]; }
                for (auto& s : s3) while (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
                t3.push_back(now_us() - h0);
                hsa_signal_store_relaxed(sig, 1);
                h0 = now_us();
                hsa_amd_memory_async_copy(hdst, ag.cpu, dsrc, ag.gpu, 464 << 10, 0, nullptr, sig);
                while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
                t1.push_back(now_us() - h0);
            }
            stats("SDMA: a frame's three row ranges as three copies", t3);
            stats("SDMA: the same bytes as one copy", t1);
        }
        // does the data arrive?  (hdst must hold the source pattern)
        bool ok = true; for (size_t i = 0; i < (464 << 10); i += 4097) ok &= hdst[i] == 1;
        printf("SDMA copy landed: %s\n", ok ? "yes" : "NO");
    }
    return 0;
}
