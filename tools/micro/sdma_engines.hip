// Which SDMA engine does a device -> pinned-host copy of a frame's rows (300 KB) take how long on?  hsa_amd_memory_async_copy (the runtime picks) against
// hsa_amd_memory_async_copy_on_engine for every engine the status call reports, the preferred mask, and the same again after the process has allocated,
// touched and freed 8 GB of device memory (what torch.cuda.empty_cache() behind a throwaway map does).
// hipcc --offload-arch=gfx950 -O3 sdma_engines.hip -o sdma_engines -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0);
    char* dsrc; hipMalloc(&dsrc, 8 << 20); hipMemset(dsrc, 1, 8 << 20);
    char* hdst; hipHostMalloc(&hdst, 8 << 20);
    hipDeviceSynchronize();
    hsa_init();
    hsa_amd_pointer_info_t si{}, di{}; si.size = sizeof(si); di.size = sizeof(di);
    hsa_amd_pointer_info(dsrc, &si, nullptr, nullptr, nullptr); hsa_amd_pointer_info(hdst, &di, nullptr, nullptr, nullptr);
    const hsa_agent_t gpu = si.agentOwner, cpu = di.agentOwner;
    hsa_signal_t sig; hsa_signal_create(1, 0, nullptr, &sig);
    auto timed = [&](int engine, size_t sz) {
        std::vector<double> t;
        for (int i = 0; i < 30; ++i) {
            hsa_signal_store_relaxed(sig, 1);
            const double h0 = now_us();
            hsa_status_t st = engine < 0 ? hsa_amd_memory_async_copy(hdst, cpu, dsrc, gpu, sz, 0, nullptr, sig)
                                         : hsa_amd_memory_async_copy_on_engine(hdst, cpu, dsrc, gpu, sz, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)(1u << engine), false);
            if (st != HSA_STATUS_SUCCESS) return -1.0;
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
            t.push_back(now_us() - h0);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    };
    for (int phase = 0; phase < 2; ++phase) {
        uint32_t avail = 0, pref = 0;
        hsa_amd_memory_copy_engine_status(cpu, gpu, &avail);
        hsa_status_t ps = hsa_amd_memory_get_preferred_copy_engine(cpu, gpu, &pref);
        printf("%s: available engines 0x%x, preferred (status %d) 0x%x\n", phase ? "after allocating, touching and freeing 8 GB" : "fresh process", avail, (int)ps, pref);
        printf("  runtime's choice: 300 KB %.1f us, 64 KB %.1f us\n", timed(-1, 300 << 10), timed(-1, 64 << 10));
        for (int e = 0; e < 16; ++e)
            if (avail & (1u << e)) printf("  engine %2d: 300 KB %.1f us\n", e, timed(e, 300 << 10));
        printf("  runtime's choice again: 300 KB %.1f us\n", timed(-1, 300 << 10));
        if (phase == 0) {
            for (int rep = 0; rep < 2; ++rep) {
                char* big; if (hipMalloc(&big, (size_t)4 << 30) != hipSuccess) break;
                hipMemset(big, 0, (size_t)4 << 30); hipDeviceSynchronize(); hipFree(big);
            }
        }
    }
    return 0;
}
