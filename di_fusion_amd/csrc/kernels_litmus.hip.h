// Litmus kernels for the fence-free hand-overs of the product kernels (dif_test_handoff; tests/test_gpu_handoff.py).  Part of libdifusion; included
// by difusion.hip inside its anonymous namespace.  TEST INFRASTRUCTURE: nothing on the fusion path launches these.
//
// The pattern under test (kernels_track.hip.h:k_sdf_hg_reduce, kernels_mesh.hip.h:extract_finish_body; a launch-wide meeting built on it was
// measured as the frame's one-launch decoder in round 6 — profiles/r06_experiments.md 1 — and not kept):
//   producer:  relaxed agent-scope (or system-scope) STORES of the payload — on gfx942 / gfx950 these are write-through stores (sc1 / sc0 sc1) —,
//              s_waitcnt vmcnt(0) in the storing wave (every store acknowledged by the memory side), THEN the counter / ticket / sequence word;
//   consumer:  sees the word through an agent-scope (system-scope: the CPU) load, THEN reads the payload with agent-scope loads (sc1: past the
//              XCD's L2, which is not coherent with the other seven).
// No fence instruction on either side (a release / acquire fence at agent scope writes back / invalidates the XCD's whole L2: ~4 us).  This is outside
// the HIP memory model: it rests on what the hardware does with sc1 accesses and on the compiler emitting them for relaxed agent-scope atomics, which
// is why the pattern is tied to the validated compiler (di_fusion_amd/_build.py:VALIDATED_HIPCC) and hammered here on every `pytest -m gpu`.
#pragma once

#define LIT_WORDS 29            /* doubles per record: the 28 sums + the count of k_sdf_hg_reduce's partial rows */

__device__ __forceinline__ double lit_value(unsigned it, unsigned g, unsigned j) {
    // distinct per (iteration, producer, word); exact in a double
    return (double)(((unsigned long long)it << 20) ^ ((unsigned long long)g << 8) ^ j) + 0.5;
}

// mode 0 — a launch-wide meeting: every workgroup writes its record, waits for the acknowledgements, adds 1 to the iteration's counter, polls
//          the counter until all G have arrived, then checks the record of ANOTHER workgroup (a different one every iteration, so that every pair of
//          XCDs is crossed).  Two record slots: a workgroup rewrites slot (it & 1) only after it has passed iteration it + 1's counter, i.e. after
//          every reader of iteration it has arrived there.
// mode 1 — the tracker reduction's hand-over: same producers, but only the LAST arriver (ticket) reads — all G records — and then opens the next
//          iteration through a second word.
// G <= the number of workgroups the GPU (or the stream's CU mask) holds at once: the launch is a sequence of launch-wide meetings.
__global__ void __launch_bounds__(256) k_litmus_device(double* __restrict__ rec /* [2][G][LIT_WORDS] */, unsigned* __restrict__ counters /* [iters + 1] */,
                                                       unsigned* __restrict__ go /* mode 1: iterations released so far */, int iters, int mode,
                                                       unsigned long long* __restrict__ out /* [0] stale values seen, [1] hand-overs checked, [2] time-outs */) {
    const unsigned G = gridDim.x, g = blockIdx.x;
    __shared__ int s_last;
    unsigned long long bad = 0, seen = 0;
    for (int it = 0; it < iters; ++it) {
        double* mine = rec + ((size_t)(it & 1) * G + g) * LIT_WORDS;
        if (threadIdx.x < LIT_WORDS) __hip_atomic_store(mine + threadIdx.x, lit_value(it, g, threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mode == 0) {
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counters + it, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(counters + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) { atomicAdd(out + 2, 1ull); break; }
                }
            }
            __syncthreads();
            const unsigned other = (g + 1u + (unsigned)it * 37u) % G;
            if (threadIdx.x < LIT_WORDS) {
                const double v = __hip_atomic_load(rec + ((size_t)(it & 1) * G + other) * LIT_WORDS + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != lit_value(it, other, threadIdx.x)) ++bad;
            }
            if (threadIdx.x == 0) ++seen;
        } else {
            if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(counters + it, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1 ? 1 : 0;
            __syncthreads();
            if (s_last) {
                for (unsigned o = threadIdx.x / 32; o < G; o += blockDim.x / 32) {
                    const unsigned j = threadIdx.x & 31;
                    if (j < LIT_WORDS) {
                        const double v = __hip_atomic_load(rec + ((size_t)(it & 1) * G + o) * LIT_WORDS + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v != lit_value(it, o, j)) ++bad;
                    }
                    if (j == 0) ++seen;
                }
                __syncthreads();
                if (threadIdx.x == 0) __hip_atomic_store(go, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (threadIdx.x == 0) {
                int spins = 0;
                while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) { atomicAdd(out + 2, 1ull); break; }
                }
            }
            __syncthreads();
        }
    }
    if (bad) atomicAdd(out + 0, bad);
    if (seen) atomicAdd(out + 1, seen);
}

// The device -> host hand-over (k_sdf_hg_reduce's 44 doubles + sequence word, extract_finish_body's counter snapshot + stamp): one workgroup per
// mailbox writes 44 doubles with system-scope stores, waits for the acknowledgements, writes the sequence number; the CPU polls the sequence word,
// checks the 44 doubles, and answers through an acknowledgement word in the same pinned allocation, which the workgroup polls before the next round.
#define LIT_HOST_WORDS 44
__global__ void __launch_bounds__(64) k_litmus_host(double* __restrict__ box /* pinned: per mailbox [LIT_HOST_WORDS doubles | seq int64 | ack int64 | pad] */, int rounds,
                                                    unsigned long long* __restrict__ out) {
    double* mine = box + (size_t)blockIdx.x * 64;
    long long* seq = reinterpret_cast<long long*>(mine + LIT_HOST_WORDS);
    long long* ack = seq + 1;
    for (int r = 1; r <= rounds; ++r) {
        if (threadIdx.x < LIT_HOST_WORDS) __hip_atomic_store(mine + threadIdx.x, lit_value(r, blockIdx.x, threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            __hip_atomic_store(seq, (long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            int spins = 0;
            while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (long long)r) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1 << 24)) { atomicAdd(out + 2, 1ull); return; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// keeps the memory system busy beside the litmus kernels: streams through `n` float4 for ~`ticks` x 10 ns
__global__ void __launch_bounds__(256) k_litmus_hog(float4* __restrict__ buf, size_t n, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        for (int k = 0; k < 64; ++k) {
            const float4 v = buf[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            i += (size_t)gridDim.x * blockDim.x;
            if (i >= n) i -= n;
        }
        buf[i] = acc;
    }
}
