// f1 : the SDF term of the tracker's Gauss-Newton step — reference tracker.py:174-218 (SDFTracker.compute_sdf_Hg)
//
//   cur = (last_pose . delta) @ obs                     tracker.py:181, motion_util.py:322-327
//   sdf, std, mask = map.get_sdf(cur)                   tracker.py:184   (dif_query_select + dif_query_decode, unchanged)
//   s = sdf / std ;  d = d s / d cur                    tracker.py:186-192 (std detached: d = grad(sdf) / std)
//   Lai = d @ last_R^T ; Lbi = (delta @ obs) x Lai      tracker.py:195-199
//   w = robust(s) ; Wf = s w ; JW = J w                 tracker.py:203-207, 58-72
//   H = sum JW (x) J / M ; g = sum J Wf / M ; e = sum s Wf / M        tracker.py:209-218
//
// What the reference spreads over ~25 torch launches, the autograd engine and three host round trips per iteration is TWO launches: the decoder
// kernel over all N points of the cloud (DecodeArgs mode 4: pose, validity test and latent look-up in its row fetch; invalid rows get std = 0) and
// the reduction below; the 44 numbers come back through pinned host memory.
#pragma once

struct HgArgs {
    float Tc[12];        // last_pose . delta : rows of [R | t]
    float Td[12];        // delta
    float Lt[9];         // last_pose's R^T, row-major
    int robust;          // 0 none, 1 huber, 2 tukey
    float k;
    int no_grad;
};

#define HG_TERMS 29      /* 21 (upper triangle of H) + 6 (g) + 1 (e) + 1 (the number of valid points) */
#define HG_BLOCKS 256     /* most workgroups the reduction is launched with (rows of the partial-sum buffer) */

struct HgPoint {
    float sd, sf, g0, g1, g2, x, y, z;
    // row i of the decoder's outputs = point i of the cloud; sd == 0: not a valid point (the decoder kernel's mark)
    __device__ __forceinline__ void load(int i, const float* __restrict__ obs, const float* __restrict__ sdf, const float* __restrict__ std_,
                                         const float* __restrict__ grad, int no_grad) {
        sd = std_[i]; sf = sdf[i];
        g0 = g1 = g2 = x = y = z = 0.0f;
        if (!no_grad) {
            g0 = grad[(int64_t)i * 3]; g1 = grad[(int64_t)i * 3 + 1]; g2 = grad[(int64_t)i * 3 + 2];
            x = obs[(int64_t)i * 3]; y = obs[(int64_t)i * 3 + 1]; z = obs[(int64_t)i * 3 + 2];
        }
    }
};

__device__ __forceinline__ void hg_accumulate(double* acc, const HgArgs& a, const HgPoint& q) {
    const float sd = q.sd;
    if (sd == 0.0f) return;
    acc[28] += 1.0;
    const float s = q.sf / sd;
    float w = 1.0f;
    if (a.robust == 1) {
        const float ab = fabsf(s);
        if (ab > a.k) w = a.k / ab;
    } else if (a.robust == 2) {
        w = 0.0f;
        if (fabsf(s) <= a.k) {
            const float r = s / a.k, u = 1.0f - r * r;
            w = u * u;
        }
    }
    const float wf = s * w;
    acc[27] += (double)s * (double)wf;
    if (a.no_grad) return;
    const float d0 = q.g0 / sd, d1 = q.g1 / sd, d2 = q.g2 / sd;
    float J[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) J[j] = (d0 * a.Lt[j] + d1 * a.Lt[3 + j]) + d2 * a.Lt[6 + j];
    const float c0 = pose_row(a.Td, 0, q.x, q.y, q.z), c1 = pose_row(a.Td, 1, q.x, q.y, q.z), c2 = pose_row(a.Td, 2, q.x, q.y, q.z);
    J[3] = c1 * J[2] - c2 * J[1];
    J[4] = c2 * J[0] - c0 * J[2];
    J[5] = c0 * J[1] - c1 * J[0];
    int t = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double jw = (double)(J[r] * w);
#pragma unroll
        for (int c = r; c < 6; ++c) acc[t++] += jw * (double)J[c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[21 + r] += (double)J[r] * (double)wf;
}

// Fixed reduction tree (thread: grid-stride order; 16-lane row; workgroup: rows in order; grid: workgroups in order, summed by whichever workgroup
// arrives last): the same inputs give the same 44 numbers, whatever the order the workgroups run in.
__global__ void __launch_bounds__(DIF_BLOCK) k_sdf_hg_reduce(int N, const float* __restrict__ obs, const float* __restrict__ sdf, const float* __restrict__ std_,
                                                           const float* __restrict__ grad, HgArgs a, double* partial, int* ticket, double* out,
                                                           double* out_host, int64_t seq) {
    __shared__ double red[DIF_BLOCK / 16][HG_TERMS];
    __shared__ int last;
    double acc[HG_TERMS];
#pragma unroll
    for (int t = 0; t < HG_TERMS; ++t) acc[t] = 0.0;
    // two points per thread and trip, their loads issued together (the decoder's outputs were written by other XCDs a moment ago: L2 misses)
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int m = (int)(blockIdx.x * blockDim.x + threadIdx.x); m < N; m += 2 * stride) {
        const int m1 = m + stride < N ? m + stride : m;
        HgPoint p0, p1;
        p0.load(m, obs, sdf, std_, grad, a.no_grad);
        p1.load(m1, obs, sdf, std_, grad, a.no_grad);
        hg_accumulate(acc, a, p0);
        if (m1 != m) hg_accumulate(acc, a, p1);
    }
    // 16-lane rows summed on the VALU (four DPP row shifts: lane 15 of a row ends with the row's sum; a 64-bit __shfl costs two trips through
    // the LDS crossbar and there are 28 values), the 16 row sums of the workgroup added in order by one thread per term
    const int lane = lane_id(), wid = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < HG_TERMS; ++t) {
        double v = acc[t];
        v += __longlong_as_double(dpp_mov64<0x111, 0xF>(__double_as_longlong(v)));
        v += __longlong_as_double(dpp_mov64<0x112, 0xF>(__double_as_longlong(v)));
        v += __longlong_as_double(dpp_mov64<0x114, 0xF>(__double_as_longlong(v)));
        v += __longlong_as_double(dpp_mov64<0x118, 0xF>(__double_as_longlong(v)));
        if ((lane & 15) == 15) red[wid * 4 + (lane >> 4)][t] = v;
    }
    __syncthreads();
    // The hand-over to the last workgroup without agent-scope FENCES (a release / acquire fence writes back / invalidates the XCD's whole L2:
    // ~4 us each, three of them were 12 of this kernel's 16 us): the partial sums are written through (agent-scope stores), the wave waits for
    // their acknowledgement, and only then thread 0 — a lane of the same wave — takes its ticket; the last workgroup reads them with
    // agent-scope loads.
    if (threadIdx.x < HG_TERMS) {
        double v = 0.0;
        for (int q = 0; q < DIF_BLOCK / 16; ++q) v += red[q][threadIdx.x];
        __hip_atomic_store(partial + (int64_t)blockIdx.x * HG_TERMS + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static_assert(HG_TERMS <= 64, "the partial sums and the ticket must come from one wave");
    if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // vmcnt(0)
    if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!last) return;
    // (the partial sums of the other workgroups sit in other XCDs' L2s or in HBM: ~0.5 us per dependent load, so the rows are read by eight
    // groups of lanes, eight loads in flight each, and the eight sub-totals are added in order)
    __shared__ double fin[DIF_BLOCK / 32][HG_TERMS];
    __shared__ double tot[HG_TERMS];
    {
        const int t = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
        const int per = ((int)gridDim.x + DIF_BLOCK / 32 - 1) / (DIF_BLOCK / 32), hi = min((q + 1) * per, (int)gridDim.x);
        if (t < HG_TERMS) {
            double a8 = 0.0;
            for (int b = q * per; b < hi; b += 8) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = b + j < hi ? __hip_atomic_load(partial + (int64_t)(b + j) * HG_TERMS + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) a8 += v[j];
            }
            fin[q][t] = a8;
        }
    }
    __syncthreads();
    if (threadIdx.x < HG_TERMS) {
        double v = 0.0;
        for (int q = 0; q < DIF_BLOCK / 32; ++q) v += fin[q][threadIdx.x];
        tot[threadIdx.x] = v;
    }
    __syncthreads();
    const double Mv = tot[28];                   // the number of valid points; the sums are scaled by 1 / M (tracker.py:209-218)
    if (threadIdx.x < 44) {
        double v;
        const int e = (int)threadIdx.x;
        if (e < 36) {
            int r = e / 6, c = e % 6;
            if (r > c) { const int q = r; r = c; c = q; }
            v = tot[r * 6 - r * (r - 1) / 2 + (c - r)];
        } else if (e < 42) v = tot[21 + (e - 36)];
        else if (e == 42) v = tot[27];
        else v = Mv;
        if (e < 43) v = Mv > 0.0 ? v / Mv : 0.0;
        out[e] = v;
        if (out_host) {                          // written through to the host and acknowledged ...
            __hip_atomic_store(out_host + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *ticket = 0;
        if (out_host) __hip_atomic_store((int64_t*)out_host + 44, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // ... before the sequence number follows
    }
}
