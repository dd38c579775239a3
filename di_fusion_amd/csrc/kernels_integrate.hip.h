// integrate_keyframe kernels: a3-a10 voxel ids, prune, allocate, gather, encoder (MFMA), fusion  (part of libdifusion; included by difusion.hip inside its anonymous namespace)
#pragma once

// =================================================================================================================
// a3..a6 : voxel ids, prune, allocate   (map.py:366-387)
// =================================================================================================================
// K1: per-point voxel id + per-voxel point count of this frame.
// Per-voxel point counts of the frame (map.py:374) with wave-run aggregation: pixels of a row that fall into the same voxel are
// neighbours in the wave, so one atomic per run of equal ids.  Also zeroes the per-call counters.
__device__ __forceinline__ void voxel_count_point(const Geo& g, bool in_range, float x, float y, float z, int64_t i, int* __restrict__ pt_lin,
                                                  int* __restrict__ frame_count, int* __restrict__ counters, int px_lo, int px_hi) {
    const int lane = lane_id();
    if (i < 4) counters[DIF_C_ALLOC_NEW + i] = 0;                   // ALLOC_NEW, M, C, ITEMS of this call
    int lin = -2;                                                    // -2: beyond N, -1: invalid point
    if (in_range) {
        float xn, yn, zn; int ix, iy, iz;
        bool ok = voxel_of(g, x, y, z, xn, yn, zn, ix, iy, iz);
        ok = ok && ix >= px_lo && ix < px_hi;                         // spatial tiling: own slab + halo only
        lin = ok ? linearize(g, ix, iy, iz) : -1;
        pt_lin[i] = lin;
    }
    int prev = __shfl_up(lin, 1);
    bool head = (lane == 0) || (prev != lin);
    unsigned long long heads = __ballot(head);
    if (head && lin >= 0) {
        unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
        int run = above ? __ffsll((long long)above) : (64 - lane);
        atomicAdd(frame_count + lin, run);
    }
}

__global__ void __launch_bounds__(DIF_BLOCK) k_voxel_count(Geo g, const float* __restrict__ xyz, int64_t N, int* __restrict__ pt_lin,
                                                         int* __restrict__ frame_count, int* __restrict__ counters, int px_lo, int px_hi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // grid covers N rounded up to a wave
    const bool in = i < N;
    voxel_count_point(g, in, in ? xyz[i * 3 + 0] : 0.f, in ? xyz[i * 3 + 1] : 0.f, in ? xyz[i * 3 + 2] : 0.f, i, pt_lin, frame_count, counters, px_lo, px_hi);
}

// a1 + a2 + a3 in one pass for a streaming caller: depth pixel -> world point and normal (written out for the later stages) -> voxel id
// and per-voxel count, without re-reading the points.
__global__ void __launch_bounds__(DIF_BLOCK) k_unproject_voxel_count(Geo g, const dif_frame_t* __restrict__ frame, int H, int W, float fx, float fy,
                                                                   float cx, float cy, float* __restrict__ xyz, float* __restrict__ nrm,
                                                                   int* __restrict__ pt_lin, int* __restrict__ frame_count, int* __restrict__ counters,
                                                                   int px_lo, int px_hi) {
    const int64_t N = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < N;
    float p[3] = {0.f, 0.f, 0.f}, nv[3];
    if (in) {
        Pose P;
#pragma unroll
        for (int k = 0; k < 9; ++k) P.r[k] = frame->pose[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) P.t[k] = frame->pose[9 + k];
        unproject_point(frame->depth, frame->normal_cam, i, W, fx, fy, cx, cy, P, p, nv);
        xyz[i * 3 + 0] = p[0]; xyz[i * 3 + 1] = p[1]; xyz[i * 3 + 2] = p[2];
        nrm[i * 3 + 0] = nv[0]; nrm[i * 3 + 1] = nv[1]; nrm[i * 3 + 2] = nv[2];
    }
    voxel_count_point(g, in, p[0], p[1], p[2], i, pt_lin, frame_count, counters, px_lo, px_hi);
}

// K2: prune mask + candidate voxels.  mask[i] = count(voxel of i) > prune_min_vox_obs (map.py:375).  A kept point whose
// voxel has no slot marks that voxel and its 6 clamped neighbours (if empty) in the bitmap (map.py:383-386).
__global__ void __launch_bounds__(DIF_BLOCK) k_prune_mark(Geo g, int prune_min, const int* __restrict__ pt_lin, int64_t N,
                                                        const int* __restrict__ frame_count, const int64_t* __restrict__ indexer,
                                                        uint8_t* __restrict__ unq_mask, uint32_t* __restrict__ bits,
                                                        int* __restrict__ counters) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int lane = lane_id();
    int lin = (i < N) ? pt_lin[i] : -2;
    bool keep = false;
    const int cnt = (lin >= 0) ? frame_count[lin] : 0;            // both look-ups hang off `lin` only: issued together
    const int64_t own = (lin >= 0) ? indexer[lin] : 0;
    if (lin >= 0) keep = (prune_min > 0) ? (cnt > prune_min) : true;
    if (i < N) unq_mask[i] = keep ? 1 : 0;
    int prev = __shfl_up(lin, 1);
    bool head = (lane == 0) || (prev != lin);
    if (head && keep && own == -1) {
        int ix, iy, iz;
        unlinearize(g, lin, ix, iy, iz);
        int cand[7];
        cand[0] = lin;
        cand[1] = linearize(g, clampi(ix - 1, 0, g.nx - 1), iy, iz);
        cand[2] = linearize(g, clampi(ix + 1, 0, g.nx - 1), iy, iz);
        cand[3] = linearize(g, ix, clampi(iy - 1, 0, g.ny - 1), iz);
        cand[4] = linearize(g, ix, clampi(iy + 1, 0, g.ny - 1), iz);
        cand[5] = linearize(g, ix, iy, clampi(iz - 1, 0, g.nz - 1));
        cand[6] = linearize(g, ix, iy, clampi(iz + 1, 0, g.nz - 1));
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            int v = cand[c];
            if (indexer[v] != -1) continue;
            uint32_t b = 1u << (v & 31);
            if (!(bits[v >> 5] & b)) atomicOr(bits + (v >> 5), b);
        }
    }
}

// K3: ordered compaction of the candidate bitmap -> slots n_occupied, n_occupied+1, ... in ASCENDING lin order
// (torch.unique order, map.py:385-387, 310-319).  Clears the bitmap as it goes.
struct AllocFunctor {
    uint32_t* bits;
    int64_t* indexer;
    int64_t* pos;
    int* counters;
    int64_t capacity;
    __device__ int count(int w) const { return __popc(bits[w]); }
    __device__ void emit(int w, int offset) const {
        uint32_t word = bits[w];
        bits[w] = 0u;
        int base = counters[DIF_C_N_OCCUPIED] + offset;
        while (word) {
            int b = __ffs((int)word) - 1;
            word &= word - 1;
            int lin = w * 32 + b;
            if (base < capacity) {
                indexer[lin] = base;
                pos[base] = lin;
            }
            ++base;
        }
    }
    __device__ void finish(int total) const { counters[DIF_C_ALLOC_NEW] = total; }
};

// K4: (i) commit n_occupied += newly allocated (all pass-2 blocks of K3 have read the old value by now),
// (ii) restore frame_count to zero, (iii) focus mask + 8-offset gather keys (map.py:389-433).
// Key of pair (offset o, point i), stored at o*N + i (the reference's concatenation order): slot of the neighbour voxel
// if that voxel is in the encode set {obs_count < encoder_count_th}, else DIF_INVALID_KEY.  Rows per slot are counted here
// (seg_cnt = the reference's `pcounts`, map.py:437-439) with one atomic per distinct slot per wave.
__device__ __forceinline__ bool in_encode_set(int64_t slot, const float* __restrict__ obs, float th) { return slot >= 0 && obs[slot] < th; }

// Wave-aggregated "fetch-add 1" on counter[key] for every lane whose key is valid; returns the lane's unique offset
// (base + rank among the lanes of the wave that share the key).  One atomic per distinct key per wave, and all of a wave's atomics
// are in flight together: the grouping loop is pure ALU, the leaders then issue their atomics in ONE instruction and the wave waits
// for a single round trip however many distinct keys it holds.
__device__ __forceinline__ int wave_grouped_fetch_add(int* __restrict__ counter, uint32_t key, bool valid) {
    const int lane = lane_id();
    int my_leader = lane, my_rank = 0, my_group = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
        const unsigned long long same = __ballot(valid && key == k0);
        if (valid && key == k0) {
            my_leader = leader;
            my_rank = __popcll(same & ((1ull << lane) - 1ull));
            my_group = __popcll(same);
        }
        todo &= ~same;
    }
    int base = 0;
    if (valid && lane == my_leader) base = atomicAdd(counter + key, my_group);
    base = __shfl(base, my_leader);
    return base + my_rank;
}

// Fire-and-forget counting at workgroup level: the wave's groups go into a small LDS table first, one global atomic per distinct key
// per WORKGROUP.
// Same-address global atomics are what bounds the row counting (hundreds per voxel per frame, ~12 ns each on one L2 channel).
#define FG_TABLE 256
__device__ __forceinline__ void block_grouped_add_lds(unsigned* __restrict__ tkey, int* __restrict__ tcnt, int* __restrict__ counter, uint32_t key,
                                                      bool valid) {
    const int lane = lane_id();
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
        const unsigned long long same = __ballot(valid && key == k0);
        if (lane == leader) {
            unsigned h = (k0 * 2654435761u) >> 24;                     // FG_TABLE = 256 slots; a workgroup normally holds far fewer distinct keys
            int probes = 0;
            for (; probes < FG_TABLE; ++probes) {
                const unsigned old = atomicCAS(tkey + h, DIF_INVALID_KEY, k0);
                if (old == DIF_INVALID_KEY || old == k0) break;
                h = (h + 1) & (FG_TABLE - 1);
            }
            if (probes < FG_TABLE) atomicAdd(tcnt + h, __popcll(same));
            else atomicAdd(counter + k0, __popcll(same));             // table full (a workgroup spanning > 256 voxels): count directly
        }
        todo &= ~same;
    }
}

__global__ void __launch_bounds__(DIF_BLOCK) k_focus_gather(Geo g, float enc_th, const float* __restrict__ xyz, const int* __restrict__ pt_lin,
                                                          const uint8_t* __restrict__ unq_mask, int64_t N, int* __restrict__ frame_count,
                                                          const int64_t* __restrict__ indexer, const float* __restrict__ obs,
                                                          uint32_t* __restrict__ pair_key, int* __restrict__ seg_cnt,
                                                          int* __restrict__ counters, int64_t capacity, int img_w) {
    __shared__ unsigned tkey[FG_TABLE];
    __shared__ int tcnt[FG_TABLE];
    tkey[threadIdx.x] = DIF_INVALID_KEY;                             // DIF_BLOCK == FG_TABLE
    tcnt[threadIdx.x] = 0;
    // Points of a frame (img_w > 0, N = H * img_w with both multiples of 16) are walked in 16 x 16 pixel tiles: a workgroup then touches
    // ~10 voxels instead of the ~90 that a 256-pixel piece of an image row does, so its rows collapse into few counters.
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // grid covers N rounded up to whole workgroups
    if (img_w > 0) {
        const int tiles_x = img_w >> 4;
        const int ty = (int)(blockIdx.x / tiles_x), tx = (int)(blockIdx.x % tiles_x);
        i = (int64_t)(ty * 16 + (int)(threadIdx.x >> 4)) * img_w + tx * 16 + (int)(threadIdx.x & 15);
    }
    __syncthreads();
    if (i == 0) {
        int n = counters[DIF_C_N_OCCUPIED] + counters[DIF_C_ALLOC_NEW];
        if (n > capacity) { n = (int)capacity; counters[DIF_C_OVERFLOW] = 1; }
        counters[DIF_C_N_OCCUPIED] = n;
    }
    const int lin = (i < N) ? pt_lin[i] : -1;
    uint32_t key[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) key[o] = DIF_INVALID_KEY;
    // get_pruned_surface: own voxel in expand(encode set) <=> own voxel or an in-grid 6-neighbour is in the set.  That is a property of the
    // VOXEL, and neighbouring pixels share voxels: the first lane of every run of equal ids does the seven look-ups, the rest of the run
    // takes its answer (in steady state ~97 % of the points fail this test, so it is most of the kernel's gathers).
    const int lane = lane_id();
    const bool kept = lin >= 0 && unq_mask[i];
    if (lin >= 0) frame_count[lin] = 0;
    const int prev = __shfl_up(lin, 1);
    const bool head = (lane == 0) || (prev != lin);
    const unsigned long long heads = __ballot(head);
    bool focus = false;
    if (head && kept) {
        int ix, iy, iz;
        unlinearize(g, lin, ix, iy, iz);
        // all seven indexer entries first, then all seven observation counts: two dependent loads deep instead of fourteen
        const int plane = g.ny * g.nz;
        const int nl[7] = {lin, lin - plane, lin + plane, lin - g.nz, lin + g.nz, lin - 1, lin + 1};
        const bool in_grid[7] = {true, ix > 0, ix < g.nx - 1, iy > 0, iy < g.ny - 1, iz > 0, iz < g.nz - 1};
        int64_t slot[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) slot[q] = in_grid[q] ? indexer[nl[q]] : -1;
        float w[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) w[q] = slot[q] >= 0 ? obs[slot[q]] : enc_th;
#pragma unroll
        for (int q = 0; q < 7; ++q) focus |= w[q] < enc_th;
    }
    const int my_head = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));      // nearest run start at or below this lane
    focus = __shfl((int)focus, my_head) != 0;
    if (kept && focus) {
        float xn, yn, zn; int ix, iy, iz;
        voxel_of(g, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], xn, yn, zn, ix, iy, iz);
        int64_t slot[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float ox = (o & 4) ? 0.5f : -0.5f, oy = (o & 2) ? 0.5f : -0.5f, oz = (o & 1) ? 0.5f : -0.5f;   // map.py:186-189
            int gx = clampi((int)(ceilf(xn + ox) - 1.0f), 0, g.nx - 1);                                   // map.py:422-424
            int gy = clampi((int)(ceilf(yn + oy) - 1.0f), 0, g.ny - 1);
            int gz = clampi((int)(ceilf(zn + oz) - 1.0f), 0, g.nz - 1);
            slot[o] = indexer[linearize(g, gx, gy, gz)];
        }
        float w[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) w[o] = slot[o] >= 0 ? obs[slot[o]] : enc_th;      // all eight counts in flight together
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (w[o] < enc_th) key[o] = (uint32_t)slot[o];
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (i < N) pair_key[(int64_t)o * N + i] = key[o];
        block_grouped_add_lds(tkey, tcnt, seg_cnt, key[o], key[o] != DIF_INVALID_KEY);
    }
    __syncthreads();
    if (tkey[threadIdx.x] != DIF_INVALID_KEY) atomicAdd(seg_cnt + tkey[threadIdx.x], tcnt[threadIdx.x]);
}

// K5: per-slot encoder work items (ceil(cnt / ITEM_ROWS)) and the item -> slot table.  A slot's rows live in the row table at
// [item_start*ITEM_ROWS, ...) (padded to whole items).
__global__ void __launch_bounds__(DIF_BLOCK) k_alloc_items(const int* __restrict__ seg_cnt, int* __restrict__ item_start, int* __restrict__ item_slot,
                                                         int* __restrict__ counters, int64_t max_items) {
    // Items are handed out with one atomic per workgroup instead of an ordered scan: WHICH items a voxel gets only decides where its
    // partial sums live and which wave encodes them; k_fuse adds a voxel's partials in its own fixed order, so results do not change.
    __shared__ int smem[8];
    __shared__ int s_base;
    const int n = counters[DIF_C_N_OCCUPIED];
    for (int s0 = (int)(blockIdx.x * blockDim.x); s0 < n; s0 += (int)(gridDim.x * blockDim.x)) {
        const int s = s0 + (int)threadIdx.x;
        const int cnt = (s < n) ? seg_cnt[s] : 0;
        const int nit = (cnt + ITEM_ROWS - 1) / ITEM_ROWS;
        int total;
        const int ex = block_excl_scan(nit, smem, total);
        const int rows = block_sum(cnt, smem), vox = block_sum(cnt > 0 ? 1 : 0, smem);      // M and C of map.py:434-437
        if (threadIdx.x == 0) {
            s_base = total ? atomicAdd(counters + DIF_C_ITEMS, total) : 0;
            if (vox) { atomicAdd(counters + DIF_C_C, vox); atomicAdd(counters + DIF_C_M, rows); }
        }
        __syncthreads();
        if (nit) {
            const int off = s_base + ex;
            item_start[s] = off;
            if ((int64_t)off + nit > max_items) counters[DIF_C_OVERFLOW] = 4;
            else for (int k = 0; k < nit; ++k) item_slot[off + k] = s;
        }
        __syncthreads();
    }
}

// K6: place every valid (offset, point) pair into its slot's rows.  Order inside a slot is arrival order — harmless, because
// the per-voxel sum is accumulated in exact fixed point (order-independent, see k_encode).
__global__ void __launch_bounds__(DIF_BLOCK) k_scatter_rows(const uint32_t* __restrict__ pair_key, int64_t n_pairs, const int* __restrict__ item_start,
                                                          int* __restrict__ seg_cursor, uint32_t* __restrict__ row_val, int64_t max_rows) {
    const int64_t n_pad = (n_pairs + 63) / 64 * 64;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_pad; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t key = (j < n_pairs) ? pair_key[j] : DIF_INVALID_KEY;
        const bool valid = key != DIF_INVALID_KEY;
        const int first_item = valid ? item_start[key] : 0;          // in flight together with the atomics below
        const int r = wave_grouped_fetch_add(seg_cursor, key, valid);
        if (valid) {
            const int64_t pos = (int64_t)first_item * ITEM_ROWS + r;
            if (pos < max_rows) row_val[pos] = (uint32_t)j;
        }
    }
}

// =================================================================================================================
// a7..a9 : gather + encoder (MFMA) + per-voxel sums
// =================================================================================================================
// Persistent: one 512-thread workgroup per CU keeps the 107 KB of packed encoder weights in LDS; each wave pulls work
// items (slot, 32 rows), runs the tile through the MFMA chain and reduces the 29 output features over the rows.
// The reduction is done in 2^-30 FIXED POINT (int64): integer addition is associative, so the per-voxel sum does not
// depend on row order, tile grouping or the order partials are added in => bit-reproducible, and more accurate than an
// fp32 running sum (the reference sums with float atomics in arbitrary order, indexing.cu:59-71).
#define DIF_FIX_SCALE 1073741824.0f          /* 2^30: |enc| < 2^12 and < 2^21 rows per voxel keep the sum inside int64 */
__global__ void __launch_bounds__(512, 2)
k_encode(Geo g, const float* __restrict__ wblob, const float* __restrict__ xyz, const float* __restrict__ normal, int64_t N,
         const uint32_t* __restrict__ row_val, const int* __restrict__ seg_cnt, const int* __restrict__ item_start,
         const int* __restrict__ item_slot, const int* __restrict__ counters, long long* __restrict__ partial /* [items][32] */, int max_items) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_weights(lds, wblob, ENC_FLOATS);
    const int lane = lane_id(), half = lane >> 5, col = lane & 31;
    // work item w goes to wave (w / #blocks) of block (w % #blocks): a partly filled launch spreads over all CUs and SIMDs first
    const int wave = (int)((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
    const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    const int n_items = min(counters[DIF_C_ITEMS], max_items);
    for (int item = wave; item < n_items; item += nwaves) {
        const int slot = item_slot[item];
        const int chunk = item - item_start[slot];
        const bool live = chunk * ITEM_ROWS + col < seg_cnt[slot];
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (live) {
            uint32_t v = row_val[(int64_t)item * ITEM_ROWS + col];
            int o = 0;
#pragma unroll
            for (int k = 1; k < 8; ++k) o += ((int64_t)v >= (int64_t)k * N) ? 1 : 0;
            int64_t i = (int64_t)v - (int64_t)o * N;
            float xn = normalize1(xyz[i * 3 + 0], g.bx, g.vs);
            float yn = normalize1(xyz[i * 3 + 1], g.by, g.vs);
            float zn = normalize1(xyz[i * 3 + 2], g.bz, g.vs);
            float ox = (o & 4) ? 0.5f : -0.5f, oy = (o & 2) ? 0.5f : -0.5f, oz = (o & 1) ? 0.5f : -0.5f;
            float gx = fminf(fmaxf(ceilf(xn + ox) - 1.0f, 0.0f), (float)(g.nx - 1));
            float gy = fminf(fmaxf(ceilf(yn + oy) - 1.0f, 0.0f), (float)(g.ny - 1));
            float gz = fminf(fmaxf(ceilf(zn + oz) - 1.0f, 0.0f), (float)(g.nz - 1));
            float rx = (xn - gx) - 0.5f, ry = (yn - gy) - 0.5f, rz = (zn - gz) - 0.5f;      // map.py:425
            float nxv = normal[i * 3 + 0], nyv = normal[i * 3 + 1], nzv = normal[i * 3 + 2];
            x0 = half ? ry : rx;
            x1 = half ? nxv : rz;
            x2 = half ? nzv : nyv;
        }
        f16v out = encoder_tile(lds, x0, x1, x2, lane);
        long long* p = partial + (int64_t)item * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long long v = live ? __float2ll_rn(out[r] * DIF_FIX_SCALE) : 0ll;
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            if (col == 0) p[(r & 3) + 8 * (r >> 2) + 4 * half] = v;
        }
    }
}

// a10: fusion update (map.py:448-452).  One 32-lane group per slot.
__global__ void __launch_bounds__(DIF_BLOCK) k_fuse(const long long* __restrict__ partial, const int* __restrict__ item_start, const int* __restrict__ item_slot,
                                                  int* __restrict__ seg_cnt, int* __restrict__ seg_cursor, float* __restrict__ latent, float* __restrict__ obs,
                                                  uint8_t* __restrict__ dirty, int* __restrict__ counters, int max_items) {
    const int n_items = min(counters[DIF_C_ITEMS], max_items);
    const int grp = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 5), ngrp = (int)((gridDim.x * blockDim.x) >> 5);
    const int f = threadIdx.x & 31;
    // walk the work items (a few thousand) instead of every allocated slot: the first item of a slot fuses the whole slot
    for (int it0 = grp; it0 < n_items; it0 += ngrp) {
        const int s = item_slot[it0];
        if (item_start[s] != it0) continue;
        const int cnt = seg_cnt[s];
        const int nit = (cnt + ITEM_ROWS - 1) / ITEM_ROWS;
        if (f < L) {
            long long Si = 0;
            for (int k = 0; k < nit; ++k) Si += partial[(int64_t)(it0 + k) * 32 + f];
            float S = (float)Si * (1.0f / DIF_FIX_SCALE);    // one rounding: exact integer sum -> nearest float
            float w_old = obs[s];
            float z_old = latent[(int64_t)s * L + f];
            S = S + z_old * w_old;                           // map.py:449
            float w_new = w_old + (float)cnt;                // map.py:450
            latent[(int64_t)s * L + f] = S / w_new;          // map.py:451
        }
        __builtin_amdgcn_wave_barrier();
        if (f == 31) {                                       // after every lane of the group has read obs[s]
            obs[s] = obs[s] + (float)cnt;
            dirty[s] = 1;                                    // map.py:452
            seg_cnt[s] = 0;
            seg_cursor[s] = 0;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[DIF_C_N_FUSED] = counters[DIF_C_N_OCCUPIED];   // the slots an overlapped extract may look at
}

