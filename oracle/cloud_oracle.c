/* CPU ORACLE (test infrastructure only): exhaustive k-nearest-neighbour search of a point cloud inside itself.
 * Restates what the reference's kd-tree search returns (ext/pcproc/cuda_kdtree.cu:1173-1240 knnSearch, exact, ascending;
 * squared L2 of CudaL2::dist, :1152-1155) without any index structure: every pair is examined.
 * Order: (d2, index); self included; float32 arithmetic, unfused (build with -ffp-contract=off).
 * Points with a non-finite coordinate have no neighbours and are nobody's neighbour.
 * Entries with d2 >= radius^2, and missing ones, are (-1, +inf). */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

void cloud_oracle_knn(const float* pc, int64_t n, int stride, int k, float radius, int32_t* out_idx, float* out_dist) {
    const float r2 = radius * radius;
    for (int64_t i = 0; i < n; ++i) {
        int32_t* oi = out_idx + (size_t)i * k;
        float* od = out_dist + (size_t)i * k;
        for (int j = 0; j < k; ++j) { oi[j] = -1; od[j] = INFINITY; }
        const float qx = pc[i * stride], qy = pc[i * stride + 1], qz = pc[i * stride + 2];
        if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) continue;
        int filled = 0;
        for (int64_t c = 0; c < n; ++c) {
            const float px = pc[c * stride], py = pc[c * stride + 1], pz = pc[c * stride + 2];
            if (!(isfinite(px) && isfinite(py) && isfinite(pz))) continue;
            const float dx = px - qx, dy = py - qy, dz = pz - qz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (!(d2 < r2)) continue;
            /* candidates arrive in ascending index: a tie with the current worst never displaces it */
            if (filled == k && !(d2 < od[k - 1])) continue;
            int pos = filled < k ? filled : k - 1;
            while (pos > 0 && od[pos - 1] > d2) { od[pos] = od[pos - 1]; oi[pos] = oi[pos - 1]; --pos; }
            od[pos] = d2; oi[pos] = (int32_t)c;
            if (filled < k) ++filled;
        }
    }
}
