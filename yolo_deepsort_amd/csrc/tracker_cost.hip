// Cost matrices of the association (appearance: deep_sort/sort/nn_matching.py:30-53,77-100,158-187 with the gate and threshold of
// linear_assignment.py:52,147-203 fused; IOU: iou_matching.py:5-91) and the tracker-side NMS (preprocessing.py:6-73).
#include "tracker_dev.h"

namespace yds {

// ------------------------------------------------------------------------------------- appearance cost
// cost[t][d] = min over the gallery rows of track t of 1 - <g/|g|, f/|f|>, then Mahalanobis gate and the
// min_cost_matching clamp.  Gallery rows are normalised once when they are appended and detections once per frame
// (normalize_rows_kernel) - the same division the reference repeats on every call.  One workgroup per (track,
// 16-detection slab): the slab and 16 gallery rows at a time sit in LDS (rows padded by one float: conflict free),
// thread (r, d) owns one dot product per chunk and keeps a running minimum.
__global__ void normalize_rows_kernel(const float *src, const int *src_idx, float *dst, int n, int normalise) {
    // one wave per row: dst[row] = src[idx[row]] / ||src[idx[row]]||  (plain gather when !normalise: euclidean metric)
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float *f = src + (size_t)(src_idx ? src_idx[row] : row) * EMB;
    float v[EMB / 64], ss = 0.f;
#pragma unroll
    for (int k = 0; k < EMB / 64; ++k) { v[k] = f[lane + 64 * k]; ss += v[k] * v[k]; }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = normalise ? sqrtf(ss) : 1.f;
#pragma unroll
    for (int k = 0; k < EMB / 64; ++k) dst[(size_t)row * EMB + lane + 64 * k] = v[k] / nrm;
}

// Track rows are named either directly (slots / n_rows, the stand-alone entry) or through the device-resident track table:
// row t = track idx[t] of the table (tab_slot / tab_nfeat), t < *count_p (device-side count; surplus blocks return).
__global__ __launch_bounds__(256) void appearance_cost_kernel(const float *gallery_n, const int *slots, const int *n_rows, int budget,
                                                             const float *feats_n, int D, const float *mean, const float *cov,
                                                             const float *tlwh, float max_dist, float flood, int do_gate, int euclid, float *cost,
                                                             const int *idx, const int *tab_slot, const int *tab_nfeat, const int *count_p) {
    const int t = blockIdx.x;
    if (count_p && t >= *count_p) return;
    const int slot = idx ? tab_slot[idx[t]] : slots[t], rows = idx ? tab_nfeat[idx[t]] : n_rows[t];
    appearance_cost_block(gallery_n, slot, rows, budget, feats_n, D, blockIdx.y * 16, mean + (size_t)slot * 8, cov + (size_t)slot * 64, tlwh, max_dist,
                          flood, do_gate, euclid, cost + (size_t)t * D);
}

// ------------------------------------------------------------------------------------------ IOU cost
// dims_p (optional): device-side {T, D}; cand / tab_slot / tab_tsu (optional): row t = track cand[t] of the track table
__global__ void iou_cost_kernel(const float *mean, const int *slots, const int *stale, int T, const float *tlwh, const int *det_idx,
                                int D, float max_dist, float flood, float *cost, const int *dims_p, const int *cand, const int *tab_slot,
                                const int *tab_tsu) {
    if (dims_p) { T = dims_p[0]; D = dims_p[1]; }
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * D) return;
    int t = idx / D, d = idx - t * D;
    const float *m = mean + (size_t)(cand ? tab_slot[cand[t]] : slots[t]) * 8;
    const bool is_stale = cand ? tab_tsu[cand[t]] > 1 : (stale && stale[t]);
    cost[idx] = iou_cost_entry(m, tlwh + (size_t)det_idx[d] * 4, is_stale, max_dist, flood);
}

// ------------------------------------------------------------------------------- tracker-side NMS
// deep_sort/sort/preprocessing.py:6-73 (gate: deep_sort.py:52-57): greedy suppression in float64 over tlwh boxes with
// the +1 pixel convention; walks `order` (= np.argsort(scores)) from its END, suppresses j when
// inter(i, j) / area(j) > max_overlap.  pick[] receives the surviving detection indices in pick order.
__global__ __launch_bounds__(256) void tracker_nms_kernel(const float *tlwh, const int *order, int n, double max_overlap, int *pick, int *n_pick) {
    extern __shared__ int alive[];                               // alive[k] for position k of `order`
    __shared__ int cur, count;
    for (int k = threadIdx.x; k < n; k += blockDim.x) alive[k] = 1;
    if (threadIdx.x == 0) { cur = n - 1; count = 0; }
    __syncthreads();
    while (true) {
        const int last = cur;
        if (last < 0) break;
        const int i = order[last];
        const double ix1 = tlwh[i * 4], iy1 = tlwh[i * 4 + 1], ix2 = (double)tlwh[i * 4 + 2] + ix1, iy2 = (double)tlwh[i * 4 + 3] + iy1;
        for (int k = threadIdx.x; k < last; k += blockDim.x) {
            if (!alive[k]) continue;
            const int j = order[k];
            const double x1 = tlwh[j * 4], y1 = tlwh[j * 4 + 1], x2 = (double)tlwh[j * 4 + 2] + x1, y2 = (double)tlwh[j * 4 + 3] + y1;
            const double area = (x2 - x1 + 1.0) * (y2 - y1 + 1.0);
            const double w = fmax(0.0, fmin(ix2, x2) - fmax(ix1, x1) + 1.0), h = fmax(0.0, fmin(iy2, y2) - fmax(iy1, y1) + 1.0);
            if ((w * h) / area > max_overlap) alive[k] = 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            pick[count++] = i;
            int k = last - 1;
            while (k >= 0 && !alive[k]) --k;
            cur = k;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_pick = count;
}

}  // namespace yds
