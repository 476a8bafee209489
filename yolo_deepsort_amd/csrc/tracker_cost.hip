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
    __shared__ float fs[16][EMB + 1], gs[16][EMB + 1];
    __shared__ float best[16][17];
    const int t = blockIdx.x, d0 = blockIdx.y * 16;
    if (count_p && t >= *count_p) return;
    const int nd = min(16, D - d0);
    const int slot = idx ? tab_slot[idx[t]] : slots[t], rows = idx ? tab_nfeat[idx[t]] : n_rows[t];
    for (int i = threadIdx.x; i < 16 * (EMB / 4); i += blockDim.x) {        // detection slab, float4 coalesced
        const int d = i / (EMB / 4), k4 = i % (EMB / 4);
        float4 v = d < nd ? *reinterpret_cast<const float4 *>(feats_n + (size_t)(d0 + d) * EMB + k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        fs[d][k4 * 4] = v.x; fs[d][k4 * 4 + 1] = v.y; fs[d][k4 * 4 + 2] = v.z; fs[d][k4 * 4 + 3] = v.w;
    }
    const int r = threadIdx.x >> 4, d = threadIdx.x & 15;
    float run_min = INFINITY;
    for (int g0 = 0; g0 < rows; g0 += 16) {
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * (EMB / 4); i += blockDim.x) {
            const int g = i / (EMB / 4), k4 = i % (EMB / 4);
            float4 v = g0 + g < rows ? *reinterpret_cast<const float4 *>(gallery_n + ((size_t)slot * budget + g0 + g) * EMB + k4 * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            gs[g][k4 * 4] = v.x; gs[g][k4 * 4 + 1] = v.y; gs[g][k4 * 4 + 2] = v.z; gs[g][k4 * 4 + 3] = v.w;
        }
        __syncthreads();
        if (g0 + r < rows) {
            float dot = 0.f;
            if (euclid) {                                        // _pdist nn_matching.py:4-27: sum (a - b)^2
#pragma unroll 8
                for (int k = 0; k < EMB; ++k) { const float df = gs[r][k] - fs[d][k]; dot += df * df; }
                run_min = fminf(run_min, dot);
            } else {
#pragma unroll 8
                for (int k = 0; k < EMB; ++k) dot += gs[r][k] * fs[d][k];
                run_min = fminf(run_min, 1.f - dot);
            }
        }
    }
    best[r][d] = run_min;
    __syncthreads();
    if ((int)threadIdx.x < nd) {
        float c = INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) c = fminf(c, best[q][threadIdx.x]);
        if (euclid) c = fmaxf(c, 0.f);                           // torch.clamp(min=0) nn_matching.py:74
        const int dd = d0 + threadIdx.x;
        if (do_gate) {
            float z[4];
            to_xyah(tlwh + (size_t)dd * 4, z);
            if (gate2(mean + (size_t)slot * 8, cov + (size_t)slot * 64, z) > CHI2_2DOF) c = INFTY_COST;
        }
        if (max_dist > 0.f && c > max_dist) c = flood;            // linear_assignment.py:52
        cost[(size_t)t * D + dd] = c;
    }
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
    float bw = m[2] * m[3], bh = m[3];                        // Track.to_tlwh track.py:81-94
    float bx = m[0] - bw / 2.f, by = m[1] - bh / 2.f;
    const float *c = tlwh + (size_t)det_idx[d] * 4;
    float ix0 = fmaxf(bx, c[0]), iy0 = fmaxf(by, c[1]);
    float ix1 = fminf(bx + bw, c[2] + c[0]), iy1 = fminf(by + bh, c[3] + c[1]);
    float iw = fmaxf(ix1 - ix0 + 1.f, 0.f), ih = fmaxf(iy1 - iy0 + 1.f, 0.f);      // asymmetric +1, iou_matching.py:36
    float inter = iw * ih;
    float v = 1.f - inter / (bw * bh + c[2] * c[3] - inter);
    if (cand ? tab_tsu[cand[t]] > 1 : (stale && stale[t])) v = INFTY_COST;      // time_since_update > 1, iou_matching.py:86-89
    if (max_dist > 0.f && v > max_dist) v = flood;
    cost[idx] = v;
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
