// DeepSORT association stage: batched HIP kernels over all tracks / detections + host bookkeeping.
//
// Replaces the per-object torch/numpy/scipy code of the reference:
//   KalmanFilter.initiate/predict/project/update/gating_distance   deep_sort/sort/kalman_filter.py:54-256
//   NearestNeighborDistanceMetric.distance (_nn_cosine_distance)     deep_sort/sort/nn_matching.py:30-53,77-100,158-187
//   gate_cost_matrix + min_cost_matching thresholding                deep_sort/sort/linear_assignment.py:52,147-203
//   iou / iou_cost                                                   deep_sort/sort/iou_matching.py:5-91
//   scipy.optimize.linear_sum_assignment (third party)               call site linear_assignment.py:56
//   DeepSort.update output stage                                     deep_sort/deep_sort.py:63-88,108-114
// ALL track state lives in HBM: mean, covariance and the appearance gallery ring by slot id, and the integer lifecycle
// table (ids, hits, age, time_since_update, Tentative/Confirmed/Deleted, ring position, payload) in track-list order.
// The order-sensitive list bookkeeping (linear_assignment.py:58-72, tracker.py:56-93,115-176, track.py) runs in two
// single-workgroup kernels built from ordered compactions (ballot + prefix), so a frame is a fixed sequence of launches
// whose sizes are read from device memory: the host synchronises ONCE per frame - or once per batch of frames
// (step_batch, used by the pipeline) - to fetch the int32 rows.
#include "engine.h"

#include <algorithm>
#include <math.h>
#include <string.h>

namespace yds {

constexpr int EMB = 512;
constexpr float INFTY_COST = 1e5f;
constexpr float CHI2_2DOF = 5.9915f;

// ------------------------------------------------------------------------------------------ Kalman
// std weights are fp32 roundings of 1/20 and 1/160 like the reference's tensors (kalman_filter.py:39-52)
__device__ __constant__ float kStdPos = 1.f / 20, kStdVel = 1.f / 160;

__device__ __forceinline__ void kf_predict_body(float *m, float *P) {
    const float h = m[3];
    float q[8];
    float sp = h * kStdPos, sv = h * kStdVel;
    q[0] = sp * sp; q[1] = q[0]; q[2] = 1e-2f * 1e-2f; q[3] = q[0];
    q[4] = sv * sv; q[5] = q[4]; q[6] = 1e-5f * 1e-5f; q[7] = q[4];
    float A[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) A[i][j] = i < 4 ? P[i * 8 + j] + P[(i + 4) * 8 + j] : P[i * 8 + j];       // F P
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = j < 4 ? A[i][j] + A[i][j + 4] : A[i][j];                                                 // (F P) F^T
            if (i == j) v += q[i];
            P[i * 8 + j] = v;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = m[i] + m[i + 4];
}
__global__ void kf_predict_kernel(float *mean, float *cov, const int *slots, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    kf_predict_body(mean + (size_t)slots[t] * 8, cov + (size_t)slots[t] * 64);
}

__device__ __forceinline__ void project4(const float *m, const float *P, float S[4][4]) {
    float sp = m[3] * kStdPos;
    float d[4] = {sp * sp, sp * sp, 1e-1f * 1e-1f, sp * sp};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i][j] = P[i * 8 + j] + (i == j ? d[i] : 0.f);
}

// z: xyah per match; solves S K^T = (P H)^T by LU with partial pivoting, then the K S K^T form
__device__ __forceinline__ void kf_update_body(float *m, float *P, const float *zt) {
    float S[4][4], LU[4][4], Kt[4][8];
    project4(m, P, S);
    int piv[4] = {0, 1, 2, 3};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) LU[i][j] = S[i][j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) Kt[i][j] = P[j * 8 + i];                  // (P H)^T
    for (int k = 0; k < 4; ++k) {
        int p = k;
        float best = fabsf(LU[k][k]);
        for (int r = k + 1; r < 4; ++r)
            if (fabsf(LU[r][k]) > best) { best = fabsf(LU[r][k]); p = r; }
        if (p != k) {
            for (int j = 0; j < 4; ++j) { float tmp = LU[k][j]; LU[k][j] = LU[p][j]; LU[p][j] = tmp; }
            for (int j = 0; j < 8; ++j) { float tmp = Kt[k][j]; Kt[k][j] = Kt[p][j]; Kt[p][j] = tmp; }
            int tp = piv[k]; piv[k] = piv[p]; piv[p] = tp;
        }
        for (int r = k + 1; r < 4; ++r) {
            float f = LU[r][k] / LU[k][k];
            for (int j = k + 1; j < 4; ++j) LU[r][j] -= f * LU[k][j];
            for (int j = 0; j < 8; ++j) Kt[r][j] -= f * Kt[k][j];
        }
    }
    for (int k = 3; k >= 0; --k) {
        for (int j = 0; j < 8; ++j) {
            float v = Kt[k][j];
            for (int r = k + 1; r < 4; ++r) v -= LU[k][r] * Kt[r][j];
            Kt[k][j] = v / LU[k][k];
        }
    }
    float innov[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) innov[i] = zt[i] - m[i];
    float KS[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) v += Kt[c][a] * S[c][b];
            KS[a][b] = v;
        }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) v += KS[a][c] * Kt[c][b];
            P[a * 8 + b] = P[a * 8 + b] - v;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) v += innov[i] * Kt[i][j];
        m[j] = m[j] + v;
    }
}
__global__ void kf_update_kernel(float *mean, float *cov, const int *slots, const float *z, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    kf_update_body(mean + (size_t)slots[t] * 8, cov + (size_t)slots[t] * 64, z + t * 4);
}

// new tracks from detections (kalman_filter.py:54-87 + detection.py:41-48)
__device__ __forceinline__ void kf_initiate_body(float *m, float *P, const float *b) {
    float w = b[2], h = b[3];
    float cx = b[0] + w / 2.f, cy = b[1] + h / 2.f, a = w / h;
    m[0] = cx; m[1] = cy; m[2] = a; m[3] = h; m[4] = m[5] = m[6] = m[7] = 0.f;
    const float cp = (float)(2 * (1. / 20)), cv = (float)(10 * (1. / 160));
    float sp = cp * h, sv = cv * h;
    float d[8] = {sp * sp, sp * sp, 1e-2f * 1e-2f, sp * sp, sv * sv, sv * sv, 1e-5f * 1e-5f, sv * sv};
    for (int i = 0; i < 64; ++i) P[i] = 0.f;
    for (int i = 0; i < 8; ++i) P[i * 9] = d[i];
}
__global__ void kf_initiate_kernel(float *mean, float *cov, const int *slots, const float *tlwh, const int *det_idx, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    kf_initiate_body(mean + (size_t)slots[t] * 8, cov + (size_t)slots[t] * 64, tlwh + (size_t)det_idx[t] * 4);
}

__device__ __forceinline__ void to_xyah(const float *b, float z[4]) {
    z[0] = b[0] + b[2] / 2.f; z[1] = b[1] + b[3] / 2.f; z[2] = b[2] / b[3]; z[3] = b[3];
}

__global__ void tlwh_to_xyah_kernel(const float *tlwh, const int *det_idx, float *z, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    to_xyah(tlwh + (size_t)det_idx[t] * 4, z + t * 4);
}

// squared Mahalanobis distance on (x, y) only (only_position=True, tracker.py:61-63)
__device__ __forceinline__ float gate2(const float *m, const float *P, const float *z) {
    float sp = m[3] * kStdPos;
    float s00 = P[0] + sp * sp, s01 = P[1], s10 = P[8], s11 = P[9] + sp * sp;
    float det = s00 * s11 - s01 * s10;
    float i00 = s11 / det, i01 = -s01 / det, i10 = -s10 / det, i11 = s00 / det;
    float d0 = z[0] - m[0], d1 = z[1] - m[1];
    float t0 = d0 * i00 + d1 * i10, t1 = d0 * i01 + d1 * i11;
    return t0 * d0 + t1 * d1;
}

__global__ void gating_kernel(const float *mean, const float *cov, const int *slots, int T, const float *xyah, int D, float *out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * D) return;
    int t = idx / D, d = idx - t * D;
    out[idx] = gate2(mean + (size_t)slots[t] * 8, cov + (size_t)slots[t] * 64, xyah + d * 4);
}

// squared Mahalanobis distance on all four measurement dimensions (only_position=False, kalman_filter.py:236-254):
// d S^-1 d^T with S^-1 by Gauss-Jordan on the 4x4 projected covariance (torch.inverse in the reference)
__device__ __forceinline__ float gate4(const float *m, const float *P, const float *z) {
    float S[4][4], I[4][4];
    project4(m, P, S);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) I[i][j] = i == j ? 1.f : 0.f;
    for (int k = 0; k < 4; ++k) {
        int p = k;
        float best = fabsf(S[k][k]);
        for (int r = k + 1; r < 4; ++r)
            if (fabsf(S[r][k]) > best) { best = fabsf(S[r][k]); p = r; }
        if (p != k)
            for (int j = 0; j < 4; ++j) { float a = S[k][j]; S[k][j] = S[p][j]; S[p][j] = a; float b = I[k][j]; I[k][j] = I[p][j]; I[p][j] = b; }
        const float inv = 1.f / S[k][k];
        for (int j = 0; j < 4; ++j) { S[k][j] *= inv; I[k][j] *= inv; }
        for (int r = 0; r < 4; ++r) {
            if (r == k) continue;
            const float f = S[r][k];
            for (int j = 0; j < 4; ++j) { S[r][j] -= f * S[k][j]; I[r][j] -= f * I[k][j]; }
        }
    }
    float d[4], t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = z[i] - m[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = d[0] * I[0][j] + d[1] * I[1][j] + d[2] * I[2][j] + d[3] * I[3][j];
    return t[0] * d[0] + t[1] * d[1] + t[2] * d[2] + t[3] * d[3];
}
__global__ void gating4_kernel(const float *mean, const float *cov, int T, const float *xyah, int D, float *out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * D) return;
    int t = idx / D, d = idx - t * D;
    out[idx] = gate4(mean + (size_t)t * 8, cov + (size_t)t * 64, xyah + d * 4);
}
// KalmanFilter.initiate from (x, y, a, h) rows (kalman_filter.py:54-87) and KalmanFilter.project (:125-158), stand-alone
__global__ void kf_initiate_xyah_kernel(const float *xyah, float *mean, float *cov, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float *z = xyah + (size_t)t * 4;
    const float b[4] = {z[0] - z[2] * z[3] / 2.f, z[1] - z[3] / 2.f, z[2] * z[3], z[3]};
    kf_initiate_body(mean + (size_t)t * 8, cov + (size_t)t * 64, b);
    float *m = mean + (size_t)t * 8;
    m[0] = z[0]; m[1] = z[1]; m[2] = z[2]; m[3] = z[3];        // the measurement itself, not a tlwh round trip
}
__global__ void kf_project_kernel(const float *mean, const float *cov, float *mean4, float *cov16, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float S[4][4];
    project4(mean + (size_t)t * 8, cov + (size_t)t * 64, S);
    for (int i = 0; i < 4; ++i) {
        mean4[(size_t)t * 4 + i] = mean[(size_t)t * 8 + i];
        for (int j = 0; j < 4; ++j) cov16[(size_t)t * 16 + i * 4 + j] = S[i][j];
    }
}

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

// ------------------------------------------------------------------------------------------ LSAP
// scipy.optimize.linear_sum_assignment (rectangular_lsap.cpp, Crouse 2016) on ONE workgroup of four wavefronts.  The
// augmenting-path search is sequential over rows; its column scan is spread over 256 lanes and the sequential tie-break of
// the scalar scan is reproduced exactly:
//   index = last unassigned column (in `remaining` order) among the minimum, else the first minimum
// encoded as a key so that ONE lexicographic (cost, key) reduction per Dijkstra step finds it.  Arithmetic is fp64 in the
// same order as scipy (minVal + c - u[i] - v[j]).  Tall matrices are solved transposed.  Position `it` of `remaining` is
// always scanned - and rewritten - by thread it % 256, the winner's column travels with the reduction, and the per-wave
// partial results are double buffered, so a Dijkstra step costs one barrier.  Solver state lives in LDS (and the cost
// matrix too when it fits); beyond ~3000 rows/columns it moves to a global scratch buffer - no size limit.
// dims_p (optional): device-side {nr, nc}.  row_out/col_out: min(nr,nc) pairs sorted by row, *n_out = that count.
constexpr int LSAP_NT = 256, LSAP_NW = LSAP_NT / 64;
#ifndef YDS_LSAP_REG
#define YDS_LSAP_REG 1                      // experiment builds: 0 = LDS-state workgroup form for every size above 64 columns
#endif
#ifndef YDS_LSAP_WAVE_COLS
#define YDS_LSAP_WAVE_COLS 64
#endif
constexpr int LSAP_WAVE_COLS = YDS_LSAP_WAVE_COLS;         // problems up to this many columns go to the single-wavefront kernel (below)
constexpr size_t LSAP_STATE_BYTES = 3 * sizeof(double) + 6 * sizeof(int);      // per row / column
constexpr size_t LSAP_LDS_MAX = 150 * 1024;

// Cross-lane helpers of the LSAP kernels.  Everything is passed as scalars: with the candidate in a struct handled through
// references the compiler kept it in scratch memory (a global-memory round trip per use inside a latency-bound loop).
template <int CTRL> __device__ __forceinline__ int lsap_dpp_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ double lsap_dpp_d(double x) {
    union { double d; int i[2]; } u, w;
    u.d = x;
    w.i[0] = lsap_dpp_i<CTRL>(u.i[0]);
    w.i[1] = lsap_dpp_i<CTRL>(u.i[1]);
    return w.d;
}
__device__ __forceinline__ double lsap_readlane_d(double x, int lane) {
    union { double d; int i[2]; } u, w;
    u.d = x;
    w.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    w.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return w.d;
}
// (v, key) <- lexicographic minimum with (ov, ok)
#define LSAP_TAKE_MIN(v, key, ov, ok)                                      \
    do {                                                                   \
        const double _ov = (ov);                                           \
        const int _ok = (ok);                                              \
        const bool _t = _ov < (v) || (_ov == (v) && _ok < (key));          \
        (v) = _t ? _ov : (v);                                              \
        (key) = _t ? _ok : (key);                                          \
    } while (0)
// lexicographic (cost, key) minimum over the wavefront, result in every lane (uniform).  Two phases instead of one
// 96-bit lexicographic butterfly (round 2: ~80 dependent instructions, 1000+ cycles of every Dijkstra step):
//   1. the minimum VALUE alone: v_min_f64 over four DPP butterflies inside each row of 16 lanes (quad_perm xor 1, xor 2,
//      row_half_mirror, row_mirror), then the four row results through v_readlane;
//   2. lanes holding that value keep their key, the others 0x7fffffff; the minimum KEY with v_min_i32 on DPP operands,
//      rows combined on the scalar ALU.
// Identical result: (min value, smallest key among the lanes that attain it).  Values are never NaN; +inf marks dead lanes.
// (__shfl_xor lowers to ds_bpermute_b32 here: dependent LDS-crossbar round trips.)
__device__ __forceinline__ void lsap_wave_min(double &v, int &key) {
    double m = v;
    m = fmin(m, lsap_dpp_d<0xB1>(m));
    m = fmin(m, lsap_dpp_d<0x4E>(m));
    m = fmin(m, lsap_dpp_d<0x141>(m));
    m = fmin(m, lsap_dpp_d<0x140>(m));
    const double r = fmin(fmin(lsap_readlane_d(m, 0), lsap_readlane_d(m, 16)), fmin(lsap_readlane_d(m, 32), lsap_readlane_d(m, 48)));
    int k = v == r ? key : 0x7fffffff;
    k = min(k, lsap_dpp_i<0xB1>(k));
    k = min(k, lsap_dpp_i<0x4E>(k));
    k = min(k, lsap_dpp_i<0x141>(k));
    k = min(k, lsap_dpp_i<0x140>(k));
    const int rk = min(min(__builtin_amdgcn_readlane(k, 0), __builtin_amdgcn_readlane(k, 16)),
                       min(__builtin_amdgcn_readlane(k, 32), __builtin_amdgcn_readlane(k, 48)));
    v = r;
    key = rk;
}

// GSTATE: solver state in the global scratch buffer (huge problems) instead of LDS - a compile-time choice, so that the LDS
// version addresses its state with ds_read / ds_write (a pointer that may be either makes every access a flat_load)
template <bool GSTATE, bool COST_LDS>
__device__ __forceinline__ void lsap_wg_solve(const float *cost, int nr0, int nc0, int *row_out, int *col_out, int *n_out, char *state_global) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    extern __shared__ __attribute__((aligned(16))) char lsap_smem[];
    __shared__ double red_val[2][LSAP_NW];
    __shared__ int red_key[2][LSAP_NW], red_col[2][LSAP_NW];
    const int n = max(nr, nc);
    double *u, *v, *spc;
    int *path, *col4row, *row4col, *remaining, *SR, *SC;
    auto carve = [&](char *base) {
        u = reinterpret_cast<double *>(base); v = u + n; spc = v + n;
        path = reinterpret_cast<int *>(spc + n); col4row = path + n; row4col = col4row + n; remaining = row4col + n; SR = remaining + n; SC = SR + n;
    };
    if (GSTATE) carve(state_global); else carve(lsap_smem);
    float *cost_lds = reinterpret_cast<float *>(lsap_smem + (GSTATE ? 0 : (size_t)n * LSAP_STATE_BYTES));
    if (COST_LDS) {
        for (int i = tid; i < nr0 * nc0; i += LSAP_NT) cost_lds[i] = cost[i];
    }
    // (two typed accesses, not one pointer that may be LDS or global: that would be a flat_load in the inner loop)
    auto C = [&](int i, int j) -> double {
        const int at = transpose ? j * nc0 + i : i * nc0 + j;
        return (double)(COST_LDS ? cost_lds[at] : cost[at]);
    };
    for (int i = tid; i < nr; i += LSAP_NT) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = tid; j < nc; j += LSAP_NT) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    __syncthreads();
    int parity = 0;
    for (int cur = 0; cur < nr; ++cur) {
        for (int i = tid; i < nr; i += LSAP_NT) SR[i] = 0;
        for (int j = tid; j < nc; j += LSAP_NT) { SC[j] = 0; spc[j] = INFINITY; remaining[j] = nc - j - 1; }   // position it <-> thread it % 256
        __syncthreads();
        double minVal = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            if (tid == 0) SR[i] = 1;
            const double ui = u[i];
            // candidate = lexicographic minimum of (shortest path cost, key): among equal costs the LAST unassigned column in
            // `remaining` order, otherwise the FIRST column: unassigned -> 0x3fffffff - it, assigned -> 0x40000000 + it
            double cv = INFINITY;
            int ckey = 0x7fffffff, cj = -1;
            for (int it = tid; it < num_remaining; it += LSAP_NT) {
                const int j = remaining[it];
                const double r = minVal + C(i, j) - ui - v[j];
                double sv = spc[j];
                if (r < sv) { path[j] = i; spc[j] = r; sv = r; }
                const int key = row4col[j] == -1 ? 0x3fffffff - it : 0x40000000 + it;
                const bool t = sv < cv || (sv == cv && key < ckey);
                cv = t ? sv : cv; ckey = t ? key : ckey; cj = t ? j : cj;
            }
            const int my_key = ckey;
            lsap_wave_min(cv, ckey);
            if (my_key == ckey && ckey != 0x7fffffff) red_col[parity][wave] = cj;      // keys are unique: exactly one lane of the wave
            if (lane == 0) { red_val[parity][wave] = cv; red_key[parity][wave] = ckey; }
            __syncthreads();
            int win = 0;
            cv = red_val[parity][0]; ckey = red_key[parity][0];
#pragma unroll
            for (int w = 1; w < LSAP_NW; ++w) {
                const double ov = red_val[parity][w];
                const int ok = red_key[parity][w];
                const bool t = ov < cv || (ov == cv && ok < ckey);
                cv = t ? ov : cv; ckey = t ? ok : ckey; win = t ? w : win;
            }
            const int j = red_col[parity][win];
            parity ^= 1;
            minVal = cv;
            const int index = ckey < 0x40000000 ? 0x3fffffff - ckey : ckey - 0x40000000;
            const int owner = row4col[j];
            if (owner == -1) sink = j; else i = owner;
            // swap-with-last removal, done by the thread that owns position `index` (the only future reader of it)
            if (tid == (index & (LSAP_NT - 1))) {
                SC[j] = 1;
                remaining[index] = remaining[num_remaining - 1];
            }
            --num_remaining;
        }
        __syncthreads();
        // dual update
        for (int r = tid; r < nr; r += LSAP_NT) {
            if (r == cur) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc[col4row[r]];
        }
        for (int j = tid; j < nc; j += LSAP_NT)
            if (SC[j]) v[j] -= minVal - spc[j];
        __syncthreads();
        if (tid == 0) {
            int j = sink;
            while (true) {
                int r = path[j];
                row4col[j] = r;
                int t = col4row[r]; col4row[r] = j; j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    if (transpose) {
        if (tid == 0) {
            int k = 0;
            for (int r = 0; r < nc; ++r) {          // nc == original row count
                int who = row4col[r];
                if (who >= 0) { row_out[k] = r; col_out[k] = who; ++k; }
            }
        }
    } else {
        for (int r = tid; r < nr; r += LSAP_NT) { row_out[r] = r; col_out[r] = col4row[r]; }
    }
    if (tid == 0 && n_out) *n_out = nr;
}


// ---- register-resident workgroup form for 64 < columns <= 256 (the crowd configuration: 200 tracks x 150 detections).
// Same algorithm, arithmetic and tie-break key as lsap_wg_solve, but position `it` of `remaining` IS thread `it`: the
// column it holds, its shortest-path cost, column dual, owner row, that row's dual and the path predecessor stay in
// registers, so the scan of a Dijkstra step is ONE LDS read (the cost entry) instead of five dependent ones.  Per step:
// scan -> DPP wave minimum -> one 16-byte candidate + the owner's row dual per wave through LDS -> one barrier -> every
// thread picks the winner among four.  The swap-with-last removal hands the state of the last position to the winner's
// position through a double-buffered LDS mailbox written BEFORE the barrier (who is last does not depend on the winner).
// Selected columns leave their registers, so their final path / shortest-path cost (= minVal at selection) go to LDS at
// that moment for the dual update and the augmentation.  ~2.7x fewer cycles per step than the LDS-state form.
struct __attribute__((aligned(16))) LsapCand { double v; int key; unsigned colown; };      // column | (owner row + 1) << 16
struct __attribute__((aligned(16))) LsapMail { double spc, vj, uo; int j, own, pth, pad; };
constexpr int LSAP_REG_COLS = 256;

template <bool COST_LDS>
__device__ __forceinline__ void lsap_reg_solve(const float *cost, int nr0, int nc0, int *row_out, int *col_out, int *n_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;      // nr <= nc <= 256
    extern __shared__ __attribute__((aligned(16))) char lsap_smem[];
    __shared__ LsapCand cand[2][LSAP_NW];
    __shared__ double cand_u[2][LSAP_NW];
    __shared__ LsapMail mail[2];
    constexpr int N = LSAP_REG_COLS;
    double *u = reinterpret_cast<double *>(lsap_smem), *v = u + N, *spc_sel = v + N;
    int *path = reinterpret_cast<int *>(spc_sel + N), *col4row = path + N, *row4col = col4row + N, *SR = row4col + N, *SC = SR + N;
    float *cost_lds = reinterpret_cast<float *>(SC + N);
    if (COST_LDS)
        for (int i = tid; i < nr0 * nc0; i += LSAP_NT) cost_lds[i] = cost[i];
    auto C = [&](int i, int j) -> double {
        const int at = transpose ? j * nc0 + i : i * nc0 + j;
        return (double)(COST_LDS ? cost_lds[at] : cost[at]);
    };
    // SR / SC hold the number (cur + 1) of the row iteration that set them: nothing to clear between iterations
    if (tid < nr) { u[tid] = 0.0; col4row[tid] = -1; SR[tid] = 0; }
    if (tid < nc) { v[tid] = 0.0; row4col[tid] = -1; path[tid] = -1; SC[tid] = 0; }
    __syncthreads();
    int parity = 0;
    const int it = tid;
    for (int cur = 0; cur < nr; ++cur) {
        const int stamp = cur + 1;
        // position it holds column nc - it - 1 (scipy fills `remaining` in reverse order); everything read here was final
        // before the barrier that closed the previous iteration, and the scan below only reads the cost matrix
        int j = nc - it - 1, own = -1, pth = -1;
        double spc = INFINITY, vj = 0.0, uo = 0.0;
        if (it < nc) {
            vj = v[j];
            own = row4col[j];
            uo = own >= 0 ? u[own] : 0.0;
        }
        const int c4r = tid < nr ? col4row[tid] : -1;                 // this row's column BEFORE the augmentation (dual update)
        double minVal = 0.0, ui = u[cur];
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            if (tid == 0) SR[i] = stamp;
            double cv = INFINITY;
            int ckey = 0x7fffffff;
            if (it < num_remaining) {
                const double r = minVal + C(i, j) - ui - vj;
                if (r < spc) { pth = i; spc = r; }
                cv = spc;
                ckey = own == -1 ? 0x3fffffff - it : 0x40000000 + it;
            }
            const int my_key = ckey;
            lsap_wave_min(cv, ckey);
            if (ckey == 0x7fffffff) {                                  // no live position in this wave
                if (lane == 0) { cand[parity][wave].v = INFINITY; cand[parity][wave].key = 0x7fffffff; }
            } else if (my_key == ckey) {                               // keys are unique: exactly one lane of the wave
                LsapCand c;
                c.v = cv; c.key = ckey; c.colown = (unsigned)j | ((unsigned)(own + 1) << 16);
                cand[parity][wave] = c;
                cand_u[parity][wave] = uo;
            }
            if (it == num_remaining - 1) {                             // the state the winner's position inherits
                LsapMail m;
                m.spc = spc; m.vj = vj; m.uo = uo; m.j = j; m.own = own; m.pth = pth; m.pad = 0;
                mail[parity] = m;
            }
            __syncthreads();
            int win = 0;
            LsapCand best = cand[parity][0];
#pragma unroll
            for (int w = 1; w < LSAP_NW; ++w) {
                const LsapCand o = cand[parity][w];
                const bool t = o.v < best.v || (o.v == best.v && o.key < best.key);
                best.v = t ? o.v : best.v; best.key = t ? o.key : best.key; best.colown = t ? o.colown : best.colown; win = t ? w : win;
            }
            const int jw = (int)(best.colown & 0xffffu), owner = (int)(best.colown >> 16) - 1;
            minVal = best.v;
            const int index = best.key < 0x40000000 ? 0x3fffffff - best.key : best.key - 0x40000000;
            if (owner == -1) sink = jw;
            else { i = owner; ui = cand_u[parity][win]; }
            if (it == index) {
                // this thread holds the selected column: its path / cost are final (spc == minVal), then swap-with-last
                path[jw] = pth; spc_sel[jw] = spc; SC[jw] = stamp;
                if (index != num_remaining - 1) {
                    const LsapMail m = mail[parity];
                    spc = m.spc; vj = m.vj; uo = m.uo; j = m.j; own = m.own; pth = m.pth;
                }
            }
            --num_remaining;
            parity ^= 1;
        }
        __syncthreads();
        // dual update (selected columns: spc_sel; the sink's entry equals minVal) - and, concurrently on thread 0, the
        // augmentation: the dual update reads the pre-augmentation columns from registers (c4r), so the two do not interfere
        if (tid < nr) {
            if (tid == cur) u[tid] += minVal;
            else if (SR[tid] == stamp) u[tid] += minVal - spc_sel[c4r];
        }
        if (tid < nc && SC[tid] == stamp) v[tid] -= minVal - spc_sel[tid];
        if (tid == 0) {
            int jj = sink;
            while (true) {
                const int r = path[jj];
                row4col[jj] = r;
                const int t = col4row[r]; col4row[r] = jj; jj = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    if (transpose) {
        if (tid == 0) {
            int k = 0;
            for (int r = 0; r < nc; ++r) {          // nc == original row count
                const int who = row4col[r];
                if (who >= 0) { row_out[k] = r; col_out[k] = who; ++k; }
            }
        }
    } else if (tid < nr) { row_out[tid] = tid; col_out[tid] = col4row[tid]; }
    if (tid == 0 && n_out) *n_out = nr;
}
constexpr size_t LSAP_REG_STATE = (size_t)LSAP_REG_COLS * (3 * sizeof(double) + 5 * sizeof(int));

// The launch sizes its LDS from host-side upper bounds (inside a batch the live-track count is only bounded by T + sum D);
// whether the cost matrix is copied into LDS is decided HERE from the actual sizes - a 200 x 150 problem launched under a
// bound of 2600 x 150 must not fall back to reading its costs from global memory in the Dijkstra step.
template <bool GSTATE>
__global__ __launch_bounds__(LSAP_NT) void lsap_kernel(const float *cost, int nr0, int nc0, const int *dims_p, int *row_out, int *col_out,
                                                       int *n_out, char *state_global, int smem_bytes) {
    if (dims_p) { nr0 = dims_p[0]; nc0 = dims_p[1]; }
    if (nr0 <= 0 || nc0 <= 0) {                                  // linear_assignment.py:48-49 early-out
        if (threadIdx.x == 0 && n_out) *n_out = 0;
        return;
    }
    if (max(nr0, nc0) <= LSAP_WAVE_COLS) return;                 // solved by lsap_wave_kernel (launched in front of this one)
    if (YDS_LSAP_REG && max(nr0, nc0) <= LSAP_REG_COLS && LSAP_REG_STATE <= (size_t)smem_bytes) {       // register-resident form
        if (LSAP_REG_STATE + (size_t)nr0 * nc0 * sizeof(float) <= (size_t)smem_bytes) lsap_reg_solve<true>(cost, nr0, nc0, row_out, col_out, n_out);
        else lsap_reg_solve<false>(cost, nr0, nc0, row_out, col_out, n_out);
        return;
    }
    const size_t state = GSTATE ? 0 : (size_t)max(nr0, nc0) * LSAP_STATE_BYTES;
    if (state + (size_t)nr0 * nc0 * sizeof(float) <= (size_t)smem_bytes) lsap_wg_solve<GSTATE, true>(cost, nr0, nc0, row_out, col_out, n_out, state_global);
    else lsap_wg_solve<GSTATE, false>(cost, nr0, nc0, row_out, col_out, n_out, state_global);
}

// ---- single-wavefront form for problems with at most 256 columns (after the tall->wide transposition): every lane keeps
// the scan state of up to four `remaining` positions (column, shortest-path cost, column dual, owner row) in REGISTERS, so a
// Dijkstra step is one LDS cost read per position, the DPP reduction and a register hand-over for the swap-with-last
// removal - no barrier and no LDS round trip on the critical path (the workgroup form above spends ~1 us per step).
// Same arithmetic, same tie-break key, same result.
constexpr int LSAP_WAVE_SLOTS = LSAP_WAVE_COLS / 64;
constexpr size_t LSAP_WAVE_COST_MAX = 128 * 1024;               // cost matrix copied to LDS when it fits

#ifndef YDS_LSAP_PROF
#define YDS_LSAP_PROF 0
#endif
__device__ unsigned long long yds_lsap_prof[8];
// SLOTS positions per lane (1: up to 64 columns, 4: up to 256).  A single wavefront issues one instruction every few cycles,
// so the step is written for instruction count: only (cost, key) travel through the reduction - the key names the position,
// whose lane then hands out column and owner - and inactive positions are masked with selects instead of branches.
template <bool COST_LDS, int SLOTS>
__device__ __forceinline__ void lsap_wave_solve(const float *cost, const float *cost_lds, int nr0, int nc0, char *smem, int *row_out, int *col_out) {
    const int lane = threadIdx.x;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    double *u = reinterpret_cast<double *>(smem), *v = u + LSAP_WAVE_COLS, *spc_rm = v + LSAP_WAVE_COLS;
    int *path = reinterpret_cast<int *>(spc_rm + LSAP_WAVE_COLS), *col4row = path + LSAP_WAVE_COLS, *row4col = col4row + LSAP_WAVE_COLS,
        *SR = row4col + LSAP_WAVE_COLS, *SC = SR + LSAP_WAVE_COLS;
    const int sj = transpose ? nc0 : 1, si = transpose ? 1 : nc0;       // cost(i, j) at i * si + j * sj
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    __syncthreads();
    for (int cur = 0; cur < nr; ++cur) {
        int jj[SLOTS], ow[SLOTS], cofs[SLOTS];
        double sp[SLOTS], vv[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int it = lane + 64 * s, j = nc - 1 - it;          // `remaining` starts as nc-1 .. 0
            const bool in = it < nc;
            jj[s] = in ? j : 0; cofs[s] = jj[s] * sj; sp[s] = INFINITY;
            vv[s] = v[jj[s]]; ow[s] = row4col[jj[s]];
        }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        for (int j = lane; j < nc; j += 64) SC[j] = 0;
        __syncthreads();
        double minVal = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        unsigned long long t_loop = YDS_LSAP_PROF ? __builtin_amdgcn_s_memtime() : 0, n_it = 0;
        while (sink == -1) {
            if (YDS_LSAP_PROF) ++n_it;
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            const int row_off = i * si;
            double cv = INFINITY;
            int ckey = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int it = lane + 64 * s;
                const bool active = it < num_remaining;
                const int at = row_off + cofs[s];
                const double cij = (double)(COST_LDS ? cost_lds[at] : cost[at]);
                const double r = minVal + cij - ui - vv[s];            // scipy's order: minVal + cost - u[i] - v[j]
                if (active && r < sp[s]) { path[jj[s]] = i; sp[s] = r; }
                const int key = ow[s] == -1 ? 0x3fffffff - it : 0x40000000 + it;
                LSAP_TAKE_MIN(cv, ckey, active ? sp[s] : (double)INFINITY, active ? key : 0x7fffffff);
            }
            lsap_wave_min(cv, ckey);
            minVal = cv;
            const int index = ckey < 0x40000000 ? 0x3fffffff - ckey : ckey - 0x40000000;
            const int is = SLOTS == 1 ? 0 : index >> 6, il = index & 63;
            // the position's lane hands out its column and owner
            int j = 0, own = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (s == is) { j = __builtin_amdgcn_readlane(jj[s], il); own = __builtin_amdgcn_readlane(ow[s], il); }
            if (own == -1) sink = j; else i = own;
            // swap-with-last removal: position `index` takes over the registers of position num_remaining - 1
            const int last = __builtin_amdgcn_readfirstlane(num_remaining - 1);
            const int ls = SLOTS == 1 ? 0 : last >> 6, ll = last & 63;
            int t_j = 0, t_o = 0, t_c = 0;
            union { double d; int w[2]; } t_sp, t_vv, a;
            t_sp.d = 0.0; t_vv.d = 0.0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (s == ls) {                                       // uniform
                    t_j = __builtin_amdgcn_readlane(jj[s], ll);
                    t_o = __builtin_amdgcn_readlane(ow[s], ll);
                    t_c = __builtin_amdgcn_readlane(cofs[s], ll);
                    a.d = sp[s]; t_sp.w[0] = __builtin_amdgcn_readlane(a.w[0], ll); t_sp.w[1] = __builtin_amdgcn_readlane(a.w[1], ll);
                    a.d = vv[s]; t_vv.w[0] = __builtin_amdgcn_readlane(a.w[0], ll); t_vv.w[1] = __builtin_amdgcn_readlane(a.w[1], ll);
                }
            if (lane == il) { SC[j] = 1; spc_rm[j] = minVal; }       // the removed column keeps its shortest-path cost for the dual update
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const bool here = s == is && lane == il;
                jj[s] = here ? t_j : jj[s]; ow[s] = here ? t_o : ow[s]; cofs[s] = here ? t_c : cofs[s];
                sp[s] = here ? t_sp.d : sp[s]; vv[s] = here ? t_vv.d : vv[s];
            }
            --num_remaining;
        }
        if (YDS_LSAP_PROF && lane == 0) { yds_lsap_prof[0] += __builtin_amdgcn_s_memtime() - t_loop; yds_lsap_prof[1] += n_it; }
        __syncthreads();
        for (int r = lane; r < nr; r += 64) {
            if (r == cur) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc_rm[col4row[r]];
        }
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc_rm[j];
        __syncthreads();
        if (lane == 0) {
            int j = sink;
            while (true) {
                int r = path[j];
                row4col[j] = r;
                int t = col4row[r]; col4row[r] = j; j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    if (transpose) {
        if (lane == 0) {
            int k = 0;
            for (int r = 0; r < nc; ++r) {          // nc == original row count
                int who = row4col[r];
                if (who >= 0) { row_out[k] = r; col_out[k] = who; ++k; }
            }
        }
    } else {
        for (int r = lane; r < nr; r += 64) { row_out[r] = r; col_out[r] = col4row[r]; }
    }
}

constexpr size_t LSAP_WAVE_STATE = (size_t)LSAP_WAVE_COLS * (3 * sizeof(double) + 5 * sizeof(int));
// big = 0: this launch solves problems with <= 256 columns and leaves larger ones to the workgroup kernel (which is launched
// with skip_small = 1 right behind it): the sizes are only known on the device, the host picks nothing.
__global__ __launch_bounds__(64) void lsap_wave_kernel(const float *cost, int nr0, int nc0, const int *dims_p, int *row_out, int *col_out, int *n_out,
                                                      int cost_lds_floats) {
    if (dims_p) { nr0 = dims_p[0]; nc0 = dims_p[1]; }
    extern __shared__ __attribute__((aligned(16))) char lsap_smem[];
    if (nr0 <= 0 || nc0 <= 0) {
        if (threadIdx.x == 0 && n_out) *n_out = 0;
        return;
    }
    if (max(nr0, nc0) > LSAP_WAVE_COLS) return;                 // the workgroup kernel's case
    const unsigned long long t_k = YDS_LSAP_PROF ? __builtin_amdgcn_s_memtime() : 0, w_k = YDS_LSAP_PROF ? wall_clock64() : 0;
    float *cost_lds = reinterpret_cast<float *>(lsap_smem + LSAP_WAVE_STATE);
    const bool narrow = max(nr0, nc0) <= 64;                     // one position per lane
    if (nr0 * nc0 <= cost_lds_floats) {
        for (int i = threadIdx.x; i < nr0 * nc0; i += 64) cost_lds[i] = cost[i];
        if (narrow) lsap_wave_solve<true, 1>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
        else lsap_wave_solve<true, LSAP_WAVE_SLOTS>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
    } else {
        if (narrow) lsap_wave_solve<false, 1>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
        else lsap_wave_solve<false, LSAP_WAVE_SLOTS>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
    }
    if (threadIdx.x == 0 && n_out) *n_out = min(nr0, nc0);
    if (YDS_LSAP_PROF && threadIdx.x == 0) {
        yds_lsap_prof[2] += __builtin_amdgcn_s_memtime() - t_k; yds_lsap_prof[3] += wall_clock64() - w_k; yds_lsap_prof[4] += 1;
    }
}

// nr_max / nc_max: upper bounds known on the host (they size the LDS / scratch); the real sizes may come from dims_dev
static void launch_lsap(const float *cost_dev, int nr_max, int nc_max, const int *dims_dev, int *rows_dev, int *cols_dev, int *n_out_dev,
                        DevBuf<char> &scratch, hipStream_t s) {
    {
        static bool wave_attr = false;
        if (!wave_attr) {
            YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(LSAP_WAVE_STATE + LSAP_WAVE_COST_MAX)));
            wave_attr = true;
        }
        const size_t want = (size_t)std::min(nr_max, LSAP_WAVE_COLS) * std::min(nc_max, LSAP_WAVE_COLS) * sizeof(float);
        const size_t cost_lds = std::min(want, LSAP_WAVE_COST_MAX);
        hipLaunchKernelGGL(lsap_wave_kernel, dim3(1), dim3(64), LSAP_WAVE_STATE + cost_lds, s, cost_dev, nr_max, nc_max, dims_dev, rows_dev, cols_dev,
                           n_out_dev, (int)(cost_lds / sizeof(float)));
        YDS_HIP(hipGetLastError());
        if (std::max(nr_max, nc_max) <= LSAP_WAVE_COLS) return;
    }
    const size_t n = (size_t)std::max(std::max(nr_max, nc_max), 1);
    const size_t state = n * LSAP_STATE_BYTES, cost_bytes = (size_t)nr_max * nc_max * sizeof(float);
    const bool state_lds = state <= LSAP_LDS_MAX;
    // LDS: the state of the largest possible problem, plus the cost matrix if the bounds allow it; when they do not, the whole
    // LDS is requested anyway and the kernel decides from the actual sizes
    // (the register-resident form for <= 256 columns keeps a fixed 11 KB of state: make room for it and its cost copy too)
    const size_t reg_want = LSAP_REG_STATE + (size_t)std::min(nr_max, LSAP_REG_COLS) * std::min(nc_max, LSAP_REG_COLS) * sizeof(float);
    const size_t smem = std::max(state_lds ? std::min(state + cost_bytes, LSAP_LDS_MAX) : std::min(cost_bytes, LSAP_LDS_MAX),
                                 std::min(reg_want, LSAP_LDS_MAX));
    if (!state_lds && scratch.n < state) {
        YDS_HIP(hipStreamSynchronize(s));                        // nothing may still use the old scratch
        scratch.alloc(state);
    }
    static bool attr_set = false;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSAP_LDS_MAX));
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSAP_LDS_MAX));
        attr_set = true;
    }
    auto kern = state_lds ? lsap_kernel<false> : lsap_kernel<true>;
    hipLaunchKernelGGL(kern, dim3(1), dim3(LSAP_NT), smem, s, cost_dev, nr_max, nc_max, dims_dev, rows_dev, cols_dev, n_out_dev,
                       state_lds ? (char *)nullptr : scratch.p, (int)smem);
    YDS_HIP(hipGetLastError());
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

// ============================================================================================ device-resident tracker
enum { TENTATIVE = 1, CONFIRMED = 2, DELETED = 3 };
enum { METRIC_COSINE = 0, METRIC_EUCLIDEAN = 1 };

// Integer lifecycle table, one entry per track in track-list order (deep_sort/sort/track.py:63-79), in device memory.
struct TrackTable {
    int *slot, *id, *hits, *age, *tsu, *state, *n_feat, *head;
    float *payload;
};
constexpr int TAB_FIELDS = 9;

// device-side counters of one frame (ints)
enum Meta {
    M_T = 0,        // live tracks
    M_NEXT_ID,      // Tracker._next_id (tracker.py:47)
    M_NFREE,        // free gallery / Kalman slots
    M_TC, M_D,      // confirmed tracks, detections      (stage A problem: dims at &meta[M_TC])
    M_TB, M_DB,     // IOU-stage candidates, leftover detections (stage B problem: dims at &meta[M_TB])
    M_NA, M_NB,     // assigned pairs returned by the two LSAP solves
    M_NM_A,         // matches of stage A
    M_NUT_A,        // unmatched confirmed tracks of stage A
    M_NKEEP,        // of those, not eligible for the IOU stage
    M_NM,           // matches (both stages)
    M_NUT,          // unmatched tracks (final)
    M_NUD,          // unmatched detections (final) = new tracks
    M_NOUT,         // output rows
    M_MAXFEAT,      // largest gallery row count of any live track (sizes the nn_budget=None galleries)
    M_COUNT = 20
};

struct TrkDev {
    TrackTable tab, tmp;
    int *meta, *free_slots;
    float *mean, *cov, *gallery, *feats_n, *cost;
    const float *tlwh, *payload;                    // this frame's detections
    int *conf_idx, *unconf_idx, *rows, *cols, *flag_r, *flag_c, *rej;
    int *matches, *um_t_a, *um_t_keep, *um_t, *um_d, *um_d2, *iou_cand;
    int *upd_slot, *upd_det, *upd_pos, *new_slot, *out_slot, *out_id;
    float *out_payload;
    int budget, unbounded, n_init, max_age;
    float max_dist, max_iou;
    // per-frame result block (device copy of what goes back to the host): header, rows, debug lists
    int *res;
    int res_out6, res_matches, res_um_t, res_um_d;   // int offsets inside res
};

// ordered compaction over [0, n) by a 256-thread workgroup: emit(i, rank) for every i with pred(i), ranks ascending
// with i; returns the number of hits (to every thread).  s_cnt: LDS int[4].
template <class Pred, class Emit> __device__ __forceinline__ int compact_ordered(int n, int *s_cnt, Pred pred, Emit emit) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int total = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const bool p = i < n && pred(i);
        const unsigned long long b = __ballot(p);
        const int lane_rank = __popcll(b & ((1ull << lane) - 1ull)), wave_cnt = __popcll(b);
        if (lane == 0) s_cnt[wave] = wave_cnt;
        __syncthreads();
        int woff = 0, chunk = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = s_cnt[w]; if (w < wave) woff += c; chunk += c; }
        if (p) emit(i, total + woff + lane_rank);
        __syncthreads();
        total += chunk;
    }
    return total;
}

// Tracker.predict bookkeeping (track.py:110-123: age += 1, time_since_update += 1) and the confirmed / unconfirmed index
// lists of Tracker._match (tracker.py:65-67), in list order.
__global__ __launch_bounds__(256) void trk_begin_kernel(TrkDev d, int D) {
    __shared__ int s_cnt[4];
    const int T = d.meta[M_T];
    for (int t = threadIdx.x; t < T; t += 256) { d.tab.age[t]++; d.tab.tsu[t]++; }
    __syncthreads();
    const int Tc = compact_ordered(T, s_cnt, [&](int t) { return d.tab.state[t] == CONFIRMED; }, [&](int t, int r) { d.conf_idx[r] = t; });
    const int Tu = compact_ordered(T, s_cnt, [&](int t) { return d.tab.state[t] != CONFIRMED; }, [&](int t, int r) { d.unconf_idx[r] = t; });
    if (threadIdx.x == 0) { d.meta[M_TC] = Tc; d.meta[M_D] = D; d.meta[M_TB] = Tu; }      // M_TB holds the unconfirmed count until stage A adds to it
}

__global__ void trk_predict_kernel(TrkDev d) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.meta[M_T]) return;
    const int slot = d.tab.slot[t];
    kf_predict_body(d.mean + (size_t)slot * 8, d.cov + (size_t)slot * 64);
}

// min_cost_matching list bookkeeping (linear_assignment.py:58-72) for one solved assignment problem:
//   unmatched detections = [columns not assigned, ascending] then [rejected pairs in row order], unmatched tracks likewise,
//   matches in row order; a pair is rejected when cost[row, col] > max_distance.
// row_name(r) / col_name(c) translate problem rows / columns to track indices / detection indices.
template <class RowName, class ColName>
__device__ __forceinline__ void assign_lists(const TrkDev &d, int *s_cnt, int nr, int nc, int n_pairs, float max_distance, RowName row_name,
                                             ColName col_name, int *matches, int &n_matches, int *um_t, int &n_um_t, int *um_d, int &n_um_d) {
    const int tid = threadIdx.x;
    for (int r = tid; r < nr; r += 256) d.flag_r[r] = 0;
    for (int c = tid; c < nc; c += 256) d.flag_c[c] = 0;
    __syncthreads();
    for (int k = tid; k < n_pairs; k += 256) {
        const int r = d.rows[k], c = d.cols[k];
        d.flag_r[r] = 1;
        d.flag_c[c] = 1;
        d.rej[k] = d.cost[(size_t)r * nc + c] > max_distance ? 1 : 0;
    }
    __syncthreads();
    const int d1 = compact_ordered(nc, s_cnt, [&](int c) { return d.flag_c[c] == 0; }, [&](int c, int q) { um_d[q] = col_name(c); });
    const int d2 = compact_ordered(n_pairs, s_cnt, [&](int k) { return d.rej[k] != 0; }, [&](int k, int q) { um_d[d1 + q] = col_name(d.cols[k]); });
    const int t1 = compact_ordered(nr, s_cnt, [&](int r) { return d.flag_r[r] == 0; }, [&](int r, int q) { um_t[q] = row_name(r); });
    const int t2 = compact_ordered(n_pairs, s_cnt, [&](int k) { return d.rej[k] != 0; }, [&](int k, int q) { um_t[t1 + q] = row_name(d.rows[k]); });
    const int nm = compact_ordered(n_pairs, s_cnt, [&](int k) { return d.rej[k] == 0; },
                                   [&](int k, int q) { matches[2 * (n_matches + q)] = row_name(d.rows[k]); matches[2 * (n_matches + q) + 1] = col_name(d.cols[k]); });
    n_um_d = d1 + d2;
    n_um_t = t1 + t2;
    n_matches += nm;
}

// after the appearance-stage assignment: its lists, then the IOU-stage candidates (tracker.py:80-85):
// unconfirmed tracks (index order) + unmatched confirmed tracks with time_since_update == 1
__global__ __launch_bounds__(256) void trk_match_a_kernel(TrkDev d) {
    __shared__ int s_cnt[4];
    const int Tc = d.meta[M_TC], D = d.meta[M_D], Tu = d.meta[M_TB];
    const int n_pairs = (Tc > 0 && D > 0) ? d.meta[M_NA] : 0;              // either side empty: nothing was solved (linear_assignment.py:48-49)
    int n_matches = 0, n_um_t = 0, n_um_d = 0;
    assign_lists(d, s_cnt, Tc, D, n_pairs, d.max_dist, [&](int r) { return d.conf_idx[r]; }, [&](int c) { return c; }, d.matches, n_matches,
                 d.um_t_a, n_um_t, d.um_d, n_um_d);
    __syncthreads();
    for (int q = threadIdx.x; q < Tu; q += 256) d.iou_cand[q] = d.unconf_idx[q];
    const int c1 = compact_ordered(n_um_t, s_cnt, [&](int q) { return d.tab.tsu[d.um_t_a[q]] == 1; }, [&](int q, int r) { d.iou_cand[Tu + r] = d.um_t_a[q]; });
    const int k1 = compact_ordered(n_um_t, s_cnt, [&](int q) { return d.tab.tsu[d.um_t_a[q]] != 1; }, [&](int q, int r) { d.um_t_keep[r] = d.um_t_a[q]; });
    if (threadIdx.x == 0) {
        d.meta[M_NM_A] = n_matches; d.meta[M_NUT_A] = n_um_t; d.meta[M_NKEEP] = k1;
        d.meta[M_TB] = Tu + c1; d.meta[M_DB] = n_um_d;
    }
}

// after the IOU-stage assignment: final lists, Tracker.update (tracker.py:129-176: Track.update / mark_missed /
// _initiate_track, deleted tracks dropped) on the integer table, the lists the Kalman / gallery kernels consume,
// and the output selection of DeepSort.update (deep_sort.py:67-71).
__global__ __launch_bounds__(256) void trk_match_b_kernel(TrkDev d) {
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x;
    const int Tb = d.meta[M_TB], Db = d.meta[M_DB], k_keep = d.meta[M_NKEEP];
    const int n_pairs = (Tb > 0 && Db > 0) ? d.meta[M_NB] : 0;
    int n_matches = d.meta[M_NM_A], n_um_t_b = 0, n_um_d = 0;
    int *um_t_b = d.um_t + k_keep;                                  // final unmatched tracks = [kept from stage A] + [stage B]
    assign_lists(d, s_cnt, Tb, Db, n_pairs, d.max_iou, [&](int r) { return d.iou_cand[r]; }, [&](int c) { return d.um_d[c]; }, d.matches, n_matches,
                 um_t_b, n_um_t_b, d.um_d2, n_um_d);
    for (int q = tid; q < k_keep; q += 256) d.um_t[q] = d.um_t_keep[q];
    __syncthreads();
    const int M = n_matches, n_um_t = k_keep + n_um_t_b, Nn = n_um_d;
    int T = d.meta[M_T];
    // ---- Track.update (track.py:125-144) for every match
    for (int k = tid; k < M; k += 256) {
        const int t = d.matches[2 * k], det = d.matches[2 * k + 1];
        int pos;
        if (d.unbounded) { pos = d.tab.n_feat[t]; d.tab.n_feat[t] = pos + 1; }
        else {
            pos = d.tab.head[t];
            d.tab.head[t] = (pos + 1) % d.budget;
            d.tab.n_feat[t] = min(d.tab.n_feat[t] + 1, d.budget);
        }
        const int hits = ++d.tab.hits[t];
        d.tab.tsu[t] = 0;
        if (d.tab.state[t] == TENTATIVE && hits >= d.n_init) d.tab.state[t] = CONFIRMED;
        d.tab.payload[t] = d.payload[det];
        d.upd_slot[k] = d.tab.slot[t]; d.upd_det[k] = det; d.upd_pos[k] = pos;
    }
    // ---- Track.mark_missed (track.py:146-152)
    for (int q = tid; q < n_um_t; q += 256) {
        const int t = d.um_t[q];
        if (d.tab.state[t] == TENTATIVE) d.tab.state[t] = DELETED;
        else if (d.tab.tsu[t] > d.max_age) d.tab.state[t] = DELETED;
    }
    // ---- Tracker._initiate_track (tracker.py:49-54) for the unmatched detections, in list order
    const int n_free = d.meta[M_NFREE], next_id = d.meta[M_NEXT_ID];
    for (int k = tid; k < Nn; k += 256) {
        const int t = T + k, slot = d.free_slots[n_free - 1 - k], det = d.um_d2[k];
        d.tab.slot[t] = slot; d.tab.id[t] = next_id + k; d.tab.hits[t] = 1; d.tab.age[t] = 1; d.tab.tsu[t] = 0; d.tab.state[t] = TENTATIVE;
        d.tab.n_feat[t] = 1; d.tab.head[t] = 1 % d.budget;
        d.tab.payload[t] = d.payload[det];
        d.new_slot[k] = slot;
        // the feature / Kalman kernels take one combined list: entries [M, M + Nn) are the new tracks
        d.upd_slot[M + k] = slot; d.upd_det[M + k] = det; d.upd_pos[M + k] = 0;
    }
    __syncthreads();
    T += Nn;
    // ---- drop deleted tracks (tracker.py:162), slots go back to the free list
    const int alive = compact_ordered(T, s_cnt, [&](int t) { return d.tab.state[t] != DELETED; }, [&](int t, int r) {
        d.tmp.slot[r] = d.tab.slot[t]; d.tmp.id[r] = d.tab.id[t]; d.tmp.hits[r] = d.tab.hits[t]; d.tmp.age[r] = d.tab.age[t]; d.tmp.tsu[r] = d.tab.tsu[t];
        d.tmp.state[r] = d.tab.state[t]; d.tmp.n_feat[r] = d.tab.n_feat[t]; d.tmp.head[r] = d.tab.head[t]; d.tmp.payload[r] = d.tab.payload[t];
    });
    const int freed = compact_ordered(T, s_cnt, [&](int t) { return d.tab.state[t] == DELETED; },
                                      [&](int t, int r) { d.free_slots[n_free - Nn + r] = d.tab.slot[t]; });
    __syncthreads();
    for (int t = tid; t < alive; t += 256) {
        d.tab.slot[t] = d.tmp.slot[t]; d.tab.id[t] = d.tmp.id[t]; d.tab.hits[t] = d.tmp.hits[t]; d.tab.age[t] = d.tmp.age[t]; d.tab.tsu[t] = d.tmp.tsu[t];
        d.tab.state[t] = d.tmp.state[t]; d.tab.n_feat[t] = d.tmp.n_feat[t]; d.tab.head[t] = d.tmp.head[t]; d.tab.payload[t] = d.tmp.payload[t];
    }
    __syncthreads();
    // ---- output selection (deep_sort.py:67-71): confirmed and time_since_update <= 1, in list order
    const int n_out = compact_ordered(alive, s_cnt, [&](int t) { return d.tab.state[t] == CONFIRMED && d.tab.tsu[t] <= 1; },
                                      [&](int t, int r) { d.out_slot[r] = d.tab.slot[t]; d.out_id[r] = d.tab.id[t]; d.out_payload[r] = d.tab.payload[t]; });
    // ---- largest gallery of a live track: the host grows unbounded galleries by what is really held, not by frames seen
    __shared__ int s_maxfeat;
    if (tid == 0) s_maxfeat = 0;
    __syncthreads();
    {
        int mf = 0;
        for (int t = tid; t < alive; t += 256) mf = max(mf, d.tab.n_feat[t]);
        if (mf) atomicMax(&s_maxfeat, mf);
    }
    __syncthreads();
    // ---- debug lists for the parity tests
    for (int k = tid; k < 2 * M; k += 256) d.res[d.res_matches + k] = d.matches[k];
    for (int q = tid; q < n_um_t; q += 256) d.res[d.res_um_t + q] = d.um_t[q];
    for (int q = tid; q < Nn; q += 256) d.res[d.res_um_d + q] = d.um_d2[q];
    if (tid == 0) {
        d.meta[M_T] = alive; d.meta[M_NEXT_ID] = next_id + Nn; d.meta[M_NFREE] = n_free - Nn + freed;
        d.meta[M_NM] = M; d.meta[M_NUT] = n_um_t; d.meta[M_NUD] = Nn; d.meta[M_NOUT] = n_out;
        d.meta[M_MAXFEAT] = s_maxfeat;
        for (int k = 0; k < M_COUNT; ++k) d.res[k] = d.meta[k];
    }
}

// KalmanFilter.update for the matches (kalman_filter.py:161-204 via tracker.py:143-150), initiate for the new tracks
// (:54-87), and the gallery rows of both (tracker.py:165-176 + nn_matching.py:152-155)
__global__ void trk_kalman_kernel(TrkDev d) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = d.meta[M_NM], Nn = d.meta[M_NUD];
    if (k >= M + Nn) return;
    const int slot = d.upd_slot[k];
    const float *b = d.tlwh + (size_t)d.upd_det[k] * 4;
    float *m = d.mean + (size_t)slot * 8, *P = d.cov + (size_t)slot * 64;
    if (k < M) {
        float z[4];
        to_xyah(b, z);
        kf_update_body(m, P, z);
    } else {
        kf_initiate_body(m, P, b);
    }
}
__global__ void trk_append_kernel(TrkDev d) {
    const int k = blockIdx.x;
    if (k >= d.meta[M_NM] + d.meta[M_NUD]) return;
    float *dst = d.gallery + ((size_t)d.upd_slot[k] * d.budget + d.upd_pos[k]) * EMB;
    const float *src = d.feats_n + (size_t)d.upd_det[k] * EMB;
    for (int c = threadIdx.x; c < EMB; c += blockDim.x) dst[c] = src[c];
}
// deep_sort.py:73-87 on the selected tracks -> int32 rows in the result block
__global__ void trk_output_kernel(TrkDev d) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.meta[M_NOUT]) return;
    const float *m = d.mean + (size_t)d.out_slot[t] * 8;
    float w = m[2] * m[3], h = m[3];
    float x = m[0] - w / 2.f, y = m[1] - h / 2.f;
    float x2 = w + x, y2 = h + y;
    x = fmaxf(x, 0.f); y = fmaxf(y, 0.f);
    int *o = d.res + d.res_out6 + t * 6;
    o[0] = (int)x; o[1] = (int)y; o[2] = (int)x2; o[3] = (int)y2; o[4] = d.out_id[t]; o[5] = (int)d.out_payload[t];
}

// ============================================================================================ host
class Tracker : public TrackerIface {
public:
    // budget <= 0: nn_budget=None, every track keeps all its features (nn_matching.py:152-154) - the per-track row
    // capacity `budget` then doubles whenever a gallery could fill up; metric: cosine | euclidean (nn_matching.py:128-134)
    Tracker(double max_dist, double max_iou, int max_age, int n_init, int budget, int metric = METRIC_COSINE)
        : max_dist(max_dist), max_iou(max_iou), max_age(max_age), n_init(n_init), budget(budget > 0 ? budget : 32), unbounded(budget <= 0),
          metric(metric) {
        if (metric != METRIC_COSINE && metric != METRIC_EUCLIDEAN) fail("Invalid metric; must be either 'euclidean' or 'cosine'");
        stream = make_stream(true);
        feats_stage.st = feats_n.st = cost_dev.st = &stream;
        det_lists.st = &stream;
        meta.alloc(M_COUNT);
        int init[M_COUNT] = {};
        init[M_NEXT_ID] = 1;
        YDS_HIP(hipMemcpy(meta.p, init, sizeof init, hipMemcpyHostToDevice));
        grow(256);
    }
    ~Tracker() override {
        if (res_host) (void)hipHostFree(res_host);
        if (in_host) (void)hipHostFree(in_host);
        if (stream) (void)hipStreamDestroy(stream);
    }
    int num_tracks() const override { return T_host; }

    static TrackTable table_at(int *base, int cap) {
        TrackTable t;
        t.slot = base; t.id = base + cap; t.hits = base + 2 * cap; t.age = base + 3 * cap; t.tsu = base + 4 * cap; t.state = base + 5 * cap;
        t.n_feat = base + 6 * cap; t.head = base + 7 * cap; t.payload = reinterpret_cast<float *>(base + 8 * cap);
        return t;
    }

    // capacity = number of slots = maximum number of live tracks.  Nothing may be in flight on the stream.
    void grow(int cap) {
        YDS_HIP(hipStreamSynchronize(stream));
        DevBuf<float> m((size_t)cap * 8), c((size_t)cap * 64), g((size_t)cap * budget * EMB);
        DevBuf<int> tab((size_t)cap * TAB_FIELDS), tmp((size_t)cap * TAB_FIELDS), fs(cap), lists((size_t)cap * 12);
        if (capacity) {
            YDS_HIP(hipMemcpy(m.p, mean.p, (size_t)capacity * 8 * 4, hipMemcpyDeviceToDevice));
            YDS_HIP(hipMemcpy(c.p, cov.p, (size_t)capacity * 64 * 4, hipMemcpyDeviceToDevice));
            YDS_HIP(hipMemcpy(g.p, gallery.p, (size_t)capacity * budget * EMB * 4, hipMemcpyDeviceToDevice));
            for (int f = 0; f < TAB_FIELDS; ++f)
                YDS_HIP(hipMemcpy(tab.p + (size_t)f * cap, table.p + (size_t)f * capacity, (size_t)capacity * 4, hipMemcpyDeviceToDevice));
        }
        // free list: existing entries, then the new slots on top (popped first); the count lives in meta[M_NFREE]
        int n_free = 0;
        YDS_HIP(hipMemcpy(&n_free, meta.p + M_NFREE, 4, hipMemcpyDeviceToHost));
        std::vector<int> fl(cap);
        if (capacity && n_free) YDS_HIP(hipMemcpy(fl.data(), free_slots.p, (size_t)n_free * 4, hipMemcpyDeviceToHost));
        for (int s = cap - 1; s >= capacity; --s) fl[n_free++] = s;
        YDS_HIP(hipMemcpy(fs.p, fl.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
        YDS_HIP(hipMemcpy(meta.p + M_NFREE, &n_free, 4, hipMemcpyHostToDevice));
        mean = std::move(m); cov = std::move(c); gallery = std::move(g); table = std::move(tab); table_tmp = std::move(tmp);
        free_slots = std::move(fs); track_lists = std::move(lists);
        capacity = cap;
    }
    // nn_budget=None: double the per-track row capacity, keeping every slot's rows
    void grow_budget() {
        YDS_HIP(hipStreamSynchronize(stream));
        const int nb = budget * 2;
        size_t free_b = 0, total_b = 0;
        YDS_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t need = (size_t)capacity * nb * EMB * 4;
        if (need > free_b)
            fail("tracker: nn_budget=None galleries would need %zu MB for %d slots x %d rows (%zu MB free); use a finite nn_budget",
                 need >> 20, capacity, nb, free_b >> 20);
        DevBuf<float> g((size_t)capacity * nb * EMB);
        YDS_HIP(hipMemcpy2D(g.p, (size_t)nb * EMB * 4, gallery.p, (size_t)budget * EMB * 4, (size_t)budget * EMB * 4, capacity, hipMemcpyDeviceToDevice));
        gallery = std::move(g);
        budget = nb;
    }

    struct FrameIn { const float *tlwh; const float *feats; bool feats_on_device; const int *feat_rows; const float *payload; int D; };

    // Enqueues one frame; T_ub = host-side upper bound of the live track count when it starts.  Returns the int offset of
    // this frame's result block inside res_host / res_dev.
    size_t enqueue(const FrameIn &f, int T_ub, size_t in_off, size_t res_off, size_t *res_len, int *out_cap) {
        const int D = f.D, Dn = std::max(D, 1), Tn = std::max(T_ub, 1);
        // ---- inputs: tlwh, payload (and feat_rows) were packed into in_host by the caller; one H2D per batch
        TrkDev d;
        d.tab = table_at(table.p, capacity); d.tmp = table_at(table_tmp.p, capacity);
        d.meta = meta.p; d.free_slots = free_slots.p;
        d.mean = mean.p; d.cov = cov.p; d.gallery = gallery.p;
        feats_n.ensure_keep((size_t)Dn * EMB);
        cost_dev.ensure_keep((size_t)(Tn + Dn) * Dn);
        det_lists.ensure_keep((size_t)Dn * 8 + (size_t)(Tn + Dn) * 8);
        d.feats_n = feats_n.p; d.cost = cost_dev.p;
        d.tlwh = reinterpret_cast<const float *>(in_dev.p + in_off);
        d.payload = d.tlwh + (size_t)D * 4;
        const int *feat_rows_dev = f.feat_rows ? reinterpret_cast<const int *>(d.payload + D) : nullptr;
        int *tl = track_lists.p;                                  // 12 lists of `capacity` ints
        const int cap = capacity;
        d.conf_idx = tl; d.unconf_idx = tl + cap; d.flag_r = tl + 2 * cap; d.um_t_a = tl + 3 * cap; d.um_t_keep = tl + 4 * cap; d.um_t = tl + 5 * cap;
        d.iou_cand = tl + 6 * cap; d.out_slot = tl + 7 * cap; d.out_id = tl + 8 * cap; d.out_payload = reinterpret_cast<float *>(tl + 9 * cap);
        d.rows = tl + 10 * cap; d.cols = tl + 11 * cap;
        int *dl = det_lists.p;                                    // per-detection lists (D) and per-(track+det) lists
        d.flag_c = dl; d.um_d = dl + Dn; d.um_d2 = dl + 2 * Dn; d.new_slot = dl + 3 * Dn; d.rej = dl + 4 * Dn;
        int *pl = dl + 8 * (size_t)Dn;
        const int P = Tn + Dn;
        d.matches = pl; d.upd_slot = pl + 2 * P; d.upd_det = pl + 3 * P; d.upd_pos = pl + 4 * P;
        d.budget = budget; d.unbounded = unbounded ? 1 : 0; d.n_init = n_init; d.max_age = max_age;
        d.max_dist = (float)max_dist; d.max_iou = (float)max_iou;
        // ---- result block layout: header | out6 rows | matches | unmatched tracks | unmatched detections
        const int rows_cap = T_ub + D, mcap = std::min(T_ub + D, T_ub + D);
        d.res_out6 = M_COUNT; d.res_matches = d.res_out6 + rows_cap * 6; d.res_um_t = d.res_matches + 2 * mcap; d.res_um_d = d.res_um_t + T_ub + D;
        *res_len = (size_t)d.res_um_d + D + 1;
        *out_cap = rows_cap;
        d.res = res_dev.p + res_off;

        // ---- kernels (sizes come from device memory; the grids use the host-side upper bounds)
        const float *feats_dev = f.feats;
        if (!f.feats_on_device && D) {
            int n_rows = D;
            if (f.feat_rows) for (int k = 0; k < D; ++k) n_rows = std::max(n_rows, f.feat_rows[k] + 1);
            feats_stage.ensure_keep((size_t)n_rows * EMB);
            YDS_HIP(hipMemcpyAsync(feats_stage.p, f.feats, (size_t)n_rows * EMB * 4, hipMemcpyHostToDevice, stream));
            feats_dev = feats_stage.p;
        }
        if (D) hipLaunchKernelGGL(normalize_rows_kernel, dim3((D + 3) / 4), dim3(256), 0, stream, feats_dev, feat_rows_dev, feats_n.p, D,
                                  metric == METRIC_COSINE ? 1 : 0);     // x / ||x|| once per frame (nn_matching.py:50-52); euclidean: as is
        hipLaunchKernelGGL(trk_begin_kernel, dim3(1), dim3(256), 0, stream, d, D);
        if (T_ub) hipLaunchKernelGGL(trk_predict_kernel, dim3((T_ub + 63) / 64), dim3(64), 0, stream, d);
        if (T_ub && D) {
            hipLaunchKernelGGL(appearance_cost_kernel, dim3(T_ub, (D + 15) / 16), dim3(256), 0, stream, gallery.p, (const int *)nullptr,
                               (const int *)nullptr, budget, feats_n.p, D, mean.p, cov.p, d.tlwh, (float)max_dist, (float)(max_dist + 1e-5), 1,
                               metric == METRIC_EUCLIDEAN ? 1 : 0, cost_dev.p, d.conf_idx, d.tab.slot, d.tab.n_feat, meta.p + M_TC);
            launch_lsap(cost_dev.p, T_ub, D, meta.p + M_TC, d.rows, d.cols, meta.p + M_NA, lsap_scratch, stream);
        }
        hipLaunchKernelGGL(trk_match_a_kernel, dim3(1), dim3(256), 0, stream, d);
        if (T_ub && D) {
            hipLaunchKernelGGL(iou_cost_kernel, dim3(((size_t)T_ub * D + 255) / 256), dim3(256), 0, stream, mean.p, (const int *)nullptr,
                               (const int *)nullptr, 0, d.tlwh, d.um_d, 0, (float)max_iou, (float)(max_iou + 1e-5), cost_dev.p, meta.p + M_TB,
                               d.iou_cand, d.tab.slot, d.tab.tsu);
            launch_lsap(cost_dev.p, T_ub, D, meta.p + M_TB, d.rows, d.cols, meta.p + M_NB, lsap_scratch, stream);
        }
        hipLaunchKernelGGL(trk_match_b_kernel, dim3(1), dim3(256), 0, stream, d);
        if (D) {
            hipLaunchKernelGGL(trk_kalman_kernel, dim3((D + 63) / 64), dim3(64), 0, stream, d);
            hipLaunchKernelGGL(trk_append_kernel, dim3(D), dim3(128), 0, stream, d);
        }
        if (rows_cap) hipLaunchKernelGGL(trk_output_kernel, dim3((rows_cap + 63) / 64), dim3(64), 0, stream, d);
        YDS_HIP(hipGetLastError());
        return res_off;
    }

    // One or several frames, in order, with ONE host synchronisation at the end.  counts[b] = rows of frame b.
    void run(const FrameIn *frames, int n_frames, int32_t *const *out6, const int *caps, int *counts) {
        // ---- capacity for the worst case: every detection of the batch starts a track, every frame adds a gallery row
        int D_sum = 0;
        for (int b = 0; b < n_frames; ++b) D_sum += frames[b].D;
        if (T_host + D_sum > capacity) { int c = capacity; while (c < T_host + D_sum) c *= 2; grow(c); }
        // (max_rows = the largest gallery any LIVE track held after the last synchronised frame, from the result header;
        //  every frame of this call can add one row to it)
        if (unbounded) while (max_rows + n_frames + 1 > budget) grow_budget();
        // ---- inputs of all frames in one pinned block, one upload
        std::vector<size_t> in_off(n_frames);
        size_t in_total = 0;
        for (int b = 0; b < n_frames; ++b) { in_off[b] = in_total; in_total += (size_t)frames[b].D * 6 + 4; }
        if (in_total > in_cap) {
            if (in_host) (void)hipHostFree(in_host);
            in_cap = in_total * 2;
            YDS_HIP(hipHostMalloc((void **)&in_host, in_cap * sizeof(int)));
            in_dev.alloc(in_cap);
        }
        for (int b = 0; b < n_frames; ++b) {
            const FrameIn &f = frames[b];
            float *dst = reinterpret_cast<float *>(in_host + in_off[b]);
            if (f.D) {
                memcpy(dst, f.tlwh, (size_t)f.D * 16);
                memcpy(dst + (size_t)f.D * 4, f.payload, (size_t)f.D * 4);
                if (f.feat_rows) memcpy(dst + (size_t)f.D * 5, f.feat_rows, (size_t)f.D * 4);
            }
        }
        YDS_HIP(hipMemcpyAsync(in_dev.p, in_host, in_total * sizeof(int), hipMemcpyHostToDevice, stream));
        // ---- result blocks
        std::vector<size_t> res_off(n_frames), res_len(n_frames);
        std::vector<int> out_cap(n_frames);
        size_t res_total = 0;
        {
            int T_ub = T_host;
            for (int b = 0; b < n_frames; ++b) { res_off[b] = res_total; res_total += (size_t)M_COUNT + (size_t)(T_ub + frames[b].D) * 10 + frames[b].D + 8; T_ub += frames[b].D; }
        }
        if (res_total > res_cap) {
            YDS_HIP(hipStreamSynchronize(stream));
            if (res_host) (void)hipHostFree(res_host);
            res_cap = res_total * 2;
            YDS_HIP(hipHostMalloc((void **)&res_host, res_cap * sizeof(int)));
            res_dev.alloc(res_cap);
        }
        int T_ub = T_host;
        for (int b = 0; b < n_frames; ++b) {
            enqueue(frames[b], T_ub, in_off[b], res_off[b], &res_len[b], &out_cap[b]);
            T_ub += frames[b].D;
        }
        YDS_HIP(hipMemcpyAsync(res_host, res_dev.p, res_total * sizeof(int), hipMemcpyDeviceToHost, stream));
        YDS_HIP(hipStreamSynchronize(stream));
        for (int b = 0; b < n_frames; ++b) {
            const int *r = res_host + res_off[b];
            const int m = r[M_NOUT];
            if (m > caps[b]) fail("tracker: %d output rows exceed the caller's capacity %d", m, caps[b]);
            if (m) memcpy(out6[b], r + M_COUNT, (size_t)m * 6 * sizeof(int));
            counts[b] = m;
        }
        // ---- host-side mirror of the last frame (debug lists for the parity tests, live track count)
        {
            const int b = n_frames - 1, D = frames[b].D;
            int T_before = T_host;
            for (int k = 0; k < b; ++k) T_before += frames[k].D;         // upper bound used for that frame's layout
            const int *r = res_host + res_off[b];
            const int rows_cap = T_before + D;
            const int *pm = r + M_COUNT + rows_cap * 6, *pt = pm + 2 * rows_cap, *pd = pt + rows_cap;
            last_matches.assign(r[M_NM], {0, 0});
            for (int k = 0; k < r[M_NM]; ++k) last_matches[k] = {pm[2 * k], pm[2 * k + 1]};
            last_um_t.assign(pt, pt + r[M_NUT]);
            std::sort(last_um_t.begin(), last_um_t.end());
            last_um_d.assign(pd, pd + r[M_NUD]);
            T_host = r[M_T];
            max_rows = r[M_MAXFEAT];
        }
    }

    int step(const float *tlwh_host, const float *feats, bool feats_on_device, const float *payload, int D, int32_t *out6, int cap) override {
        return step_sel(tlwh_host, feats, feats_on_device, nullptr, payload, D, out6, cap);
    }
    // feat_rows (optional): detection d uses row feat_rows[d] of `feats` (tracker-side NMS keeps a subset in pick order)
    int step_sel(const float *tlwh_host, const float *feats, bool feats_on_device, const int *feat_rows, const float *payload, int D,
                 int32_t *out6, int cap) {
        FrameIn f{tlwh_host, feats, feats_on_device, feat_rows, payload, D};
        int count = 0;
        run(&f, 1, &out6, &cap, &count);
        return count;
    }
    // frames of one batch, in order, one synchronisation (the pipeline's association stage); skip[b]: tracker not called
    void step_batch(int n, const float *tlwh_host, const int *first, const float *feats_dev, const float *payload, const char *skip, int32_t *out6,
                    int cap, int32_t *counts) override {
        std::vector<FrameIn> fr;
        std::vector<int32_t *> outs;
        std::vector<int> caps, cnt, which;
        for (int b = 0; b < n; ++b) {
            if (skip && skip[b]) { counts[b] = -1; continue; }
            const int D = first[b + 1] - first[b];
            fr.push_back(FrameIn{tlwh_host + (size_t)first[b] * 4, feats_dev + (size_t)first[b] * EMB, true, nullptr, payload + first[b], D});
            outs.push_back(out6 + (size_t)b * cap * 6);
            caps.push_back(cap);
            which.push_back(b);
        }
        cnt.assign(fr.size(), 0);
        if (!fr.empty()) run(fr.data(), (int)fr.size(), outs.data(), caps.data(), cnt.data());
        for (size_t k = 0; k < which.size(); ++k) counts[which[k]] = cnt[k];
    }

    // host copy of the integer table (parity tests, DeepSort.tracker.tracks)
    struct HostTable { std::vector<int> slot, id, hits, age, tsu, state, n_feat; };
    HostTable read_table() {
        HostTable h;
        const int T = T_host;
        auto get = [&](int field, std::vector<int> &v) {
            v.resize(T);
            if (T) YDS_HIP(hipMemcpy(v.data(), table.p + (size_t)field * capacity, (size_t)T * 4, hipMemcpyDeviceToHost));
        };
        get(0, h.slot); get(1, h.id); get(2, h.hits); get(3, h.age); get(4, h.tsu); get(5, h.state); get(6, h.n_feat);
        return h;
    }

    double max_dist, max_iou;       // python floats in the reference; fp32 roundings are taken where torch/numpy take them
    int max_age, n_init, budget;
    bool unbounded = false;
    int metric = METRIC_COSINE;
    int capacity = 0;
    int T_host = 0;                 // live tracks after the last synchronised frame
    int max_rows = 1;               // largest gallery row count of a live track after the last synchronised frame
    template <class T> struct GrowBuf : DevBuf<T> {
        // ensure() that never shrinks and - unlike DevBuf::ensure - may only be called while nothing that uses the old buffer is
        // in flight; growth is rare (sizes follow the largest frame seen), so it simply drains the stream first
        hipStream_t *st = nullptr;
        void ensure_keep(size_t count) {
            if (count <= this->n) return;
            if (st && *st) (void)hipStreamSynchronize(*st);
            this->alloc(count + count / 2);
        }
    };
    DevBuf<float> mean, cov, gallery /* cosine: rows stored normalised */;
    DevBuf<int> table, table_tmp, free_slots, track_lists, meta, in_dev, res_dev;
    GrowBuf<float> feats_stage, feats_n, cost_dev;
    GrowBuf<int> det_lists;
    DevBuf<char> lsap_scratch;
    int *res_host = nullptr, *in_host = nullptr;
    size_t res_cap = 0, in_cap = 0;
    std::vector<std::pair<int, int>> last_matches;
    std::vector<int> last_um_t, last_um_d;
    hipStream_t stream = nullptr;
};

}  // namespace yds

// ============================================================================================ C ABI
static inline yds::Tracker *impl(yds_trk *h) { return static_cast<yds::Tracker *>(h->t); }
static inline const yds::Tracker *impl(const yds_trk *h) { return static_cast<const yds::Tracker *>(h->t); }

namespace {
struct Scratch {
    hipStream_t s = nullptr;
    ~Scratch() { if (s) (void)hipStreamDestroy(s); }
    hipStream_t stream() { if (!s) YDS_HIP(hipStreamCreate(&s)); return s; }
};
Scratch g_scratch;
std::vector<int> iota(int n) { std::vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = i; return v; }
}  // namespace

extern "C" {

yds_trk *yds_tracker_create(double max_dist, double max_iou_distance, int max_age, int n_init, int nn_budget) {
    YDS_API_BEGIN
    return new yds_trk{new yds::Tracker(max_dist, max_iou_distance, max_age, n_init, nn_budget)};
    YDS_API_END_PTR
}
yds_trk *yds_tracker_create_ex(double max_dist, double max_iou_distance, int max_age, int n_init, int nn_budget, int metric) {
    YDS_API_BEGIN
    return new yds_trk{new yds::Tracker(max_dist, max_iou_distance, max_age, n_init, nn_budget, metric)};
    YDS_API_END_PTR
}
int yds_tracker_step_sel(yds_trk *t, const float *tlwh, const float *feats, int feats_on_device, const int32_t *feat_rows, const float *payload,
                         int D, int32_t *out6, int cap, int *m_out, int32_t *dbg_matches, int dbg_cap, int *n_matches) {
    YDS_API_BEGIN
    *m_out = impl(t)->step_sel(tlwh, feats, feats_on_device != 0, feat_rows, payload, D, out6, cap);
    if (n_matches) {
        const auto &lm = impl(t)->last_matches;
        *n_matches = (int)lm.size();
        if (dbg_matches) {
            if ((int)lm.size() > dbg_cap) yds::fail("tracker: %zu matches exceed dbg_cap %d", lm.size(), dbg_cap);
            for (size_t k = 0; k < lm.size(); ++k) { dbg_matches[2 * k] = lm[k].first; dbg_matches[2 * k + 1] = lm[k].second; }
        }
    }
    YDS_API_END
}
void yds_tracker_destroy(yds_trk *t) {
    if (t) { delete t->t; delete t; }
}
int yds_tracker_step(yds_trk *t, const float *tlwh, const float *feats, const float *payload, int D, int32_t *out6, int cap, int *m_out,
                     int32_t *dbg_matches, int dbg_cap, int *n_matches) {
    YDS_API_BEGIN
    *m_out = impl(t)->step(tlwh, feats, false, payload, D, out6, cap);
    if (n_matches) {
        const auto &lm = impl(t)->last_matches;
        *n_matches = (int)lm.size();
        if (dbg_matches) {
            if ((int)lm.size() > dbg_cap) yds::fail("tracker: %zu matches exceed dbg_cap %d", lm.size(), dbg_cap);
            for (size_t k = 0; k < lm.size(); ++k) { dbg_matches[2 * k] = lm[k].first; dbg_matches[2 * k + 1] = lm[k].second; }
        }
    }
    YDS_API_END
}
int yds_tracker_step_dev(yds_trk *t, const float *tlwh, const float *feats_dev, const float *payload, int D, int32_t *out6, int cap, int *m_out) {
    YDS_API_BEGIN
    *m_out = impl(t)->step(tlwh, feats_dev, true, payload, D, out6, cap);
    YDS_API_END
}
int yds_tracker_num_tracks(const yds_trk *t) { return t->t->num_tracks(); }
int yds_tracker_get_state(yds_trk *t, int32_t *ids, int32_t *state, int32_t *tsu, int32_t *hits, float *mean8, float *cov64, int cap, int *T) {
    YDS_API_BEGIN
    yds::Tracker *k = impl(t);
    const int n = k->T_host;
    if (n > cap) yds::fail("tracker: %d tracks exceed cap %d", n, cap);
    *T = n;
    const yds::Tracker::HostTable h = k->read_table();          // the table lives on the device; this is a debug / parity read-back
    std::vector<float> m, c;
    if (mean8 && n) { m.resize((size_t)k->capacity * 8); YDS_HIP(hipMemcpy(m.data(), k->mean.p, m.size() * 4, hipMemcpyDeviceToHost)); }
    if (cov64 && n) { c.resize((size_t)k->capacity * 64); YDS_HIP(hipMemcpy(c.data(), k->cov.p, c.size() * 4, hipMemcpyDeviceToHost)); }
    for (int i = 0; i < n; ++i) {
        if (ids) ids[i] = h.id[i];
        if (state) state[i] = h.state[i];
        if (tsu) tsu[i] = h.tsu[i];
        if (hits) hits[i] = h.hits[i];
        if (mean8) memcpy(mean8 + (size_t)i * 8, &m[(size_t)h.slot[i] * 8], 32);
        if (cov64) memcpy(cov64 + (size_t)i * 64, &c[(size_t)h.slot[i] * 64], 256);
    }
    YDS_API_END
}
int yds_tracker_gallery_rows(const yds_trk *t) { return impl(t)->budget; }
int yds_tracker_get_payload(yds_trk *t, float *payload, int cap) {
    YDS_API_BEGIN
    yds::Tracker *k = impl(t);
    const int n = k->T_host;
    if (n > cap) yds::fail("tracker: %d tracks exceed cap %d", n, cap);
    if (n) YDS_HIP(hipMemcpy(payload, k->table.p + (size_t)8 * k->capacity, (size_t)n * 4, hipMemcpyDeviceToHost));
    YDS_API_END
}
int yds_tracker_get_age(yds_trk *t, int32_t *age, int cap) {
    YDS_API_BEGIN
    yds::Tracker *k = impl(t);
    const int n = k->T_host;
    if (n > cap) yds::fail("tracker: %d tracks exceed cap %d", n, cap);
    if (n) YDS_HIP(hipMemcpy(age, k->table.p + (size_t)3 * k->capacity, (size_t)n * 4, hipMemcpyDeviceToHost));
    YDS_API_END
}
int yds_tracker_last_unmatched(yds_trk *t, int32_t *um_tracks, int cap_t, int *n_t, int32_t *um_dets, int cap_d, int *n_d) {
    YDS_API_BEGIN
    const auto &a = impl(t)->last_um_t, &b = impl(t)->last_um_d;
    if ((int)a.size() > cap_t || (int)b.size() > cap_d) yds::fail("tracker: unmatched lists exceed caller capacity");
    *n_t = (int)a.size(); *n_d = (int)b.size();
    for (size_t i = 0; i < a.size(); ++i) um_tracks[i] = a[i];
    for (size_t i = 0; i < b.size(); ++i) um_dets[i] = b[i];
    YDS_API_END
}

// ---- stand-alone primitives for parity tests ------------------------------------------------------
int yds_lsap(const float *cost_host, int nr, int nc, int32_t *rows, int32_t *cols, int *n_out) {
    YDS_API_BEGIN
    using namespace yds;
    int n = std::min(nr, nc);
    *n_out = n;
    if (n == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> c; c.upload(cost_host, (size_t)nr * nc, s);
    DevBuf<int> r(n), cc(n);
    DevBuf<char> scratch;
    launch_lsap(c.p, nr, nc, nullptr, r.p, cc.p, nullptr, scratch, s);
    YDS_HIP(hipMemcpyAsync(rows, r.p, n * sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cols, cc.p, n * sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_lsap_bench(const float *cost_host, int nr, int nc, int iters, double *avg_us) {
    YDS_API_BEGIN
    using namespace yds;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> c; c.upload(cost_host, (size_t)nr * nc, s);
    const int n = std::max(std::min(nr, nc), 1);
    DevBuf<int> r(n), cc(n);
    DevBuf<char> scratch;
    hipEvent_t e0, e1;
    YDS_HIP(hipEventCreate(&e0)); YDS_HIP(hipEventCreate(&e1));
    launch_lsap(c.p, nr, nc, nullptr, r.p, cc.p, nullptr, scratch, s);
    YDS_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_lsap(c.p, nr, nc, nullptr, r.p, cc.p, nullptr, scratch, s);
    YDS_HIP(hipEventRecord(e1, s));
    YDS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    YDS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1e3 / iters;
    if (YDS_LSAP_PROF) {
        unsigned long long v[8];
        YDS_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(yds_lsap_prof), sizeof v));
        fprintf(stderr, "lsap prof: loop cycles/iter %.0f, iters/launch %.0f, kernel cycles %.0f, wall ticks(100MHz) %.0f -> %.2f GHz\n",
                (double)v[0] / v[1], (double)v[1] / v[4], (double)v[2] / v[4], (double)v[3] / v[4], (double)v[2] / ((double)v[3] * 10.0));
        unsigned long long z[8] = {};
        YDS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(yds_lsap_prof), z, sizeof z));
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    YDS_API_END
}
int yds_kalman_predict(float *mean_host, float *cov_host, int T) {
    YDS_API_BEGIN
    using namespace yds;
    if (T == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c; DevBuf<int> sl;
    m.upload(mean_host, (size_t)T * 8, s); c.upload(cov_host, (size_t)T * 64, s);
    auto v = iota(T); sl.upload(v.data(), T, s);
    hipLaunchKernelGGL(kf_predict_kernel, dim3((T + 63) / 64), dim3(64), 0, s, m.p, c.p, sl.p, T);
    YDS_HIP(hipMemcpyAsync(mean_host, m.p, (size_t)T * 32, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cov_host, c.p, (size_t)T * 256, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_update(float *mean_host, float *cov_host, const float *xyah_host, int M) {
    YDS_API_BEGIN
    using namespace yds;
    if (M == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c, z; DevBuf<int> sl;
    m.upload(mean_host, (size_t)M * 8, s); c.upload(cov_host, (size_t)M * 64, s); z.upload(xyah_host, (size_t)M * 4, s);
    auto v = iota(M); sl.upload(v.data(), M, s);
    hipLaunchKernelGGL(kf_update_kernel, dim3((M + 63) / 64), dim3(64), 0, s, m.p, c.p, sl.p, z.p, M);
    YDS_HIP(hipMemcpyAsync(mean_host, m.p, (size_t)M * 32, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cov_host, c.p, (size_t)M * 256, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_gating(const float *mean_host, const float *cov_host, int T, const float *xyah_host, int D, float *out) {
    YDS_API_BEGIN
    using namespace yds;
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c, z, o((size_t)T * D); DevBuf<int> sl;
    m.upload(mean_host, (size_t)T * 8, s); c.upload(cov_host, (size_t)T * 64, s); z.upload(xyah_host, (size_t)D * 4, s);
    auto v = iota(T); sl.upload(v.data(), T, s);
    hipLaunchKernelGGL(gating_kernel, dim3((T * D + 255) / 256), dim3(256), 0, s, m.p, c.p, sl.p, T, z.p, D, o.p);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_gating_ex(const float *mean_host, const float *cov_host, int T, const float *xyah_host, int D, int only_position, float *out) {
    if (only_position) return yds_kalman_gating(mean_host, cov_host, T, xyah_host, D, out);
    YDS_API_BEGIN
    using namespace yds;
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c, z, o((size_t)T * D);
    m.upload(mean_host, (size_t)T * 8, s); c.upload(cov_host, (size_t)T * 64, s); z.upload(xyah_host, (size_t)D * 4, s);
    hipLaunchKernelGGL(gating4_kernel, dim3((T * D + 255) / 256), dim3(256), 0, s, m.p, c.p, T, z.p, D, o.p);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_initiate(const float *xyah_host, int n, float *mean_host, float *cov_host) {
    YDS_API_BEGIN
    using namespace yds;
    if (n == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> z, m((size_t)n * 8), c((size_t)n * 64);
    z.upload(xyah_host, (size_t)n * 4, s);
    hipLaunchKernelGGL(kf_initiate_xyah_kernel, dim3((n + 63) / 64), dim3(64), 0, s, z.p, m.p, c.p, n);
    YDS_HIP(hipMemcpyAsync(mean_host, m.p, (size_t)n * 32, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cov_host, c.p, (size_t)n * 256, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_kalman_project(const float *mean_host, const float *cov_host, int n, float *mean4_host, float *cov16_host) {
    YDS_API_BEGIN
    using namespace yds;
    if (n == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> m, c, m4((size_t)n * 4), c16((size_t)n * 16);
    m.upload(mean_host, (size_t)n * 8, s); c.upload(cov_host, (size_t)n * 64, s);
    hipLaunchKernelGGL(kf_project_kernel, dim3((n + 63) / 64), dim3(64), 0, s, m.p, c.p, m4.p, c16.p, n);
    YDS_HIP(hipMemcpyAsync(mean4_host, m4.p, (size_t)n * 16, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cov16_host, c16.p, (size_t)n * 64, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_iou_cost(const float *track_tlwh_host, int T, const float *det_tlwh_host, int D, float *out) {
    YDS_API_BEGIN
    using namespace yds;
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    // tracks are given as tlwh: rebuild the (x, y, a, h) means the kernel reads
    std::vector<float> mean((size_t)T * 8, 0.f);
    for (int t = 0; t < T; ++t) {
        const float *b = track_tlwh_host + t * 4;
        mean[t * 8 + 0] = b[0] + b[2] / 2.f; mean[t * 8 + 1] = b[1] + b[3] / 2.f; mean[t * 8 + 2] = b[2] / b[3]; mean[t * 8 + 3] = b[3];
    }
    DevBuf<float> m, dt, o((size_t)T * D); DevBuf<int> sl, di;
    m.upload(mean.data(), mean.size(), s); dt.upload(det_tlwh_host, (size_t)D * 4, s);
    auto v = iota(T); sl.upload(v.data(), T, s);
    auto w = iota(D); di.upload(w.data(), D, s);
    hipLaunchKernelGGL(iou_cost_kernel, dim3((T * D + 255) / 256), dim3(256), 0, s, m.p, sl.p, (const int *)nullptr, T, dt.p, di.p, D, 0.f, 0.f, o.p, (const int *)nullptr, (const int *)nullptr,
                       (const int *)nullptr, (const int *)nullptr);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
static int nn_min_cost(const float *gallery_host, const int32_t *seg_offsets_host, int T, const float *feats_host, int D, int dim, float *out,
                       int euclid) {
    YDS_API_BEGIN
    using namespace yds;
    // nn_matching.py:4-53 is dimension agnostic; the kernel reduces 512-wide rows, so narrower embeddings are zero padded (zeros
    // change neither a dot product nor a norm) and wider ones are refused
    if (dim < 1 || dim > EMB) fail("nn distance: feature dimension %d outside [1, %d]", dim, EMB);
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    int budget = 1;
    for (int t = 0; t < T; ++t) {
        if (seg_offsets_host[t + 1] < seg_offsets_host[t]) fail("nn distance: segment offsets must not decrease");
        budget = std::max(budget, seg_offsets_host[t + 1] - seg_offsets_host[t]);
    }
    std::vector<float> g((size_t)T * budget * EMB, 0.f), fpad;
    std::vector<int> rows(T);
    for (int t = 0; t < T; ++t) {
        rows[t] = seg_offsets_host[t + 1] - seg_offsets_host[t];
        for (int r = 0; r < rows[t]; ++r)
            memcpy(&g[((size_t)t * budget + r) * EMB], gallery_host + ((size_t)seg_offsets_host[t] + r) * dim, (size_t)dim * 4);
    }
    if (dim != EMB) {
        fpad.assign((size_t)D * EMB, 0.f);
        for (int d = 0; d < D; ++d) memcpy(&fpad[(size_t)d * EMB], feats_host + (size_t)d * dim, (size_t)dim * 4);
        feats_host = fpad.data();
    }
    DevBuf<float> gd, fd, gn(g.size()), fn((size_t)D * EMB), o((size_t)T * D); DevBuf<int> sl, nr;
    gd.upload(g.data(), g.size(), s); fd.upload(feats_host, (size_t)D * EMB, s);
    auto v = iota(T); sl.upload(v.data(), T, s); nr.upload(rows.data(), T, s);
    const int G = T * budget;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((G + 3) / 4), dim3(256), 0, s, gd.p, (const int *)nullptr, gn.p, G, euclid ? 0 : 1);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((D + 3) / 4), dim3(256), 0, s, fd.p, (const int *)nullptr, fn.p, D, euclid ? 0 : 1);
    hipLaunchKernelGGL(appearance_cost_kernel, dim3(T, (D + 15) / 16), dim3(256), 0, s, gn.p, sl.p, nr.p, budget, fn.p, D,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.f, 0.f, 0, euclid, o.p, (const int *)nullptr,
                       (const int *)nullptr, (const int *)nullptr, (const int *)nullptr);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
int yds_cosine_min_cost(const float *gallery_host, const int32_t *seg_offsets_host, int T, const float *feats_host, int D, int dim, float *out) {
    return nn_min_cost(gallery_host, seg_offsets_host, T, feats_host, D, dim, out, 0);
}
int yds_euclidean_min_cost(const float *gallery_host, const int32_t *seg_offsets_host, int T, const float *feats_host, int D, int dim, float *out) {
    return nn_min_cost(gallery_host, seg_offsets_host, T, feats_host, D, dim, out, 1);
}
int yds_tracker_nms(const float *tlwh_host, const int32_t *order_host, int D, double max_overlap, int32_t *pick_host, int *n_pick) {
    YDS_API_BEGIN
    using namespace yds;
    *n_pick = 0;
    if (D == 0) return 0;
    for (int k = 0; k < D; ++k) if (order_host[k] < 0 || order_host[k] >= D) fail("tracker_nms: order[%d] = %d outside [0,%d)", k, order_host[k], D);
    hipStream_t s = g_scratch.stream();
    DevBuf<float> b; DevBuf<int> ord, pick(D), cnt(1);
    b.upload(tlwh_host, (size_t)D * 4, s); ord.upload(order_host, D, s);
    hipLaunchKernelGGL(tracker_nms_kernel, dim3(1), dim3(256), (size_t)D * sizeof(int), s, b.p, ord.p, D, max_overlap, pick.p, cnt.p);
    YDS_HIP(hipGetLastError());
    YDS_HIP(hipMemcpyAsync(n_pick, cnt.p, sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(pick_host, pick.p, (size_t)D * sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}

}  // extern "C"
