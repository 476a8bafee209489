// Device-side pieces shared by the tracker translation units (tracker.hip, tracker_kalman.hip, tracker_cost.hip, tracker_lsap.hip):
// constants, the Kalman step bodies (deep_sort/sort/kalman_filter.py:54-256, one thread per track) and the kernel / launcher
// declarations.  A kernel defined in one unit is launched from another through its host stub (no relocatable device code needed).
#pragma once
#include "engine.h"

#include <algorithm>
#include <math.h>
#include <string.h>

namespace yds {

constexpr int EMB = 512;
constexpr float INFTY_COST = 1e5f;
constexpr float CHI2_2DOF = 5.9915f;

// std weights are fp32 roundings of 1/20 and 1/160 like the reference's tensors (kalman_filter.py:39-52)
constexpr float kStdPos = 1.f / 20, kStdVel = 1.f / 160;

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

__device__ __forceinline__ void to_xyah(const float *b, float z[4]) {
    z[0] = b[0] + b[2] / 2.f; z[1] = b[1] + b[3] / 2.f; z[2] = b[2] / b[3]; z[3] = b[3];
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


// ---- cost entries shared by the stand-alone kernels (tracker_cost.hip) and the fused per-frame kernels (tracker.hip)
// One (track, 16-detection slab) block of the appearance cost: min over the track's gallery rows of the cosine / squared euclidean
// distance, Mahalanobis gate on the (predicted) state mean_t / cov_t, min_cost_matching clamp; 256 threads, 66 KB of static LDS.
__device__ __forceinline__ void appearance_cost_block(const float *gallery_n, int slot, int rows, int budget, const float *feats_n, int D, int d0,
                                                      const float *mean_t, const float *cov_t, const float *tlwh, float max_dist, float flood,
                                                      int do_gate, int euclid, float *cost_row) {
    const int nd = min(16, D - d0);
    __shared__ float fs[16][EMB + 1], gs[16][EMB + 1];
    __shared__ float best[16][17];
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
            if (gate2(mean_t, cov_t, z) > CHI2_2DOF) c = INFTY_COST;
        }
        if (max_dist > 0.f && c > max_dist) c = flood;            // linear_assignment.py:52
        cost_row[dd] = c;
    }
}

// IOU cost of one (track, detection) pair (iou_matching.py:5-91): m = the track's state mean, c = the detection's tlwh
__device__ __forceinline__ float iou_cost_entry(const float *m, const float *c, bool stale, float max_dist, float flood) {
    float bw = m[2] * m[3], bh = m[3];                        // Track.to_tlwh track.py:81-94
    float bx = m[0] - bw / 2.f, by = m[1] - bh / 2.f;
    float ix0 = fmaxf(bx, c[0]), iy0 = fmaxf(by, c[1]);
    float ix1 = fminf(bx + bw, c[2] + c[0]), iy1 = fminf(by + bh, c[3] + c[1]);
    float iw = fmaxf(ix1 - ix0 + 1.f, 0.f), ih = fmaxf(iy1 - iy0 + 1.f, 0.f);      // asymmetric +1, iou_matching.py:36
    float inter = iw * ih;
    float v = 1.f - inter / (bw * bh + c[2] * c[3] - inter);
    if (stale) v = INFTY_COST;                                // time_since_update > 1, iou_matching.py:86-89
    if (max_dist > 0.f && v > max_dist) v = flood;
    return v;
}

// ---- kernels (definitions: tracker_kalman.hip, tracker_cost.hip)
__global__ void kf_predict_kernel(float *mean, float *cov, const int *slots, int n);
__global__ void kf_update_kernel(float *mean, float *cov, const int *slots, const float *z, int n);
__global__ void kf_initiate_kernel(float *mean, float *cov, const int *slots, const float *tlwh, const int *det_idx, int n);
__global__ void tlwh_to_xyah_kernel(const float *tlwh, const int *det_idx, float *z, int n);
__global__ void gating_kernel(const float *mean, const float *cov, const int *slots, int T, const float *xyah, int D, float *out);
__global__ void gating4_kernel(const float *mean, const float *cov, int T, const float *xyah, int D, float *out);
__global__ void kf_initiate_xyah_kernel(const float *xyah, float *mean, float *cov, int n);
__global__ void kf_project_kernel(const float *mean, const float *cov, float *mean4, float *cov16, int n);
__global__ void normalize_rows_kernel(const float *src, const int *src_idx, float *dst, int n, int normalise);
__global__ __launch_bounds__(256) void appearance_cost_kernel(const float *gallery_n, const int *slots, const int *n_rows, int budget,
                                                             const float *feats_n, int D, const float *mean, const float *cov,
                                                             const float *tlwh, float max_dist, float flood, int do_gate, int euclid, float *cost,
                                                             const int *idx, const int *tab_slot, const int *tab_nfeat, const int *count_p);
__global__ void iou_cost_kernel(const float *mean, const int *slots, const int *stale, int T, const float *tlwh, const int *det_idx,
                                int D, float max_dist, float flood, float *cost, const int *dims_p, const int *cand, const int *tab_slot,
                                const int *tab_tsu);
__global__ __launch_bounds__(256) void tracker_nms_kernel(const float *tlwh, const int *order, int n, double max_overlap, int *pick, int *n_pick);

// ---- linear assignment (tracker_lsap.hip): scipy.optimize.linear_sum_assignment on the device
// nr_max / nc_max: upper bounds known on the host (they size the LDS / scratch); the real sizes may come from dims_dev
void launch_lsap(const float *cost_dev, int nr_max, int nc_max, const int *dims_dev, int *rows_dev, int *cols_dev, int *n_out_dev,
                 DevBuf<char> &scratch, hipStream_t s);

}  // namespace yds
