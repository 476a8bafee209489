// DeepSORT association stage: batched HIP kernels over all tracks / detections + host bookkeeping.
//
// Replaces the per-object torch/numpy/scipy code of the reference:
//   KalmanFilter.initiate/predict/project/update/gating_distance   deep_sort/sort/kalman_filter.py:54-256
//   NearestNeighborDistanceMetric.distance (_nn_cosine_distance)     deep_sort/sort/nn_matching.py:30-53,77-100,158-187
//   gate_cost_matrix + min_cost_matching thresholding                deep_sort/sort/linear_assignment.py:52,147-203
//   iou / iou_cost                                                   deep_sort/sort/iou_matching.py:5-91
//   scipy.optimize.linear_sum_assignment (third party)               call site linear_assignment.py:56
//   DeepSort.update output stage                                     deep_sort/deep_sort.py:63-88,108-114
// Track state (mean, covariance, appearance gallery ring) lives in HBM, addressed through slot ids;
// integer lifecycle state (ids, hits, age, time_since_update, Tentative/Confirmed/Deleted) and the
// order-sensitive list bookkeeping (linear_assignment.py:58-72, tracker.py:56-93,115-176, track.py) stay
// on the host: they are a few hundred integer operations per frame and decide track ids.
#include "engine.h"

#include <algorithm>
#include <math.h>
#include <string.h>

namespace yds {

constexpr int EMB = 512;
constexpr float INFTY_COST = 1e5f;
constexpr float CHI2_2DOF = 5.9915f;
constexpr int LSAP_MAX = 1024;

// ------------------------------------------------------------------------------------------ Kalman
// std weights are fp32 roundings of 1/20 and 1/160 like the reference's tensors (kalman_filter.py:39-52)
__device__ __constant__ float kStdPos = 1.f / 20, kStdVel = 1.f / 160;

__global__ void kf_predict_kernel(float *mean, float *cov, const int *slots, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float *m = mean + (size_t)slots[t] * 8, *P = cov + (size_t)slots[t] * 64;
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
__global__ void kf_update_kernel(float *mean, float *cov, const int *slots, const float *z, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float *m = mean + (size_t)slots[t] * 8, *P = cov + (size_t)slots[t] * 64;
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
    for (int i = 0; i < 4; ++i) innov[i] = z[t * 4 + i] - m[i];
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
__global__ void kf_initiate_kernel(float *mean, float *cov, const int *slots, const float *tlwh, const int *det_idx, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float *b = tlwh + (size_t)det_idx[t] * 4;
    float *m = mean + (size_t)slots[t] * 8, *P = cov + (size_t)slots[t] * 64;
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

__global__ __launch_bounds__(256) void appearance_cost_kernel(const float *gallery_n, const int *slots, const int *n_rows, int budget,
                                                             const float *feats_n, int D, const float *mean, const float *cov,
                                                             const float *tlwh, float max_dist, float flood, int do_gate, int euclid, float *cost) {
    __shared__ float fs[16][EMB + 1], gs[16][EMB + 1];
    __shared__ float best[16][17];
    const int t = blockIdx.x, d0 = blockIdx.y * 16;
    const int nd = min(16, D - d0);
    const int slot = slots[t], rows = n_rows[t];
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
__global__ void iou_cost_kernel(const float *mean, const int *slots, const int *stale, int T, const float *tlwh, const int *det_idx,
                                int D, float max_dist, float flood, float *cost) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * D) return;
    int t = idx / D, d = idx - t * D;
    const float *m = mean + (size_t)slots[t] * 8;
    float bw = m[2] * m[3], bh = m[3];                        // Track.to_tlwh track.py:81-94
    float bx = m[0] - bw / 2.f, by = m[1] - bh / 2.f;
    const float *c = tlwh + (size_t)det_idx[d] * 4;
    float ix0 = fmaxf(bx, c[0]), iy0 = fmaxf(by, c[1]);
    float ix1 = fminf(bx + bw, c[2] + c[0]), iy1 = fminf(by + bh, c[3] + c[1]);
    float iw = fmaxf(ix1 - ix0 + 1.f, 0.f), ih = fmaxf(iy1 - iy0 + 1.f, 0.f);      // asymmetric +1, iou_matching.py:36
    float inter = iw * ih;
    float v = 1.f - inter / (bw * bh + c[2] * c[3] - inter);
    if (stale && stale[t]) v = INFTY_COST;                    // time_since_update > 1, iou_matching.py:86-89
    if (max_dist > 0.f && v > max_dist) v = flood;
    cost[idx] = v;
}

// ------------------------------------------------------------------------------------------ LSAP
// scipy.optimize.linear_sum_assignment (rectangular_lsap.cpp, Crouse 2016) on ONE wavefront: the
// augmenting-path search is sequential over rows, its column scan is spread over the 64 lanes and the
// sequential tie-break of the scalar scan is reproduced exactly:
//   index = last unassigned column (in `remaining` order) among the minimum, else the first minimum.
// Arithmetic is fp64 in the same order as scipy (minVal + c - u[i] - v[j]).  Tall matrices are solved
// transposed.  row_out/col_out: min(nr,nc) pairs sorted by row.
__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(64) void lsap_kernel(const float *cost, int nr0, int nc0, int *row_out, int *col_out, int cost_in_lds) {
    const int lane = threadIdx.x;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    // all solver state lives in LDS; when it fits, so does the cost matrix (the augmenting-path scan is a chain of
    // dependent lookups: one L2 round trip per Dijkstra step was most of the kernel's time)
    extern __shared__ __attribute__((aligned(16))) char lsap_smem[];
    const int n = max(nr, nc);
    double *u = reinterpret_cast<double *>(lsap_smem), *v = u + n, *spc = v + n;
    int *path = reinterpret_cast<int *>(spc + n), *col4row = path + n, *row4col = col4row + n, *remaining = row4col + n, *SR = remaining + n,
        *SC = SR + n;
    float *cost_lds = reinterpret_cast<float *>(SC + n);
    if (cost_in_lds) {
        for (int i = lane; i < nr0 * nc0; i += 64) cost_lds[i] = cost[i];
        __syncthreads();
    }
    const float *cm = cost_in_lds ? cost_lds : cost;
    auto C = [&](int i, int j) -> double { return (double)(transpose ? cm[(size_t)j * nc0 + i] : cm[(size_t)i * nc0 + j]); };
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    __syncthreads();
    for (int cur = 0; cur < nr; ++cur) {
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        for (int j = lane; j < nc; j += 64) { SC[j] = 0; spc[j] = INFINITY; remaining[j] = nc - j - 1; }
        __syncthreads();
        double minVal = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            // one pass, one reduction: the candidate is the lexicographic minimum of (shortest path cost, key) where the key
            // encodes scipy's tie-break - among equal costs the LAST unassigned column in `remaining` order, otherwise the
            // FIRST column: unassigned -> 0x3fffffff - it (larger it = smaller key), assigned -> 0x40000000 + it
            double lmin = INFINITY;
            int lkey = 0x7fffffff;
            for (int it = lane; it < num_remaining; it += 64) {
                int j = remaining[it];
                double r = minVal + C(i, j) - ui - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const int key = row4col[j] == -1 ? 0x3fffffff - it : 0x40000000 + it;
                if (s < lmin || (s == lmin && key < lkey)) { lmin = s; lkey = key; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double os = __shfl_xor(lmin, o, 64);
                const int ok = __shfl_xor(lkey, o, 64);
                if (os < lmin || (os == lmin && ok < lkey)) { lmin = os; lkey = ok; }
            }
            const double lowest = lmin;
            const int index = lkey < 0x40000000 ? 0x3fffffff - lkey : lkey - 0x40000000;
            minVal = lowest;
            const int j = remaining[index];
            const int owner = row4col[j];
            if (owner == -1) sink = j; else i = owner;
            __syncthreads();
            if (lane == 0) {
                SC[j] = 1;
                remaining[index] = remaining[num_remaining - 1];
            }
            --num_remaining;
            __syncthreads();
        }
        // dual update
        for (int r = lane; r < nr; r += 64) {
            if (r == cur) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc[col4row[r]];
        }
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
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
    if (lane == 0) {
        if (transpose) {
            int k = 0;
            for (int r = 0; r < nc; ++r) {          // nc == original row count
                int who = row4col[r];
                if (who >= 0) { row_out[k] = r; col_out[k] = who; ++k; }
            }
        } else {
            for (int r = 0; r < nr; ++r) { row_out[r] = r; col_out[r] = col4row[r]; }
        }
    }
}

static void launch_lsap(const float *cost_dev, int nr, int nc, int *rows_dev, int *cols_dev, hipStream_t s) {
    const int n = nr > nc ? nr : nc;
    size_t state = (size_t)n * (3 * sizeof(double) + 6 * sizeof(int)), with_cost = state + (size_t)nr * nc * sizeof(float);
    const bool in_lds = with_cost <= 150 * 1024;
    const size_t smem = in_lds ? with_cost : state;
    static bool attr_set = false;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(lsap_kernel, dim3(1), dim3(64), smem, s, cost_dev, nr, nc, rows_dev, cols_dev, in_lds ? 1 : 0);
    YDS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ misc
// appends the (already normalised) detection feature det_idx[t] to the gallery ring of slot slots[t]
__global__ void feature_append_kernel(float *gallery_n, int budget, const int *slots, const int *pos, const float *feats_n, const int *det_idx,
                                      int n) {
    int t = blockIdx.x;
    if (t >= n) return;
    float *dst = gallery_n + ((size_t)slots[t] * budget + pos[t]) * EMB;
    const float *src = feats_n + (size_t)det_idx[t] * EMB;
    for (int k = threadIdx.x; k < EMB; k += blockDim.x) dst[k] = src[k];
}

// deep_sort.py:73-87: w = a*h; xy -= wh/2; x2y2 = wh + xy; x1y1 = max(.,0); int32 truncation
__global__ void output_kernel(const float *mean, const int *slots, const int *ids, const float *payload, int n, int *out6) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float *m = mean + (size_t)slots[t] * 8;
    float w = m[2] * m[3], h = m[3];
    float x = m[0] - w / 2.f, y = m[1] - h / 2.f;
    float x2 = w + x, y2 = h + y;
    x = fmaxf(x, 0.f); y = fmaxf(y, 0.f);
    int *o = out6 + t * 6;
    o[0] = (int)x; o[1] = (int)y; o[2] = (int)x2; o[3] = (int)y2; o[4] = ids[t]; o[5] = (int)payload[t];
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

// ============================================================================================ host
enum { TENTATIVE = 1, CONFIRMED = 2, DELETED = 3 };
enum { METRIC_COSINE = 0, METRIC_EUCLIDEAN = 1 };

struct Track {
    int slot, id, hits = 1, age = 1, tsu = 0, state = TENTATIVE;
    int n_feat = 0, head = 0;        // gallery ring fill / next write position
    float payload = 0.f;
};

class Tracker : public TrackerIface {
public:
    // budget <= 0: nn_budget=None, every track keeps all its features (nn_matching.py:152-154) - the per-track row
    // capacity `budget` then doubles whenever a gallery fills up; metric: cosine | euclidean (nn_matching.py:128-134)
    Tracker(double max_dist, double max_iou, int max_age, int n_init, int budget, int metric = METRIC_COSINE)
        : max_dist(max_dist), max_iou(max_iou), max_age(max_age), n_init(n_init), budget(budget > 0 ? budget : 32), unbounded(budget <= 0),
          metric(metric) {
        if (metric != METRIC_COSINE && metric != METRIC_EUCLIDEAN) fail("Invalid metric; must be either 'euclidean' or 'cosine'");
        YDS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        grow(256);
        YDS_HIP(hipHostMalloc((void **)&ibuf_host, IBUF_INTS * sizeof(int), hipHostMallocMapped));
        YDS_HIP(hipHostGetDevicePointer((void **)&ibuf_dev, ibuf_host, 0));
        lsap_rows.alloc(LSAP_MAX);
        lsap_cols.alloc(LSAP_MAX);
    }
    ~Tracker() override {
        if (ibuf_host) (void)hipHostFree(ibuf_host);
        if (stream) (void)hipStreamDestroy(stream);
    }
    int num_tracks() const override { return (int)tracks.size(); }

    void grow(int cap) {
        DevBuf<float> m((size_t)cap * 8), c((size_t)cap * 64), g((size_t)cap * budget * EMB);
        if (capacity) {
            YDS_HIP(hipMemcpyAsync(m.p, mean.p, (size_t)capacity * 8 * 4, hipMemcpyDeviceToDevice, stream));
            YDS_HIP(hipMemcpyAsync(c.p, cov.p, (size_t)capacity * 64 * 4, hipMemcpyDeviceToDevice, stream));
            YDS_HIP(hipMemcpyAsync(g.p, gallery.p, (size_t)capacity * budget * EMB * 4, hipMemcpyDeviceToDevice, stream));
            YDS_HIP(hipStreamSynchronize(stream));
        }
        mean = std::move(m); cov = std::move(c); gallery = std::move(g);
        for (int s = cap - 1; s >= capacity; --s) free_slots.push_back(s);
        capacity = cap;
    }
    // nn_budget=None: double the per-track row capacity, keeping every slot's rows
    void grow_budget() {
        const int nb = budget * 2;
        DevBuf<float> g((size_t)capacity * nb * EMB);
        YDS_HIP(hipMemcpy2DAsync(g.p, (size_t)nb * EMB * 4, gallery.p, (size_t)budget * EMB * 4, (size_t)budget * EMB * 4, capacity,
                                 hipMemcpyDeviceToDevice, stream));
        YDS_HIP(hipStreamSynchronize(stream));
        gallery = std::move(g);
        budget = nb;
    }
    // ring position for the next feature of a track (tracker.py:165-176 + nn_matching.py:152-155: last `budget` rows)
    int next_row(Track &t) {
        if (unbounded) {
            if (t.n_feat == budget) grow_budget();
            return t.n_feat++;
        }
        const int pos = t.head;
        t.head = (t.head + 1) % budget;
        t.n_feat = std::min(t.n_feat + 1, budget);
        return pos;
    }

    // uploads an int vector into a scratch region and returns the device pointer
    // Small index lists go through a pinned, device-mapped host ring: the kernels read them in place (a few hundred
    // bytes over the host link) instead of paying one hipMemcpyAsync per list.  The ring is rewound once per step,
    // after the step's final stream synchronisation.
    const int *up(const std::vector<int> &v) {
        if (v.empty()) return nullptr;
        if (ibuf_used + v.size() > IBUF_INTS) fail("tracker: index scratch exhausted");
        int *h = ibuf_host + ibuf_used;
        memcpy(h, v.data(), v.size() * sizeof(int));
        const int *d = ibuf_dev + ibuf_used;
        ibuf_used += (v.size() + 3) / 4 * 4;
        return d;
    }

    struct Assignment { std::vector<int> rows, cols; std::vector<float> cost; };

    void solve(const float *cost_dev, int nr, int nc, Assignment &a) {
        if (nr > LSAP_MAX || nc > LSAP_MAX) fail("tracker: assignment problem %dx%d exceeds %d", nr, nc, LSAP_MAX);
        int n = std::min(nr, nc);
        launch_lsap(cost_dev, nr, nc, lsap_rows.p, lsap_cols.p, stream);
        a.rows.resize(n); a.cols.resize(n); a.cost.resize((size_t)nr * nc);
        YDS_HIP(hipMemcpyAsync(a.rows.data(), lsap_rows.p, n * sizeof(int), hipMemcpyDeviceToHost, stream));
        YDS_HIP(hipMemcpyAsync(a.cols.data(), lsap_cols.p, n * sizeof(int), hipMemcpyDeviceToHost, stream));
        YDS_HIP(hipMemcpyAsync(a.cost.data(), cost_dev, a.cost.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        YDS_HIP(hipStreamSynchronize(stream));
    }

    // linear_assignment.py:58-72 list bookkeeping
    void bookkeeping(const Assignment &a, int nc, float max_distance, const std::vector<int> &track_idx, const std::vector<int> &det_idx,
                     std::vector<std::pair<int, int>> &matches, std::vector<int> &um_t, std::vector<int> &um_d) {
        std::vector<char> col_used(det_idx.size(), 0), row_used(track_idx.size(), 0);
        for (size_t k = 0; k < a.rows.size(); ++k) { row_used[a.rows[k]] = 1; col_used[a.cols[k]] = 1; }
        for (size_t c = 0; c < det_idx.size(); ++c) if (!col_used[c]) um_d.push_back(det_idx[c]);
        for (size_t r = 0; r < track_idx.size(); ++r) if (!row_used[r]) um_t.push_back(track_idx[r]);
        for (size_t k = 0; k < a.rows.size(); ++k) {
            int r = a.rows[k], c = a.cols[k];
            if (a.cost[(size_t)r * nc + c] > max_distance) { um_t.push_back(track_idx[r]); um_d.push_back(det_idx[c]); }
            else matches.emplace_back(track_idx[r], det_idx[c]);
        }
    }

    int step(const float *tlwh_host, const float *feats, bool feats_on_device, const float *payload, int D, int32_t *out6, int cap) override {
        return step_sel(tlwh_host, feats, feats_on_device, nullptr, payload, D, out6, cap);
    }
    // feat_rows (optional): detection d uses row feat_rows[d] of `feats` (tracker-side NMS keeps a subset in pick order)
    int step_sel(const float *tlwh_host, const float *feats, bool feats_on_device, const int *feat_rows, const float *payload, int D,
                 int32_t *out6, int cap) {
        ibuf_used = 0;
        const int T = (int)tracks.size();
        tlwh_dev.ensure((size_t)std::max(D, 1) * 4);
        if (D) YDS_HIP(hipMemcpyAsync(tlwh_dev.p, tlwh_host, (size_t)D * 16, hipMemcpyHostToDevice, stream));
        const float *feats_dev = feats;
        if (!feats_on_device && D) {
            int n_rows = D;
            if (feat_rows) for (int d = 0; d < D; ++d) n_rows = std::max(n_rows, feat_rows[d] + 1);
            feats_stage.ensure((size_t)n_rows * EMB);
            YDS_HIP(hipMemcpyAsync(feats_stage.p, feats, (size_t)n_rows * EMB * 4, hipMemcpyHostToDevice, stream));
            feats_dev = feats_stage.p;
        }
        if (D) {                                              // x / ||x|| once per frame (nn_matching.py:50-52); euclidean: as is
            feats_n.ensure((size_t)D * EMB);
            const int *rows_dev = nullptr;
            if (feat_rows) { std::vector<int> r(feat_rows, feat_rows + D); rows_dev = up(r); }
            hipLaunchKernelGGL(normalize_rows_kernel, dim3((D + 3) / 4), dim3(256), 0, stream, feats_dev, rows_dev, feats_n.p, D,
                               metric == METRIC_COSINE ? 1 : 0);
        }
        // ---- Tracker.predict (tracker.py:95-113)
        if (T) {
            std::vector<int> slots(T);
            for (int i = 0; i < T; ++i) slots[i] = tracks[i].slot;
            hipLaunchKernelGGL(kf_predict_kernel, dim3((T + 63) / 64), dim3(64), 0, stream, mean.p, cov.p, up(slots), T);
            for (Track &t : tracks) { t.age++; t.tsu++; }
        }
        // ---- Tracker._match (tracker.py:56-93)
        std::vector<int> confirmed, unconfirmed, all_dets(D);
        for (int i = 0; i < T; ++i) (tracks[i].state == CONFIRMED ? confirmed : unconfirmed).push_back(i);
        for (int d = 0; d < D; ++d) all_dets[d] = d;
        std::vector<std::pair<int, int>> matches;
        std::vector<int> um_t_a, um_d;
        if (D == 0 || confirmed.empty()) {
            um_t_a = confirmed;
            um_d = all_dets;
        } else {
            const int Tc = (int)confirmed.size();
            std::vector<int> slots(Tc), rows(Tc);
            for (int r = 0; r < Tc; ++r) { slots[r] = tracks[confirmed[r]].slot; rows[r] = tracks[confirmed[r]].n_feat; }
            cost_dev.ensure((size_t)Tc * D);
            hipLaunchKernelGGL(appearance_cost_kernel, dim3(Tc, (D + 15) / 16), dim3(256), 0, stream, gallery.p, up(slots), up(rows), budget,
                               feats_n.p, D, mean.p, cov.p, tlwh_dev.p, (float)max_dist, (float)(max_dist + 1e-5), 1, metric == METRIC_EUCLIDEAN ? 1 : 0, cost_dev.p);
            Assignment a;
            solve(cost_dev.p, Tc, D, a);
            bookkeeping(a, D, (float)max_dist, confirmed, all_dets, matches, um_t_a, um_d);
        }
        std::vector<int> iou_cand = unconfirmed, um_t_keep;
        for (int k : um_t_a) (tracks[k].tsu == 1 ? iou_cand : um_t_keep).push_back(k);
        std::vector<int> um_t_b;
        if (um_d.empty() || iou_cand.empty()) {
            um_t_b = iou_cand;
        } else {
            const int Tb = (int)iou_cand.size(), Db = (int)um_d.size();
            std::vector<int> slots(Tb), stale(Tb);
            for (int r = 0; r < Tb; ++r) { slots[r] = tracks[iou_cand[r]].slot; stale[r] = tracks[iou_cand[r]].tsu > 1; }
            cost_dev.ensure((size_t)Tb * Db);
            hipLaunchKernelGGL(iou_cost_kernel, dim3((Tb * Db + 255) / 256), dim3(256), 0, stream, mean.p, up(slots), up(stale), Tb, tlwh_dev.p,
                               up(um_d), Db, (float)max_iou, (float)(max_iou + 1e-5), cost_dev.p);
            Assignment a;
            solve(cost_dev.p, Tb, Db, a);
            std::vector<int> um_d2;
            bookkeeping(a, Db, (float)max_iou, iou_cand, um_d, matches, um_t_b, um_d2);
            um_d = um_d2;
        }
        std::vector<int> unmatched_tracks = um_t_keep;
        unmatched_tracks.insert(unmatched_tracks.end(), um_t_b.begin(), um_t_b.end());
        last_matches = matches;
        last_um_t = unmatched_tracks;
        std::sort(last_um_t.begin(), last_um_t.end());
        last_um_d = um_d;
        // ---- Tracker.update (tracker.py:129-176)
        const int M = (int)matches.size();
        if (M) {
            std::vector<int> slots(M), dets(M), pos(M);
            for (int k = 0; k < M; ++k) {
                Track &t = tracks[matches[k].first];
                slots[k] = t.slot; dets[k] = matches[k].second; pos[k] = next_row(t);
                t.hits++; t.tsu = 0;
                if (t.state == TENTATIVE && t.hits >= n_init) t.state = CONFIRMED;
                t.payload = payload[matches[k].second];
            }
            z_dev.ensure((size_t)M * 4);
            const int *dslots = up(slots), *ddets = up(dets);
            hipLaunchKernelGGL(tlwh_to_xyah_kernel, dim3((M + 63) / 64), dim3(64), 0, stream, tlwh_dev.p, ddets, z_dev.p, M);
            hipLaunchKernelGGL(kf_update_kernel, dim3((M + 63) / 64), dim3(64), 0, stream, mean.p, cov.p, dslots, z_dev.p, M);
            hipLaunchKernelGGL(feature_append_kernel, dim3(M), dim3(128), 0, stream, gallery.p, budget, dslots, up(pos), feats_n.p, ddets, M);
        }
        for (int k : unmatched_tracks) {                                   // Track.mark_missed track.py:146-152
            Track &t = tracks[k];
            if (t.state == TENTATIVE) t.state = DELETED;
            else if (t.tsu > max_age) t.state = DELETED;
        }
        const int Nn = (int)um_d.size();
        if (Nn) {                                                           // Tracker._initiate_track tracker.py:49-54
            while ((int)free_slots.size() < Nn) grow(capacity * 2);
            std::vector<int> slots(Nn), pos(Nn, 0);
            for (int k = 0; k < Nn; ++k) {
                Track t;
                t.slot = free_slots.back(); free_slots.pop_back();
                t.id = next_id++;
                t.n_feat = 1; t.head = 1 % budget;
                t.payload = payload[um_d[k]];
                slots[k] = t.slot;
                tracks.push_back(t);
            }
            const int *dslots = up(slots), *ddets = up(um_d);
            hipLaunchKernelGGL(kf_initiate_kernel, dim3((Nn + 63) / 64), dim3(64), 0, stream, mean.p, cov.p, dslots, tlwh_dev.p, ddets, Nn);
            hipLaunchKernelGGL(feature_append_kernel, dim3(Nn), dim3(128), 0, stream, gallery.p, budget, dslots, up(pos), feats_n.p, ddets, Nn);
        }
        std::vector<Track> alive;
        for (const Track &t : tracks) {
            if (t.state == DELETED) free_slots.push_back(t.slot);
            else alive.push_back(t);
        }
        tracks.swap(alive);
        // ---- output stage (deep_sort.py:63-88)
        std::vector<int> slots, ids;
        std::vector<float> pl;
        for (const Track &t : tracks)
            if (t.state == CONFIRMED && t.tsu <= 1) { slots.push_back(t.slot); ids.push_back(t.id); pl.push_back(t.payload); }
        int m = (int)slots.size();
        if (m > cap) fail("tracker: %d output rows exceed the caller's capacity %d", m, cap);
        if (m) {
            out_dev.ensure((size_t)m * 6);
            pl_dev.upload(pl.data(), pl.size(), stream);
            hipLaunchKernelGGL(output_kernel, dim3((m + 63) / 64), dim3(64), 0, stream, mean.p, up(slots), up(ids), pl_dev.p, m, out_dev.p);
            YDS_HIP(hipMemcpyAsync(out6, out_dev.p, (size_t)m * 6 * sizeof(int), hipMemcpyDeviceToHost, stream));
        }
        YDS_HIP(hipGetLastError());
        YDS_HIP(hipStreamSynchronize(stream));
        return m;
    }

    double max_dist, max_iou;       // python floats in the reference; fp32 roundings are taken where torch/numpy take them
    int max_age, n_init, budget;
    bool unbounded = false;
    int metric = METRIC_COSINE;
    int capacity = 0, next_id = 1;
    std::vector<Track> tracks;
    std::vector<int> free_slots;
    DevBuf<float> mean, cov, gallery /* rows stored normalised */, tlwh_dev, feats_stage, feats_n, cost_dev, z_dev, pl_dev;
    DevBuf<int> lsap_rows, lsap_cols, out_dev;
    static constexpr size_t IBUF_INTS = 1 << 16;
    int *ibuf_host = nullptr, *ibuf_dev = nullptr;
    size_t ibuf_used = 0;
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
    int n = (int)k->tracks.size();
    if (n > cap) yds::fail("tracker: %d tracks exceed cap %d", n, cap);
    *T = n;
    for (int i = 0; i < n; ++i) {
        const yds::Track &tr = k->tracks[i];
        if (ids) ids[i] = tr.id;
        if (state) state[i] = tr.state;
        if (tsu) tsu[i] = tr.tsu;
        if (hits) hits[i] = tr.hits;
        if (mean8) YDS_HIP(hipMemcpy(mean8 + (size_t)i * 8, k->mean.p + (size_t)tr.slot * 8, 32, hipMemcpyDeviceToHost));
        if (cov64) YDS_HIP(hipMemcpy(cov64 + (size_t)i * 64, k->cov.p + (size_t)tr.slot * 64, 256, hipMemcpyDeviceToHost));
    }
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
    if (nr > LSAP_MAX || nc > LSAP_MAX) fail("lsap: %dx%d exceeds %d", nr, nc, LSAP_MAX);
    int n = std::min(nr, nc);
    *n_out = n;
    if (n == 0) return 0;
    hipStream_t s = g_scratch.stream();
    DevBuf<float> c; c.upload(cost_host, (size_t)nr * nc, s);
    DevBuf<int> r(n), cc(n);
    launch_lsap(c.p, nr, nc, r.p, cc.p, s);
    YDS_HIP(hipMemcpyAsync(rows, r.p, n * sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipMemcpyAsync(cols, cc.p, n * sizeof(int), hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
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
    hipLaunchKernelGGL(iou_cost_kernel, dim3((T * D + 255) / 256), dim3(256), 0, s, m.p, sl.p, (const int *)nullptr, T, dt.p, di.p, D, 0.f, 0.f, o.p);
    YDS_HIP(hipMemcpyAsync(out, o.p, (size_t)T * D * 4, hipMemcpyDeviceToHost, s));
    YDS_HIP(hipStreamSynchronize(s));
    YDS_API_END
}
static int nn_min_cost(const float *gallery_host, const int32_t *seg_offsets_host, int T, const float *feats_host, int D, int dim, float *out,
                       int euclid) {
    YDS_API_BEGIN
    using namespace yds;
    if (dim != EMB) fail("cosine: feature dimension must be %d", EMB);
    if (T == 0 || D == 0) return 0;
    hipStream_t s = g_scratch.stream();
    int budget = 1;
    for (int t = 0; t < T; ++t) budget = std::max(budget, seg_offsets_host[t + 1] - seg_offsets_host[t]);
    std::vector<float> g((size_t)T * budget * EMB, 0.f);
    std::vector<int> rows(T);
    for (int t = 0; t < T; ++t) {
        rows[t] = seg_offsets_host[t + 1] - seg_offsets_host[t];
        memcpy(&g[(size_t)t * budget * EMB], gallery_host + (size_t)seg_offsets_host[t] * EMB, (size_t)rows[t] * EMB * 4);
    }
    DevBuf<float> gd, fd, gn(g.size()), fn((size_t)D * EMB), o((size_t)T * D); DevBuf<int> sl, nr;
    gd.upload(g.data(), g.size(), s); fd.upload(feats_host, (size_t)D * EMB, s);
    auto v = iota(T); sl.upload(v.data(), T, s); nr.upload(rows.data(), T, s);
    const int G = T * budget;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((G + 3) / 4), dim3(256), 0, s, gd.p, (const int *)nullptr, gn.p, G, euclid ? 0 : 1);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((D + 3) / 4), dim3(256), 0, s, fd.p, (const int *)nullptr, fn.p, D, euclid ? 0 : 1);
    hipLaunchKernelGGL(appearance_cost_kernel, dim3(T, (D + 15) / 16), dim3(256), 0, s, gn.p, sl.p, nr.p, budget, fn.p, D,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.f, 0.f, 0, euclid, o.p);
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
