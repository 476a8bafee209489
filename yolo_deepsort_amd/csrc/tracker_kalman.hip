// Stand-alone Kalman kernels of the tracker (deep_sort/sort/kalman_filter.py:54-256): one thread per track / (track, detection) pair;
// the step bodies are tracker_dev.h's (the lifecycle kernels of tracker.hip call the same ones).
#include "tracker_dev.h"

namespace yds {

__global__ void kf_predict_kernel(float *mean, float *cov, const int *slots, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    kf_predict_body(mean + (size_t)slots[t] * 8, cov + (size_t)slots[t] * 64);
}

__global__ void kf_update_kernel(float *mean, float *cov, const int *slots, const float *z, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    kf_update_body(mean + (size_t)slots[t] * 8, cov + (size_t)slots[t] * 64, z + t * 4);
}

__global__ void kf_initiate_kernel(float *mean, float *cov, const int *slots, const float *tlwh, const int *det_idx, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    kf_initiate_body(mean + (size_t)slots[t] * 8, cov + (size_t)slots[t] * 64, tlwh + (size_t)det_idx[t] * 4);
}

__global__ void tlwh_to_xyah_kernel(const float *tlwh, const int *det_idx, float *z, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    to_xyah(tlwh + (size_t)det_idx[t] * 4, z + t * 4);
}

__global__ void gating_kernel(const float *mean, const float *cov, const int *slots, int T, const float *xyah, int D, float *out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * D) return;
    int t = idx / D, d = idx - t * D;
    out[idx] = gate2(mean + (size_t)slots[t] * 8, cov + (size_t)slots[t] * 64, xyah + d * 4);
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

}  // namespace yds
