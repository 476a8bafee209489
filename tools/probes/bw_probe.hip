// HBM bandwidth probe (tuning aid): write-only, read-only and copy streams with 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void wr(float4 *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1, 2, 3, 4); }
__global__ void rd(const float4 *p, size_t n, float *o) { float s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; } if (s == 12345.f) *o = s; }
__global__ void cp(const float4 *a, float4 *b, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }
int main() {
    for (size_t mb : {64, 256, 757, 2048}) {
        size_t n = mb * 1024 * 1024 / 16;
        float4 *a, *b; float *o;
        hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, 4);
        hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int which = 0; which < 3; ++which) {
            for (int g : {2048, 8192, 65536}) {
                float best = 1e9;
                for (int r = 0; r < 5; ++r) {
                    hipEventRecord(e0);
                    if (which == 0) wr<<<g, 256>>>(a, n); else if (which == 1) rd<<<g, 256>>>(a, n, o); else cp<<<g, 256>>>(a, b, n);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                double bytes = (double)n * 16 * (which == 2 ? 2 : 1);
                printf("%5zu MB %s grid %6d: %8.1f us  %6.2f TB/s\n", mb, which == 0 ? "write" : which == 1 ? "read " : "copy ", g, best * 1e3, bytes / best / 1e9);
            }
        }
        hipFree(a); hipFree(b); hipFree(o);
    }
    return 0;
}
