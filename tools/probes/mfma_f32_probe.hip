// fp32-input matrix pipe probe: v_mfma_f32_32x32x2_f32 (and the 16x16x4 form) alone, sustained launches, random / flat operands.
// Reports TFLOP/s against the 157.3 TFLOP/s peak of MI355X_MICROARCH.md (64 FLOP/clk/SIMD at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int steps, int random_data) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        unsigned h = (tid * 16 + i) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        a[i] = random_data ? (float)(h & 0xffff) / 65536.f - 0.5f : 0.001f;
        b[i] = random_data ? (float)(h >> 16) / 65536.f - 0.5f : 0.002f;
    }
    float s = 0;
    if (MODE == 0) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
        for (int k = 0; k < steps; ++k) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * 4 + c], b[j * 4 + c], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0;
        for (int k = 0; k < steps; ++k) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i * 2 + c], b[j * 2 + c], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    if (s == 123.456f) out[0] = s;
}
template <int MODE> void run(const char *name, int wg, int rnd, int steps) {
    float *o; hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); probe<MODE><<<256 * wg, 256>>>(o, steps, rnd); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
    }
    // MODE 0: 16 MFMAs of 4096 flops per step; MODE 1: 32 MFMAs of 2048 flops
    double flops = (double)256 * wg * 4 * steps * 65536.0;
    printf("%-28s %d WG/CU %s: %9.1f us %7.1f TFLOP/s (%.3f of 157.3)\n", name, wg, rnd ? "random" : "flat  ", best * 1e3, flops / best / 1e9, flops / best / 1e9 / 157.3);
    hipFree(o);
}
int main() {
    for (int rnd : {1, 0}) for (int wg : {1, 2}) { run<0>("v_mfma_f32_32x32x2_f32", wg, rnd, 200000); run<1>("v_mfma_f32_16x16x4_f32", wg, rnd, 200000); }
    return 0;
}
