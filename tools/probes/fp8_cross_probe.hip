// Power-limited throughput of two ways to do the f16x3 arithmetic (tuning aid):
//   MODE 0: three v_mfma_f32_32x32x16_f16 per k16 block (hi x hi, hi x lo, lo x hi) - what the kernels do
//   MODE 1: hi x hi on the fp16 pipe, both cross terms of TWO k16 blocks as ONE v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3,
//           K = 64 = [xh8 | xl8] . [wl8 ; wh8]) - the "fp8 cross terms" idea: 2 + 1 MFMA instructions instead of 6 per 32 channels
// Registers only (no LDS, no memory): the matrix pipe under the chip's power limit, random operand bits.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int steps, unsigned seed, unsigned lo_mask) {
    const int tid = threadIdx.x;
    unsigned h = (tid + 977u * blockIdx.x + seed) * 2654435761u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return h; };
    h8 ah[2], al[2], bh[2], bl[2];
    i8v a8[2], b8[2];
    for (int i = 0; i < 2; ++i) {
        union { h8 v; unsigned u[4]; } t;
        for (int k = 0; k < 4; ++k) t.u[k] = ((rnd() >> 3) & 0x03ff03ffu) | 0x38003800u | (rnd() & 0x80008000u);
        ah[i] = t.v;
        for (int k = 0; k < 4; ++k) t.u[k] = ((rnd() >> 3) & 0x03ff03ffu) | 0x38003800u | (rnd() & 0x80008000u);
        for (int k = 0; k < 4; ++k) t.u[k] &= lo_mask;          // MODE 0 experiment: lo halves with their low mantissa bits cleared
        al[i] = t.v;
        for (int k = 0; k < 4; ++k) t.u[k] = ((rnd() >> 3) & 0x03ff03ffu) | 0x38003800u | (rnd() & 0x80008000u);
        bh[i] = t.v;
        for (int k = 0; k < 4; ++k) t.u[k] = ((rnd() >> 3) & 0x03ff03ffu) | 0x38003800u | (rnd() & 0x80008000u);
        for (int k = 0; k < 4; ++k) t.u[k] &= lo_mask;
        bl[i] = t.v;
        for (int k = 0; k < 8; ++k) { a8[i][k] = (int)((rnd() & 0x87878787u) | 0x30303030u); b8[i][k] = (int)((rnd() & 0x87878787u) | 0x30303030u); }   // e4m3 in [0.5, 2), random signs
    }
    f32x16 acc1[2][2], acc2[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) { acc1[i][j][e] = 0; acc2[i][j][e] = 0; }
    for (int s = 0; s < steps; ++s) {
        // one step = 32 original channels (two k16 blocks) of a 64x64 wave tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc1[i][j], 0, 0, 0);
                    if (MODE == 0) {
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
                    }
                }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc2[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc2[i][j], 0, 0, 0, 127, 0, 127);
        }
        // rotate operands a little so that the compiler cannot hoist and the data keeps toggling
        ah[0][0] = (_Float16)((float)ah[0][0] * 1.0009765625f); bh[1][3] = (_Float16)((float)bh[1][3] * 0.9990234375f);
    }
    float t = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) t += acc1[i][j][e] + acc2[i][j][e];
    if (t == 12345.678f) out[0] = t;
}

template <int MODE> double run(int blocks, int steps, unsigned lo_mask = 0xffffffffu) {
    float *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, steps / 8, 1u, lo_mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, steps, 7u + r, lo_mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms / 5;
}
int main() {
    const int blocks = 512, steps = 200000;
    // fp32-equivalent MACs per step per wave: 64 x 64 x 32
    const double macs = (double)blocks * 4 * steps * 64.0 * 64.0 * 32.0;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = run<0>(blocks, steps), t1 = run<1>(blocks, steps);
        printf("3 x fp16 MFMA per k16        : %8.2f ms  %7.1f TFLOP/s fp32-equivalent\n", t0, 2 * macs / t0 / 1e9);
        printf("fp16 hi x hi + fp8 cross terms: %8.2f ms  %7.1f TFLOP/s fp32-equivalent   (x%.2f)\n", t1, 2 * macs / t1 / 1e9, t0 / t1);
    }
    // does the energy of an fp16 MFMA depend on how many mantissa bits of an operand are populated?  (lo halves truncated to n bits)
    for (int bits = 10; bits >= 0; bits -= 2) {
        const unsigned m16 = 0xffffu & ~((1u << (10 - bits)) - 1u), mask = m16 | (m16 << 16);
        double t = run<0>(blocks, steps, mask);
        printf("3 x fp16 MFMA, lo halves with %2d mantissa bits: %8.2f ms  %7.1f TFLOP/s\n", bits, t, 2 * macs / t / 1e9);
    }
    return 0;
}
