// Matrix-pipe energy probe (tuning aid, round 4): the chip is power limited under the f16x3 kernels, so what counts is
// joules per multiply-accumulate.  Same 64x64 wave tile, same three-term product, sustained launches on random operands:
//   MODE 0  v_mfma_f32_32x32x16_f16, block order (i, j) row-major               (what the kernels do)
//   MODE 1  v_mfma_f32_32x32x16_f16, snake order, term-major (one operand changes between consecutive MFMAs)
//   MODE 2  v_mfma_f32_16x16x32_f16 (4x4 blocks: a quarter of the accumulator read/write per flop, twice the operand reads)
//   MODE 3  v_mfma_f32_16x16x32_f16, snake order
// Reports TFLOP/s of MFMA work at the sustained (power limited) clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline h8 rnd_h8(unsigned seed) {
    union { h8 v; unsigned u[4]; } r;
    for (int i = 0; i < 4; ++i) {
        unsigned h = (seed + 0x9e3779b9u * (i + 1)) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        r.u[i] = ((h >> 3) & 0x03ff03ffu) | 0x38003800u | (h & 0x80008000u);       // two fp16 in [0.5, 1), random signs
    }
    return r.v;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int steps, int random_data) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    float s = 0;
    if (MODE < 2) {
        f32x16 acc1[2][2], acc2[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) { acc1[i][j][e] = 0; acc2[i][j][e] = 0; }
        h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        for (int q = 0; q < 2; ++q) for (int i = 0; i < 2; ++i) {
            unsigned b = random_data ? tid * 64 + q * 8 + i * 4 : 7;
            ah[q][i] = rnd_h8(b); al[q][i] = rnd_h8(b + 1); bh[q][i] = rnd_h8(b + 2); bl[q][i] = rnd_h8(b + 3);
        }
        for (int k = 0; k < steps; ++k) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][i], bh[q][j], acc1[i][j], 0, 0, 0);
                            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][i], bl[q][j], acc2[i][j], 0, 0, 0);
                            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q][i], bh[q][j], acc2[i][j], 0, 0, 0);
                        }
                } else {
                    // term-major, snake: (a0,b0) (a0,b1) (a1,b1) (a1,b0) per term - one operand register changes per MFMA
                    const int si[4] = {0, 0, 1, 1}, sj[4] = {0, 1, 1, 0};
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc1[si[t]][sj[t]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][si[t]], bh[q][sj[t]], acc1[si[t]][sj[t]], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc2[si[3 - t]][sj[3 - t]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][si[3 - t]], bl[q][sj[3 - t]], acc2[si[3 - t]][sj[3 - t]], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc2[si[t]][sj[t]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q][si[t]], bh[q][sj[t]], acc2[si[t]][sj[t]], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc1[i][j][e] + acc2[i][j][e];
    } else {
        f32x4 acc1[4][4], acc2[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) { acc1[i][j][e] = 0; acc2[i][j][e] = 0; }
        h8 ah[4], al[4], bh[4], bl[4];          // one K = 32 step: 4 row blocks, 4 column blocks
        for (int i = 0; i < 4; ++i) {
            unsigned b = random_data ? tid * 64 + i * 4 : 7;
            ah[i] = rnd_h8(b); al[i] = rnd_h8(b + 1); bh[i] = rnd_h8(b + 2); bl[i] = rnd_h8(b + 3);
        }
        for (int k = 0; k < steps; ++k) {
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc1[i][j], 0, 0, 0);
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = (i & 1) ? 3 - jj : jj;
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc1[i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 3; i >= 0; --i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = (i & 1) ? 3 - jj : jj;
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = (i & 1) ? 3 - jj : jj;
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc1[i][j][e] + acc2[i][j][e];
    }
    if (s == 123.456f) out[0] = s;
}


// MODE 4 / 5: the K loop of the f16x3 kernels (16 ds_read_b128 per K = 32 step from random LDS rows, reads of the next step's
// first half issued under this step's MFMAs, one barrier per step) with 32x32x16 (4) or 16x16x32 (5) MFMAs
constexpr int ROWB = 144;
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe_lds(float *out, int steps, int random_data) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 256 * ROWB / 4; i += 256) {
        unsigned h = (unsigned)(i + 977 * blockIdx.x) * 2654435761u;
        unsigned bits = ((h >> 3) & 0x03ff03ffu) | 0x38003800u | (h & 0x80008000u);
        if (random_data) reinterpret_cast<unsigned *>(lds)[i] = bits; else reinterpret_cast<float *>(lds)[i] = 0.001f * (i & 15);
    }
    __syncthreads();
    float s = 0;
    if (MODE == 4) {
        f32x16 acc1[2][2], acc2[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) { acc1[i][j][e] = 0; acc2[i][j][e] = 0; }
        const int frag = (lane & 31) * ROWB + (lane >> 5) * 16;
        const char *a_lds = lds + (wave >> 1) * 64 * ROWB + frag, *b_lds = lds + 128 * ROWB + (wave & 1) * 64 * ROWB + frag;
        h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        auto read = [&](int q, int buf) {
            const char *a = a_lds + buf * 256 * ROWB, *b = b_lds + buf * 256 * ROWB;
            for (int i = 0; i < 2; ++i) {
                ah[q][i] = *reinterpret_cast<const h8 *>(a + i * 32 * ROWB + q * 32);
                al[q][i] = *reinterpret_cast<const h8 *>(a + i * 32 * ROWB + q * 32 + 64);
                bh[q][i] = *reinterpret_cast<const h8 *>(b + i * 32 * ROWB + q * 32);
                bl[q][i] = *reinterpret_cast<const h8 *>(b + i * 32 * ROWB + q * 32 + 64);
            }
        };
        auto mma = [&](int q) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][i], bh[q][j], acc1[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][i], bl[q][j], acc2[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q][i], bh[q][j], acc2[i][j], 0, 0, 0);
                }
        };
        read(0, 0); read(1, 0);
        for (int k = 0; k < steps; ++k) {
            mma(0);
            __builtin_amdgcn_sched_barrier(0);
            read(0, (k + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            read(1, (k + 1) & 1);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc1[i][j][e] + acc2[i][j][e];
    } else {
        f32x4 acc1[4][4], acc2[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) { acc1[i][j][e] = 0; acc2[i][j][e] = 0; }
        const int frag = (lane & 15) * ROWB + (lane >> 4) * 16;
        const char *a_lds = lds + (wave >> 1) * 64 * ROWB + frag, *b_lds = lds + 128 * ROWB + (wave & 1) * 64 * ROWB + frag;
        h8 ah[4], al[4], bh[4], bl[4];
        // half q of the step's reads: row blocks 2q, 2q+1 of A and B
        auto read = [&](int q, int buf) {
            const char *a = a_lds + buf * 256 * ROWB, *b = b_lds + buf * 256 * ROWB;
            for (int i = 2 * q; i < 2 * q + 2; ++i) {
                ah[i] = *reinterpret_cast<const h8 *>(a + i * 16 * ROWB);
                al[i] = *reinterpret_cast<const h8 *>(a + i * 16 * ROWB + 64);
                bh[i] = *reinterpret_cast<const h8 *>(b + i * 16 * ROWB);
                bl[i] = *reinterpret_cast<const h8 *>(b + i * 16 * ROWB + 64);
            }
        };
        auto mma = [&](int i0, int i1, int j0, int j1) {
#pragma unroll
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int j = j0; j < j1; ++j) {
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc1[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
                }
        };
        read(0, 0); read(1, 0);
        for (int k = 0; k < steps; ++k) {
            // quadrant order: (A01 x B01) needs half 0 only; then the rest; half 0 of the next step is re-read once its last user is done
            mma(0, 2, 0, 2);
            mma(0, 2, 2, 4);
            mma(2, 4, 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            read(0, (k + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(2, 4, 2, 4);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            read(1, (k + 1) & 1);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc1[i][j][e] + acc2[i][j][e];
    }
    if (s == 123.456f) out[0] = s;
}

template <int MODE> void run_lds(const char *name, int wg_per_cu, int random_data, int steps) {
    float *o; hipMalloc(&o, 4);
    const int grid = 256 * wg_per_cu;
    size_t smem = 2 * 256 * ROWB;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_lds<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        probe_lds<MODE><<<grid, 256, smem>>>(o, steps, random_data);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
    }
    double flops = (double)grid * 4 * steps * 24 * 32768.0;
    printf("%-52s %d WG/CU %s data: %9.1f us  %7.1f TFLOP/s MFMA  (%.1f fp32-equivalent)\n", name, wg_per_cu, random_data ? "random" : "flat  ", best * 1e3,
           flops / best / 1e9, flops / best / 1e9 / 3);
    hipFree(o);
}

template <int MODE> void run(const char *name, int wg_per_cu, int random_data, int steps) {
    float *o; hipMalloc(&o, 4);
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        probe<MODE><<<grid, 256>>>(o, steps, random_data);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
    }
    // per step and wave: MODE 0/1: 24 MFMAs of 32768 flops; MODE 2/3: 48 MFMAs of 16384 flops
    double flops = (double)grid * 4 * steps * 24 * 32768.0;
    printf("%-52s %d WG/CU %s data: %9.1f us  %7.1f TFLOP/s MFMA  (%.1f fp32-equivalent)\n", name, wg_per_cu, random_data ? "random" : "flat  ", best * 1e3,
           flops / best / 1e9, flops / best / 1e9 / 3);
    hipFree(o);
}
int main() {
    const int steps = 600000;     // ~0.5 s per launch: sustained clocks
    for (int rnd : {1, 0})
        for (int w : {2}) {
            run<0>("32x32x16, row-major block order", w, rnd, steps);
            run<1>("32x32x16, term-major snake order", w, rnd, steps);
            run<2>("16x16x32, row-major block order", w, rnd, steps);
            run<3>("16x16x32, term-major snake order", w, rnd, steps);
        }
    for (int rnd : {1, 0}) {
        run_lds<4>("32x32x16 + 16 ds_read_b128 / step + barrier", 2, rnd, 300000);
        run_lds<5>("16x16x32 + 16 ds_read_b128 / step + barrier", 2, rnd, 300000);
    }
    return 0;
}
