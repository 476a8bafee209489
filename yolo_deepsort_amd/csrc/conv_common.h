// Pieces shared by the implicit-GEMM convolution kernels (conv.hip: fp32 MFMA, conv_f16x3.hip: split-fp16 MFMA).
#pragma once
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace yds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvKernelArgs {
    const float *x, *w, *bias, *res;
    float *y;
    int H, W, Cin, ldx;
    int Ho, Wo, Cout, ldy, ldr;
    int ksize, stride, pad;
    int K, Kpad, M;
    int act, res_mode;
    // XCD-aware tile map: the 8 XCDs own an xm x xn grid of rectangles of rm x rn tiles (workgroup id % 8 = XCD)
    int tiles_m, tiles_n, xm, rm, rn;
};

constexpr int KALIGN = 32;                // weight rows are zero padded to a multiple of this
#ifndef YDS_STAGGER
#define YDS_STAGGER 24
#endif
constexpr int STAGGER = YDS_STAGGER;     // s_sleep units of 64 clocks

template <int ACT> __device__ __forceinline__ float apply_act(float v) {
    if (ACT == ACT_LEAKY) return v > 0.f ? v : v * 0.1f;
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_MISH) {
        float sp = v > 20.f ? v : log1pf(expf(v));
        return v * tanhf(sp);
    }
    return v;
}

// Workgroup -> tile map.  The dispatcher places workgroup b on XCD b % 8 and every XCD has a private 4 MiB L2,
// so each XCD gets a compact rm x rn rectangle of tiles (small A-rows + B-columns footprint per K step) instead
// of a stripe through the whole problem.  Placement only affects speed, never results.
__device__ __forceinline__ bool tile_of_block(const ConvKernelArgs &p, int &tm, int &tn) {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    tm = (xcd % p.xm) * p.rm + idx % p.rm;
    tn = (xcd / p.xm) * p.rn + idx / p.rm;
    return tm < p.tiles_m && tn < p.tiles_n;
}

// host: choose the XCD grid (xm x xn = 8) with the smallest per-K-step footprint (rows*BM + cols*BN) per XCD;
// returns the grid size in workgroups
inline int plan_tile_map(ConvKernelArgs &k, int BM, int BN) {
    k.tiles_m = (k.M + BM - 1) / BM;
    k.tiles_n = (k.Cout + BN - 1) / BN;
    long best = -1;
    for (int xm = 1; xm <= 8; xm *= 2) {
        int xn = 8 / xm;
        int rm = (k.tiles_m + xm - 1) / xm, rn = (k.tiles_n + xn - 1) / xn;
        long waste = (long)rm * rn * 8 - (long)k.tiles_m * k.tiles_n;       // idle workgroup slots
        long cost = ((long)rm * BM + (long)rn * BN) * 64 + waste * (BM + BN);
        if (best < 0 || cost < best) { best = cost; k.xm = xm; k.rm = rm; k.rn = rn; }
    }
    return 8 * k.rm * k.rn;
}

// Epilogue over 32x32 MFMA accumulator fragments.  C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
// ACT / RES are compile-time so the 16 elements of a fragment are straight-line code: residual loads are issued
// together, then bias + activation, then the stores (128-byte row segments).
template <int TM, int TN, int ACT, int RES>
__device__ __forceinline__ void conv_epilogue(const ConvKernelArgs &p, f32x16 (&acc)[TM][TN], int m_wave, int n_wave, int lane) {
    const int col = lane & 31, rsel = (lane >> 5) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_wave + j * 32 + col;
        if (n >= p.Cout) continue;
        const float bias = p.bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m_wave + i * 32 + rsel;
            float r[16];
            if (RES != RES_NONE) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    r[e] = m < p.M ? p.res[(size_t)m * p.ldr + n] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                float v = acc[i][j][e] + bias;
                if (RES == RES_BEFORE_ACT) v += r[e];
                v = apply_act<ACT>(v);
                if (RES == RES_AFTER_ACT) v += r[e];
                if (m < p.M) p.y[(size_t)m * p.ldy + n] = v;
            }
        }
    }
}

ConvKernelArgs make_conv_args(const ConvArgs &a);

// dispatch on (activation, residual mode) to the compile-time epilogue instantiation of launcher L<ACT, RES>
#define YDS_DISPATCH_ACT_RES(k, CALL)                                                                     \
    switch ((k).act * 4 + (k).res_mode) {                                                                 \
        case ACT_LINEAR * 4 + RES_NONE: CALL(ACT_LINEAR, RES_NONE); break;                                \
        case ACT_LEAKY * 4 + RES_NONE: CALL(ACT_LEAKY, RES_NONE); break;                                  \
        case ACT_LEAKY * 4 + RES_AFTER_ACT: CALL(ACT_LEAKY, RES_AFTER_ACT); break;                        \
        case ACT_MISH * 4 + RES_NONE: CALL(ACT_MISH, RES_NONE); break;                                    \
        case ACT_MISH * 4 + RES_AFTER_ACT: CALL(ACT_MISH, RES_AFTER_ACT); break;                          \
        case ACT_RELU * 4 + RES_NONE: CALL(ACT_RELU, RES_NONE); break;                                    \
        case ACT_RELU * 4 + RES_BEFORE_ACT: CALL(ACT_RELU, RES_BEFORE_ACT); break;                        \
        default: fail("conv: unsupported activation/residual combination (%d, %d)", (k).act, (k).res_mode); \
    }

// split-fp16 path (conv_f16x3.hip)
constexpr int kF16Variants = 4;
const char *conv_f16x3_variant_name(int v);
void launch_conv_f16x3(ConvKernelArgs k, int variant, hipStream_t s);

}  // namespace yds
