// Pieces shared by the implicit-GEMM convolution kernels (conv.hip: fp32 MFMA, conv_f16x3.hip: split-fp16 MFMA).
#pragma once
#include "common.h"
#include "h16.h"

#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace yds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// split-fp16 ("f16x3") operand encoding shared by conv_f16x3.hip and conv_win.hip:
//   x * A_SCALE = hi + lo / LO_SCALE,  w = hi + lo / LO_SCALE   (hi, lo fp16; see conv_f16x3.hip)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr float A_SCALE = 1.f / 256.f, LO_SCALE = 2048.f;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct ConvKernelArgs {
    const float *x, *w, *bias, *res;
    float *y;
    int H, W, Cin, ldx;
    int Ho, Wo, Cout, ldy, ldr;
    int ksize, stride, pad;
    int K, Kpad, M;
    int act, res_mode;
    int fmt_x, fmt_y, fmt_r;              // TensorFmt of input, output and residual views
    int terms;                            // 3: f16x3, 1: hi halves only (half mode; LDS-DMA and window kernels)
    int ksplit = 1;                       // LDS-DMA kernel: K ranges (grid.y); > 1 writes raw partial sums into slab blockIdx.y of y
    // two convolutions of the same input in one launch (CSP split, darknet.cpp): filters [n_split, Cout) write to y2
    float *y2 = nullptr;
    int ldy2 = 0, n_split = 0;
    // XCD-aware tile map: the 8 XCDs own an xm x xn grid of rectangles of rm x rn tiles (workgroup id % 8 = XCD)
    int tiles_m, tiles_n, xm, rm, rn;
    int gn = 1;                           // filter tiles walked together inside an XCD's rectangle (tile_of_block); set by plan_tile_map
    int gn_req = 0;                       // ... as requested by the caller (ConvArgs::tile_gn: 0 / 1 = column by column)
};

constexpr int KALIGN = 32;                // weight rows are zero padded to a multiple of this
constexpr int STAGGER = 24;              // s_sleep units of 64 clocks: start offset of every other resident workgroup (staged kernels)

template <int ACT> __device__ __forceinline__ float apply_act(float v) {
    if (ACT == ACT_LEAKY) return v > 0.f ? v : v * 0.1f;
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_MISH) {
        // x * tanh(softplus(x)) with ONE exponential:  tanh(log(1+e)) = ((1+e)^2 - 1) / ((1+e)^2 + 1) = n / (n + 2),
        // n = e*(e+2), e = exp(x).  Same threshold as torch's softplus (x > 20 -> softplus(x) = x, tanh = 1).
        // Raw hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each), branch free: the result stays within ~3e-7
        // relative of the exact value.  (Round 3: `__frcp_rn` had been expanding to the IEEE division sequence - div_scale,
        // four fmas, div_fmas, div_fixup - behind an exec-mask branch: ten instructions per value, which made the Mish layers of
        // yolov4 VALU bound in their epilogues: 304x304 64->64 181 us against 149 us with a linear epilogue.)
        const float e = __builtin_amdgcn_exp2f(fminf(v, 20.f) * 1.44269504088896341f);
        const float n = e * (e + 2.f);
        const float m = v * (n * __builtin_amdgcn_rcpf(n + 2.f));
        return v > 20.f ? v : m;
    }
    return v;
}

// Workgroup -> tile map.  The dispatcher places workgroup b on XCD b % 8 and every XCD has a private 4 MiB L2,
// so each XCD gets a compact rm x rn rectangle of tiles (small A-rows + B-columns footprint per K step) instead
// of a stripe through the whole problem.  Placement only affects speed, never results.
// Order INSIDE an XCD's rectangle (round 6): the ~32 workgroups an XCD runs at a time are consecutive ids.  gn = 1 walks the
// rectangle column by column (32 different pixel tiles against ONE filter tile at a time); gn > 1 walks it in groups of gn filter
// tiles, filter tile fastest, so that the gn workgroups of one pixel tile run together and share its input window through the L2.
__device__ __forceinline__ bool tile_of_block(const ConvKernelArgs &p, int &tm, int &tn) {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    int lm, ln;
    if (p.gn <= 1) {
        lm = idx % p.rm;
        ln = idx / p.rm;
    } else {
        const int per = p.rm * p.gn, grp = idx / per, w = idx - grp * per;
        const int g = min(p.gn, p.rn - grp * p.gn);              // the last group of columns may be narrower
        lm = w / g;
        ln = grp * p.gn + (w - lm * g);
    }
    tm = (xcd % p.xm) * p.rm + lm;
    tn = (xcd / p.xm) * p.rn + ln;
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
    k.gn = std::max(1, std::min(k.gn_req, k.rn));                 // measured per layer by conv_autotune (results do not depend on it)
    return 8 * k.rm * k.rn;
}

// Epilogue over 32x32 MFMA accumulator fragments (C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)).
// The tile is transposed through LDS (the main loop's buffers are free by now) so that global traffic is 16-byte
// and row contiguous: bias, residual and output move as float4 along the channel axis.  One pass per row of waves
// keeps the staging area at (BM/WM) x (BN+4) floats, which fits inside every variant's main-loop allocation.
// ACT / RES are compile-time.
// RowMap: tile row -> flat output pixel index (or < 0 for "no such pixel"); FlatRows = consecutive pixels from m0
struct FlatRows {
    int m0;
    __device__ __forceinline__ int operator()(int row) const { return m0 + row; }
};
// ONE_PASS: the staging area holds the WHOLE tile (BM x (BN+4) floats instead of one row of waves): every wave stores its
// accumulators at once and the workgroup sweeps all rows after a single barrier (2 barriers instead of 2 x WM, no waves
// idling while one wave row is staged).  The caller sizes the LDS for it (conv_stage_bytes).
// Store: callable (float *st) that writes THIS wave's accumulator tile into its band of the staging area (row r, column c of
// the band at st[r * (BN + 4) + c]) - the only part that depends on the MFMA shape's C/D layout.
template <int BM, int BN, int WM, int WN, int ACT, int RES, int NT, class RowMap, bool ONE_PASS, class Store>
__device__ __forceinline__ void conv_epilogue_core(const ConvKernelArgs &p, Store store, float *stage, RowMap rows, int n0, int tid) {
    constexpr int ROWS = BM / WM, LD = BN + 4, C4 = BN / 4;        // staged rows, padded row length, float4 per row
    constexpr int RSTEP = NT / C4;                                   // rows covered by one sweep of the NT threads
    static_assert(NT % C4 == 0 && C4 % 2 == 0 && ROWS % RSTEP == 0, "tile width must divide the workgroup; lane pairs share a row");
    const int wave = tid >> 6, wm = wave / WN;
    const int c4 = tid % C4, rr = tid / C4;
    const int n = n0 + c4 * 4;
    const bool n_vec = n + 3 < p.Cout;
    // output destination of this thread's four channels (merged launches: the second convolution's filters go to y2)
    const bool second = p.n_split > 0 && n >= p.n_split;
    float *const y_base = second ? p.y2 : p.y;
    const int y_ld = second ? p.ldy2 : p.ldy, ny = second ? n - p.n_split : n;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n_vec) bias4 = *reinterpret_cast<const float4 *>(p.bias + n);
    else {
        if (n < p.Cout) bias4.x = p.bias[n];
        if (n + 1 < p.Cout) bias4.y = p.bias[n + 1];
        if (n + 2 < p.Cout) bias4.z = p.bias[n + 2];
    }
    // The tensor formats are uniform run-time flags; the sweep is instantiated per (output H16?, residual H16?) and entered
    // through ONE uniform branch, so its loops carry no format tests (the generic form spent a third of its instructions on
    // exec-mask bookkeeping and re-tested the formats in every iteration).
    constexpr int PER_PASS = ROWS / RSTEP, NRES = RES != RES_NONE ? WM * PER_PASS : 1;
    auto sweep = [&](auto yf_c, auto rf_c) {
        constexpr int YF = decltype(yf_c)::value, RF = decltype(rf_c)::value;          // TensorFmt of the output / the residual
        constexpr bool YH = YF == FMT_H16, RH = RF == FMT_H16, YS = YF == FMT_F16, RS = RF == FMT_F16;
        // The residual rows of ALL passes are fetched up front: the loads are in flight while the tile goes through LDS,
        // instead of paying one exposed global-load latency per pass.
        float4 rraw[NRES];
        if (RES != RES_NONE) {
#pragma unroll
            for (int q = 0; q < NRES; ++q) {
                const int m = rows((q / PER_PASS) * ROWS + rr + (q % PER_PASS) * RSTEP);
                rraw[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m >= 0 && m < p.M && n < p.Cout) {
                    const float *rp = p.res + (size_t)m * p.ldr;
                    if (RH) {                                                      // H16 tensors have Cout % 32 == 0: [hi 8 B | lo 8 B]
                        const char *g = reinterpret_cast<const char *>(rp + (n & ~31)) + (n & 31) * 2;
                        const float2 hi = *reinterpret_cast<const float2 *>(g), lo = *reinterpret_cast<const float2 *>(g + 64);
                        rraw[q] = make_float4(hi.x, hi.y, lo.x, lo.y);
                    } else if (RS) {                                               // FMT_F16: four channels = 8 bytes, ldr in float slots
                        const float2 h = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(rp) + n * 2);
                        rraw[q] = make_float4(h.x, h.y, 0.f, 0.f);
                    } else if (n_vec) {
                        rraw[q] = *reinterpret_cast<const float4 *>(rp + n);
                    } else {
                        rraw[q].x = rp[n];
                        if (n + 1 < p.Cout) rraw[q].y = rp[n + 1];
                        if (n + 2 < p.Cout) rraw[q].z = rp[n + 2];
                    }
                }
            }
        }
#pragma unroll
        for (int pass = 0; pass < WM; ++pass) {
            if (ONE_PASS ? pass == 0 : wm == pass) store(stage + (ONE_PASS ? wm * ROWS * LD : 0));
            if (!ONE_PASS || pass == 0) __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < PER_PASS; ++k2) {
                const int r = rr + k2 * RSTEP;
                const int m = rows(pass * ROWS + r);
                const bool live = m >= 0 && m < p.M && n < p.Cout;   // no early exit: lane pairs trade halves below
                float4 v = *reinterpret_cast<const float4 *>(stage + (ONE_PASS ? pass * ROWS * LD : 0) + r * LD + c4 * 4);
                float rs[4] = {0.f, 0.f, 0.f, 0.f};
                if (RES != RES_NONE) {
                    const float4 t = rraw[RES != RES_NONE ? pass * PER_PASS + k2 : 0];     // zeros where the row is not live
                    if (RH) {
                        union { float2 f; h16x4 h; } uh, ul;
                        uh.f = make_float2(t.x, t.y);
                        ul.f = make_float2(t.z, t.w);
                        h16_decode4(uh.h, ul.h, rs);
                    } else if (RS) {
                        union { float2 f; h16x4 h; } uh;
                        uh.f = make_float2(t.x, t.y);
#pragma unroll
                        for (int k = 0; k < 4; ++k) rs[k] = (float)uh.h[k] * (1.f / H16_A_SCALE);
                    } else { rs[0] = t.x; rs[1] = t.y; rs[2] = t.z; rs[3] = t.w; }
                }
                float o[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (RES == RES_BEFORE_ACT) o[k] += rs[k];
                    o[k] = apply_act<ACT>(o[k]);
                    if (RES == RES_AFTER_ACT) o[k] += rs[k];
                }
                float *yp = y_base + (size_t)m * y_ld;
                if (YH) {
                    // neighbouring lanes hold neighbouring channel quads of the same pixel: they trade halves so that each lane
                    // issues ONE 16-byte store (even lane: 8 hi halves, odd lane: 8 lo halves) instead of two 8-byte ones.
                    // The trade is a DPP quad permute [1,0,3,2] on the vector ALU (__shfl_xor goes through the LDS crossbar).
                    h16x4 hi, lo;
                    h16_encode4(o, hi, lo);
                    const bool odd = c4 & 1;
                    union { h16x4 h; int i[2]; } send, recv;
                    send.h = odd ? hi : lo;
                    recv.i[0] = __builtin_amdgcn_update_dpp(send.i[0], send.i[0], 0xB1, 0xF, 0xF, false);
                    recv.i[1] = __builtin_amdgcn_update_dpp(send.i[1], send.i[1], 0xB1, 0xF, 0xF, false);
                    union { h16x4 h[2]; float4 f; } out;
                    out.h[0] = odd ? recv.h : hi;
                    out.h[1] = odd ? lo : recv.h;
                    const int nq = ny & ~7;                                         // first channel of the lane pair
                    char *g = reinterpret_cast<char *>(yp + (nq & ~31)) + (nq & 31) * 2 + (odd ? 64 : 0);
                    if (live) *reinterpret_cast<float4 *>(g) = out.f;
                } else if (YS) {
                    // FMT_F16: the even lane of a pair stores both lanes' channel quads, 8 consecutive fp16 = one 16-byte store
                    union { h16x4 h; int i[2]; } mine, other;
#pragma unroll
                    for (int k = 0; k < 4; ++k) mine.h[k] = (_Float16)(o[k] * H16_A_SCALE);
                    other.i[0] = __builtin_amdgcn_update_dpp(mine.i[0], mine.i[0], 0xB1, 0xF, 0xF, false);
                    other.i[1] = __builtin_amdgcn_update_dpp(mine.i[1], mine.i[1], 0xB1, 0xF, 0xF, false);
                    union { h16x4 h[2]; float4 f; } out;
                    out.h[0] = mine.h;
                    out.h[1] = other.h;
                    if (live && !(c4 & 1)) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(yp) + ny * 2) = out.f;
                } else if (live) {
                    if (n_vec) *reinterpret_cast<float4 *>(yp + ny) = make_float4(o[0], o[1], o[2], o[3]);
                    else { yp[ny] = o[0]; if (n + 1 < p.Cout) yp[ny + 1] = o[1]; if (n + 2 < p.Cout) yp[ny + 2] = o[2]; }
                }
            }
            if (!ONE_PASS && pass + 1 < WM) __syncthreads();
        }
    };
    // (FMT_F16 outputs exist in half mode only and take an FMT_F16 residual or none: the planner keeps both sides of a shortcut in
    //  one format, make_conv_args refuses anything else)
    using F32c = std::integral_constant<int, FMT_F32>;
    using H16c = std::integral_constant<int, FMT_H16>;
    using F16c = std::integral_constant<int, FMT_F16>;
    const bool yh = p.fmt_y == FMT_H16, rh = RES != RES_NONE && p.fmt_r == FMT_H16;
    if (p.fmt_y == FMT_F16) {
        sweep(F16c{}, F16c{});
    } else if (yh) {
        if (rh) sweep(H16c{}, H16c{});
        else sweep(H16c{}, F32c{});
    } else {
        if (rh) sweep(F32c{}, H16c{});
        else sweep(F32c{}, F32c{});
    }
}
constexpr size_t conv_stage_bytes(int BM, int BN) { return (size_t)BM * (BN + 4) * sizeof(float); }

// 32x32 accumulator fragments (v_mfma_f32_32x32x*): col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
template <int BM, int BN, int WM, int WN, int ACT, int RES, int TM, int TN, int NT = 256, class RowMap = FlatRows, bool ONE_PASS = false>
__device__ __forceinline__ void conv_epilogue_rows(const ConvKernelArgs &p, f32x16 (&acc)[TM][TN], float *stage, RowMap rows, int n0, int tid) {
    constexpr int LD = BN + 4;
    const int lane = tid & 63, wn = (tid >> 6) % WN;
    const int col = lane & 31, rsel = (lane >> 5) * 4;
    auto store = [&](float *st) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    st[(i * 32 + rsel + (e & 3) + 8 * (e >> 2)) * LD + wn * (BN / WN) + j * 32 + col] = acc[i][j][e];
    };
    conv_epilogue_core<BM, BN, WM, WN, ACT, RES, NT, RowMap, ONE_PASS>(p, store, stage, rows, n0, tid);
}

// 16x16 accumulator fragments (v_mfma_f32_16x16x32_f16): col = lane & 15, row = 4 * (lane >> 4) + e; TM / TN count 16-wide blocks.
// (Bank check: the four 16-lane groups of a store sit 4 rows = 4 * (BN + 4) floats apart = 16 banks mod 64: conflict free.)
template <int BM, int BN, int WM, int WN, int ACT, int RES, int TM, int TN, int NT = 256, class RowMap = FlatRows, bool ONE_PASS = false>
__device__ __forceinline__ void conv_epilogue16_rows(const ConvKernelArgs &p, f32x4 (&acc)[TM][TN], float *stage, RowMap rows, int n0, int tid) {
    constexpr int LD = BN + 4;
    const int lane = tid & 63, wn = (tid >> 6) % WN;
    const int col = lane & 15, rsel = (lane >> 4) * 4;
    auto store = [&](float *st) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st[(i * 16 + rsel + e) * LD + wn * (BN / WN) + j * 16 + col] = acc[i][j][e];
    };
    conv_epilogue_core<BM, BN, WM, WN, ACT, RES, NT, RowMap, ONE_PASS>(p, store, stage, rows, n0, tid);
}
template <int BM, int BN, int WM, int WN, int ACT, int RES, int TM, int TN, int NT = 256, bool ONE_PASS = false>
__device__ __forceinline__ void conv_epilogue16(const ConvKernelArgs &p, f32x4 (&acc)[TM][TN], float *stage, int m0, int n0, int tid) {
    conv_epilogue16_rows<BM, BN, WM, WN, ACT, RES, TM, TN, NT, FlatRows, ONE_PASS>(p, acc, stage, FlatRows{m0}, n0, tid);
}

template <int BM, int BN, int WM, int WN, int ACT, int RES, int TM, int TN, int NT = 256, bool ONE_PASS = false>
__device__ __forceinline__ void conv_epilogue(const ConvKernelArgs &p, f32x16 (&acc)[TM][TN], float *stage, int m0, int n0, int tid) {
    conv_epilogue_rows<BM, BN, WM, WN, ACT, RES, TM, TN, NT, FlatRows, ONE_PASS>(p, acc, stage, FlatRows{m0}, n0, tid);
}

ConvKernelArgs make_conv_args(const ConvArgs &a);
// fused detector stem (conv_stem2.hip): 3x3/s1 RGB conv (direct, vector ALU) + 3x3/s2 32->64 conv (f16x3 MFMA) in one kernel
bool conv_stem2_applicable(const ConvKernelArgs &k0, const ConvKernelArgs &k1);
void launch_conv_stem2(const ConvKernelArgs &k0, const ConvKernelArgs &k1, hipStream_t s);
// fused first residual block (conv_block1.hip): conv1x1 64->32 + conv3x3 32->64 + shortcut to the block input
bool conv_block1_applicable(const ConvKernelArgs &k2, const ConvKernelArgs &k3);
void launch_conv_block1(const ConvKernelArgs &k2, const ConvKernelArgs &k3, hipStream_t s);

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

// direct first-layer kernel (conv_first.hip)
bool conv_direct_applicable(const ConvKernelArgs &k);
void launch_conv_direct(const ConvKernelArgs &k, hipStream_t s);
// ReID stem: the direct kernel fused with MaxPool2d(3, 2, 1); k.y is the pooled tensor
bool conv_pool_applicable(const ConvKernelArgs &k);
void launch_conv_pool(const ConvKernelArgs &k, hipStream_t s);

// split-fp16 path (conv_f16x3.hip)
constexpr int kF16Variants = 16;           // 0-3 register-staged tiles, 4-7 and 11-12 LDS-DMA ring, 8-10 window-resident 3x3, 13 window-resident with two workgroups per CU, 14-15 LDS-DMA ring with split-K + reduce pass (pre-split inputs only)
inline bool f16_variant_is_dma(int v) { return (v >= 4 && v <= 7) || v == 11 || v == 12; }
inline bool f16_variant_is_win(int v) { return v >= 8 && v <= 10; }
inline bool f16_variant_is_win2(int v) { return v == 13; }
inline bool f16_variant_is_splitk(int v) { return v == 14 || v == 15; }
bool conv_splitk_applicable(const ConvKernelArgs &k, int fv);     // a split of >= 2 K ranges exists
bool conv_splitk_preferred(const ConvKernelArgs &k);              // the deterministic rule that selects variant 14 (never timed against the others)
// two-workgroup window kernel (conv_win2.hip): 128x128 tiles, 4 waves, 16-channel K steps
bool conv_win2_applicable(const ConvKernelArgs &k);
void launch_conv_win2(ConvKernelArgs k, hipStream_t s);
// window-resident 3x3 stride-1 kernel (conv_win.hip)
bool conv_win_applicable(const ConvKernelArgs &k);
void launch_conv_win(ConvKernelArgs k, int shape, hipStream_t s);   // shape 0: 256x128 (4x2 waves), 1: 256x64 (8x1), 2: 128x64 (4x1, two workgroups per CU; default arithmetic only)
const char *conv_f16x3_variant_name(int v);
void launch_conv_f16x3(ConvKernelArgs k, int variant, hipStream_t s);
// sampled (shader cycles, 100 MHz ticks) accumulated inside the window kernels since the last reset
void conv_win_clock(unsigned long long *cycles_ticks, bool reset);   // in-kernel sampling: -DYDS_CLOCK_PROBE builds only (zeros otherwise)
void conv_win16_clock(unsigned long long *cycles_ticks, bool reset);
bool conv_win16_small_applicable(const ConvKernelArgs &k);           // shape 2 below
void launch_conv_win16(ConvKernelArgs k, int shape, hipStream_t s);  // the f16x3 (default arithmetic) form on v_mfma_f32_16x16x32_f16 (conv_win16.hip)
void conv_win2_clock(unsigned long long *cycles_ticks, bool reset);

}  // namespace yds
