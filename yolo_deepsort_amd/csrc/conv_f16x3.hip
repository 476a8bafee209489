// Implicit-GEMM convolution on the fp16 matrix cores with fp32-class accuracy ("f16x3": split-fp16, 3 MFMAs).
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the fp16/bf16 MFMA rate, so the exact-fp32 kernel in conv.hip is
// matrix-pipe bound.  This kernel feeds v_mfma_f32_32x32x16_f16 with a two-term fp16 expansion of both operands:
//     x * 2^-8 = xh + xl * 2^-11        xh = fp16(x * 2^-8),  xl = fp16((x * 2^-8 - xh) * 2^11)
//     w        = wh + wl * 2^-11        (pre-split on the host, BN already folded)
//     x.w * 2^-8 ~= sum xh*wh  +  2^-11 * sum (xh*wl + xl*wh)          (the xl*wl term, 2^-22 relative, is dropped)
// fp16 products are exact in the fp32 accumulators; two accumulator sets keep the 2^-11 scaled cross terms apart
// until the epilogue.  Each operand carries 22 significant bits, so the result differs from an fp32 fma chain by
// ~1e-6 relative - the same class as a different fp32 summation order (measured through all 75 / 110 layers of
// yolov3 / yolov4: max 5e-5 of the output scale; a 3-term bf16 split was 10x worse).  The 2^11 pre-scale keeps
// the low halves in fp16's normal range, the 2^-8 activation scale moves overflow out to 1.6e7.
//
// Structure is the fp32 kernel's: NHWC fp32 activations are split on the fly while being staged to LDS (the
// producer keeps writing plain fp32), weights arrive pre-split as [Cout][K/32][32 hi | 32 lo] fp16 (same bytes as
// fp32), LDS rows are [64 B hi | 64 B lo | 16 B pad] = 144 B which keeps ds_read_b128 fragment reads conflict
// free, XCD-aware tile map, compile-time epilogue.
#include "conv_common.h"

#include <math.h>
#include <string.h>

namespace yds {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 144;                 // bytes per LDS row
constexpr float A_SCALE = 1.f / 256.f, LO_SCALE = 2048.f;

__device__ __forceinline__ void split4(const f32x4 &v, h4 &hi, h4 &lo) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float xs = v[c] * A_SCALE;
        _Float16 h = (_Float16)xs;                         // round to nearest even
        hi[c] = h;
        lo[c] = (_Float16)((xs - (float)h) * LO_SCALE);
    }
}

// AIN: format of the input tensor (compile time: the staging differs); output / residual formats are runtime flags
// of the epilogue.
template <int BM, int BN, int ACT, int RES, int AIN>
__global__ __launch_bounds__(256, 2) void conv_igemm_f16x3(ConvKernelArgs p) {
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;      // 16-byte chunks per thread per K step (32 k)
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char *As = smem16;                                     // [2][BM][ROWB]
    char *Bs = smem16 + 2 * BM * ROWB;                     // [2][BN][ROWB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int m0, n0;
    {
        int tm, tn;
        if (!tile_of_block(p, tm, tn)) return;
        m0 = tm * BM;
        n0 = tn * BN;
    }
    const int cq = tid & 7;          // which 4 of the 32 k (A, fp32) / which 16-byte chunk of the 128-byte weight row
    const int r0 = tid >> 3;         // first staged row; further rows at +32

    int a_base[A_ROWS], a_iy[A_ROWS], a_ix[A_ROWS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            int img = m / HoWo, rem = m - img * HoWo;
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
            a_base[i] = ((img * p.H + a_iy[i]) * p.W + a_ix[i]) * p.ldx;
        } else {
            a_iy[i] = -(1 << 28);
            a_ix[i] = 0;
            a_base[i] = 0;
        }
    }
    // K position of the tile being loaded.  With a pre-split input Cin % 32 == 0, so a 32-wide K step is one
    // 32-channel group of ONE filter tap for every lane: tap and group offset are wave-uniform (scalar registers),
    // the per-row pixel offset + bounds test are recomputed only when the tap changes, and the loads take a scalar
    // base + 32-bit per-lane offset (no per-step address arithmetic on the vector ALU).  fp32 inputs (the image,
    // narrow nets) keep the general per-lane decomposition: there a K step can straddle taps.
    int kk = cq * 4, kh = 0, kw = 0, kc = AIN == FMT_H16 ? 0 : kk;
    if (AIN != FMT_H16)
        while (kc >= p.Cin) { kc -= p.Cin; if (++kw == p.ksize) { kw = 0; ++kh; } }
    unsigned w_off[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) w_off[i] = (unsigned)min(n0 + r0 + 32 * i, p.Cout - 1) * (unsigned)p.Kpad * 4u + cq * 16;
    const char *w_bytes = reinterpret_cast<const char *>(p.w), *x_bytes = reinterpret_cast<const char *>(p.x);

    f32x4 a_reg[A_ROWS], b_reg[B_ROWS];
    unsigned a_ok = 0, a_off[A_ROWS], tap_ok = 0;
    int kt_load = 0;
    auto set_tap = [&]() {                                     // H16 path: per-row byte offset + bounds of tap (kh, kw)
        tap_ok = 0;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            int iy = a_iy[i] + kh, ix = a_ix[i] + kw;
            bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            tap_ok |= (ok ? 1u : 0u) << i;
            a_off[i] = ok ? (unsigned)(a_base[i] + (kh * p.W + kw) * p.ldx + cq * 4) * 4u : cq * 16u;
        }
    };
    if (AIN == FMT_H16) set_tap();
    auto load_tiles = [&]() {
        if (AIN == FMT_H16) {
            a_ok = tap_ok;
            const char *xg = x_bytes + (size_t)kc * 4;           // uniform: channel-group base of this K step
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) a_reg[i] = *reinterpret_cast<const f32x4 *>(xg + a_off[i]);
        } else {
            const int tap_off = (kh * p.W + kw) * p.ldx + kc;
            const bool k_ok = kk < p.K;
            a_ok = 0;
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                int iy = a_iy[i] + kh, ix = a_ix[i] + kw;
                bool ok = k_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                a_ok |= (ok ? 1u : 0u) << i;
                a_reg[i] = *reinterpret_cast<const f32x4 *>(p.x + (ok ? a_base[i] + tap_off : 0));
            }
        }
        const char *wg = w_bytes + (size_t)kt_load * 128;        // uniform
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) b_reg[i] = *reinterpret_cast<const f32x4 *>(wg + w_off[i]);
    };
    auto advance_k = [&]() {
        ++kt_load;
        kk += 32;
        kc += 32;
        if (AIN == FMT_H16) {
            if (kc >= p.Cin) { kc = 0; if (++kw == p.ksize) { kw = 0; ++kh; } set_tap(); }
        } else {
            while (kc >= p.Cin) { kc -= p.Cin; if (++kw == p.ksize) { kw = 0; ++kh; } }
        }
    };
    auto store_tiles = [&](int buf) {
        char *a = As + buf * BM * ROWB, *b = Bs + buf * BN * ROWB;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            f32x4 v = a_reg[i];
            if (!((a_ok >> i) & 1u)) v = f32x4{0, 0, 0, 0};
            if (AIN == FMT_H16) {
                // pre-split input: chunk cq of the 128-byte group is already [hi | lo] fp16 - plain copy
                *reinterpret_cast<f32x4 *>(a + (r0 + 32 * i) * ROWB + cq * 16) = v;
            } else {
                h4 hi, lo;
                split4(v, hi, lo);
                char *row = a + (r0 + 32 * i) * ROWB + cq * 8;
                *reinterpret_cast<h4 *>(row) = hi;
                *reinterpret_cast<h4 *>(row + 64) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<f32x4 *>(b + (r0 + 32 * i) * ROWB + cq * 16) = b_reg[i];
    };

    f32x16 acc1[TM][TN], acc2[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc1[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

    const int nk = p.Kpad / 32;
    load_tiles();
    store_tiles(0);
    if (STAGGER && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_sleep(STAGGER / 4);
    __syncthreads();

    const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
    const char *a_lds = As + wm * (BM / WM) * ROWB + frag_off;
    const char *b_lds = Bs + wn * (BN / WN) * ROWB + frag_off;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) { advance_k(); load_tiles(); }
        const char *a = a_lds + cur * BM * ROWB, *b = b_lds + cur * BN * ROWB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                      // two MFMA k-steps of 16 per staged tile of 32
            h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const h8 *>(a + i * 32 * ROWB + s * 32);
                al[i] = *reinterpret_cast<const h8 *>(a + i * 32 * ROWB + s * 32 + 64);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const h8 *>(b + j * 32 * ROWB + s * 32);
                bl[j] = *reinterpret_cast<const h8 *>(b + j * 32 * ROWB + s * 32 + 64);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc1[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
                }
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
    }
    // recombine the two accumulator sets and undo the activation scale
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[i][j][e] = (acc1[i][j][e] + acc2[i][j][e] * (1.f / LO_SCALE)) * (1.f / A_SCALE);
    static_assert((BM / WM) * (BN + 4) * 4 <= 2 * (BM + BN) * ROWB, "epilogue staging must fit the main-loop LDS");
    conv_epilogue<BM, BN, WM, WN, ACT, RES>(p, acc1, reinterpret_cast<float *>(smem16), m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for pre-split (H16) inputs: both operands are opaque 16-byte chunks, so they go global -> LDS with
// global_load_lds_dwordx4 (no staging VGPRs, no ds_write, no conversion) through an NS-deep ring of K=16 stages.
// The staged kernel above keeps exactly one tile of loads in flight per workgroup and its K loop runs at the
// latency of that batch; here NS-1 stages are in flight and the wave only waits (counted vmcnt) for the oldest.
//   stage  = BM + BN rows of 64 B: [16 hi | 16 lo] fp16 = one MFMA k-step; row r keeps chunk c at position
//            c ^ ((r >> 2) & 3) (the DMA writes lane-linear, so the swizzle is applied to the SOURCE address and
//            again by the fragment reads; 16 consecutive rows then hit 16 distinct 16-byte bank slots)
//   ring   : iteration t waits for tile t (vmcnt = DMAs of the younger tiles), one s_barrier, re-fills the stage
//            that iteration t-1 finished reading, then 12 MFMAs per wave on stage t % NS
//   zero padding: out-of-image taps fetch from a 16-byte zero page instead of branching.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int NS, int ACT, int RES>
__global__ __launch_bounds__(256, 2) void conv_igemm_f16x3_dma(ConvKernelArgs p, const void *zero_page) {
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int ROW = 64;
    constexpr int A_BYTES = BM * ROW, STAGE = (BM + BN) * ROW;
    constexpr int A_INST = BM / 64, B_INST = BN / 64;          // DMA instructions per wave per stage (16 rows each)
    constexpr int IN = A_INST + B_INST;
    static_assert(NS >= 3 && NS <= 6, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char ring[];     // [NS][STAGE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int m0, n0;
    {
        int tm, tn;
        if (!tile_of_block(p, tm, tn)) return;
        m0 = tm * BM;
        n0 = tn * BN;
    }
    // DMA lane roles: instruction q of this wave fills rows (q*4 + wave)*16 .. +15; lane -> (row, position)
    const int drow = lane >> 2, dpos = lane & 3;
    int a_base[A_INST], a_iy[A_INST], a_ix[A_INST], a_sc[A_INST];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int q = 0; q < A_INST; ++q) {
        const int row = (q * 4 + wave) * 16 + drow;
        a_sc[q] = dpos ^ ((row >> 2) & 3);                      // logical chunk this lane fetches
        const int m = m0 + row;
        if (m < p.M) {
            int img = m / HoWo, rem = m - img * HoWo;
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy[q] = oy * p.stride - p.pad;
            a_ix[q] = ox * p.stride - p.pad;
            a_base[q] = ((img * p.H + a_iy[q]) * p.W + a_ix[q]) * p.ldx;
        } else {
            a_iy[q] = -(1 << 28);
            a_ix[q] = 0;
            a_base[q] = 0;
        }
    }
    const char *w_row[B_INST];
    int b_sc[B_INST];
#pragma unroll
    for (int q = 0; q < B_INST; ++q) {
        const int row = (q * 4 + wave) * 16 + drow;
        b_sc[q] = dpos ^ ((row >> 2) & 3);
        w_row[q] = reinterpret_cast<const char *>(p.w) + (size_t)min(n0 + row, p.Cout - 1) * p.Kpad * 4;
    }
    // byte offset of logical chunk sc (0,1: hi k0-7 / k8-15; 2,3: lo) of half h inside a 128-byte group
    auto chunk_off = [](int h, int sc) { return (sc < 2 ? 0 : 64) + (2 * h + (sc & 1)) * 16; };

    int kt_issue = 0, kh = 0, kw = 0, kc = 0;                   // next K16 tile to issue and its (tap, channel) position
    auto issue = [&](int stage) {
        const int h = (kc >> 4) & 1, gb = kc & ~31;
        char *sa = ring + stage * STAGE, *sb = sa + A_BYTES;
        const int tap_off = (kh * p.W + kw) * p.ldx + gb;
#pragma unroll
        for (int q = 0; q < A_INST; ++q) {
            int iy = a_iy[q] + kh, ix = a_ix[q] + kw;
            bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const char *src = ok ? reinterpret_cast<const char *>(p.x + (a_base[q] + tap_off)) + chunk_off(h, a_sc[q])
                                 : reinterpret_cast<const char *>(zero_page);
            __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(sa + (q * 4 + wave) * 16 * ROW), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < B_INST; ++q) {
            const char *src = w_row[q] + (size_t)(kt_issue >> 1) * 128 + chunk_off(kt_issue & 1, b_sc[q]);
            __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(sb + (q * 4 + wave) * 16 * ROW), 16, 0, 0);
        }
        ++kt_issue;
        kc += 16;
        if (kc >= p.Cin) { kc = 0; if (++kw == p.ksize) { kw = 0; ++kh; } }
    };

    f32x16 acc1[TM][TN], acc2[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc1[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

    const int nk = p.K / 16;                                    // Cin % 32 == 0 on this path
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s);

    // fragment addressing: row = lane & 31 (+ tile offsets, multiples of 32), chunk kb / 2+kb at swizzled position
    const int swz = (lane >> 2) & 3, kb = lane >> 5;
    const int a_frag = (wm * (BM / WM) + (lane & 31)) * ROW, b_frag = A_BYTES + (wn * (BN / WN) + (lane & 31)) * ROW;
    const int pos_hi = (kb ^ swz) * 16, pos_lo = ((2 + kb) ^ swz) * 16;

    for (int t = 0; t < nk; ++t) {
        // tile t must have landed; the NS-2 younger tiles may stay in flight
        const int younger = min(NS - 2, nk - 1 - t);
        switch (younger) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<IN>(); break;
            case 2: wait_vmcnt<2 * IN>(); break;
            case 3: wait_vmcnt<3 * IN>(); break;
            default: wait_vmcnt<4 * IN>(); break;
        }
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < nk) issue((t + NS - 1) % NS);
        const char *st = ring + (t % NS) * STAGE;
        h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            ah[i] = *reinterpret_cast<const h8 *>(st + a_frag + i * 32 * ROW + pos_hi);
            al[i] = *reinterpret_cast<const h8 *>(st + a_frag + i * 32 * ROW + pos_lo);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const h8 *>(st + b_frag + j * 32 * ROW + pos_hi);
            bl[j] = *reinterpret_cast<const h8 *>(st + b_frag + j * 32 * ROW + pos_lo);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc1[i][j], 0, 0, 0);
                acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
                acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
            }
    }
    __syncthreads();                                            // every wave is done with the ring
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[i][j][e] = (acc1[i][j][e] + acc2[i][j][e] * (1.f / LO_SCALE)) * (1.f / A_SCALE);
    static_assert((BM / WM) * (BN + 4) * 4 <= NS * STAGE, "epilogue staging must fit the ring");
    conv_epilogue<BM, BN, WM, WN, ACT, RES>(p, acc1, reinterpret_cast<float *>(ring), m0, n0, tid);
}

static const void *zero_page_dev() {
    static void *z = nullptr;
    if (!z) {
        YDS_HIP(hipMalloc(&z, 256));
        YDS_HIP(hipMemset(z, 0, 256));
    }
    return z;
}

template <int BM, int BN, int NS, int ACT, int RES> static void launch_inst_dma(ConvKernelArgs k, hipStream_t s) {
    constexpr size_t smem = (size_t)NS * (BM + BN) * 64;
    static bool attr_set = false;
    auto kern = conv_igemm_f16x3_dma<BM, BN, NS, ACT, RES>;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid(plan_tile_map(k, BM, BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, k, zero_page_dev());
    YDS_HIP(hipGetLastError());
}

template <int BM, int BN, int NS> static void launch_cfg_dma(const ConvKernelArgs &k, hipStream_t s) {
    if (k.fmt_x != FMT_H16 || k.Cin % 32) fail("conv: the LDS-DMA kernel needs a pre-split (H16) input");
#define YDS_CALL(A, R) launch_inst_dma<BM, BN, NS, A, R>(k, s)
    YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
}

template <int BM, int BN, int ACT, int RES, int AIN> static void launch_inst16(ConvKernelArgs k, hipStream_t s) {
    constexpr size_t smem = 2ull * (BM + BN) * ROWB;
    static bool attr_set = false;
    auto kern = conv_igemm_f16x3<BM, BN, ACT, RES, AIN>;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid(plan_tile_map(k, BM, BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, k);
    YDS_HIP(hipGetLastError());
}

template <int BM, int BN> static void launch_cfg16(const ConvKernelArgs &k, hipStream_t s) {
    if (k.fmt_x == FMT_H16) {
#define YDS_CALL(A, R) launch_inst16<BM, BN, A, R, FMT_H16>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
    } else {
#define YDS_CALL(A, R) launch_inst16<BM, BN, A, R, FMT_F32>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
    }
}

const char *conv_f16x3_variant_name(int v) {
    static const char *names[kF16Variants] = {"conv_igemm_f16x3<128,128>", "conv_igemm_f16x3<64,128>", "conv_igemm_f16x3<128,64>",
                                              "conv_igemm_f16x3<64,64>", "conv_igemm_f16x3_dma<128,128,5>", "conv_igemm_f16x3_dma<128,128,4>",
                                              "conv_igemm_f16x3_dma<64,128,5>", "conv_igemm_f16x3_dma<64,64,6>"};
    return v >= 0 && v < kF16Variants ? names[v] : "?";
}

void launch_conv_f16x3(ConvKernelArgs k, int variant, hipStream_t s) {
    switch (variant) {
        case 0: launch_cfg16<128, 128>(k, s); break;
        case 1: launch_cfg16<64, 128>(k, s); break;
        case 2: launch_cfg16<128, 64>(k, s); break;
        case 3: launch_cfg16<64, 64>(k, s); break;
        case 4: launch_cfg_dma<128, 128, 5>(k, s); break;
        case 5: launch_cfg_dma<128, 128, 4>(k, s); break;
        case 6: launch_cfg_dma<64, 128, 5>(k, s); break;
        default: launch_cfg_dma<64, 64, 6>(k, s); break;
    }
}

// host: fold-free split of already BN-folded fp32 weights [cout][kpad] into [cout][kpad/32][32 hi | 32 lo] fp16
void pack_weights_f16x3(const float *w, int cout, int kpad, std::vector<uint16_t> &out) {
    out.assign((size_t)cout * kpad * 2, 0);
    for (int o = 0; o < cout; ++o)
        for (int k = 0; k < kpad; ++k) {
            float x = w[(size_t)o * kpad + k];
            if (!(fabsf(x) < 65504.f)) fail("conv: weight %g does not fit the fp16 split (|w| must be < 65504)", (double)x);
            _Float16 h = (_Float16)x;
            _Float16 l = (_Float16)((x - (float)h) * LO_SCALE);
            size_t base = ((size_t)o * (kpad / 32) + k / 32) * 64;
            uint16_t hb, lb;
            memcpy(&hb, &h, 2);
            memcpy(&lb, &l, 2);
            out[base + (k & 31)] = hb;
            out[base + 32 + (k & 31)] = lb;
        }
}

}  // namespace yds
