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

#include <map>
#include <mutex>

#include <math.h>
#include <string.h>

namespace yds {


constexpr int ROWB = 144;                 // bytes per LDS row
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
    fp16_saturate_on();
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;      // 16-byte chunks per thread per K step (32 k)
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char *As = smem16;                                     // [2][BM][ROWB]
    char *Bs = smem16 + 2 * BM * ROWB;                     // [2][BN][ROWB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): LDS-DMA destinations need no v_readfirstlane per piece
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
    static_assert(BM * (BN + 4) * 4 <= 2 * (BM + BN) * ROWB, "whole-tile epilogue staging must fit the main-loop LDS");
    conv_epilogue<BM, BN, WM, WN, ACT, RES, BM / WM / 32, BN / WN / 32, 256, true>(p, acc1, reinterpret_cast<float *>(smem16), m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for pre-split (H16) inputs: both operands are opaque 16-byte chunks, so they go global -> LDS with
// global_load_lds_dwordx4 (no staging VGPRs, no ds_write, no conversion) through an NS-deep ring of K=32 stages.
//   stage  = BM + BN rows of 128 B = one whole [32 hi | 32 lo] group per row (a full cache line per pixel / filter
//            row, 8 rows per wave instruction); row r keeps its 16-byte chunk c at position c ^ ((r >> 1) & 7): the
//            DMA writes lane-linear, so the swizzle is applied to the SOURCE address and again by the fragment reads
//            (a ds_read_b128 lane group then hits 16 distinct 16-byte bank slots)
//   ring   : iteration t waits for tile t (vmcnt = DMAs of the younger tiles), one s_barrier, re-fills the stage
//            that iteration t-1 finished reading, then 24 MFMAs per wave on stage t % NS
//   zero padding: out-of-image taps fetch from a zero page instead of branching
//   WM x WN waves of 64 threads, each owning a 64x64 (TM = TN = 2) or smaller sub-tile.

constexpr int ZERO_PAGE_BYTES = 64 * 1024;                      // >= Cin * 4 + 128 for every layer (checked at launch)

constexpr size_t dma_smem_bytes(int BM, int BN, int NS) {
    return (size_t)NS * (BM + BN) * 128 > conv_stage_bytes(BM, BN) ? (size_t)NS * (BM + BN) * 128 : conv_stage_bytes(BM, BN);
}
template <int BM, int BN, int WM, int WN, int NS, int ACT, int RES, int TERMS>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN != 4 || dma_smem_bytes(BM, BN, NS) > 80 * 1024) ? 1 : (dma_smem_bytes(BM, BN, NS) <= 53 * 1024 ? 3 : 2))
void conv_igemm_f16x3_dma(ConvKernelArgs p, const char *zero_page) {
    fp16_saturate_on();
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int ROW = 128;
    constexpr int A_BYTES = BM * ROW, STAGE = (BM + BN) * ROW;
    constexpr int A_INST = BM / (8 * NW), B_INST = BN / (8 * NW);   // DMA instructions per wave per stage (8 rows each)
    constexpr int IN = A_INST + B_INST;
    static_assert(NS >= 2 && NS <= 4 && A_INST >= 1 && B_INST >= 1, "ring shape");
    extern __shared__ __attribute__((aligned(16))) char ring[];     // [NS][STAGE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): LDS-DMA destinations need no v_readfirstlane per piece
    const int wm = wave / WN, wn = wave % WN;
    int m0, n0;
    {
        int tm, tn;
        if (!tile_of_block(p, tm, tn)) return;
        m0 = tm * BM;
        n0 = tn * BN;
    }
    // DMA lane roles: instruction q of this wave fills rows (q*NW + wave)*8 .. +7; lane -> (row, 16-byte position)
    const int drow = lane >> 3, dpos = lane & 7;
    int a_base[A_INST], a_iy[A_INST], a_ix[A_INST];
    const char *a_src[A_INST];                                  // this tap: group-0 address of the lane's chunk, or the zero page
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int q = 0; q < A_INST; ++q) {
        const int row = (q * NW + wave) * 8 + drow;
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
    // logical chunk fetched by this lane: position dpos of row r holds chunk dpos ^ ((r >> 1) & 7); r % 8 == drow for
    // every instruction (row bases are multiples of 8) and bit 3 of r is bit 0 of (q*NW + wave)
    // TERMS = 4 (half mode, 64 channels per K step): an LDS row is gathered from the hi halves of two consecutive channel groups -
    // logical chunks 0-3 = chunks 0-3 of group 2q's record, 4-7 = chunks 0-3 of group 2q+1's (see conv_win.hip)
    // An FMT_F16 activation tensor (2-byte half-mode activations, TERMS = 4 only) already IS that gathered row: 64 channels = 128
    // contiguous bytes, chunk c in place; its strides are in float slots, so only the byte step per channel differs (xbpc).
    constexpr int KC = TERMS == 4 ? 64 : 32;                    // channels per K step
    const bool xs = TERMS == 4 && p.fmt_x == FMT_F16;
    const int xbpc = xs ? 2 : 4;                                // bytes per channel of the activation tensor
    auto src_chunk = [&](int q, bool act) {                     // act: activation row (filter rows always come from the split record)
        const int c = dpos ^ ((((q * NW + wave) * 8 + drow) >> 1) & 7);
        return ((TERMS == 4 && !(act && xs)) ? ((c >> 2) << 3) + (c & 3) : c) * 16;
    };
    const char *w_src[B_INST];
#pragma unroll
    for (int q = 0; q < B_INST; ++q) {
        const int row = (q * NW + wave) * 8 + drow;
        w_src[q] = reinterpret_cast<const char *>(p.w) + (size_t)min(n0 + row, p.Cout - 1) * p.Kpad * 4 + src_chunk(q, false);
    }
    // split-K (gridDim.y > 1): this workgroup accumulates K tiles [t_begin, t_begin + nk) only and writes raw fp32 partial
    // sums into slab blockIdx.y of the workspace p.y (the launcher passes zero bias / linear / no residual / fp32 output);
    // splitk_reduce_kernel adds the slabs in a fixed order and applies the real epilogue.
    const int nk_all = p.K / KC;                                // Cin % KC == 0 on this path
    int t_begin = 0, nk = nk_all;
    if (gridDim.y > 1) {
        const int per = (nk_all + gridDim.y - 1) / gridDim.y;
        t_begin = blockIdx.y * per;
        nk = max(0, min(nk_all, t_begin + per) - t_begin);
        p.y += (size_t)blockIdx.y * (size_t)p.M * p.ldy;
    }
    const int kcs = p.Cin / KC;
    int kh = (t_begin / kcs) / p.ksize, kw = (t_begin / kcs) % p.ksize, kc = (t_begin % kcs) * KC;   // (tap, channel group) of the next tile to issue
#pragma unroll
    for (int q = 0; q < B_INST; ++q) w_src[q] += (size_t)t_begin * (KC * 4);
    auto set_tap = [&]() {
#pragma unroll
        for (int q = 0; q < A_INST; ++q) {
            int iy = a_iy[q] + kh, ix = a_ix[q] + kw;
            bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            a_src[q] = ok ? reinterpret_cast<const char *>(p.x + (a_base[q] + (kh * p.W + kw) * p.ldx)) + src_chunk(q, true) : zero_page;
        }
    };
    set_tap();
    // one DMA instruction of the tile being issued: pieces 0..A_INST-1 fetch activation rows, the rest filter rows
    auto piece = [&](int stage, int q) {
        char *sa = ring + stage * STAGE, *sb = sa + A_BYTES;
        if (q < A_INST) {
            __builtin_amdgcn_global_load_lds((glb_void_t *)(a_src[q] + kc * xbpc), (lds_void_t *)(sa + (q * NW + wave) * 8 * ROW), 16, 0, 0);
        } else {
            const int b = q - A_INST;
            __builtin_amdgcn_global_load_lds((glb_void_t *)w_src[b], (lds_void_t *)(sb + (b * NW + wave) * 8 * ROW), 16, 0, 0);
            w_src[b] += KC * 4;
        }
    };
    auto advance_tile = [&]() {
        kc += KC;
        if (kc >= p.Cin) { kc = 0; if (++kw == p.ksize) { kw = 0; ++kh; } set_tap(); }
    };

    f32x16 acc1[TM][TN], acc2[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc1[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) {
#pragma unroll
            for (int q = 0; q < IN; ++q) piece(s, q);
            advance_tile();
        }
    if (STAGGER && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_sleep(STAGGER / 4);

    // fragment addressing: row = lane & 31 (+ tile offsets, multiples of 32); chunk ids inside a row: hi k 0-7 / 8-15 /
    // 16-23 / 24-31 = 0..3, lo = 4..7; k-substep s, lane half kb -> chunks 2s+kb (hi) and 4+2s+kb (lo)
    const int swz = (lane >> 1) & 7, kb = lane >> 5;
    const int a_frag = (wm * (BM / WM) + (lane & 31)) * ROW, b_frag = A_BYTES + (wn * (BN / WN) + (lane & 31)) * ROW;
    int pos_hi[2], pos_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { pos_hi[s] = ((2 * s + kb) ^ swz) * 16; pos_lo[s] = ((4 + 2 * s + kb) ^ swz) * 16; }

    // The wave issues in order, so memory instructions are placed BETWEEN the MFMAs of a k-substep (each MFMA keeps the
    // matrix pipe busy for 32 cycles, which pays for one LDS read or one DMA piece): substep 0 carries the fragment
    // reads of substep 1 and the first half of the next tile's DMA pieces, substep 1 the second half.
    // sched_barrier(0) pins that order; fragments are double buffered in registers.
    constexpr int NF = 2 * (TM + TN);                           // fragment reads per substep
    constexpr int NM = 3 * TM * TN;                             // MFMAs per substep
    h8 fr[2][NF];                                               // [substep][ah.., al.., bh.., bl..]
    auto frag_read = [&](const char *st, int s, int f) {
        const int which = f / 2, lo = f & 1;                    // f = 2*tile + (hi|lo), A tiles first
        if (TERMS == 1 && lo) return;                           // half mode: hi halves only
        const int off = which < TM ? a_frag + which * 32 * ROW : b_frag + (which - TM) * 32 * ROW;
        fr[s][f] = *reinterpret_cast<const h8 *>(st + off + (lo ? pos_lo[s] : pos_hi[s]));
    };
    auto mfma = [&](int s, int m) {
        const int ij = m / 3, term = m % 3, i = ij / TN, j = ij % TN;
        if (TERMS == 1 && term != 0) return;
        const h8 ah = fr[s][2 * i], al = fr[s][2 * i + 1], bh = fr[s][2 * (TM + j)], bl = fr[s][2 * (TM + j) + 1];
        if (TERMS == 4) {                                        // both slots hold hi values (of two channel groups)
            if (term == 0) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[i][j], 0, 0, 0);
            else if (term == 1) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc1[i][j], 0, 0, 0);
            return;
        }
        if (term == 0) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[i][j], 0, 0, 0);
        else if (term == 1) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[i][j], 0, 0, 0);
        else acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[i][j], 0, 0, 0);
    };
    // slot plan (one memory instruction after each MFMA): substep 0 first issues the DMA pieces of the next tile (as
    // early as possible: they have to land before the next step), then the first fragments substep 1 needs; substep 1
    // starts with the MFMAs of tile (0,0) and reads the fragments of the other tiles in its first slots.
    constexpr int P0 = IN < NM - 4 ? IN : NM - 4;               // pieces issued in substep 0 (4 slots stay for fragments)
    constexpr int F0 = NM - P0 < NF ? NM - P0 : NF;             // substep-1 fragments read during substep 0
    static_assert(IN - P0 + NF - F0 <= NM, "not enough MFMA slots for the interleaved memory instructions");
    // fragment order for substep 1: ah0, al0, bh0, bl0 (tile (0,0)), then bh1, bl1, ..., then ah1, al1, ...
    auto frag_order = [&](int k) {
        if (k < 2) return k;                                    // ah0, al0
        if (k < 4) return 2 * TM + (k - 2);                     // bh0, bl0
        const int r = k - 4, nb = 2 * (TN - 1);
        return r < nb ? 2 * TM + 2 + r : 2 + (r - nb);          // remaining B tiles, then remaining A tiles
    };
    // one K step; REFILL (compile time) = a further tile exists and is fetched into the stage freed by step t-1
    auto step = [&](int t, auto refill_c) {
        constexpr bool REFILL = decltype(refill_c)::value;
        const char *st = ring + (t % NS) * STAGE;
        const int fill_stage = (t + NS - 1) % NS;
#pragma unroll
        for (int f = 0; f < NF; ++f) frag_read(st, 0, f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma(0, m);
            __builtin_amdgcn_sched_barrier(0);
            if (m < P0) { if (REFILL) piece(fill_stage, m); }
            else if (m - P0 < F0) frag_read(st, 1, frag_order(m - P0));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma(1, m);
            __builtin_amdgcn_sched_barrier(0);
            if (m < NF - F0) frag_read(st, 1, frag_order(F0 + m));
            else if (REFILL && P0 + (m - (NF - F0)) < IN) piece(fill_stage, P0 + (m - (NF - F0)));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (REFILL) advance_tile();
    };
    int t = 0;
    for (; t + NS - 1 < nk; ++t) {                             // steady state: NS-2 younger tiles stay in flight
        wait_vmcnt<(NS - 2) * IN>();
        __builtin_amdgcn_s_barrier();
        step(t, std::true_type{});
    }
    for (; t < nk; ++t) {                                       // drain: nothing left to fetch
        const int younger = nk - 1 - t;
        if (younger >= 1 && NS >= 3) wait_vmcnt<(NS >= 3 ? IN : 0)>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        step(t, std::false_type{});
    }
    __syncthreads();                                            // every wave is done with the ring
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc1[i][j][e] = TERMS != 3 ? acc1[i][j][e] * (1.f / A_SCALE) : (acc1[i][j][e] + acc2[i][j][e] * (1.f / LO_SCALE)) * (1.f / A_SCALE);
    // whole-tile staging: the launcher sizes the LDS as max(ring, BM x (BN+4) floats)
    conv_epilogue<BM, BN, WM, WN, ACT, RES, TM, TN, NT, true>(p, acc1, reinterpret_cast<float *>(ring), m0, n0, tid);
}

static const char *zero_page_dev() {
    static void *z = nullptr;
    if (!z) {
        YDS_HIP(hipMalloc(&z, ZERO_PAGE_BYTES));
        YDS_HIP(hipMemset(z, 0, ZERO_PAGE_BYTES));
    }
    return static_cast<const char *>(z);
}

template <int BM, int BN, int WM, int WN, int NS, int ACT, int RES, int TERMS> static void launch_inst_dma(ConvKernelArgs k, hipStream_t s) {
    constexpr size_t smem = dma_smem_bytes(BM, BN, NS);
    static bool attr_set = false;
    auto kern = conv_igemm_f16x3_dma<BM, BN, WM, WN, NS, ACT, RES, TERMS>;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid(plan_tile_map(k, BM, BN), k.ksplit > 1 ? k.ksplit : 1);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, s, k, zero_page_dev());
    YDS_HIP(hipGetLastError());
}

template <int BM, int BN, int WM, int WN, int NS> static void launch_cfg_dma(const ConvKernelArgs &k, hipStream_t s) {
    const bool xs = k.fmt_x == FMT_F16;                          // 2-byte activations (half mode): the 64-channel K steps read them in place
    if (xs ? (k.terms != 1 || k.Cin % 64) : (k.fmt_x != FMT_H16 || k.Cin % 32)) fail("conv: the LDS-DMA kernel needs a pre-split (H16, or F16 in half mode) input");
    if ((size_t)k.Cin * 4 + 128 > (size_t)ZERO_PAGE_BYTES) fail("conv: %d input channels exceed the zero page of the LDS-DMA kernel", k.Cin);
    if (k.terms == 1 && k.Cin % 64 == 0 && (xs || !getenv("YDS_HALF_NARROW"))) {      // half mode, 64 channels per K step
#define YDS_CALL(A, R) launch_inst_dma<BM, BN, WM, WN, NS, A, R, 4>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
        return;
    }
    if (k.terms == 1) {
#define YDS_CALL(A, R) launch_inst_dma<BM, BN, WM, WN, NS, A, R, 1>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
        return;
    }
#define YDS_CALL(A, R) launch_inst_dma<BM, BN, WM, WN, NS, A, R, 3>(k, s)
    YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
}

// ---------------------------------------------------------------------------------------------------------------
// Split-K (small batches: the frame-by-frame API runs the detector on ONE image, where a 19x19 layer has 24-48 tiles for
// 256 CUs and every workgroup streams its whole filter slice - megabytes - through a ring that holds two K steps; the
// launch is bound by the latency of that stream, not by the matrix cores).  The K tiles are cut into `ksplit` ranges,
// grid.y = ksplit workgroups per output tile accumulate one range each (more independent filter streams in flight) and
// write raw fp32 partial sums into a workspace slab; splitk_reduce_kernel adds the slabs in slab order (deterministic:
// the same bits on every run) and applies bias, activation, residual and the output format.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *ws, int ksplit, int M, int ldw, ConvKernelArgs p) {
    fp16_saturate_on();
    const int c4 = (p.Cout + 3) / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * c4) return;
    const int m = idx / c4, n = (idx - m * c4) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < ksplit; ++z) {
        const float4 t = *reinterpret_cast<const float4 *>(ws + ((size_t)z * M + m) * ldw + n);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = n + 3 < p.Cout;
    if (p.res_mode != RES_NONE) {
        const float *rp = p.res + (size_t)m * p.ldr;
        if (p.fmt_r == FMT_H16 || full) load4(rp, n, p.fmt_r, r);
        else for (int k = 0; k < 4; ++k) if (n + k < p.Cout) r[k] = rp[n + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float o = v[k] + (n + k < p.Cout ? p.bias[n + k] : 0.f);
        if (p.res_mode == RES_BEFORE_ACT) o += r[k];
        o = p.act == ACT_LEAKY ? apply_act<ACT_LEAKY>(o) : p.act == ACT_MISH ? apply_act<ACT_MISH>(o) : p.act == ACT_RELU ? apply_act<ACT_RELU>(o) : o;
        if (p.res_mode == RES_AFTER_ACT) o += r[k];
        v[k] = o;
    }
    float *yp = p.y + (size_t)m * p.ldy;
    if (p.fmt_y == FMT_H16 || full) store4(yp, n, p.fmt_y, v);
    else for (int k = 0; k < 4; ++k) if (n + k < p.Cout) yp[n + k] = v[k];
}

// number of K ranges for a BM x BN tiling: enough workgroups for two per CU, at least four K tiles per range
int conv_splitk_factor(const ConvKernelArgs &k, int BM, int BN) {
    const long tiles = (long)((k.M + BM - 1) / BM) * ((k.Cout + BN - 1) / BN);
    const int nk = k.K / 32;
    int s = (int)((512 + tiles - 1) / tiles);
    s = std::min(s, std::min(16, nk / 4));
    return s >= 2 ? s : 1;
}

static float *splitk_workspace(size_t floats, hipStream_t s) {
    // one grow-only slab set per stream (the detector and the ReID network run concurrently on their own streams)
    // (handles may be driven from different host threads: the map is guarded; a slab is only ever used on its own stream)
    static std::mutex mu;
    static std::map<hipStream_t, DevBuf<float>> bufs;
    std::lock_guard<std::mutex> lock(mu);
    DevBuf<float> &b = bufs[s];
    if (b.n < floats) {
        YDS_HIP(hipStreamSynchronize(s));
        b.alloc(floats + floats / 2);
    }
    return b.p;
}

template <int BM, int BN> static void launch_splitk(const ConvKernelArgs &k, hipStream_t s) {
    if (k.fmt_x != FMT_H16 || k.Cin % 32) fail("conv: the split-K LDS-DMA kernel needs a pre-split (H16) input");
    if ((size_t)k.Cout * 4 > (size_t)ZERO_PAGE_BYTES) fail("conv: %d filters exceed the zero page used as the split-K bias", k.Cout);
    const int ksplit = conv_splitk_factor(k, BM, BN);
    if (ksplit < 2) fail("conv: split-K does not apply (the layer already has enough tiles or too few K steps)");
    const int ldw = (k.Cout + 3) / 4 * 4;
    float *ws = splitk_workspace((size_t)ksplit * k.M * ldw, s);
    ConvKernelArgs part = k;
    part.y = ws; part.ldy = ldw; part.fmt_y = FMT_F32;
    part.bias = reinterpret_cast<const float *>(zero_page_dev());
    part.res = nullptr; part.res_mode = RES_NONE; part.fmt_r = FMT_F32; part.act = ACT_LINEAR;
    part.ksplit = ksplit;
    launch_cfg_dma<BM, BN, 2, 2, 2>(part, s);
    const int c4 = (k.Cout + 3) / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(((size_t)k.M * c4 + 255) / 256), dim3(256), 0, s, ws, ksplit, k.M, ldw, k);
    YDS_HIP(hipGetLastError());
}

// Split-K is chosen by RULE, not by the autotuner's stopwatch: it changes the fp32 summation order, so a timing-dependent
// choice would let the low bits of a result vary from run to run.  Few tiles (less than half the chip at 64 x 128) and a
// long K (>= 32 steps): the 19x19 / 38x38 layers of a one-image detector pass, the last ReID stages of a small crop batch.
bool conv_splitk_preferred(const ConvKernelArgs &k) {
    if (k.fmt_x != FMT_H16 || k.Cin % 32) return false;
    const long tiles = (long)((k.M + 63) / 64) * ((k.Cout + 127) / 128);
    return tiles <= 128 && k.K / 32 >= 32 && conv_splitk_factor(k, 64, 128) >= 2;
}

bool conv_splitk_applicable(const ConvKernelArgs &k, int fv) {
    if (k.fmt_x != FMT_H16 || k.Cin % 32) return false;
    return fv == 14 ? conv_splitk_factor(k, 64, 128) >= 2 : conv_splitk_factor(k, 128, 128) >= 2;
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
    if (k.fmt_x == FMT_F16) fail("conv: the register-staged kernel does not read 2-byte (F16) activations");
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
                                              "conv_igemm_f16x3<64,64>", "conv_igemm_f16x3_dma<128,128,2x2,2>", "conv_igemm_f16x3_dma<256,128,4x2,3>",
                                              "conv_igemm_f16x3_dma<128,256,2x4,3>", "conv_igemm_f16x3_dma<128,128,2x2,3>", "conv3x3_f16x3_win<256,128,4x2>",
                                              "conv3x3_f16x3_win<256,64,8x1>", "conv3x3_f16x3_win<128,64,4x1>",
                                              "conv_igemm_f16x3_dma<128,64,2x2,2>", "conv_igemm_f16x3_dma<64,128,2x2,2>", "conv3x3_f16x3_win2<128,128,2x2>",
                                              "conv_igemm_f16x3_dma<64,128,2x2,2>+splitK", "conv_igemm_f16x3_dma<128,128,2x2,2>+splitK"};
    return v >= 0 && v < kF16Variants ? names[v] : "?";
}

void launch_conv_f16x3(ConvKernelArgs k, int variant, hipStream_t s) {
    switch (variant) {
        case 0: launch_cfg16<128, 128>(k, s); break;
        case 1: launch_cfg16<64, 128>(k, s); break;
        case 2: launch_cfg16<128, 64>(k, s); break;
        case 3: launch_cfg16<64, 64>(k, s); break;
        case 4: launch_cfg_dma<128, 128, 2, 2, 2>(k, s); break;
        case 5: launch_cfg_dma<256, 128, 4, 2, 3>(k, s); break;
        case 6: launch_cfg_dma<128, 256, 2, 4, 3>(k, s); break;
        case 7: launch_cfg_dma<128, 128, 2, 2, 3>(k, s); break;
        case 8: launch_conv_win(k, 0, s); break;
        case 9: launch_conv_win(k, 1, s); break;
        case 10: launch_conv_win(k, 2, s); break;
        // small LDS-DMA tiles (48 KB of LDS: three workgroups per CU): layers whose 128x128 tile count leaves a long tail -
        // the 1x1 layers at 76^2 / 38^2 / 19^2 run 1.4 rounds of 128x128 tiles on 512 slots, i.e. pay for 2
        case 11: launch_cfg_dma<128, 64, 2, 2, 2>(k, s); break;
        case 13: launch_conv_win2(k, s); break;
        case 14: launch_splitk<64, 128>(k, s); break;
        case 15: launch_splitk<128, 128>(k, s); break;
        default: launch_cfg_dma<64, 128, 2, 2, 2>(k, s); break;
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
