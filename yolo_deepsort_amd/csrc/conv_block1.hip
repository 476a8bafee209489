// Fused first residual block of yolov3 (config/yolov3.cfg layers 2-4; yolo3/models/models.py:36-56, shortcut :83-85,:304-306):
//   h = leaky(bn(conv1x1 64 -> 32 (x)))     y = leaky(bn(conv3x3 32 -> 64 (h))) + x        at 304 x 304
// Unfused, the 32-channel tensor h (189 MB per 16 frames) is written by one launch and read back nine taps deep by the
// next, and both launches have only 2 / 9 K steps per tile, so they are all prologue and epilogue (141 + 391 us).  Here a
// 512-thread persistent workgroup owns an 8 x 16 patch of outputs, like the fused stem (conv_stem2.hip):
//   load     the 10 x 18 input pixels (64 channels = two pre-split 128-byte groups each) -> LDS, zeros outside the image
//   phase A  h for those 180 pixels: f16x3 MFMAs with the 1x1 filters as register-resident first operand (swapped operands:
//            a lane gets 4 consecutive channels of one pixel), bias + activation, zero outside the image (the 3x3 pads with
//            zeros), pre-split rows -> LDS patch
//   phase B  the 3x3 as 9 taps x 2 k-substeps from the patch; its 64 x 288 filters stay in LDS for the kernel's lifetime
//   epilogue shared conv_epilogue with the residual add (x is re-read from L2 for the 128 centre pixels)
#include "conv_common.h"

#include <algorithm>

namespace yds {

namespace {

constexpr int TH = 8, TW = 16;                      // output patch
constexpr int BM = TH * TW, BN = 64, WM = 4, WN = 2, NW = 8, NT = NW * 64;
constexpr int PR = TH + 2, PC = TW + 2, PPIX = PR * PC;     // 10 x 18 = 180 patch pixels
constexpr int CMID = 32, CIN = 64, GIN = CIN / 32;
constexpr int W3_BYTES = 9 * BN * 128, XP_BYTES = GIN * PPIX * 128, HP_BYTES = PPIX * 128;
constexpr int SMEM = W3_BYTES + XP_BYTES + HP_BYTES;
constexpr int XCHUNKS = GIN * PPIX * 8, XLOADS = (XCHUNKS + NT - 1) / NT;

struct PatchRows {                                  // tile row -> flat output pixel
    int img, oy0, ox0, H, W;
    __device__ __forceinline__ int operator()(int row) const {
        const int oy = oy0 + row / TW, ox = ox0 + row % TW;
        return oy < H && ox < W ? (img * H + oy) * W + ox : -1;
    }
};

// HALF (Darknet.half(), round 4): single-term fp16 operands in both convolutions, hi halves only in the patch (as in conv_stem2.hip)
template <int ACT, bool HALF>
__global__ __launch_bounds__(NT, 1) void conv_block1_f16x3(ConvKernelArgs p2, ConvKernelArgs p3, int tiles_y, int tiles_x, int n_tiles) {
    fp16_saturate_on();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *w3 = smem, *xp = smem + W3_BYTES, *hp = smem + W3_BYTES + XP_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN, kb = lane >> 5;
    const int H = p2.H, W = p2.W;

    // 3x3 filters -> LDS once: row (tap, cout) at (tap * 64 + cout) * 128, chunk c at position c ^ ((row >> 1) & 7)
    for (int i = tid; i < 9 * BN * 8; i += NT) {
        const int row = i >> 3, pos = i & 7, tap = row / BN, co = row - tap * BN;
        const int c = pos ^ ((row >> 1) & 7);
        *reinterpret_cast<f32x4 *>(w3 + row * 128 + pos * 16) =
            *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p3.w) + ((size_t)co * 9 + tap) * 128 + c * 16);
    }
    // 1x1 filter fragments (first MFMA operand: row = mid channel lane & 31; k-substep sb covers input channels
    // 16 sb + 8 kb + 0..7, i.e. chunk (2 (sb & 1) + kb) of group sb >> 1) and the biases of this lane's 16 accumulator rows
    h8 w2h[4], w2l[4];
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
        const char *g = reinterpret_cast<const char *>(p2.w) + ((size_t)(lane & 31) * GIN + (sb >> 1)) * 128;
        w2h[sb] = *reinterpret_cast<const h8 *>(g + (2 * (sb & 1) + kb) * 16);
        w2l[sb] = *reinterpret_cast<const h8 *>(g + 64 + (2 * (sb & 1) + kb) * 16);
    }
    float bias2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bias2[e] = p2.bias[(e & 3) + 8 * (e >> 2) + 4 * kb] * A_SCALE;      // in the accumulators' scaled domain (exact: a power of two)
    // phase-B fragment bookkeeping
    const int r = wm * 32 + (lane & 31), py = r / TW, px = r - py * TW;
    const int brow = wn * 32 + (lane & 31);

    // the next tile's input pixels are fetched into registers while this tile is computed
    // (half mode with 2-byte activations: the block input is an FMT_F16 tensor - a pixel's 64 channels are 128 contiguous bytes; its
    //  eight 16-byte pieces become the hi chunks 0-3 of the two LDS rows, the lo chunks are zero, phase A runs unchanged)
    const bool xs = p2.fmt_x == FMT_F16;
    f32x4 nxt[XLOADS];
    auto fetch = [&](int tl) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int iy0 = (rem / tiles_x) * TH - 1, ix0 = (rem % tiles_x) * TW - 1;
#pragma unroll
        for (int l = 0; l < XLOADS; ++l) {
            const int i = tid + l * NT, c = i & 7, row = i >> 3, g = row / PPIX, pr = row - g * PPIX;
            const int iy = iy0 + pr / PC, ix = ix0 + pr % PC;
            nxt[l] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < XCHUNKS && tl < n_tiles && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && !(xs && c >= 4))
                nxt[l] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p2.x + ((size_t)(img * H + iy) * W + ix) * p2.ldx) +
                                                          (xs ? (g * 4 + c) * 16 : g * 128 + c * 16));
        }
    };
    fetch(blockIdx.x);
    for (int tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int oy0 = (rem / tiles_x) * TH, ox0 = (rem % tiles_x) * TW;
        __syncthreads();                                        // previous tile: patches read, epilogue staging consumed
#pragma unroll
        for (int l = 0; l < XLOADS; ++l) {
            const int i = tid + l * NT, c = i & 7, row = i >> 3;
            if (i < XCHUNKS) *reinterpret_cast<f32x4 *>(xp + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = nxt[l];
        }
        __syncthreads();
        fetch(tl + gridDim.x);
        // ---- phase A: the 1x1 conv for the 180 patch pixels, 6 fragments of 32 pixels (waves 0-5)
        if (wave < (PPIX + 31) / 32) {
            const int pix = wave * 32 + (lane & 31), pr = min(pix, PPIX - 1);
            f32x16 c1, c2;
#pragma unroll
            for (int e = 0; e < 16; ++e) { c1[e] = 0.f; c2[e] = 0.f; }
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
                const int row = (sb >> 1) * PPIX + pr, sw = (row >> 1) & 7;
                const char *ap = xp + row * 128;
                const h8 xh = *reinterpret_cast<const h8 *>(ap + (((2 * (sb & 1) + kb) ^ sw) << 4));
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[sb], xh, c1, 0, 0, 0);
                if (!HALF) {
                    const h8 xl = *reinterpret_cast<const h8 *>(ap + (((4 + 2 * (sb & 1) + kb) ^ sw) << 4));
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[sb], xl, c2, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l[sb], xh, c2, 0, 0, 0);
                }
            }
            const int ry = pr / PC, rc = pr - ry * PC;
            const bool inside = (unsigned)(oy0 - 1 + ry) < (unsigned)H && (unsigned)(ox0 - 1 + rc) < (unsigned)W;
            const int sw = (pr >> 1) & 7;
            // (round 6, as in conv_stem2.hip: bias and LeakyReLU in the scaled domain, encoded from there; pixels outside the image - the
            //  3x3 zero-pads h - take a branch of their own instead of a select per value)
            if (pix < PPIX) {
                if (inside) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int e = g * 4 + c;
                            const float os = (HALF ? c1[e] : c1[e] + c2[e] * (1.f / LO_SCALE)) + bias2[e];
                            v[c] = ACT == ACT_LEAKY ? fmaxf(os, os * 0.1f) : apply_act<ACT>(os * (1.f / A_SCALE)) * A_SCALE;
                        }
                        h16x4 hi, lo;
                        if (HALF) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) hi[c] = (_Float16)v[c];
                        } else {
                            h16_encode4_scaled(v, hi, lo);
                        }
                        *reinterpret_cast<h16x4 *>(hp + pr * 128 + ((g ^ sw) << 4) + kb * 8) = hi;
                        if (!HALF) *reinterpret_cast<h16x4 *>(hp + pr * 128 + (((4 + g) ^ sw) << 4) + kb * 8) = lo;
                    }
                } else {
                    const h16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
                    for (int g = 0; g < (HALF ? 4 : 8); ++g) *reinterpret_cast<h16x4 *>(hp + pr * 128 + ((g ^ sw) << 4) + kb * 8) = z;
                }
            }
        }
        __syncthreads();
        // ---- phase B: the 3x3 conv, 9 taps x 2 k-substeps on the patch
        f32x16 acc1[1][1], acc2[1][1];
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc1[0][0][e] = 0.f; acc2[0][0][e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int j = (py + t / 3) * PC + px + t % 3;
            const int jsw = (j >> 1) & 7, wrow = t * BN + brow, wsw = (wrow >> 1) & 7;
            const char *ap = hp + j * 128, *bp = w3 + wrow * 128;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const h8 ah = *reinterpret_cast<const h8 *>(ap + (((2 * s + kb) ^ jsw) << 4));
                const h8 bh = *reinterpret_cast<const h8 *>(bp + (((2 * s + kb) ^ wsw) << 4));
                acc1[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[0][0], 0, 0, 0);
                if (!HALF) {
                    const h8 al = *reinterpret_cast<const h8 *>(ap + (((4 + 2 * s + kb) ^ jsw) << 4));
                    const h8 bl = *reinterpret_cast<const h8 *>(bp + (((4 + 2 * s + kb) ^ wsw) << 4));
                    acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[0][0], 0, 0, 0);
                    acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[0][0], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[0][0][e] = (HALF ? acc1[0][0][e] : acc1[0][0][e] + acc2[0][0][e] * (1.f / LO_SCALE)) * (1.f / A_SCALE);
        // whole-tile staging (128 x 68 floats) in the input patch, which phase A is done with: one barrier pair, not four passes (round 6)
        conv_epilogue_rows<BM, BN, WM, WN, ACT, RES_AFTER_ACT, 1, 1, NT, PatchRows, true>(p3, acc1, reinterpret_cast<float *>(xp),
                                                                                           PatchRows{img, oy0, ox0, H, W}, 0, tid);
    }
}

template <int ACT, bool HALF> void launch_inst(const ConvKernelArgs &k2, const ConvKernelArgs &k3, hipStream_t s) {
    static_assert(BM * (BN + 4) * 4 <= XP_BYTES, "whole-tile epilogue staging must fit the input patch");
    const int n_img = k2.M / (k2.H * k2.W);
    const int tiles_y = (k2.H + TH - 1) / TH, tiles_x = (k2.W + TW - 1) / TW, n_tiles = n_img * tiles_y * tiles_x;
    static bool attr_set = false;
    auto kern = conv_block1_f16x3<ACT, HALF>;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(std::min(n_tiles, 256)), dim3(NT), SMEM, s, k2, k3, tiles_y, tiles_x, n_tiles);
    YDS_HIP(hipGetLastError());
}

}  // namespace

bool conv_block1_applicable(const ConvKernelArgs &k2, const ConvKernelArgs &k3) {
    return k2.Cin == CIN && k2.ksize == 1 && k2.stride == 1 && k2.Cout == CMID && k2.res_mode == RES_NONE && (k2.fmt_x == FMT_H16 || k2.fmt_x == FMT_F16) &&
           k3.Cin == CMID && k3.ksize == 3 && k3.stride == 1 && k3.pad == 1 && k3.Cout == BN && k3.res_mode == RES_AFTER_ACT &&
           k3.fmt_r == k2.fmt_x && k3.fmt_y == k2.fmt_x && k3.res == k2.x && k3.ldr == k2.ldx && k3.H == k2.H && k3.W == k2.W &&
           k2.act == k3.act && (k2.act == ACT_LEAKY || k2.act == ACT_MISH);
}

// k2 / k3: the two layers' own arguments (both .w = their pre-split f16x3 weights); k2's output tensor is never written
void launch_conv_block1(const ConvKernelArgs &k2, const ConvKernelArgs &k3, hipStream_t s) {
    if (!conv_block1_applicable(k2, k3)) fail("conv: the fused residual block takes conv1x1 64->32 + conv3x3 32->64 + shortcut to the block input");
    const bool half = k3.terms == 1;                             // Darknet.half(): single-term instantiation
    if (k2.act == ACT_LEAKY) { if (half) launch_inst<ACT_LEAKY, true>(k2, k3, s); else launch_inst<ACT_LEAKY, false>(k2, k3, s); }
    else { if (half) launch_inst<ACT_MISH, true>(k2, k3, s); else launch_inst<ACT_MISH, false>(k2, k3, s); }
}

}  // namespace yds
