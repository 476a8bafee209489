// Fused detector stem: layer 0 (3x3 / stride 1, RGB -> 32 channels) and layer 1 (3x3 / stride 2, 32 -> 64) in one kernel.
// yolov3 / yolov4: yolo3/models/models.py:36-56 applied to the first two [convolutional] blocks of config/yolov3.cfg /
// yolov4.cfg.
//
// Unfused, layer 0 writes 608*608*32 values per image (757 MB per 16 frames) that layer 1 reads straight back: the two
// launches are HBM bound on that tensor (322 + 349 us per 16 frames, ~340 us of it for the round trip).  Here a 512-thread
// persistent workgroup owns an 8 x 16 patch of LAYER-1 outputs:
//   phase A  the 17 x 33 layer-0 pixels it needs are computed as an f16x3 MFMA product too (K = 9 taps x 4 = 36 -> 48): the
//            filter fragments (32 x 48, split on the fly) live in registers, the pixel fragments are gathered from a split
//            RGB tile in LDS ([hi r g b 0 | lo r g b 0] per pixel, two 8-byte reads per 8 k), operands swapped so that a lane
//            ends up with 4 consecutive channels of ONE pixel, which go to LDS as pre-split H16 rows (128 B = [32 hi | 32 lo]
//            fp16 per pixel, 8-byte writes); pixels outside the image are written as zeros (layer 1 pads).  (A vector-ALU
//            version of this phase cost 255 of 451 us: 27-deep fma chains at two waves per SIMD.)
//   phase B  layer 1 as 9 taps x 2 k-substeps of f16x3 MFMAs (3 per 32x32 tile and substep) straight from that LDS patch;
//            the 64 x 288 filter matrix (74 KB pre-split) is loaded into LDS once per workgroup
//   epilogue shared conv_epilogue (bias, activation, H16 encode, 16-byte stores) with a patch row map
// Stride 2 makes neighbouring output pixels read every other patch column, which would put a 16-lane ds_read_b128 group on
// 8 bank slots; the patch is therefore stored as two column-parity planes (even columns: 17 x 17 rows, odd: 17 x 16), so a
// tap reads CONSECUTIVE rows of one plane (dx = 0, 2: even plane at column px + dx/2; dx = 1: odd plane at px).
// Results equal the unfused pair up to the summation order inside the MFMAs.
#include "conv_common.h"

#include <algorithm>


namespace yds {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 8, TW = 16;                      // layer-1 output patch
constexpr int BM = TH * TW, BN = 64, WM = 4, WN = 2, NW = 8, NT = NW * 64;
constexpr int PR = 2 * TH + 1, PC = 2 * TW + 1;     // layer-0 patch 17 x 33
constexpr int EVEN_COLS = TW + 1, ODD_COLS = TW;    // column-parity planes
constexpr int ODD_BASE = PR * EVEN_COLS;            // 289
constexpr int PATCH_ROWS = PR * PC;                 // 561
constexpr int RR = PR + 2, RC = PC + 2;             // RGB tile 19 x 35
constexpr int W_BYTES = 9 * BN * 128, PATCH_BYTES = PATCH_ROWS * 128, RGB_BYTES = RR * RC * 16;
constexpr int SMEM = W_BYTES + PATCH_BYTES + RGB_BYTES;

struct StemRows {                                   // tile row -> flat layer-1 output pixel
    int img, oy0, ox0, Ho, Wo;
    __device__ __forceinline__ int operator()(int row) const {
        const int oy = oy0 + row / TW, ox = ox0 + row % TW;
        return oy < Ho && ox < Wo ? (img * Ho + oy) * Wo + ox : -1;
    }
};

__device__ __forceinline__ int patch_row(int ry, int rc) {      // LDS row of layer-0 patch pixel (ry, rc)
    return (rc & 1) ? ODD_BASE + ry * ODD_COLS + (rc >> 1) : ry * EVEN_COLS + (rc >> 1);
}

// HALF (Darknet.half(), round 4): single-term fp16 operands in both layers - one MFMA per product block instead of three, the
// patch rows carry hi halves only (no lo encode in phase A: 7 instead of 14 vector instructions per layer-0 value).
template <int ACT0, int ACT1, bool HALF>
__global__ __launch_bounds__(NT, 1) void conv_stem2_f16x3(ConvKernelArgs p0, ConvKernelArgs p1, int tiles_y, int tiles_x, int n_tiles) {
    fp16_saturate_on();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *wreg = smem, *patch = smem + W_BYTES;
    float4 *rgb = reinterpret_cast<float4 *>(smem + W_BYTES + PATCH_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;

    // layer-1 filters -> LDS once: row (tap, cout) at (tap * 64 + cout) * 128, chunk c at position c ^ ((row >> 1) & 7)
    for (int i = tid; i < 9 * BN * 8; i += NT) {
        const int row = i >> 3, pos = i & 7, tap = row / BN, co = row - tap * BN;
        const int c = pos ^ ((row >> 1) & 7);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p1.w) + ((size_t)min(co, p1.Cout - 1) * 9 + tap) * 128 + c * 16);
        *reinterpret_cast<f32x4 *>(wreg + row * 128 + pos * 16) = v;
    }
    // layer-0 filter fragments (first MFMA operand: row = output channel lane & 31, k = 16 s + 8 kb + 0..7 = taps 4s + 2kb,
    // 4s + 2kb + 1 x (r, g, b, pad)), split hi / lo on the fly, and the biases of the 16 channels this lane's accumulators hold
    const int kb0 = lane >> 5;
    h8 w0h[3], w0l[3];
#pragma unroll
    for (int sb = 0; sb < 3; ++sb)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = 4 * sb + 2 * kb0 + (e >> 2), c = e & 3;
            // (round 6: the pad channel of the CENTRE tap carries the bias - the RGB tile holds 1.0 there for every pixel inside the
            //  image - so the accumulators come out with the bias added, in the scaled domain)
            const float x = (t < 9 && c < 3) ? p0.w[(size_t)(lane & 31) * p0.Kpad + t * 4 + c] : (t == 4 && c == 3 ? p0.bias[lane & 31] : 0.f);
            const _Float16 h = (_Float16)x;
            w0h[sb][e] = h;
            w0l[sb][e] = (_Float16)((x - (float)h) * LO_SCALE);
        }
    // phase-B fragment bookkeeping
    const int kb = lane >> 5, r = wm * 32 + (lane & 31), py = r / TW, px = r - py * TW;
    const int brow = wn * 32 + (lane & 31);

    // the next tile's RGB pixels are fetched into registers while this tile is computed (two 16-byte loads per thread)
    constexpr int LOADS = (RR * RC + NT - 1) / NT;
    float4 nxt[LOADS];
    auto fetch = [&](int tl) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int iy0 = 2 * ((rem / tiles_x) * TH) - 2, ix0 = 2 * ((rem % tiles_x) * TW) - 2;
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const int i = tid + l * NT, rr = i / RC, cc = i - rr * RC;
            const int iy = iy0 + rr, ix = ix0 + cc;
            nxt[l] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < RR * RC && tl < n_tiles && (unsigned)iy < (unsigned)p0.H && (unsigned)ix < (unsigned)p0.W) {
                nxt[l] = *reinterpret_cast<const float4 *>(p0.x + ((size_t)(img * p0.H + iy) * p0.W + ix) * p0.ldx);
                nxt[l].w = 1.f;                                 // the bias rides on the pad channel (see w0h / w0l)
            }
        }
    };
    fetch(blockIdx.x);
    for (int tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int oy0 = (rem / tiles_x) * TH, ox0 = (rem % tiles_x) * TW;
        const int ly0 = 2 * oy0 - 1, lx0 = 2 * ox0 - 1;           // layer-0 pixel of patch (0, 0); the RGB tile starts one before
        __syncthreads();                                        // previous tile: fragments read, epilogue staging consumed
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const int i = tid + l * NT;
            if (i < RR * RC) {                                  // [hi r g b 0 | lo r g b 0], x * 2^-8 = hi + lo * 2^-11
                const float v[4] = {nxt[l].x, nxt[l].y, nxt[l].z, nxt[l].w};
                union { h16x4 h[2]; float4 f; } sp;
                h16_encode4(v, sp.h[0], sp.h[1]);
                rgb[i] = sp.f;
            }
        }
        __syncthreads();
        fetch(tl + gridDim.x);
        // ---- phase A: layer 0 for the 17 x 33 patch as 18 fragments of 32 pixels (wave w takes fragments w, w + 8, w + 16)
#pragma unroll 1
        for (int fr = wave; fr < (PATCH_ROWS + 31) / 32; fr += NW) {
            const int pix = fr * 32 + (lane & 31), pc = min(pix, PATCH_ROWS - 1);
            const int ry = pc / PC, rc = pc - ry * PC;
            f32x16 c1, c2;
#pragma unroll
            for (int e = 0; e < 16; ++e) { c1[e] = 0.f; c2[e] = 0.f; }
#pragma unroll
            for (int sb = 0; sb < 3; ++sb) {
                union { h16x4 q[2]; h8 v; } xh, xl;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = 4 * sb + 2 * kb0 + u;             // tap of this half of the lane's 8 k
                    union { float4 f; h16x4 h[2]; } px4;
                    px4.f = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (t < 9) px4.f = rgb[(ry + t / 3) * RC + rc + t % 3];
                    xh.q[u] = px4.h[0];
                    xl.q[u] = px4.h[1];
                }
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h[sb], xh.v, c1, 0, 0, 0);
                if (!HALF) {
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h[sb], xl.v, c2, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l[sb], xh.v, c2, 0, 0, 0);
                }
            }
            // this lane: pixel `pix`, channels (e & 3) + 8 (e >> 2) + 4 kb0 -> four 8-byte pieces of hi and of lo.
            // Round 6: ~5.5 instead of ~9.5 vector instructions per value - the bias is already in the accumulators, LeakyReLU is
            // applied in the scaled domain (positively homogeneous: max(o, 0.1 o), bit-identical to the unscaled form) and encoded
            // from there, pixels outside the image take a branch of their own (zeros: layer 1 pads) instead of a select per value.
            const bool inside = (unsigned)(ly0 + ry) < (unsigned)p0.H && (unsigned)(lx0 + rc) < (unsigned)p0.W;
            const int j = patch_row(ry, rc), jsw = (j >> 1) & 7;
            if (pix < PATCH_ROWS) {
                if (inside) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int e = g * 4 + c;
                            const float os = HALF ? c1[e] : c1[e] + c2[e] * (1.f / LO_SCALE);             // o * 2^-8, bias included
                            v[c] = ACT0 == ACT_LEAKY ? fmaxf(os, os * 0.1f) : apply_act<ACT0>(os * (1.f / A_SCALE)) * A_SCALE;
                        }
                        h16x4 hi, lo;
                        if (HALF) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) hi[c] = (_Float16)v[c];
                        } else {
                            h16_encode4_scaled(v, hi, lo);
                        }
                        *reinterpret_cast<h16x4 *>(patch + j * 128 + ((g ^ jsw) << 4) + kb0 * 8) = hi;          // chunk g: channels 8g .. 8g+7
                        if (!HALF) *reinterpret_cast<h16x4 *>(patch + j * 128 + (((4 + g) ^ jsw) << 4) + kb0 * 8) = lo;
                    }
                } else {
                    const h16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
                    for (int g = 0; g < (HALF ? 4 : 8); ++g) *reinterpret_cast<h16x4 *>(patch + j * 128 + ((g ^ jsw) << 4) + kb0 * 8) = z;
                }
            }
        }
        __syncthreads();
        // ---- phase B: layer 1, 9 taps x 2 k-substeps on the patch
        f32x16 acc1[1][1], acc2[1][1];
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc1[0][0][e] = 0.f; acc2[0][0][e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t % 3;
            const int j = (dx & 1) ? ODD_BASE + (2 * py + dy) * ODD_COLS + px : (2 * py + dy) * EVEN_COLS + px + (dx >> 1);
            const int jsw = (j >> 1) & 7, wrow = t * BN + brow, wsw = (wrow >> 1) & 7;
            const char *ap = patch + j * 128, *bp = wreg + wrow * 128;
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
        // whole-tile staging (128 x 68 floats) in the layer-0 patch once every wave is done reading it: one barrier pair instead of
        // the four passes of 32 rows that the RGB tile could hold (round 6)
        __syncthreads();
        conv_epilogue_rows<BM, BN, WM, WN, ACT1, RES_NONE, 1, 1, NT, StemRows, true>(p1, acc1, reinterpret_cast<float *>(patch),
                                                                                       StemRows{img, oy0, ox0, p1.Ho, p1.Wo}, 0, tid);
    }
}

template <int ACT0, int ACT1, bool HALF> void launch_inst(const ConvKernelArgs &k0, const ConvKernelArgs &k1, hipStream_t s) {
    static_assert(BM * (BN + 4) * 4 <= PATCH_BYTES, "whole-tile epilogue staging must fit the layer-0 patch");
    const int n_img = k0.M / (k0.H * k0.W);
    const int tiles_y = (k1.Ho + TH - 1) / TH, tiles_x = (k1.Wo + TW - 1) / TW, n_tiles = n_img * tiles_y * tiles_x;
    static bool attr_set = false;
    auto kern = conv_stem2_f16x3<ACT0, ACT1, HALF>;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(std::min(n_tiles, 256)), dim3(NT), SMEM, s, k0, k1, tiles_y, tiles_x, n_tiles);
    YDS_HIP(hipGetLastError());
}

}  // namespace

bool conv_stem2_applicable(const ConvKernelArgs &k0, const ConvKernelArgs &k1) {
    return k0.Cin == 4 && k0.ksize == 3 && k0.stride == 1 && k0.pad == 1 && k0.Cout == 32 && k0.res_mode == RES_NONE && k0.fmt_x == FMT_F32 &&
           k1.Cin == 32 && k1.ksize == 3 && k1.stride == 2 && k1.pad == 1 && k1.Cout == 64 && k1.res_mode == RES_NONE && k1.K == 288 &&
           k1.H == k0.Ho && k1.W == k0.Wo && k0.act == k1.act && (k0.act == ACT_LEAKY || k0.act == ACT_MISH);
}

// k0 / k1: the two layers' own arguments (k1.w = its pre-split f16x3 weights); k0's output tensor is never written
void launch_conv_stem2(const ConvKernelArgs &k0, const ConvKernelArgs &k1, hipStream_t s) {
    if (!conv_stem2_applicable(k0, k1)) fail("conv: the fused stem takes a 3x3/s1 RGB->32 conv followed by a 3x3/s2 32->64 conv");
    const bool half = k1.terms == 1;                             // Darknet.half(): single-term instantiation
    if (k0.act == ACT_LEAKY) { if (half) launch_inst<ACT_LEAKY, ACT_LEAKY, true>(k0, k1, s); else launch_inst<ACT_LEAKY, ACT_LEAKY, false>(k0, k1, s); }
    else { if (half) launch_inst<ACT_MISH, ACT_MISH, true>(k0, k1, s); else launch_inst<ACT_MISH, ACT_MISH, false>(k0, k1, s); }
}

}  // namespace yds
