// Fused detector stem: layer 0 (3x3 / stride 1, RGB -> 32 channels) and layer 1 (3x3 / stride 2, 32 -> 64) in one kernel.
// yolov3 / yolov4: yolo3/models/models.py:36-56 applied to the first two [convolutional] blocks of config/yolov3.cfg /
// yolov4.cfg.
//
// Unfused, layer 0 writes 608*608*32 values per image (757 MB per 16 frames) that layer 1 reads straight back: the two
// launches are HBM bound on that tensor (322 + 349 us per 16 frames, ~340 us of it for the round trip).  Here a 512-thread
// persistent workgroup owns an 8 x 16 patch of LAYER-1 outputs:
//   phase A  the 17 x 33 layer-0 pixels it needs are computed on the vector ALU from an RGB tile in LDS (the fma chain and
//            register-resident weights of conv_first.hip: 8 lanes per pixel, 4 channels each) and written to LDS as pre-split
//            H16 rows (128 B = [32 hi | 32 lo] fp16 per pixel); pixels outside the image are written as zeros (layer 1 pads)
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

template <int ACT0, int ACT1>
__global__ __launch_bounds__(NT, 1) void conv_stem2_f16x3(ConvKernelArgs p0, ConvKernelArgs p1, int tiles_y, int tiles_x, int n_tiles) {
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
    // layer-0 weights of this lane: 4 output channels x 9 taps x 3 input channels (+ bias), as channel pairs (v_pk_fma_f32)
    const int q = tid & 7, slot = tid >> 3;
    f32x2 w0[2][9][3], b0[2];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float *wr = p0.w + (size_t)(q * 4 + o) * p0.Kpad;
        b0[o / 2][o % 2] = p0.bias[q * 4 + o];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) w0[o / 2][t][c][o % 2] = wr[t * 4 + c];
    }
    // phase-B fragment bookkeeping
    const int kb = lane >> 5, r = wm * 32 + (lane & 31), py = r / TW, px = r - py * TW;
    const int brow = wn * 32 + (lane & 31);

    for (int tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int oy0 = (rem / tiles_x) * TH, ox0 = (rem % tiles_x) * TW;
        const int ly0 = 2 * oy0 - 1, lx0 = 2 * ox0 - 1;           // layer-0 pixel of patch (0, 0); the RGB tile starts one before
        __syncthreads();                                        // previous tile: fragments read, epilogue staging consumed
        for (int i = tid; i < RR * RC; i += NT) {
            const int rr = i / RC, cc = i - rr * RC;
            const int iy = ly0 - 1 + rr, ix = lx0 - 1 + cc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)p0.H && (unsigned)ix < (unsigned)p0.W)
                v = *reinterpret_cast<const float4 *>(p0.x + ((size_t)(img * p0.H + iy) * p0.W + ix) * p0.ldx);
            rgb[i] = v;
        }
        __syncthreads();
        // ---- phase A: layer 0 for the 17 x 33 patch, 64 pixels per pass
#pragma unroll 1
        for (int pass = 0; pass < (PATCH_ROWS + 63) / 64; ++pass) {
            const int pix = pass * 64 + slot;
            const int ry = min(pix, PATCH_ROWS - 1) / PC, rc = min(pix, PATCH_ROWS - 1) % PC;
            f32x2 a2[2] = {b0[0], b0[1]};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 v = rgb[(ry + t / 3) * RC + rc + t % 3];
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    a2[o] = __builtin_elementwise_fma(f32x2{v.x, v.x}, w0[o][t][0], a2[o]);
                    a2[o] = __builtin_elementwise_fma(f32x2{v.y, v.y}, w0[o][t][1], a2[o]);
                    a2[o] = __builtin_elementwise_fma(f32x2{v.z, v.z}, w0[o][t][2], a2[o]);
                }
            }
            float acc[4] = {a2[0][0], a2[0][1], a2[1][0], a2[1][1]};
            const bool inside = (unsigned)(ly0 + ry) < (unsigned)p0.H && (unsigned)(lx0 + rc) < (unsigned)p0.W;
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = inside ? apply_act<ACT0>(acc[o]) : 0.f;      // layer 1 zero-pads layer 0's output
            h16x4 hi, lo;
            h16_encode4(acc, hi, lo);
            const bool odd = q & 1;
            union { h16x4 h; int i[2]; } send, recv;
            send.h = odd ? hi : lo;
            recv.i[0] = __shfl_xor(send.i[0], 1);
            recv.i[1] = __shfl_xor(send.i[1], 1);
            union { h16x4 h[2]; f32x4 f; } out;
            out.h[0] = odd ? recv.h : hi;
            out.h[1] = odd ? lo : recv.h;
            const int chunk = (odd ? 4 : 0) + (q >> 1);           // hi chunks 0-3 (8 channels each), lo chunks 4-7
            const int j = patch_row(ry, rc);
            if (pix < PATCH_ROWS) *reinterpret_cast<f32x4 *>(patch + j * 128 + ((chunk ^ ((j >> 1) & 7)) << 4)) = out.f;
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
                const h8 al = *reinterpret_cast<const h8 *>(ap + (((4 + 2 * s + kb) ^ jsw) << 4));
                const h8 bh = *reinterpret_cast<const h8 *>(bp + (((2 * s + kb) ^ wsw) << 4));
                const h8 bl = *reinterpret_cast<const h8 *>(bp + (((4 + 2 * s + kb) ^ wsw) << 4));
                acc1[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[0][0], 0, 0, 0);
                acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[0][0], 0, 0, 0);
                acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[0][0], 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[0][0][e] = (acc1[0][0][e] + acc2[0][0][e] * (1.f / LO_SCALE)) * (1.f / A_SCALE);
        // the staging area (32 x 68 floats) lives in the RGB tile, which phase A is done with
        conv_epilogue_rows<BM, BN, WM, WN, ACT1, RES_NONE, 1, 1, NT, StemRows>(p1, acc1, reinterpret_cast<float *>(rgb),
                                                                                 StemRows{img, oy0, ox0, p1.Ho, p1.Wo}, 0, tid);
    }
}

template <int ACT0, int ACT1> void launch_inst(const ConvKernelArgs &k0, const ConvKernelArgs &k1, hipStream_t s) {
    static_assert((BM / WM) * (BN + 4) * 4 <= RGB_BYTES, "epilogue staging must fit the RGB tile");
    const int n_img = k0.M / (k0.H * k0.W);
    const int tiles_y = (k1.Ho + TH - 1) / TH, tiles_x = (k1.Wo + TW - 1) / TW, n_tiles = n_img * tiles_y * tiles_x;
    static bool attr_set = false;
    auto kern = conv_stem2_f16x3<ACT0, ACT1>;
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
    if (k0.act == ACT_LEAKY) launch_inst<ACT_LEAKY, ACT_LEAKY>(k0, k1, s);
    else launch_inst<ACT_MISH, ACT_MISH>(k0, k1, s);
}

}  // namespace yds
