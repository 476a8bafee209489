// First-layer convolution: 3x3, stride 1, RGB (channel-padded to 4) -> 32 or 64 channels.
//
// The implicit-GEMM kernels are a poor fit here: K = 27 (padded to 64) and the layer is bound by writing
// 608*608*32 outputs per image, not by arithmetic.  This kernel is a direct convolution on the vector ALU:
// COUT/4 neighbouring lanes own one output pixel (4 output channels each, their 4x27 weights live in registers
// for the whole kernel), so a wave's store instruction writes whole contiguous pixel rows (128 B / 256 B per
// pixel); the input tile (8x32 pixels + halo) is staged in LDS and the 9 taps of a pixel are LDS broadcast reads.  Arithmetic is a plain fp32 fma chain in both math
// modes (yolov3/yolov4 layer 0: yolo3/models/models.py:36-56; ReID stem: deep_sort/deep/model.py:52-60).
#include "conv_common.h"

#include <algorithm>

namespace yds {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 8, TW = 32;                      // output pixels per tile (rows x columns)

template <int COUT, int ACT>
__global__ __launch_bounds__(256) void conv3x3_rgb_direct(ConvKernelArgs p, int tiles_y, int tiles_x, int n_tiles) {
    fp16_saturate_on();
    constexpr int QUADS = COUT / 4;                 // lanes per pixel
    constexpr int PIX_PER_PASS = 256 / QUADS, PASSES = TH * TW / PIX_PER_PASS;
    __shared__ float4 tile[TH + 2][TW + 2];         // input tile + halo (zero outside the image): the nine taps of a pixel
                                                    // are LDS reads (4 cycles per wave instruction) instead of vector-memory
                                                    // loads (16 cycles each on the texture path, where this kernel was bound)
    const int q = threadIdx.x % QUADS, slot = threadIdx.x / QUADS;
    // this lane's weights: 4 output channels x 9 taps x 3 input channels (+ bias), loaded once (persistent blocks)
    f32x2 w[2][9][3], b[2];                         // channel pairs: the fma chain runs on v_pk_fma_f32 (two channels per instruction)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float *wr = p.w + (size_t)(q * 4 + o) * p.Kpad;
        b[o / 2][o % 2] = p.bias[q * 4 + o];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) w[o / 2][t][c][o % 2] = wr[t * 4 + c];
    }
    // the next tile's input is fetched into registers while this tile is computed (two 16-byte loads per thread)
    constexpr int LOADS = ((TH + 2) * (TW + 2) + 255) / 256;
    float4 nxt[LOADS];
    auto fetch = [&](int tl) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const int i = threadIdx.x + l * 256;
            const int r = i / (TW + 2), c = i - r * (TW + 2);
            const int iy = y0 - 1 + r, ix = x0 - 1 + c;
            nxt[l] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < (TH + 2) * (TW + 2) && tl < n_tiles && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                nxt[l] = *reinterpret_cast<const float4 *>(p.x + ((size_t)(img * p.H + iy) * p.W + ix) * p.ldx);
        }
    };
    fetch(blockIdx.x);
    for (int tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
        __syncthreads();                            // previous tile fully consumed
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const int i = threadIdx.x + l * 256;
            if (i < (TH + 2) * (TW + 2)) (&tile[0][0])[i] = nxt[l];
        }
        __syncthreads();
        fetch(tl + gridDim.x);
#pragma unroll 2
        for (int pass = 0; pass < PASSES; ++pass) {
            const int pi = pass * PIX_PER_PASS + slot, py = pi / TW, px = pi - py * TW;
            const int oy = y0 + py, ox = x0 + px;
            f32x2 a2[2] = {b[0], b[1]};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 v = tile[py + t / 3][px + t % 3];
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    a2[o] = __builtin_elementwise_fma(f32x2{v.x, v.x}, w[o][t][0], a2[o]);
                    a2[o] = __builtin_elementwise_fma(f32x2{v.y, v.y}, w[o][t][1], a2[o]);
                    a2[o] = __builtin_elementwise_fma(f32x2{v.z, v.z}, w[o][t][2], a2[o]);
                }
            }
            float acc[4] = {a2[0][0], a2[0][1], a2[1][0], a2[1][1]};
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = apply_act<ACT>(acc[o]);
            const bool inside = oy < p.H && ox < p.W;
            float *yp = p.y + ((size_t)(img * p.H + oy) * p.W + ox) * p.ldy;
            if (p.fmt_y == FMT_H16) {
                // lane pairs trade halves so that every lane issues ONE 16-byte store: the even lane writes the hi
                // halves of both lanes' channels (8 consecutive fp16), the odd lane the lo halves
                h16x4 hi, lo;
                h16_encode4(acc, hi, lo);
                const bool odd = q & 1;
                union { h16x4 h; int i[2]; } send, recv;
                send.h = odd ? hi : lo;
                recv.i[0] = __builtin_amdgcn_update_dpp(send.i[0], send.i[0], 0xB1, 0xF, 0xF, false);   // lane ^ 1 (DPP quad_perm [1,0,3,2])
                recv.i[1] = __builtin_amdgcn_update_dpp(send.i[1], send.i[1], 0xB1, 0xF, 0xF, false);
                union { h16x4 h[2]; float4 f; } out;
                out.h[0] = odd ? recv.h : hi;
                out.h[1] = odd ? lo : recv.h;
                const int c0 = (q & ~1) * 4;                                    // first channel of the pair
                char *g = reinterpret_cast<char *>(yp + (c0 & ~31)) + (c0 & 31) * 2 + (odd ? 64 : 0);
                if (inside) *reinterpret_cast<float4 *>(g) = out.f;
            } else if (inside) {
                *reinterpret_cast<float4 *>(yp + q * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
    }
}

template <int COUT> static void launch_direct_cout(const ConvKernelArgs &k, hipStream_t s) {
    const int n_img = k.M / (k.H * k.W);
    const int tiles_y = (k.H + TH - 1) / TH, tiles_x = (k.W + TW - 1) / TW, n_tiles = n_img * tiles_y * tiles_x;
    dim3 grid((unsigned)std::min(n_tiles, 256 * 8));      // persistent blocks: the weights are loaded once per lane
    switch (k.act) {
        case ACT_LEAKY: hipLaunchKernelGGL((conv3x3_rgb_direct<COUT, ACT_LEAKY>), grid, dim3(256), 0, s, k, tiles_y, tiles_x, n_tiles); break;
        case ACT_MISH: hipLaunchKernelGGL((conv3x3_rgb_direct<COUT, ACT_MISH>), grid, dim3(256), 0, s, k, tiles_y, tiles_x, n_tiles); break;
        case ACT_RELU: hipLaunchKernelGGL((conv3x3_rgb_direct<COUT, ACT_RELU>), grid, dim3(256), 0, s, k, tiles_y, tiles_x, n_tiles); break;
        default: hipLaunchKernelGGL((conv3x3_rgb_direct<COUT, ACT_LINEAR>), grid, dim3(256), 0, s, k, tiles_y, tiles_x, n_tiles); break;
    }
    YDS_HIP(hipGetLastError());
}

// ReID stem (deep_sort/deep/model.py:52-60): the same convolution followed by MaxPool2d(3, 2, padding=1), fused.  The
// unfused pair writes and re-reads the full-resolution 64-channel tensor (2 KB per input pixel and crop: the stem was
// HBM bound on exactly that); here a lane group owns one POOLED pixel, evaluates its (up to) nine convolution outputs
// from the LDS input tile with the same fma chain as above and keeps the running maximum in registers (2.25x the
// arithmetic, 1/8 of the traffic).  Out-of-image convolution positions are skipped (the pool pads with -inf).
constexpr int PTH = 4, PTW = 16;                    // pooled pixels per tile

template <int COUT, int ACT>
__global__ __launch_bounds__(256) void conv3x3_rgb_pool(ConvKernelArgs p, int Hp, int Wp, int tiles_y, int tiles_x, int n_tiles) {
    fp16_saturate_on();
    constexpr int QUADS = COUT / 4, PIX_PER_PASS = 256 / QUADS, PASSES = PTH * PTW / PIX_PER_PASS;
    constexpr int IR = 2 * PTH + 3, IC = 2 * PTW + 3;           // input tile incl. both halos
    __shared__ float4 tile[IR][IC];
    const int q = threadIdx.x % QUADS, slot = threadIdx.x / QUADS;
    f32x2 w[2][9][3], b[2];                         // channel pairs: the fma chain runs on v_pk_fma_f32 (two channels per instruction)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float *wr = p.w + (size_t)(q * 4 + o) * p.Kpad;
        b[o / 2][o % 2] = p.bias[q * 4 + o];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) w[o / 2][t][c][o % 2] = wr[t * 4 + c];
    }
    for (int tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int py0 = (rem / tiles_x) * PTH, px0 = (rem % tiles_x) * PTW;
        const int iy0 = 2 * py0 - 2, ix0 = 2 * px0 - 2;         // input pixel of tile[0][0]
        __syncthreads();
        for (int i = threadIdx.x; i < IR * IC; i += 256) {
            const int r = i / IC, c = i - r * IC;
            const int iy = iy0 + r, ix = ix0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v = *reinterpret_cast<const float4 *>(p.x + ((size_t)(img * p.H + iy) * p.W + ix) * p.ldx);
            tile[r][c] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int pass = 0; pass < PASSES; ++pass) {
            const int pi = pass * PIX_PER_PASS + slot, ppy = pi / PTW, ppx = pi - ppy * PTW;
            const int Py = py0 + ppy, Px = px0 + ppx;
            float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll 1
            for (int d = 0; d < 9; ++d) {
                const int dy = d / 3, dx = d - dy * 3;
                const int cy = 2 * Py - 1 + dy, cx = 2 * Px - 1 + dx;           // convolution output position
                const bool valid = (unsigned)cy < (unsigned)p.H && (unsigned)cx < (unsigned)p.W;
                f32x2 a2[2] = {b[0], b[1]};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float4 v = tile[2 * ppy + dy + t / 3][2 * ppx + dx + t % 3];
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        a2[o] = __builtin_elementwise_fma(f32x2{v.x, v.x}, w[o][t][0], a2[o]);
                        a2[o] = __builtin_elementwise_fma(f32x2{v.y, v.y}, w[o][t][1], a2[o]);
                        a2[o] = __builtin_elementwise_fma(f32x2{v.z, v.z}, w[o][t][2], a2[o]);
                    }
                }
                const float acc[4] = {a2[0][0], a2[0][1], a2[1][0], a2[1][1]};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float a = apply_act<ACT>(acc[o]);
                    m[o] = valid ? fmaxf(m[o], a) : m[o];
                }
            }
            if (Py < Hp && Px < Wp) store4(p.y + ((size_t)(img * Hp + Py) * Wp + Px) * p.ldy, q * 4, p.fmt_y, m);
        }
    }
}

// The same stem on the matrix cores (f16x3 arithmetic, conv_f16x3.hip): the vector-ALU version above spends its time in
// 27-deep fma chains and evaluates every convolution output 2.25 times.  Here a 512-thread persistent workgroup owns a
// 3 x 16 patch of POOLED pixels (round 3: 4 x 16 with one workgroup per CU took 297 us per 480 crops, 3 x 16 leaves room for two
// workgroups per CU whose load / convolve / pool phases overlap: 239 us): the 7 x 33 convolution outputs it needs are computed once as MFMA products (filter
// fragments = first operand, in registers; pixel fragments gathered from a split RGB tile in LDS as in conv_stem2.hip; a
// lane ends up with 4 consecutive channels of one pixel), written to LDS as fp32 after bias + activation (-inf where the
// position lies outside the image: the pool pads with -inf), and pooled from there.
constexpr int MP_TH = 3, MP_OCC = 2, MP_TW = 16;                        // pooled patch: rows per patch, workgroups per CU, columns
constexpr int MP_CR = 2 * MP_TH + 1, MP_CC = 2 * MP_TW + 1;         // conv region 9 x 33
constexpr int MP_IR = MP_CR + 2, MP_IC = MP_CC + 2;                 // input tile 11 x 35
constexpr int MP_CONV = MP_CR * MP_CC;                              // 297
constexpr int MP_LD = 68;                                           // floats per conv row in LDS (64 + 4: conflict-free 16-byte accesses)
constexpr int MP_NT = 512;

template <int ACT>
__global__ __launch_bounds__(MP_NT, MP_OCC) void conv3x3_rgb_pool_mfma(ConvKernelArgs p, int Hp, int Wp, int tiles_y, int tiles_x, int n_tiles) {
    fp16_saturate_on();
    extern __shared__ __attribute__((aligned(16))) char mp_smem[];
    float4 *rgb = reinterpret_cast<float4 *>(mp_smem);              // [hi r g b 0 | lo r g b 0] per input pixel
    float *conv = reinterpret_cast<float *>(mp_smem + MP_IR * MP_IC * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kb = lane >> 5;
    // filter fragments of both 32-channel halves (first MFMA operand: row = channel, k = 16 s + 8 kb + 0..7 = taps 4s + 2kb,
    // 4s + 2kb + 1 x (r, g, b, pad)), split on the fly
    h8 wh[2][3], wl[2][3];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int sb = 0; sb < 3; ++sb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int t = 4 * sb + 2 * kb + (e >> 2), c = e & 3;
                const float x = (t < 9 && c < 3) ? p.w[(size_t)(n * 32 + (lane & 31)) * p.Kpad + t * 4 + c] : 0.f;
                const _Float16 h = (_Float16)x;
                wh[n][sb][e] = h;
                wl[n][sb][e] = (_Float16)((x - (float)h) * 2048.f);
            }
    constexpr int LOADS = (MP_IR * MP_IC + MP_NT - 1) / MP_NT;
    float4 nxt[LOADS];
    auto fetch = [&](int tl) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int iy0 = 2 * ((rem / tiles_x) * MP_TH) - 2, ix0 = 2 * ((rem % tiles_x) * MP_TW) - 2;
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const int i = tid + l * MP_NT, rr = i / MP_IC, cc = i - rr * MP_IC;
            const int iy = iy0 + rr, ix = ix0 + cc;
            nxt[l] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < MP_IR * MP_IC && tl < n_tiles && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                nxt[l] = *reinterpret_cast<const float4 *>(p.x + ((size_t)(img * p.H + iy) * p.W + ix) * p.ldx);
        }
    };
    fetch(blockIdx.x);
    for (int tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int img = tl / (tiles_y * tiles_x), rem = tl - img * (tiles_y * tiles_x);
        const int py0 = (rem / tiles_x) * MP_TH, px0 = (rem % tiles_x) * MP_TW;
        const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;             // convolution position of conv-region (0, 0)
        __syncthreads();                                            // previous tile pooled
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const int i = tid + l * MP_NT;
            if (i < MP_IR * MP_IC) {
                const float v[4] = {nxt[l].x, nxt[l].y, nxt[l].z, 0.f};
                union { h16x4 h[2]; float4 f; } sp;
                h16_encode4(v, sp.h[0], sp.h[1]);
                rgb[i] = sp.f;
            }
        }
        __syncthreads();
        fetch(tl + gridDim.x);
        // convolution: 10 fragments of 32 positions x 2 channel halves = 20 units over 8 waves
#pragma unroll 1
        for (int u = wave; u < 2 * ((MP_CONV + 31) / 32); u += 8) {
            const int fr = u >> 1, n = u & 1;
            const int pix = fr * 32 + (lane & 31), pc = min(pix, MP_CONV - 1);
            const int ry = pc / MP_CC, rc = pc - ry * MP_CC;
            f32x16 c1, c2;
#pragma unroll
            for (int e = 0; e < 16; ++e) { c1[e] = 0.f; c2[e] = 0.f; }
#pragma unroll
            for (int sb = 0; sb < 3; ++sb) {
                union { h16x4 q[2]; h8 v; } xh, xl;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int t = 4 * sb + 2 * kb + h;
                    union { float4 f; h16x4 hh[2]; } px4;
                    px4.f = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (t < 9) px4.f = rgb[(ry + t / 3) * MP_IC + rc + t % 3];
                    xh.q[h] = px4.hh[0];
                    xl.q[h] = px4.hh[1];
                }
                const h8 fh = n ? wh[1][sb] : wh[0][sb], fl = n ? wl[1][sb] : wl[0][sb];
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, xh.v, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, xl.v, c2, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, xh.v, c2, 0, 0, 0);
            }
            const bool inside = (unsigned)(cy0 + ry) < (unsigned)p.H && (unsigned)(cx0 + rc) < (unsigned)p.W;
            if (pix < MP_CONV) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = n * 32 + 8 * g + 4 * kb;             // this lane's 4 consecutive channels of group g
                    const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + ch);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = (c1[g * 4 + c] + c2[g * 4 + c] * (1.f / 2048.f)) * 256.f + bb[c];
                        o[c] = inside ? apply_act<ACT>(v) : -INFINITY;
                    }
                    *reinterpret_cast<float4 *>(conv + pix * MP_LD + ch) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        __syncthreads();
        // pooling: 64 pooled pixels x 16 channel quads
#pragma unroll
        for (int it = 0; it < (MP_TH * MP_TW * 16 + MP_NT - 1) / MP_NT; ++it) {
            const int idx = it * MP_NT + tid, pp = idx >> 4, cq = idx & 15;
            if (pp >= MP_TH * MP_TW) break;
            const int ppy = pp / MP_TW, ppx = pp - ppy * MP_TW;
            float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int d = 0; d < 9; ++d) {
                const float4 v = *reinterpret_cast<const float4 *>(conv + ((2 * ppy + d / 3) * MP_CC + 2 * ppx + d % 3) * MP_LD + cq * 4);
                m[0] = fmaxf(m[0], v.x); m[1] = fmaxf(m[1], v.y); m[2] = fmaxf(m[2], v.z); m[3] = fmaxf(m[3], v.w);
            }
            const int Py = py0 + ppy, Px = px0 + ppx;
            const bool live = Py < Hp && Px < Wp;
            float *yp = p.y + ((size_t)(img * Hp + (live ? Py : 0)) * Wp + (live ? Px : 0)) * p.ldy;
            if (p.fmt_y == FMT_H16) {                               // lane pairs trade halves: one 16-byte store per lane
                h16x4 hi, lo;
                h16_encode4(m, hi, lo);
                const bool odd = cq & 1;
                union { h16x4 h; int i[2]; } send, recv;
                send.h = odd ? hi : lo;
                recv.i[0] = __builtin_amdgcn_update_dpp(send.i[0], send.i[0], 0xB1, 0xF, 0xF, false);   // lane ^ 1 (DPP quad_perm [1,0,3,2])
                recv.i[1] = __builtin_amdgcn_update_dpp(send.i[1], send.i[1], 0xB1, 0xF, 0xF, false);
                union { h16x4 h[2]; float4 f; } out;
                out.h[0] = odd ? recv.h : hi;
                out.h[1] = odd ? lo : recv.h;
                const int c0 = (cq & ~1) * 4;
                char *g = reinterpret_cast<char *>(yp + (c0 & ~31)) + (c0 & 31) * 2 + (odd ? 64 : 0);
                if (live) *reinterpret_cast<float4 *>(g) = out.f;
            } else if (live) {
                *reinterpret_cast<float4 *>(yp + cq * 4) = make_float4(m[0], m[1], m[2], m[3]);
            }
        }
    }
}

bool conv_pool_applicable(const ConvKernelArgs &k) {
    return k.Cin == 4 && k.ksize == 3 && k.stride == 1 && k.pad == 1 && k.Cout == 64 && k.res_mode == RES_NONE && k.fmt_x == FMT_F32;
}

// k describes the convolution (H, W = its input size); k.y / k.ldy / k.fmt_y the POOLED output tensor
void launch_conv_pool(const ConvKernelArgs &k, hipStream_t s) {
    if (!conv_pool_applicable(k)) fail("conv+pool: only the 3x3 RGB stem with 64 filters is fused");
    const int Hp = (k.H + 2 - 3) / 2 + 1, Wp = (k.W + 2 - 3) / 2 + 1, n_img = k.M / (k.H * k.W);
    const int tiles_y = (Hp + PTH - 1) / PTH, tiles_x = (Wp + PTW - 1) / PTW, n_tiles = n_img * tiles_y * tiles_x;
    if (conv_math() == MATH_F16X3 && !getenv("YDS_POOL_VALU")) {    // matrix-core version (exact fp32 mode keeps the fma chain)
        const int ty = (Hp + MP_TH - 1) / MP_TH, tx = (Wp + MP_TW - 1) / MP_TW, nt = n_img * ty * tx;
        dim3 g((unsigned)std::min(nt, 256 * MP_OCC));
        constexpr int smem = MP_IR * MP_IC * 16 + MP_CONV * MP_LD * 4;
        static bool attr_set = false;
        if (!attr_set) {
            YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_rgb_pool_mfma<ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_rgb_pool_mfma<ACT_LEAKY>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set = true;
        }
        switch (k.act) {
            case ACT_RELU: hipLaunchKernelGGL((conv3x3_rgb_pool_mfma<ACT_RELU>), g, dim3(MP_NT), smem, s, k, Hp, Wp, ty, tx, nt); break;
            case ACT_LEAKY: hipLaunchKernelGGL((conv3x3_rgb_pool_mfma<ACT_LEAKY>), g, dim3(MP_NT), smem, s, k, Hp, Wp, ty, tx, nt); break;
            default: fail("conv+pool: unsupported activation %d", k.act);
        }
        YDS_HIP(hipGetLastError());
        return;
    }
    dim3 grid((unsigned)std::min(n_tiles, 256 * 8));
    switch (k.act) {
        case ACT_RELU: hipLaunchKernelGGL((conv3x3_rgb_pool<64, ACT_RELU>), grid, dim3(256), 0, s, k, Hp, Wp, tiles_y, tiles_x, n_tiles); break;
        case ACT_LEAKY: hipLaunchKernelGGL((conv3x3_rgb_pool<64, ACT_LEAKY>), grid, dim3(256), 0, s, k, Hp, Wp, tiles_y, tiles_x, n_tiles); break;
        default: fail("conv+pool: unsupported activation %d", k.act);
    }
    YDS_HIP(hipGetLastError());
}

bool conv_direct_applicable(const ConvKernelArgs &k) {
    return k.Cin == 4 && k.ksize == 3 && k.stride == 1 && k.pad == 1 && (k.Cout == 32 || k.Cout == 64) && k.res_mode == RES_NONE &&
           k.fmt_x == FMT_F32 && k.H == k.Ho && k.W == k.Wo;
}

void launch_conv_direct(const ConvKernelArgs &k, hipStream_t s) {
    if (k.Cout == 32) launch_direct_cout<32>(k, s);
    else launch_direct_cout<64>(k, s);
}

}  // namespace yds
