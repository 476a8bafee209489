// First-layer convolution: 3x3, stride 1, RGB (channel-padded to 4) -> 32 or 64 channels.
//
// The implicit-GEMM kernels are a poor fit here: K = 27 (padded to 64) and the layer is bound by writing
// 608*608*32 outputs per image, not by arithmetic.  This kernel is a direct convolution on the vector ALU:
// COUT/4 neighbouring lanes own one output pixel (4 output channels each, their 4x27 weights live in registers
// for the whole kernel), so a wave's store instruction writes whole contiguous pixel rows (128 B / 256 B per
// pixel) and the 9 input taps of a pixel are broadcast loads.  Arithmetic is a plain fp32 fma chain in both math
// modes (yolov3/yolov4 layer 0: yolo3/models/models.py:36-56; ReID stem: deep_sort/deep/model.py:52-60).
#include "conv_common.h"

namespace yds {

template <int COUT, int ACT>
__global__ __launch_bounds__(256) void conv3x3_rgb_direct(ConvKernelArgs p) {
    constexpr int QUADS = COUT / 4;                 // lanes per pixel
    constexpr int PIX_PER_BLOCK = 256 / QUADS;
    const int q = threadIdx.x % QUADS, slot = threadIdx.x / QUADS;
    // this lane's weights: 4 output channels x 9 taps x 3 input channels (+ bias)
    float w[4][9][3], b[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float *wr = p.w + (size_t)(q * 4 + o) * p.Kpad;
        b[o] = p.bias[q * 4 + o];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) w[o][t][c] = wr[t * 4 + c];
    }
    const int HW = p.H * p.W;
    for (int pix = blockIdx.x * PIX_PER_BLOCK + slot; pix < p.M; pix += gridDim.x * PIX_PER_BLOCK) {
        const int img = pix / HW, rem = pix - img * HW;
        const int oy = rem / p.W, ox = rem - oy * p.W;
        float acc[4] = {b[0], b[1], b[2], b[3]};
        // all nine taps are fetched branch-free (clamped address, zeroed afterwards) so the loads overlap
        float4 tap[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
            float4 v = *reinterpret_cast<const float4 *>(p.x + ((size_t)(img * p.H + cy) * p.W + cx) * p.ldx);
            // arithmetic mask, not a select: a select lets the compiler sink the load into a branch again
            // (nine serialized round trips); image values are finite, so v * 0 is an exact zero
            const float keep = ok ? 1.f : 0.f;
            tap[t] = make_float4(v.x * keep, v.y * keep, v.z * keep, 0.f);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                acc[o] = fmaf(tap[t].x, w[o][t][0], acc[o]);
                acc[o] = fmaf(tap[t].y, w[o][t][1], acc[o]);
                acc[o] = fmaf(tap[t].z, w[o][t][2], acc[o]);
            }
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = apply_act<ACT>(acc[o]);
        store4(p.y + (size_t)pix * p.ldy, q * 4, p.fmt_y, acc);
    }
}

template <int COUT> static void launch_direct_cout(const ConvKernelArgs &k, hipStream_t s) {
    const int pix_per_block = 256 / (COUT / 4);
    long blocks = ((long)k.M + pix_per_block - 1) / pix_per_block;
    if (blocks > 256 * 16) blocks = 256 * 16;             // grid-stride: weights are loaded once per lane
    dim3 grid((unsigned)blocks);
    switch (k.act) {
        case ACT_LEAKY: hipLaunchKernelGGL((conv3x3_rgb_direct<COUT, ACT_LEAKY>), grid, dim3(256), 0, s, k); break;
        case ACT_MISH: hipLaunchKernelGGL((conv3x3_rgb_direct<COUT, ACT_MISH>), grid, dim3(256), 0, s, k); break;
        case ACT_RELU: hipLaunchKernelGGL((conv3x3_rgb_direct<COUT, ACT_RELU>), grid, dim3(256), 0, s, k); break;
        default: hipLaunchKernelGGL((conv3x3_rgb_direct<COUT, ACT_LINEAR>), grid, dim3(256), 0, s, k); break;
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
