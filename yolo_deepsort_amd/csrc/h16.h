// "H16" activation format of the f16x3 convolution path.
//
// A tensor whose channel count is a multiple of 32 may be stored pre-split for the split-fp16 MFMA kernel: every
// pixel keeps, per group of 32 channels, 128 bytes = [32 x hi fp16 | 32 x lo fp16] with
//     x * 2^-8 = hi + lo * 2^-11        hi = fp16(x * 2^-8),  lo = fp16((x * 2^-8 - hi) * 2^11)
// i.e. 22 significant bits per value in the same 4 bytes per channel as fp32.  Byte-wise the tensor looks like an
// fp32 NHWC tensor whose "float slots" [32g, 32g+32) hold group g, so copies, nearest upsampling and channel views
// at 32-channel granularity are format agnostic.  The conv kernel stages such a tensor into LDS as opaque 16-byte
// chunks (no conversion arithmetic in the K loop); producers encode in their epilogues.
#pragma once
#include <hip/hip_runtime.h>

namespace yds {

// FMT_F16 (round 4, half mode only): the hi halves alone, 2 bytes per channel - per pixel and group of 64 channels 128 bytes =
// [64 x fp16(x * 2^-8)], i.e. a plain NHWC fp16 tensor of scaled values.  Views of such a tensor keep their strides and channel
// offsets in FLOAT SLOTS (4-byte units) like every other view: `ld` = channels / 2 and a channel offset c sits c / 2 slots in, so
// byte-wise copies, nearest upsampling and 64-channel-granular concatenation stay format agnostic (`fmt_slots`).
enum TensorFmt { FMT_F32 = 0, FMT_H16 = 1, FMT_F16 = 2 };
// float slots that `c` channels occupy in a pixel of format `fmt`
__host__ __device__ inline int fmt_slots(int fmt, int c) { return fmt == FMT_F16 ? c / 2 : c; }

constexpr float H16_A_SCALE = 1.f / 256.f, H16_LO_SCALE = 2048.f;

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// RANGE.  hi = fp16(x * 2^-8) holds |x| up to 65504 * 256 = 1.677e7 where the reference's fp32 tensors reach 3.4e38.  Every kernel
// that encodes H16 / F16 values starts with fp16_saturate_on(): MODE.FP16_OVFL (bit 23 of the wave's MODE register) makes a
// float -> fp16 conversion that overflows return +-65504 instead of +-inf (true infinities and NaNs pass), so an activation
// beyond the range SATURATES at +-(65504 + 32) * 256 (hi and the lo term both clamp) instead of decoding to inf - inf = NaN
// and poisoning everything downstream.  tests/test_gpu_conv_variants.py::test_h16_range_saturates pins the behaviour; activations
// of that size do not occur in a trained detector (BatchNorm keeps them O(1..1e3)), the clamp is a guard, not a feature.
__device__ __forceinline__ void fp16_saturate_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }

__device__ __forceinline__ void h16_encode4(const float v[4], h16x4 &hi, h16x4 &lo) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float xs = v[c] * H16_A_SCALE;
        _Float16 h = (_Float16)xs;                         // round to nearest even
        hi[c] = h;
        lo[c] = (_Float16)((xs - (float)h) * H16_LO_SCALE);
    }
}

// the same for values that are ALREADY in the scaled domain (xs = x * 2^-8): producers whose accumulators come out scaled and whose
// activation is positively homogeneous (LeakyReLU, ReLU, linear) skip the unscale / rescale pair
__device__ __forceinline__ void h16_encode4_scaled(const float xs[4], h16x4 &hi, h16x4 &lo) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        _Float16 h = (_Float16)xs[c];
        hi[c] = h;
        lo[c] = (_Float16)((xs[c] - (float)h) * H16_LO_SCALE);
    }
}

__device__ __forceinline__ void h16_decode4(const h16x4 &hi, const h16x4 &lo, float v[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = ((float)hi[c] + (float)lo[c] * (1.f / H16_LO_SCALE)) * (1.f / H16_A_SCALE);
}

// pixel: pointer to the pixel's first float slot; c: channel index, multiple of 4
__device__ __forceinline__ void h16_load4(const float *pixel, int c, float v[4]) {
    const char *g = reinterpret_cast<const char *>(pixel + (c & ~31)) + (c & 31) * 2;
    h16x4 hi = *reinterpret_cast<const h16x4 *>(g), lo = *reinterpret_cast<const h16x4 *>(g + 64);
    h16_decode4(hi, lo, v);
}
__device__ __forceinline__ void h16_store4(float *pixel, int c, const float v[4]) {
    char *g = reinterpret_cast<char *>(pixel + (c & ~31)) + (c & 31) * 2;
    h16x4 hi, lo;
    h16_encode4(v, hi, lo);
    *reinterpret_cast<h16x4 *>(g) = hi;
    *reinterpret_cast<h16x4 *>(g + 64) = lo;
}

// FMT_F16: four consecutive channels = 8 bytes at byte offset 2 c of the pixel
__device__ __forceinline__ void f16_load4(const float *pixel, int c, float v[4]) {
    const h16x4 h = *reinterpret_cast<const h16x4 *>(reinterpret_cast<const char *>(pixel) + c * 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (float)h[k] * (1.f / H16_A_SCALE);
}
__device__ __forceinline__ void f16_store4(float *pixel, int c, const float v[4]) {
    h16x4 h;
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = (_Float16)(v[k] * H16_A_SCALE);
    *reinterpret_cast<h16x4 *>(reinterpret_cast<char *>(pixel) + c * 2) = h;
}

// generic 4-channel accessors on a (pointer, fmt) pair
__device__ __forceinline__ void load4(const float *pixel, int c, int fmt, float v[4]) {
    if (fmt == FMT_H16) h16_load4(pixel, c, v);
    else if (fmt == FMT_F16) f16_load4(pixel, c, v);
    else {
        float4 t = *reinterpret_cast<const float4 *>(pixel + c);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
}
__device__ __forceinline__ void store4(float *pixel, int c, int fmt, const float v[4]) {
    if (fmt == FMT_H16) h16_store4(pixel, c, v);
    else if (fmt == FMT_F16) f16_store4(pixel, c, v);
    else *reinterpret_cast<float4 *>(pixel + c) = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace yds
