/* TEST ORACLE (C / OpenMP part of the CPU baseline).  Not linked into, loaded by, or shipped with the product.
 *
 * Multi-threaded pieces of the convolution path of the reference (yolo3/models/models.py:36-56 conv + BatchNorm2d +
 * LeakyReLU / Mish; deep_sort/deep/model.py:5-37 conv + BN + ReLU) on NHWC tensors, so that the oracle pipeline can serve
 * as a CPU baseline that uses every host core: the patch gather (im2col, memcpy of C-contiguous runs), the fused
 * scale/shift/activation epilogue, max pooling, nearest up-sampling and the layout changes run here under OpenMP; the
 * GEMM in between is numpy's BLAS sgemm.  oracle/fast.py drives it and tests/test_oracle_fast.py holds it to the plain
 * numpy oracle.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* x [B,H,W,C] -> cols [B*Ho*Wo, k*k*C], patch order (kh, kw, c); zero padding */
void im2col_nhwc(const float *x, int B, int H, int W, int C, int k, int stride, int pad, float *cols)
{
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const size_t K = (size_t)k * k * C;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int oy = 0; oy < Ho; ++oy) {
            float *row = cols + ((size_t)b * Ho + oy) * Wo * K;
            for (int ox = 0; ox < Wo; ++ox, row += K)
                for (int kh = 0; kh < k; ++kh) {
                    const int iy = oy * stride + kh - pad;
                    for (int kw = 0; kw < k; ++kw) {
                        const int ix = ox * stride + kw - pad;
                        float *dst = row + (size_t)(kh * k + kw) * C;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) memset(dst, 0, (size_t)C * sizeof(float));
                        else memcpy(dst, x + (((size_t)b * H + iy) * W + ix) * C, (size_t)C * sizeof(float));
                    }
                }
        }
}

/* in place on y [M, O]: y = act(y * scale[o] + shift[o] (+ res)); act 0 linear, 1 leaky 0.1, 2 mish, 3 relu;
 * res_mode 0 none, 1 add after the activation (Darknet shortcut), 2 add before it (BasicBlock) */
void scale_shift_act(float *y, int64_t M, int O, const float *scale, const float *shift, int act, const float *res, int res_mode)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float *r = y + m * O;
        const float *q = res ? res + m * O : 0;
        for (int o = 0; o < O; ++o) {
            float v = r[o] * scale[o] + shift[o];
            if (res_mode == 2) v += q[o];
            if (act == 1) v = v > 0.f ? v : v * 0.1f;
            else if (act == 3) v = v > 0.f ? v : 0.f;
            else if (act == 2) {
                const float sp = v > 20.f ? v : log1pf(expf(v));        /* torch softplus, threshold 20 */
                v = v * tanhf(sp);
            }
            if (res_mode == 1) v += q[o];
            r[o] = v;
        }
    }
}

/* x [B,H,W,C] -> y [B,Ho,Wo,C]; window k, stride, symmetric pad with -inf; zero_br: ZeroPad2d((0,1,0,1)) with ZEROS
 * first (models.py:61-63, the k=2 s=1 pool of the tiny nets) */
void maxpool_nhwc(const float *x, int B, int H, int W, int C, int k, int stride, int pad, int zero_br, float *y)
{
    const int Hp = H + (zero_br ? 1 : 0), Wp = W + (zero_br ? 1 : 0);
    const int Ho = (Hp + 2 * pad - k) / stride + 1, Wo = (Wp + 2 * pad - k) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                float *dst = y + (((size_t)b * Ho + oy) * Wo + ox) * C;
                for (int c = 0; c < C; ++c) dst[c] = -INFINITY;
                for (int kh = 0; kh < k; ++kh)
                    for (int kw = 0; kw < k; ++kw) {
                        const int iy = oy * stride + kh - pad, ix = ox * stride + kw - pad;
                        if (iy < 0 || ix < 0 || iy >= Hp || ix >= Wp) continue;      /* -inf padding */
                        if (iy >= H || ix >= W) {                                    /* the zero-padded bottom / right line */
                            for (int c = 0; c < C; ++c) dst[c] = dst[c] > 0.f ? dst[c] : 0.f;
                            continue;
                        }
                        const float *src = x + (((size_t)b * H + iy) * W + ix) * C;
                        for (int c = 0; c < C; ++c) dst[c] = dst[c] > src[c] ? dst[c] : src[c];
                    }
            }
}

void upsample_nhwc(const float *x, int B, int H, int W, int C, int s, float *y)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int oy = 0; oy < H * s; ++oy)
            for (int ox = 0; ox < W * s; ++ox)
                memcpy(y + (((size_t)b * H * s + oy) * W * s + ox) * C, x + (((size_t)b * H + oy / s) * W + ox / s) * C, (size_t)C * sizeof(float));
}

void add_inplace(float *y, const float *a, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] += a[i];
}

void nchw_to_nhwc(const float *x, int B, int C, int H, int W, float *y)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
                for (int c = 0; c < C; ++c) y[(((size_t)b * H + h) * W + w) * C + c] = x[(((size_t)b * C + c) * H + h) * W + w];
}

void nhwc_to_nchw(const float *x, int B, int H, int W, int C, float *y)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) y[(((size_t)b * C + c) * H + h) * W + w] = x[(((size_t)b * H + h) * W + w) * C + c];
}
