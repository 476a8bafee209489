// Semantics of the gfx950 fp8 pieces the f16 + fp8-cross-term convolution mode relies on (one-off check, results in
// profiles/r03_fp8_cross.txt):  v_cvt_scalef32_pk_fp8_f16 (scale direction, rounding, saturation) and the operand layout /
// scale exponents of v_mfma_scale_f32_32x32x64_f8f6f4 (lane l: row l & 31, its 32 bytes = K block l >> 5).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void cvt_kernel(const _Float16 *in, unsigned char *out, int n, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    h2 v = {in[2 * i], in[2 * i + 1]};
    s2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16((s2){0, 0}, v, scale, false);
    out[2 * i] = (unsigned char)(r[0] & 0xff);
    out[2 * i + 1] = (unsigned char)((r[0] >> 8) & 0xff);
}

// one wave: D = A (32 x 64 fp8, row major bytes) x B^T (32 x 64 fp8, row n holds column n's K values), scales given
__global__ void mfma_kernel(const unsigned char *A, const unsigned char *B, float *D, int sa, int sb) {
    const int l = threadIdx.x;
    i8v a, b;
    for (int k = 0; k < 8; ++k) {
        a[k] = *reinterpret_cast<const int *>(A + (l & 31) * 64 + (l >> 5) * 32 + 4 * k);
        b[k] = *reinterpret_cast<const int *>(B + (l & 31) * 64 + (l >> 5) * 32 + 4 * k);
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[e];
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) return NAN;
    const float mag = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6);
    return s ? -mag : mag;
}
static unsigned char q_e4m3(float x) {                       // RNE, saturating
    const unsigned char s = x < 0 ? 0x80 : 0;
    float a = fabsf(x);
    if (a >= 448.f) return s | 0x7e;
    int ex; frexpf(a, &ex); ex -= 1;
    if (ex < -6) ex = -6;
    const float step = ldexpf(1.f, ex - 3);
    const float r = nearbyintf(a / step) * step;
    if (r == 0.f) return s;
    int e2; const float fr = frexpf(r, &e2); e2 -= 1;
    if (e2 < -6) return s | (unsigned char)lrintf(r / ldexpf(1.f, -9));
    return s | (unsigned char)(((e2 + 7) << 3) | (int)lrintf((fr * 2.f - 1.f) * 8.f));
}

int main() {
    // 1. conversion
    const int n = 4096;
    _Float16 *h = (_Float16 *)malloc(n * 2), *d;
    unsigned char *o = (unsigned char *)malloc(n), *dout;
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float mag = ldexpf(1.f + (rand() % 1024) / 1024.f, rand() % 24 - 14);
        h[i] = (_Float16)((rand() & 1) ? -mag : mag);
    }
    h[0] = (_Float16)448.f; h[1] = (_Float16)449.f; h[2] = (_Float16)480.f; h[3] = (_Float16)1000.f; h[4] = (_Float16)60000.f; h[5] = (_Float16)0.f;
    hipMalloc(&d, n * 2); hipMalloc(&dout, n);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    for (float scale : {1.f, 4.f, 0.25f}) {
        hipLaunchKernelGGL(cvt_kernel, dim3(n / 2 / 64), dim3(64), 0, 0, d, dout, n, scale);
        hipMemcpy(o, dout, n, hipMemcpyDeviceToHost);
        int bad_div = 0, bad_mul = 0;
        for (int i = 0; i < n; ++i) {
            bad_div += o[i] != q_e4m3((float)h[i] / scale);
            bad_mul += o[i] != q_e4m3((float)h[i] * scale);
        }
        printf("cvt_scalef32_pk_fp8_f16 scale %.2f: mismatches against q(x / scale) %d, against q(x * scale) %d of %d\n", scale, bad_div, bad_mul, n);
        if (scale == 1.f) for (int i = 0; i < 6; ++i) printf("   %g -> 0x%02x (%g)   emulation 0x%02x\n", (float)h[i], o[i], e4m3(o[i]), q_e4m3((float)h[i]));
    }
    // 2. MFMA layout and scales
    unsigned char A[32 * 64], B[32 * 64], *dA, *dB;
    float D[32 * 32], *dD;
    for (int i = 0; i < 32 * 64; ++i) { A[i] = q_e4m3((rand() % 33 - 16) / 4.f); B[i] = q_e4m3((rand() % 29 - 14) / 8.f); }
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dD, sizeof D);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    for (int t = 0; t < 2; ++t) {
        const int sa = t ? 127 - 3 : 127, sb = t ? 127 + 1 : 127;
        hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
        hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0;
                for (int k = 0; k < 64; ++k) s += (double)e4m3(A[i * 64 + k]) * e4m3(B[j * 64 + k]);
                s *= ldexp(1.0, (sa - 127) + (sb - 127));
                worst = fmax(worst, fabs(s - D[i * 32 + j]));
            }
        printf("mfma_scale_f32_32x32x64_f8f6f4 scale bytes (%d, %d): max |D - A.B^T * 2^(sa+sb-254)| = %g\n", sa, sb, worst);
    }
    return 0;
}
