// Implicit-GEMM convolution for gfx950 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the conv -> BatchNorm -> LeakyReLU/Mish (-> shortcut add) chains that the reference
// runs as separate ATen ops (yolo3/models/models.py:36-56,298-306; deep_sort/deep/model.py:5-37).
//
//   GEMM view:  M = N*Ho*Wo output pixels, N = Cout, K = ksize*ksize*Cin, no im2col buffer.
//   Activations are NHWC fp32, so for a fixed filter tap (kh,kw) the K-slice of an output pixel is a
//   contiguous run of input channels: every global load is a 16-byte, channel-contiguous float4.
//   Weights are pre-packed [Cout][Kpad] with the same K order and BN already folded in.
//   A workgroup (4 waves, 256 threads) owns a BM x BN output tile; K is walked in steps of 32 through
//   a two-stage LDS ring (register prefetch of tile t+1 while tile t feeds the MFMAs, one barrier per
//   step).  LDS rows are padded to 36 floats so that ds_read_b128 fragment reads are conflict free.
//   Fragment trick: lane (i, kk) reads 4 consecutive k for its row with one ds_read_b128 and issues 4
//   MFMAs, MFMA c consuming component c from both operands - the K order inside a step is permuted
//   identically for A and B, which a dot product does not care about.
//   Epilogue: bias + activation (+ residual, before or after the activation) and 128-byte row stores.
#include "common.h"

namespace yds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvKernelArgs {
    const float *x, *w, *bias, *res;
    float *y;
    int H, W, Cin, ldx;
    int Ho, Wo, Cout, ldy, ldr;
    int ksize, stride, pad;
    int K, Kpad, M;
    int act, res_mode;
};

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_LEAKY: return v > 0.f ? v : v * 0.1f;
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_MISH: {
            float sp = v > 20.f ? v : log1pf(expf(v));
            return v * tanhf(sp);
        }
        default: return v;
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(ConvKernelArgs p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;   // float4 rows per thread per K step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                                   // [2][BM][LDS_LD]
    float *Bs = smem + 2 * BM * LDS_LD;                 // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // blockIdx.x walks M fastest so that neighbouring workgroups share the weight tile in L2
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    const int cq = tid & 7;          // which float4 of the 32-wide K step this thread stages
    const int r0 = tid >> 3;         // first staged row; further rows at +32

    // per staged A row: input pixel origin for filter tap (0,0)
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
            a_iy[i] = -(1 << 28);    // never in bounds
            a_ix[i] = 0;
            a_base[i] = 0;
        }
    }
    // running decomposition of this thread's k = kt*32 + cq*4 into (kh, kw, c)
    int kk = cq * 4, kh = 0, kw = 0, kc = kk;
    while (kc >= p.Cin) { kc -= p.Cin; if (++kw == p.ksize) { kw = 0; ++kh; } }

    f32x4 a_reg[A_ROWS], b_reg[B_ROWS];
    auto load_tiles = [&]() {
        const int tap_off = (kh * p.W + kw) * p.ldx + kc;
        const bool k_ok = kk < p.K;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            int iy = a_iy[i] + kh, ix = a_ix[i] + kw;
            bool ok = k_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            a_reg[i] = ok ? *reinterpret_cast<const f32x4 *>(p.x + (a_base[i] + tap_off)) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            int n = n0 + r0 + 32 * i;
            b_reg[i] = n < p.Cout ? *reinterpret_cast<const f32x4 *>(p.w + (size_t)n * p.Kpad + kk) : f32x4{0, 0, 0, 0};
        }
    };
    auto advance_k = [&]() {
        kk += BK;
        kc += BK;
        while (kc >= p.Cin) { kc -= p.Cin; if (++kw == p.ksize) { kw = 0; ++kh; } }
    };
    auto store_tiles = [&](int buf) {
        float *a = As + buf * BM * LDS_LD, *b = Bs + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<f32x4 *>(a + (r0 + 32 * i) * LDS_LD + cq * 4) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<f32x4 *>(b + (r0 + 32 * i) * LDS_LD + cq * 4) = b_reg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = p.Kpad / BK;
    load_tiles();
    store_tiles(0);
    __syncthreads();

    const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { advance_k(); load_tiles(); }
        const float *a = As + cur * BM * LDS_LD + (wm * (BM / WM) + frag_row) * LDS_LD + frag_k;
        const float *b = Bs + cur * BN * LDS_LD + (wn * (BN / WN) + frag_row) * LDS_LD + frag_k;
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(a + i * 32 * LDS_LD + ks * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(b + j * 32 * LDS_LD + ks * 8);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][c], bf[j][c], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int col = lane & 31, rsel = (lane >> 5) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + col;
        if (n >= p.Cout) continue;
        const float bias = p.bias[n];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * (BM / WM) + i * 32 + rsel;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m >= p.M) continue;
                float v = acc[i][j][e] + bias;
                if (p.res_mode == RES_BEFORE_ACT) v += p.res[(size_t)m * p.ldr + n];
                v = apply_act(v, p.act);
                if (p.res_mode == RES_AFTER_ACT) v += p.res[(size_t)m * p.ldr + n];
                p.y[(size_t)m * p.ldy + n] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN> static void launch_cfg(const ConvKernelArgs &k, hipStream_t s) {
    constexpr size_t smem = 2ull * (BM + BN) * LDS_LD * sizeof(float);
    static bool attr_set = false;
    auto kern = conv_igemm_f32<BM, BN, WM, WN>;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    dim3 grid((k.M + BM - 1) / BM, (k.Cout + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, k);
    YDS_HIP(hipGetLastError());
}

double conv_flops(const ConvArgs &a) {
    return 2.0 * (double)a.y.pixels() * a.y.c * a.ksize * a.ksize * a.x.c;
}

const char *conv_variant_name(int v) {
    static const char *names[kConvVariants] = {"conv_igemm_f32<128,128,2,2>", "conv_igemm_f32<128,64,2,2>", "conv_igemm_f32<64,64,2,2>", "conv_igemm_f32<128,32,4,1>"};
    return v >= 0 && v < kConvVariants ? names[v] : "?";
}

int launch_conv(const ConvArgs &a, hipStream_t s) {
    ConvKernelArgs k;
    k.x = a.x.p; k.w = a.w; k.bias = a.bias; k.res = a.res.p; k.y = a.y.p;
    k.H = a.x.h; k.W = a.x.w; k.Cin = a.x.c; k.ldx = a.x.ld;
    k.Ho = a.y.h; k.Wo = a.y.w; k.Cout = a.y.c; k.ldy = a.y.ld; k.ldr = a.res.ld;
    k.ksize = a.ksize; k.stride = a.stride; k.pad = a.pad;
    k.K = a.ksize * a.ksize * a.x.c; k.Kpad = a.kpad;
    k.M = (int)a.y.pixels();
    k.act = a.act; k.res_mode = a.res.p ? a.res_mode : RES_NONE;
    if (a.x.c % 4 || a.x.ld % 4 || ((uintptr_t)a.x.p & 15)) fail("conv: input channels/stride must be multiples of 4 (got c=%d ld=%d)", a.x.c, a.x.ld);
    if (a.kpad % BK || a.kpad < k.K) fail("conv: bad kpad %d for K=%d", a.kpad, k.K);
    if ((size_t)a.x.n * a.x.h * a.x.w * a.x.ld >= (1ull << 31)) fail("conv: input tensor too large for 32-bit indexing");
    // tile choice: widest N tile that the layer fills; fall back to smaller M tiles when the grid
    // would leave most of the 256 CUs idle
    const int M = k.M, N = k.Cout;
    auto blocks = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    if (N <= 32) {
        launch_cfg<128, 32, 4, 1>(k, s);
        return 3;
    }
    if (N <= 64) {
        if (blocks(128, 64) >= 256) { launch_cfg<128, 64, 2, 2>(k, s); return 1; }
        launch_cfg<64, 64, 2, 2>(k, s);
        return 2;
    }
    if (blocks(128, 128) >= 384) { launch_cfg<128, 128, 2, 2>(k, s); return 0; }
    if (blocks(128, 64) >= 384) { launch_cfg<128, 64, 2, 2>(k, s); return 1; }
    launch_cfg<64, 64, 2, 2>(k, s);
    return 2;
}

}  // namespace yds
