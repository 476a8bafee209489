// Window-resident 3x3 / stride-1 convolution on the fp16 matrix cores (f16x3 arithmetic, see conv_f16x3.hip).
//
// The LDS-DMA implicit-GEMM kernel fetches the activation tile once per filter tap: nine times the same pixels,
// shifted.  Its K loop is bound by the number of global->LDS instructions a wave has to issue between its MFMAs
// (profiles/r01_ablation_dma.txt), so this kernel stops re-fetching: for a 3x3, stride 1, pad 1 layer the output pixel
// with flat index m (over batch, rows, columns) reads input pixel m + dy*W + dx, hence the 256 output pixels of a tile
// need the CONTIGUOUS run of input pixels [m0 - W - 1, m0 + 256 + W + 1).  That window (one 32-channel group = 128 B
// per pixel) is brought into LDS once per channel group and all nine taps read their A fragments from it at a row
// offset of (dy+1)*W + (dx+1); taps that fall outside the image (left/right column, top/bottom row, tile tail) are
// redirected per lane to a zero row.  Only the filter rows (BN x 128 B per tap) still stream through a 3-stage ring.
//   K order: channel group outer, tap inner (weights stay [Cout][kh][kw][Cin]: chunk index tap*G + g)
//   global->LDS instructions per wave and K step: 2 (filters) + <= 1 (next group's window), against 6-8 before
//   LDS: two window buffers (double buffered across channel groups) + 3 filter stages + the zero row <= 160 KB, one
//   512-thread workgroup per CU (8 waves, 4x2, 64x64 accumulator tiles), W <= 95 (single-group layers: one window
//   buffer, W <= 318)
// Same swizzle as the DMA kernel: row r keeps its 16-byte chunk c at position c ^ ((r >> 1) & 7), applied on the source
// address of the DMA and again by the fragment reads; 16 consecutive rows hit 16 distinct bank slots at any base.
#include "conv_common.h"

namespace yds {

// Sustained shader clock INSIDE the kernel: one workgroup in 32 samples the shader-cycle counter (s_memtime) and the constant
// 100 MHz counter (s_memrealtime) at its start and end; cycles / ticks is the clock the chip really ran at while every CU
// was busy with this kernel (it is power limited: ~1.55 GHz, not the 2.4 GHz the MFMA peak is quoted at).
#ifdef YDS_CLOCK_PROBE
__device__ unsigned long long yds_clk_win[2];
#endif
namespace {

constexpr int BM = 256, NW = 8, NT = NW * 64;
constexpr int NSB = 3;                         // filter-stage ring depth
constexpr int ROW = 128;
constexpr int APW = 7;                         // window DMA instructions per wave per channel group (8 rows each)
constexpr int MAX_WROWS = APW * NW * 8;        // 448 window rows

// BN x (WM x WN waves): 128 x (4x2) = 64x64 accumulator tiles per wave; 64 x (8x1) / 64 x (4x2) for 64-filter layers
// This file holds the HALF-MODE tiers of the window kernel (Darknet.half()); the default f16x3 arithmetic runs conv_win16.hip.
// TERMS: 1 = half mode (hi halves of both operands only: no lo fragment reads, one MFMA per product block - compile-time
// pruning of the three-term slot plan; a round-3 tier that computed the cross terms in fp8, TERMS = 2, was removed in round 4:
// with the default kernel on v_mfma_f32_16x16x32_f16 it was no longer faster - 1487 against 1508 frames/s);
// 4 = half mode with 64 channels per K step: the LDS rows (window and filter stages alike) are GATHERED by the DMA from the hi halves
// of two consecutive 32-channel groups (a DMA lane's global address is free), [hi of group 2q | hi of group 2q+1], so the "lo" fragment
// slots hold the second group's hi values and the step does A_hi x B_hi + A_lo x B_lo - the same DMA instructions, fragment reads and
// barrier per step as TERMS = 1 for twice the channels, i.e. half the steps (a K step is bound by those, not by its MFMAs, once two
// thirds of them are gone: profiles/r03_fp8_cross.txt #5).  Needs Cin % 64 == 0; the tensors in HBM stay H16.
template <int BN, int WM, int WN, int ACT, int RES, int TERMS>
__global__ __launch_bounds__(NT, 1) void conv3x3_f16x3_win(ConvKernelArgs p, int wrows, int nbuf) {
    fp16_saturate_on();
    static_assert(WM * WN == NW, "eight waves");
    static_assert(TERMS == 1 || TERMS == 4, "the default arithmetic (three fp16 terms) is conv_win16.hip's kernel");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int B_STAGE = BN * ROW;
    constexpr int B_INST = BN / (8 * NW);      // filter DMA instructions per wave per stage (8 rows each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int WB = wrows * ROW;                                  // bytes per window buffer
    char *bring = smem + nbuf * WB;                              // [NSB][BN][128]; nbuf = 2 window buffers, 1 for single-group layers
    const int zoff = nbuf * WB + NSB * B_STAGE;                  // zero row

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
    if (tid < 32) reinterpret_cast<float *>(smem + zoff)[tid] = 0.f;
#ifdef YDS_CLOCK_PROBE
    const bool clk_sample = tid == 0 && (blockIdx.x & 31) == 0;
    unsigned long long clk_c0 = 0, clk_w0 = 0;
    if (clk_sample) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_w0 = wall_clock64(); }
#endif

    const int W = p.W, G = TERMS == 4 ? p.Cin / 64 : p.Cin / 32;       // channel groups per K step
    constexpr int GROUP_BYTES = TERMS == 4 ? 256 : 128;
    // logical 16-byte chunk c of an LDS row -> chunk of the global record(s): TERMS = 4 takes chunks 0-3 (the hi halves) of two records
    auto gchunk = [&](int c) { return TERMS == 4 ? ((c >> 2) << 3) + (c & 3) : c; };
    // An FMT_F16 activation tensor (2-byte half-mode activations, TERMS = 4 only) already is the gathered row: 64 channels = 128
    // contiguous bytes with chunk c in place; the filters keep the split record.  Strides stay in float slots (h16.h).
    const bool xs = TERMS == 4 && p.fmt_x == FMT_F16;
    const int x_group_bytes = xs ? 128 : GROUP_BYTES;
    const int drow = lane >> 3, dpos = lane & 7;
    // window pieces: piece pc covers window rows pc*8 .. pc*8+7, wave w issues pieces w, w+8, ...; row j <-> flat input
    // pixel m0 - W - 1 + j (clamped into the tensor: rows outside it are never read unmasked); offsets in 16-byte units
    const int npieces = wrows / 8;
    unsigned w_off16[B_INST];
#pragma unroll
    for (int b = 0; b < B_INST; ++b) {
        const int row = (b * NW + wave) * 8 + drow;
        w_off16[b] = (unsigned)min(n0 + row, p.Cout - 1) * (unsigned)(p.Kpad / 4) + (unsigned)gchunk(dpos ^ ((row >> 1) & 7));
    }
    const char *x_bytes = reinterpret_cast<const char *>(p.x), *w_bytes = reinterpret_cast<const char *>(p.w);
    auto a_piece = [&](int g, int k) {                           // window of channel group g -> buffer g & 1
        const int pc = min(k * NW + wave, npieces - 1);        // surplus instructions repeat the last piece (same data, same place)
        const int j = pc * 8 + drow;
        const int f = min(max(m0 - W - 1 + j, 0), p.M - 1);
        const int c = dpos ^ ((j >> 1) & 7);
        const unsigned off16 = (unsigned)f * (unsigned)(p.ldx / 4) + (unsigned)(xs ? c : gchunk(c));
        const char *src = x_bytes + (size_t)g * x_group_bytes + ((size_t)off16 << 4);
        __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(smem + (g & 1) * WB + pc * 8 * ROW), 16, 0, 0);
    };
    auto b_piece = [&](int g, int tap, int stage, int b) {       // filter rows of K chunk (tap, g)
        const char *src = w_bytes + (size_t)(tap * G + g) * GROUP_BYTES + ((size_t)w_off16[b] << 4);
        __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(bring + stage * B_STAGE + (b * NW + wave) * 8 * ROW), 16, 0, 0);
    };

    // per-lane validity of the nine taps for the two A fragments of this wave (rows wm*64 + i*32 + (lane & 31))
    int r_row[TM];
    unsigned ok9[TM];
    {
        const int HW = p.H * W;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * (BM / WM) + i * 32 + (lane & 31), m = m0 + r;
            r_row[i] = r;
            unsigned bits = 0;
            if (m < p.M) {
                const int rem = m % HW, y = rem / W, x = rem - y * W;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    bits |= ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)W ? 1u : 0u) << t;
                }
            }
            ok9[i] = bits;
        }
    }

    f32x16 acc1[TM][TN], acc2[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc1[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

    // filter fragment addressing (as in the DMA kernel)
    const int swz = (lane >> 1) & 7, kb = lane >> 5;
    const int b_frag = (wn * (BN / WN) + (lane & 31)) * ROW;
    int bpos_hi[2], bpos_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { bpos_hi[s] = ((2 * s + kb) ^ swz) * 16; bpos_lo[s] = ((4 + 2 * s + kb) ^ swz) * 16; }

    constexpr int NF = 2 * (TM + TN), NM = 3 * TM * TN;
    h8 fr[2][NF];                                               // [substep][A0h, A0l, A1h, A1l, B0h, B0l, B1h, B1l]
    int a_addr[TM], a_sw[TM];                                   // this tap: LDS byte address of the lane's window row (or the zero row), swizzle
    auto frag_read = [&](const char *bst, int s, int f) {
        const int which = f / 2, lo = f & 1;
        if (TERMS == 1 && lo) return;
        if (which < TM) {
            const int c = (lo ? 4 : 0) + 2 * s + kb;
            fr[s][f] = *reinterpret_cast<const h8 *>(smem + a_addr[which] + ((c ^ a_sw[which]) << 4));
        } else {
            fr[s][f] = *reinterpret_cast<const h8 *>(bst + b_frag + (which - TM) * 32 * ROW + (lo ? bpos_lo[s] : bpos_hi[s]));
        }
    };
    auto mfma = [&](int s, int m0) {
        const int m = m0;
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
    // substep-1 fragment order: the operands of accumulator tile (0,0) first
    auto frag_order = [&](int k) {
        if (k < 2) return k;
        if (k < 4) return 2 * TM + (k - 2);
        const int r = k - 4, nb = 2 * (TN - 1);
        return r < nb ? 2 * TM + 2 + r : 2 + (r - nb);
    };

    // One K step = tap TAP of channel group g, software pipelined across steps (the wave issues in order and an MFMA
    // occupies the matrix pipe for 32 cycles, so every memory instruction sits in the shadow of one MFMA, order pinned):
    //   substep 0   12 MFMAs on fr[0] (read during the previous step)    slots: this step's substep-1 fragments -> fr[1]
    //   mid         s_waitcnt vmcnt(0) + s_barrier: the filter stage of step t+1 (and, before a new group, its window)
    //               has landed for every wave, and every wave is done with step t-1
    //   substep 1   12 MFMAs on fr[1]      slots: window piece of group g+1, filter pieces of step t+2 (into the stage
    //               step t-1 used), then the substep-0 fragments of step t+1 -> fr[0]
    // so the only exposed latency per step is the barrier itself.
    auto tap_addr = [&](int g, int tap) {                        // A-fragment row addresses for (g, tap)
        const int shift = (tap / 3) * W + (tap % 3);            // (dy+1)*W + (dx+1)
        const int wbase = (g & 1) * WB;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            asm volatile("" : "+v"(r_row[i]), "+v"(ok9[i]));      // keep the per-tap addresses out of loop-invariant hoisting (registers)
            const bool ok = (ok9[i] >> tap) & 1u;
            const int j = r_row[i] + shift;
            a_addr[i] = ok ? wbase + j * ROW : zoff;
            a_sw[i] = ok ? (j >> 1) & 7 : 0;
        }
    };
    auto step = [&](int g, auto tap_c, auto last_c) {
        constexpr int TAP = decltype(tap_c)::value;
        constexpr bool LAST = decltype(last_c)::value;          // last channel group: no window prefetch, filter refills stop
        constexpr int AHEAD = 2;             // prefetch distance of the filter stages, in steps
        constexpr bool REFILL = !(LAST && TAP + AHEAD > 8);     // a step t+AHEAD exists
        constexpr bool NEXT = !(LAST && TAP == 8);              // a step t+1 exists
        constexpr int TAP1 = (TAP + 1) % 9, TAP2 = (TAP + AHEAD) % 9;
        constexpr int OPS = (1 + B_INST + NF + NM - 1) / NM;    // memory operations per substep-1 slot
        static_assert(NF <= NM, "not enough MFMA slots in substep 0");
        const int g1 = TAP + 1 >= 9 ? g + 1 : g, g2 = TAP + AHEAD >= 9 ? g + 1 : g;
        const char *bst = bring + (TAP % NSB) * B_STAGE, *bst1 = bring + ((TAP + 1) % NSB) * B_STAGE;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma(0, m);
            __builtin_amdgcn_sched_barrier(0);
            if (m < NF) frag_read(bst, 1, frag_order(m));
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (NEXT) tap_addr(g1, TAP1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma(1, m);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int o = m * OPS; o < (m + 1) * OPS; ++o) {     // memory operations of this slot
                if (o == 0) { if (!LAST && TAP < APW) a_piece(g + 1, TAP); }
                else if (o - 1 < B_INST) { if (REFILL) b_piece(g2, TAP2, (TAP + AHEAD) % NSB, o - 1); }
                else if (o - 1 - B_INST < NF) { if (NEXT) frag_read(bst1, 0, o - 1 - B_INST); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto group = [&](int g, auto last_c) {
        step(g, std::integral_constant<int, 0>{}, last_c);
        step(g, std::integral_constant<int, 1>{}, last_c);
        step(g, std::integral_constant<int, 2>{}, last_c);
        step(g, std::integral_constant<int, 3>{}, last_c);
        step(g, std::integral_constant<int, 4>{}, last_c);
        step(g, std::integral_constant<int, 5>{}, last_c);
        step(g, std::integral_constant<int, 6>{}, last_c);
        step(g, std::integral_constant<int, 7>{}, last_c);
        step(g, std::integral_constant<int, 8>{}, last_c);
    };

    // prologue: window of group 0, filter stages of steps 0 and 1, fragments of step 0 / substep 0
    for (int k = 0; k < (npieces + NW - 1) / NW; ++k) a_piece(0, k);
#pragma unroll
    for (int b = 0; b < B_INST; ++b) b_piece(0, 0, 0, b);
#pragma unroll
    for (int b = 0; b < B_INST; ++b) b_piece(0, 1, 1, b);
    wait_vmcnt<B_INST>();              // later stages may still be in flight: the mid-step waits cover them
    __syncthreads();                                            // window 0, stage 0 and the zero row are in LDS
    tap_addr(0, 0);
#pragma unroll
    for (int f = 0; f < NF; ++f) frag_read(bring, 0, f);
    __builtin_amdgcn_sched_barrier(0);

    for (int g = 0; g + 1 < G; ++g) group(g, std::false_type{});
    group(G - 1, std::true_type{});

    __syncthreads();                                            // every wave is done with the window and the ring
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc1[i][j][e] *= 1.f / A_SCALE;                       // every tier of this kernel keeps one accumulator set
    conv_epilogue<BM, BN, WM, WN, ACT, RES, TM, TN, NT, true>(p, acc1, reinterpret_cast<float *>(smem), m0, n0, tid);   // whole-tile staging
#ifdef YDS_CLOCK_PROBE
    if (clk_sample) {
        atomicAdd(&yds_clk_win[0], __builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(&yds_clk_win[1], wall_clock64() - clk_w0);
    }
#endif
}

int window_rows(int W) { return (BM + 2 * W + 2 + 7) / 8 * 8; }

template <int BN, int WM, int WN, int ACT, int RES, int TERMS> void launch_inst_win(ConvKernelArgs k, hipStream_t s) {
    const int wrows = window_rows(k.W), nbuf = k.Cin == (TERMS == 4 ? 64 : 32) ? 1 : 2;
    // (the epilogue stages the whole 256 x BN tile in the same LDS: narrow images need more than their windows + ring)
    const size_t smem = std::max((size_t)nbuf * wrows * ROW + (size_t)NSB * BN * ROW + ROW, conv_stage_bytes(BM, BN));
    static size_t attr_set = 0;
    auto kern = conv3x3_f16x3_win<BN, WM, WN, ACT, RES, TERMS>;
    if (smem > attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = smem;
    }
    dim3 grid(plan_tile_map(k, BM, BN));
    hipLaunchKernelGGL(kern, grid, dim3(NT), smem, s, k, wrows, nbuf);
    YDS_HIP(hipGetLastError());
}

}  // namespace

void conv_win_clock(unsigned long long *cycles_ticks, bool reset) {
#ifdef YDS_CLOCK_PROBE
    YDS_HIP(hipMemcpyFromSymbol(cycles_ticks, HIP_SYMBOL(yds_clk_win), 2 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[2] = {};
        YDS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(yds_clk_win), z, sizeof z));
    }
#else
    cycles_ticks[0] = cycles_ticks[1] = 0;     // product build: no sampling inside the kernel (clock_probe.hip measures beside it)
    (void)reset;
#endif
}

bool conv_win_applicable(const ConvKernelArgs &k) {
    const bool presplit = (k.fmt_x == FMT_H16 && k.Cin % 32 == 0) || (k.fmt_x == FMT_F16 && k.terms == 1 && k.Cin % 64 == 0);
    if (!(k.ksize == 3 && k.stride == 1 && k.pad == 1 && presplit && k.H == k.Ho && k.W == k.Wo)) return false;
    // (half mode runs 64 channels per K step - launch_conv_win below - so a 64-channel layer is a single group there)
    const bool wide_step = k.terms == 1 && k.Cin % 64 == 0 && (k.fmt_x == FMT_F16 || !getenv("YDS_HALF_NARROW"));
    const int wrows = window_rows(k.W), nbuf = k.Cin == (wide_step ? 64 : 32) ? 1 : 2;
    // several channel groups: the next group's window is fetched by at most APW instructions per wave while this one is
    // consumed (two buffers); a single group needs one buffer only, which admits much wider images
    if (nbuf == 2 && wrows > MAX_WROWS) return false;
    // (the epilogue stages BM x (BN+4) floats in the same LDS: 135 KB at BN = 128)
    return (size_t)nbuf * wrows * ROW + (size_t)NSB * 128 * ROW + ROW <= 160 * 1024 && (size_t)k.M * (k.ldx / 4) < (1ull << 32);
}

void launch_conv_win(ConvKernelArgs k, int shape, hipStream_t s) {
    if (!conv_win_applicable(k)) fail("conv: the window-resident kernel needs a 3x3 stride-1 layer with a pre-split input and W <= 95 (W <= 318 for 32 input channels)");
    if (k.terms == 1 && k.Cin % 64 == 0 && (k.fmt_x == FMT_F16 || !getenv("YDS_HALF_NARROW"))) {   // half mode, 64 channels per step (YDS_HALF_NARROW: tuning aid, the 32-channel form)
        if (shape == 0) {
#define YDS_CALL(A, R) launch_inst_win<128, 4, 2, A, R, 4>(k, s)
            YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
        } else {
#define YDS_CALL(A, R) launch_inst_win<64, 4, 2, A, R, 4>(k, s)
            YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
        }
    } else if (k.terms == 1) {                                   // half mode: one instantiation family (256x128, or 256x64 for narrow layers)
        if (shape == 0) {
#define YDS_CALL(A, R) launch_inst_win<128, 4, 2, A, R, 1>(k, s)
            YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
        } else {
#define YDS_CALL(A, R) launch_inst_win<64, 4, 2, A, R, 1>(k, s)
            YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
        }
    } else {                                                     // default arithmetic: the v_mfma_f32_16x16x32_f16 form (conv_win16.hip)
        launch_conv_win16(k, shape, s);
    }
}

}  // namespace yds
