// Window-resident 3x3 / stride-1 convolution, f16x3 arithmetic, on v_mfma_f32_16x16x32_f16 (round 4).
//
// Data movement is that of conv_win.hip (read its header first): a 256-pixel tile keeps the contiguous run of input pixels
// [m0 - W - 1, m0 + 256 + W + 1) of one 32-channel group in LDS (double buffered across groups, LDS-DMA), all nine taps read
// their A fragments from it at a row offset, the filter rows stream through a three-stage ring, one barrier per K step
// (one tap of one channel group, K = 32), same row swizzle (chunk c of row r at position c ^ ((r >> 1) & 7)).
//
// What changed is the matrix instruction.  These kernels run at the chip's power limit, so joules per
// multiply-accumulate decide their speed, and on random operands the 16x16x32 form needs fewer of them than 32x32x16
// (tools/probes/mfma_shape_probe.hip, sustained launches at the cap: 2046 against 1777 TFLOP/s for the bare pipe, 1718 against
// 1577 with this kernel's fragment reads and barrier - a quarter of the accumulator traffic per flop outweighs twice the
// operand reads; on operands that never toggle the two are equal).  One instruction now spans the whole K = 32 of a step, so
// the step is split SPATIALLY instead of into two K halves: the wave's blocks of 16 rows / 16 filters form two halves,
//     H0 = A blocks [0, TM/2) + B blocks [0, TN/2),   H1 = the rest,
// and the accumulator tile four quadrants Qab = (A half a) x (B half b):
//     Q00   on H0 (read during the previous step)                 slots: this step's B fragments of H1
//     Q01                                                               slots: this step's A fragments of H1
//     mid   s_waitcnt vmcnt(0) + s_barrier: stage t+1 (and a new group's window) landed, every wave is done with step t-1
//     Q10                                                               slots: window piece of group g+1, filter pieces of step t+2,
//                                                                              then the A fragments of H0 of step t+1 (free since Q01)
//     Q11                                                               slots: the B fragments of H0 of step t+1 (free since Q10)
// Inside a quadrant the MFMAs run term-major (all hi x hi, all hi x lo, all lo x hi): no back-to-back accumulation into one
// tile.  Measured against the 32x32x16 kernel on the same box (profiles/r04_win16_ab.txt): 3-6 % faster on the 128-filter tiles,
// 1-3 % on the 64-filter ReID tiles - after the per-step address work was cut to one register per block (a first build with two
// and 22 spilled VGPRs was 9 % SLOWER: the step is as much issue bound as power bound).
// Fragments: lane l holds 8 consecutive channels (chunk l >> 4 of the hi or lo half) of row l & 15 of its block - the 16 lanes
// of one chunk read 16 consecutive rows, which the swizzle spreads over 16 distinct 16-byte bank slots at any base.
// C/D layout of the instruction: column (filter) = lane & 15, row (pixel) = 4 * (lane >> 4) + e, e = 0..3.
#include "conv_common.h"

namespace yds {

#ifdef YDS_CLOCK_PROBE
__device__ unsigned long long yds_clk_win16[2];        // sampled (shader cycles, 100 MHz ticks) inside the kernel, see conv_win.hip
#endif

namespace {

constexpr int NSB = 3;                         // filter-stage ring depth
constexpr int ROW = 128;
constexpr int APW = 7;                         // window DMA instructions per wave per channel group (8 rows each): taps 0-6 of a group carry one
constexpr int max_wrows(int nw) { return APW * nw * 8; }   // 448 window rows for eight waves, 224 for four

// BM x BN tile on WM x WN waves.  256 x 128 / 256 x 64 on eight waves: one workgroup per CU (its LDS holds two windows + the
// ring: 154 KB at W = 76).  128 x 64 on four waves (round 4): 76 KB at W = 32, so TWO workgroups share a CU and one computes
// while the other runs its prologue / epilogue - the ReID network's 64 -> 64 layers have only 18 K steps per tile, a
// one-workgroup-per-CU tile spent ~40 % of its time outside the K loop there.
template <int BM, int BN, int WM, int WN, int ACT, int RES>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 8 ? 1 : 2) void conv3x3_f16x3_win16(ConvKernelArgs p, int wrows, int nbuf) {
    fp16_saturate_on();
    constexpr int NW = WM * WN, NT = NW * 64;
    static_assert(NW == 8 || NW == 4, "eight waves (one workgroup per CU) or four (two)");
    constexpr int RW = BM / WM, CW = BN / WN;                   // rows / filters per wave
    constexpr int TM = RW / 16, TN = CW / 16;                   // 16-row / 16-filter blocks per wave
    constexpr int HM = TM / 2, HN = TN / 2;                     // blocks per half
    static_assert(TM % 2 == 0 && TN % 2 == 0, "the wave tile splits into quadrants");
    constexpr int B_STAGE = BN * ROW;
    constexpr int B_INST = BN / (8 * NW);                       // filter DMA instructions per wave per stage (8 rows each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int WB = wrows * ROW;                                  // bytes per window buffer
    char *bring = smem + nbuf * WB;                              // [NSB][BN][128]
    const int zoff = nbuf * WB + NSB * B_STAGE;                  // zero row

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    const int W = p.W, G = p.Cin / 32;
    const int drow = lane >> 3, dpos = lane & 7;
    const int npieces = wrows / 8;
    unsigned w_off16[B_INST];
#pragma unroll
    for (int b = 0; b < B_INST; ++b) {
        const int row = (b * NW + wave) * 8 + drow;
        w_off16[b] = (unsigned)min(n0 + row, p.Cout - 1) * (unsigned)(p.Kpad / 4) + (unsigned)(dpos ^ ((row >> 1) & 7));
    }
    const char *x_bytes = reinterpret_cast<const char *>(p.x), *w_bytes = reinterpret_cast<const char *>(p.w);
    auto a_piece = [&](int g, int k) {                           // window of channel group g -> buffer g & 1
        const int pc = min(k * NW + wave, npieces - 1);        // surplus instructions repeat the last piece (same data, same place)
        const int j = pc * 8 + drow;
        const int f = min(max(m0 - W - 1 + j, 0), p.M - 1);
        const unsigned off16 = (unsigned)f * (unsigned)(p.ldx / 4) + (unsigned)(dpos ^ ((j >> 1) & 7));
        const char *src = x_bytes + (size_t)g * ROW + ((size_t)off16 << 4);
        __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(smem + (g & 1) * WB + pc * 8 * ROW), 16, 0, 0);
    };
    auto b_piece = [&](int g, int tap, int stage, int b) {       // filter rows of K chunk (tap, g)
        const char *src = w_bytes + (size_t)(tap * G + g) * ROW + ((size_t)w_off16[b] << 4);
        __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(bring + stage * B_STAGE + (b * NW + wave) * 8 * ROW), 16, 0, 0);
    };

    // per-lane validity of the nine taps for the TM row blocks of this wave (block i: row r0 + i*16), 9 bits per block, three
    // blocks per register
    const int r0 = wm * RW + (lane & 15);
    unsigned okbits[(TM + 2) / 3];
    {
        const int HW = p.H * W;
#pragma unroll
        for (int q = 0; q < (TM + 2) / 3; ++q) okbits[q] = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + r0 + i * 16;
            unsigned bits = 0;
            if (m < p.M) {
                const int rem = m % HW, y = rem / W, x = rem - y * W;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    bits |= ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)W ? 1u : 0u) << t;
                }
            }
            okbits[i / 3] |= bits << (9 * (i % 3));
        }
    }

    f32x4 acc1[TM][TN], acc2[TM][TN];                           // hi x hi; hi x lo + lo x hi
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc1[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

    const int swz = (lane >> 1) & 7, kq = lane >> 4;
    const int b_frag = (wn * CW + (lane & 15)) * ROW;
    const int b_hi = b_frag + (kq ^ swz) * 16;                  // the lo half sits 4 chunks further in the row: position ^ 4, address ^ 64

    h8 ah[TM], al[TM], bh[TN], bl[TN];
    int a_hi[TM];                                               // this tap: LDS byte address of the lane's hi chunk in its window row (or in the zero row); lo = ^ 64
    // fragment f of half `half`: f < 2*HM: A block half*HM + f/2 (hi, lo); else B block half*HN + (f - 2*HM)/2 (hi, lo)
    auto frag_read = [&](const char *bst, int half, int f) {
        if (f < 2 * HM) {
            const int i = half * HM + f / 2;
            const h8 v = *reinterpret_cast<const h8 *>(smem + (f & 1 ? a_hi[i] ^ 64 : a_hi[i]));
            if (f & 1) al[i] = v; else ah[i] = v;
        } else {
            const int g = f - 2 * HM, j = half * HN + g / 2;
            const h8 v = *reinterpret_cast<const h8 *>(bst + j * 16 * ROW + (g & 1 ? b_hi ^ 64 : b_hi));
            if (g & 1) bl[j] = v; else bh[j] = v;
        }
    };
    constexpr int NRH = 2 * (HM + HN);                           // fragment reads per half
    constexpr int NMQ = 3 * HM * HN;                             // MFMAs per quadrant
    auto mfma = [&](int qa, int qb, int m) {                     // MFMA m of quadrant (A half qa) x (B half qb)
        const int ij = m % (HM * HN), term = m / (HM * HN), i = qa * HM + ij / HN, j = qb * HN + ij % HN;   // term-major: no back-to-back accumulation into one tile
        if (term == 0) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc1[i][j], 0, 0, 0);
        else if (term == 1) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
        else acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
    };
    auto tap_addr = [&](int g, int tap) {                        // A-fragment addresses for (g, tap)
        const int shift = (tap / 3) * W + (tap % 3);            // (dy+1)*W + (dx+1)
        const int wbase = (g & 1) * WB;
        int rj = r0 + shift;
        asm volatile("" : "+v"(rj));                             // keep the per-tap addresses out of loop-invariant hoisting (registers)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bool ok = (okbits[i / 3] >> (9 * (i % 3) + tap)) & 1u;
            const int j = rj + i * 16;
            a_hi[i] = ok ? wbase + j * ROW + ((kq ^ ((j >> 1) & 7)) << 4) : zoff + (kq << 4);
        }
    };
    auto step = [&](int g, auto tap_c, auto last_c) {
        constexpr int TAP = decltype(tap_c)::value;
        constexpr bool LAST = decltype(last_c)::value;          // last channel group: no window prefetch, filter refills stop
        constexpr bool REFILL = !(LAST && TAP + 2 > 8);         // a step t+2 exists
        constexpr bool NEXT = !(LAST && TAP == 8);              // a step t+1 exists
        constexpr int TAP1 = (TAP + 1) % 9, TAP2 = (TAP + 2) % 9;
        const int g1 = TAP + 1 >= 9 ? g + 1 : g, g2 = TAP + 2 >= 9 ? g + 1 : g;
        const char *bst = bring + (TAP % NSB) * B_STAGE, *bst1 = bring + ((TAP + 1) % NSB) * B_STAGE;
        auto dma = [&](int o) {                                  // DMA operation o of this step (window piece, then the filter pieces)
            if (o == 0) { if (!LAST && TAP < APW) a_piece(g + 1, TAP); }
            else if (o - 1 < B_INST) { if (REFILL) b_piece(g2, TAP2, (TAP + 2) % NSB, o - 1); }
        };
        constexpr int NA = 2 * HM, NB = 2 * HN;                 // A / B fragment reads of one half (A fragments are f < NA)
        // Q00: this step's B fragments of half 1; Q01: its A fragments of half 1
#pragma unroll
        for (int m = 0; m < NMQ; ++m) {
            mfma(0, 0, m);
            __builtin_amdgcn_sched_barrier(0);
            if (m < NB) frag_read(bst, 1, NA + m);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < NMQ; ++m) {
            mfma(0, 1, m);
            __builtin_amdgcn_sched_barrier(0);
            if (m < NA) frag_read(bst, 1, m);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (NEXT) tap_addr(g1, TAP1);
        __builtin_amdgcn_sched_barrier(0);
        // Q10: the DMA pieces, then the next step's A fragments of half 0 (free since Q01); Q11: its B fragments of half 0
#pragma unroll
        for (int m = 0; m < NMQ; ++m) {
            mfma(1, 0, m);
            __builtin_amdgcn_sched_barrier(0);
            if (m <= B_INST) dma(m);
            else if (NEXT && m - 1 - B_INST < NA) frag_read(bst1, 0, m - 1 - B_INST);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(1 + B_INST + NA <= NMQ, "Q10 holds the DMA pieces and the A fragments");
#pragma unroll
        for (int m = 0; m < NMQ; ++m) {
            mfma(1, 1, m);
            __builtin_amdgcn_sched_barrier(0);
            if (NEXT && m < NB) frag_read(bst1, 0, NA + m);
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

    // prologue: window of group 0, filter stages of steps 0 and 1, H0 fragments of step 0
    for (int k = 0; k < (npieces + NW - 1) / NW; ++k) a_piece(0, k);
#pragma unroll
    for (int b = 0; b < B_INST; ++b) b_piece(0, 0, 0, b);
#pragma unroll
    for (int b = 0; b < B_INST; ++b) b_piece(0, 1, 1, b);
    wait_vmcnt<B_INST>();                                       // stage 1 may still be in flight: the first mid-step wait covers it
    __syncthreads();                                            // window 0, stage 0 and the zero row are in LDS
    tap_addr(0, 0);
#pragma unroll
    for (int f = 0; f < NRH; ++f) frag_read(bring, 0, f);
    __builtin_amdgcn_sched_barrier(0);

    for (int g = 0; g + 1 < G; ++g) group(g, std::false_type{});
    group(G - 1, std::true_type{});

    __syncthreads();                                            // every wave is done with the window and the ring
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1[i][j][e] = (acc1[i][j][e] + acc2[i][j][e] * (1.f / LO_SCALE)) * (1.f / A_SCALE);
    conv_epilogue16<BM, BN, WM, WN, ACT, RES, TM, TN, NT, true>(p, acc1, reinterpret_cast<float *>(smem), m0, n0, tid);   // whole-tile staging
#ifdef YDS_CLOCK_PROBE
    if (clk_sample) {
        atomicAdd(&yds_clk_win16[0], __builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(&yds_clk_win16[1], wall_clock64() - clk_w0);
    }
#endif
}

int window_rows16(int BM, int W) { return (BM + 2 * W + 2 + 7) / 8 * 8; }
size_t win16_smem(int BM, int BN, int W, int Cin) {
    const int wrows = window_rows16(BM, W), nbuf = Cin == 32 ? 1 : 2;
    // (the epilogue stages the whole BM x BN tile in the same LDS: narrow images need more than their windows + ring)
    return std::max((size_t)nbuf * wrows * ROW + (size_t)NSB * BN * ROW + ROW, conv_stage_bytes(BM, BN));
}

template <int BM, int BN, int WM, int WN, int ACT, int RES> void launch_inst_win16(ConvKernelArgs k, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    const int wrows = window_rows16(BM, k.W), nbuf = k.Cin == 32 ? 1 : 2;
    const size_t smem = win16_smem(BM, BN, k.W, k.Cin);
    static size_t attr_set = 0;
    auto kern = conv3x3_f16x3_win16<BM, BN, WM, WN, ACT, RES>;
    if (smem > attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = smem;
    }
    dim3 grid(plan_tile_map(k, BM, BN));
    hipLaunchKernelGGL(kern, grid, dim3(NT), smem, s, k, wrows, nbuf);
    YDS_HIP(hipGetLastError());
}

}  // namespace

void conv_win16_clock(unsigned long long *cycles_ticks, bool reset) {
#ifdef YDS_CLOCK_PROBE
    YDS_HIP(hipMemcpyFromSymbol(cycles_ticks, HIP_SYMBOL(yds_clk_win16), 2 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[2] = {};
        YDS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(yds_clk_win16), z, sizeof z));
    }
#else
    cycles_ticks[0] = cycles_ticks[1] = 0;     // product build: no sampling inside the kernel (clock_probe.hip measures beside it)
    (void)reset;
#endif
}

// the 128 x 64 tile with two workgroups per CU: 64-filter layers whose two windows + ring fit 80 KB and whose window is fetched by
// at most APW instructions per wave (W <= 43 with several channel groups: 216 window rows)
bool conv_win16_small_applicable(const ConvKernelArgs &k) {
    if (!conv_win_applicable(k) || k.Cout > 64) return false;
    const int wrows = window_rows16(128, k.W), nbuf = k.Cin == 32 ? 1 : 2;
    if (nbuf == 2 && wrows > max_wrows(4)) return false;
    return win16_smem(128, 64, k.W, k.Cin) <= 80 * 1024;
}

// default arithmetic (f16x3) of the window-resident kernel; shapes 0 / 1: applicability is conv_win_applicable's (same LDS plan),
// shape 2: conv_win16_small_applicable
void launch_conv_win16(ConvKernelArgs k, int shape, hipStream_t s) {
    if (shape == 0) {
#define YDS_CALL(A, R) launch_inst_win16<256, 128, 4, 2, A, R>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
    } else if (shape == 1) {
#define YDS_CALL(A, R) launch_inst_win16<256, 64, 8, 1, A, R>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
    } else {
        if (!conv_win16_small_applicable(k)) fail("conv: the 128x64 window-resident tile needs a 3x3 stride-1 layer with a pre-split input, at most 64 filters and W <= 43");
#define YDS_CALL(A, R) launch_inst_win16<128, 64, 4, 1, A, R>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
    }
}

}  // namespace yds
