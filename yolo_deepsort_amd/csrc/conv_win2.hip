// Window-resident 3x3 / stride-1 convolution, TWO workgroups per CU (f16x3 arithmetic, see conv_f16x3.hip / conv_win.hip).
//
// conv_win.hip keeps one 512-thread workgroup per CU (135-155 KB of LDS): its prologue (window + two filter stages) and its
// epilogue (128 KB written + 128 KB residual read per tile, every CU at the same time) are exposed - ~20 % of a launch
// (profiles/r01_ablation_dma.txt).  This kernel halves everything so that two independent workgroups fit on a CU and
// one computes while the other loads / stores:
//   tile 128 pixels x 128 filters, 256 threads = 4 waves (2x2, 64x64 accumulator tiles as before)
//   K step = ONE tap of a 16-channel HALF group: 12 MFMAs per wave, 8 fragment reads, 2 filter DMA pieces, <= 1 window piece
//   window rows hold one half group: 64 B per pixel [16 hi | 16 lo], two buffers (double buffered across half groups)
//   filter ring: 4 stages of 128 filters x 64 B; the stage of step t+4 is fetched during step t (three steps of latency)
//   LDS 2 x wrows x 64 + 32 KB + 64 B <= 80 KB for W <= 127 (W = 76: 68 KB), the epilogue stages 128 x 132 floats = 66 KB
// Swizzle for 64-byte rows: row r keeps its 16-byte chunk c at position c ^ ((r >> 2) & 3); a ds_read_b128 lane group
// covers 16 rows with distinct residues mod 16, i.e. all 16 (r & 3, (r >> 2) & 3) pairs once: 16 distinct bank slots at
// ANY base row, which is what the shifted tap windows need.
#include "conv_common.h"

namespace yds {

#ifdef YDS_CLOCK_PROBE
__device__ unsigned long long yds_clk_win2[2];  // sustained shader clock inside the kernel: (cycles, 100 MHz ticks) of one workgroup in 32 (see conv_win.hip)
#endif

namespace {

constexpr int BM2 = 128, BN2 = 128, NW2 = 4, NT2 = NW2 * 64;
constexpr int ROW2 = 64;                        // bytes per LDS row (one pixel or one filter, one half group: 16 hi + 16 lo fp16)
constexpr int NSB2 = 4;                         // filter-stage ring depth
constexpr int B_STAGE2 = BN2 * ROW2;            // 8 KB
constexpr int B_INST2 = BN2 / (16 * NW2);       // filter DMA instructions per wave per stage (16 rows each): 2
constexpr int MAX_WROWS2 = 384;

template <int ACT, int RES, int TERMS>
__global__ __launch_bounds__(NT2, 2) void conv3x3_f16x3_win2(ConvKernelArgs p, int wrows, int apw) {
    fp16_saturate_on();
    constexpr int WM = 2, WN = 2, TM = 2, TN = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int WB = wrows * ROW2;                                 // bytes per window buffer
    char *bring = smem + 2 * WB;                                 // [NSB2][BN2][64]
    const int zoff = 2 * WB + NSB2 * B_STAGE2;                   // zero row

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): LDS-DMA destinations need no v_readfirstlane per piece
    const int wm = wave / WN, wn = wave % WN;
    int m0, n0;
    {
        int tm, tn;
        if (!tile_of_block(p, tm, tn)) return;
        m0 = tm * BM2;
        n0 = tn * BN2;
    }
    if (tid < 16) reinterpret_cast<float *>(smem + zoff)[tid] = 0.f;
#ifdef YDS_CLOCK_PROBE
    const bool clk_sample = tid == 0 && (blockIdx.x & 31) == 0;
    unsigned long long clk_c0 = 0, clk_w0 = 0;
    if (clk_sample) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_w0 = wall_clock64(); }
#endif
    const int W = p.W, G = p.Cin / 32, HG = 2 * G;               // half groups
    const int drow = lane >> 2, dpos = lane & 3;
    const int npieces = wrows / 16;
    // offsets in 16-byte units (32 bits reach 64 GB): logical chunk c (0, 1: hi channels 0-7 / 8-15 of the half group; 2, 3: lo)
    // sits at unit (c < 2 ? 0 : 4) + (c & 1) of half 0 of a 128-byte group, half 1 two units further
    auto chunk16 = [](int c) { return (c < 2 ? 0 : 4) + (c & 1); };
    unsigned w_off16[B_INST2];                                   // filter row + chunk (half 0 of K chunk 0)
#pragma unroll
    for (int b = 0; b < B_INST2; ++b) {
        const int row = (b * NW2 + wave) * 16 + drow;
        w_off16[b] = (unsigned)min(n0 + row, p.Cout - 1) * (unsigned)(p.Kpad / 4) + (unsigned)chunk16(dpos ^ ((row >> 2) & 3));
    }
    const char *x_bytes = reinterpret_cast<const char *>(p.x), *w_bytes = reinterpret_cast<const char *>(p.w);
    auto a_piece = [&](int hg, int k) {                          // window of half group hg -> buffer hg & 1
        const int pc = min(k * NW2 + wave, npieces - 1);       // surplus instructions repeat the last piece (same data, same place)
        const int j = pc * 16 + drow;
        const int f = min(max(m0 - W - 1 + j, 0), p.M - 1);
        const unsigned off16 = (unsigned)f * (unsigned)(p.ldx / 4) + (unsigned)chunk16(dpos ^ ((j >> 2) & 3));
        const char *src = x_bytes + (size_t)((hg >> 1) * 128 + (hg & 1) * 32) + ((size_t)off16 << 4);
        __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(smem + (hg & 1) * WB + pc * 16 * ROW2), 16, 0, 0);
    };
    auto b_piece = [&](int hg, int tap, int slot, int b) {       // filter rows of K chunk (tap, hg >> 1), half hg & 1
        const char *src = w_bytes + (size_t)((tap * G + (hg >> 1)) * 128 + (hg & 1) * 32) + ((size_t)w_off16[b] << 4);
        __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(bring + slot * B_STAGE2 + (b * NW2 + wave) * 16 * ROW2), 16, 0, 0);
    };

    // per-lane validity of the nine taps for the two A fragments of this wave (rows wm*64 + i*32 + (lane & 31))
    int r_row[TM];
    unsigned ok9[TM];
    {
        const int HW = p.H * W;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * (BM2 / WM) + i * 32 + (lane & 31), m = m0 + r;
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

    const int kb = lane >> 5;
    const int bsw = ((lane & 31) >> 2) & 3;                      // swizzle of this lane's filter rows (row = 32-aligned base + lane & 31)
    const int b_hi = (wn * (BN2 / WN) + (lane & 31)) * ROW2 + ((kb ^ bsw) << 4);
    const int b_lo = (wn * (BN2 / WN) + (lane & 31)) * ROW2 + (((2 + kb) ^ bsw) << 4);

    constexpr int NF = 2 * (TM + TN), NM = 3 * TM * TN;         // 8 fragments, 12 MFMAs per step
    h8 fr[2][NF];                                               // [step parity][A0h, A0l, A1h, A1l, B0h, B0l, B1h, B1l]
    int a_addr[TM], a_sw[TM];
    auto frag_read = [&](const char *bst, int buf, int f) {
        const int which = f / 2, lo = f & 1;
        if (TERMS == 1 && lo) return;
        if (which < TM) {
            const int c = (lo ? 2 : 0) + kb;
            fr[buf][f] = *reinterpret_cast<const h8 *>(smem + a_addr[which] + ((c ^ a_sw[which]) << 4));
        } else {
            fr[buf][f] = *reinterpret_cast<const h8 *>(bst + (lo ? b_lo : b_hi) + (which - TM) * 32 * ROW2);
        }
    };
    // MFMA order: term-major (all hi x hi products, then hi x lo, then lo x hi).  The two cross terms of a tile accumulate
    // into the same registers; issued back to back (tile-major order) the second one waits for the first one's result -
    // 64 cycles of latency against 32 of issue - which a wave that has its SIMD to itself cannot hide.
    auto mfma = [&](int buf, int m) {
        const int ij = m / 3, term = m % 3, i = ij / TN, j = ij % TN;
        if (TERMS == 1 && term != 0) return;
        const h8 ah = fr[buf][2 * i], al = fr[buf][2 * i + 1], bh = fr[buf][2 * (TM + j)], bl = fr[buf][2 * (TM + j) + 1];
        if (term == 0) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[i][j], 0, 0, 0);
        else if (term == 1) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[i][j], 0, 0, 0);
        else acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[i][j], 0, 0, 0);
    };
    // fragment order: the operands of accumulator tile (0,0) first (tile-major MFMAs) / all hi halves first (term-major)
    auto frag_order = [&](int k) {
        if (k < 2) return k;                    // A0h, A0l
        if (k < 4) return 2 * TM + (k - 2);     // B0h, B0l
        if (k < 6) return 2 * TM + 2 + (k - 4); // B1h, B1l
        return 2 + (k - 6);                     // A1h, A1l
    };
    auto tap_addr = [&](int hg, int tap) {                       // A-fragment row addresses for (hg, tap)
        const int shift = (tap / 3) * W + (tap % 3);            // (dy+1)*W + (dx+1)
        const int wbase = (hg & 1) * WB;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            asm volatile("" : "+v"(r_row[i]), "+v"(ok9[i]));      // keep the per-tap addresses out of loop-invariant hoisting (registers)
            const bool ok = (ok9[i] >> tap) & 1u;
            const int j = r_row[i] + shift;
            a_addr[i] = ok ? wbase + j * ROW2 : zoff;
            a_sw[i] = ok ? (j >> 2) & 3 : 0;
        }
    };

    // Step t = (hg, TAP); fragments of step t sit in fr[PAR] (read during step t-1).
    //   top      s_waitcnt vmcnt(N) + lgkmcnt(0), s_barrier: stage t+1 (fetched during step t-3) and - before a new half group -
    //            its window have landed for every wave; every wave has finished READING stage t, so its slot can be refilled
    //   body     12 MFMAs on fr[PAR]; slots: window piece of half group hg+1, filter pieces of step t+4 (into the slot of
    //            stage t), fragments of step t+1 -> fr[PAR ^ 1]
    // N = DMA instructions this wave issued during steps t-2 and t-1 (they may stay in flight).  Window pieces are counted
    // as absent (a conservative N: with them the wait covers slightly more than it must); in the last half group the
    // refills stop four steps before the end and N shrinks with them.
    auto step = [&](int hg, auto tap_c, auto last_c, auto par_c) {
        constexpr int TAP = decltype(tap_c)::value, PAR = decltype(par_c)::value;
        constexpr bool LAST = decltype(last_c)::value;          // last half group
        constexpr bool REFILL = !(LAST && TAP >= 5);            // a step t+4 exists
        constexpr bool NEXT = !(LAST && TAP == 8);              // a step t+1 exists
        constexpr int TAP1 = (TAP + 1) % 9, TAP4 = (TAP + 4) % 9;
        constexpr int N_OUT = !LAST || TAP <= 5 ? 2 * B_INST2 : (TAP == 6 ? B_INST2 : 0);
        const int hg1 = TAP + 1 >= 9 ? hg + 1 : hg, hg4 = TAP + 4 >= 9 ? hg + 1 : hg;
        const int slot0 = (hg + TAP) & 3;                        // ring slot of stage t: t = 9*hg + TAP, 9 = 1 mod 4
        if (NEXT) {
            wait_vmcnt<N_OUT>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            tap_addr(hg1, TAP1);
        }
        const char *bst1 = bring + ((slot0 + 1) & 3) * B_STAGE2;
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NDMA = 1 + B_INST2;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma(PAR, m);
            __builtin_amdgcn_sched_barrier(0);
            const int o = m < NF ? m + NDMA : (m - NF < NDMA ? m - NF : NDMA + NF);   // operation: 0 window, 1..2 filter, 3..10 fragments; the next step's fragments go first
            if (o == 0) { if (!LAST && TAP < 6 && TAP < apw) a_piece(hg + 1, TAP); }
            else if (o - 1 < B_INST2) { if (REFILL) b_piece(hg4, TAP4, slot0, o - 1); }
            else if (o - NDMA < NF) { if (NEXT) frag_read(bst1, PAR ^ 1, frag_order(o - NDMA)); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // nine taps of one half group; the step parity alternates and 9 is odd, so half groups alternate between two bodies
    auto half_group = [&](int hg, auto last_c, auto par_c) {
        constexpr int P0 = decltype(par_c)::value;
        step(hg, std::integral_constant<int, 0>{}, last_c, std::integral_constant<int, P0>{});
        step(hg, std::integral_constant<int, 1>{}, last_c, std::integral_constant<int, P0 ^ 1>{});
        step(hg, std::integral_constant<int, 2>{}, last_c, std::integral_constant<int, P0>{});
        step(hg, std::integral_constant<int, 3>{}, last_c, std::integral_constant<int, P0 ^ 1>{});
        step(hg, std::integral_constant<int, 4>{}, last_c, std::integral_constant<int, P0>{});
        step(hg, std::integral_constant<int, 5>{}, last_c, std::integral_constant<int, P0 ^ 1>{});
        step(hg, std::integral_constant<int, 6>{}, last_c, std::integral_constant<int, P0>{});
        step(hg, std::integral_constant<int, 7>{}, last_c, std::integral_constant<int, P0 ^ 1>{});
        step(hg, std::integral_constant<int, 8>{}, last_c, std::integral_constant<int, P0>{});
    };

    // prologue: window of half group 0, filter stages of steps 0..3, fragments of step 0
    for (int k = 0; k < (npieces + NW2 - 1) / NW2; ++k) a_piece(0, k);
#pragma unroll
    for (int t = 0; t < NSB2; ++t)
#pragma unroll
        for (int b = 0; b < B_INST2; ++b) b_piece(0, t, t, b);
    wait_vmcnt<(NSB2 - 1) * B_INST2>();                         // window 0 and stage 0 have landed (stages 1..3 may be in flight)
    __syncthreads();
    tap_addr(0, 0);
#pragma unroll
    for (int f = 0; f < NF; ++f) frag_read(bring, 0, f);
    __builtin_amdgcn_sched_barrier(0);

    // half groups 0 .. HG-1: parity of the first step of half group hg is hg & 1 (9 steps each); HG is even
    for (int hg = 0; hg + 2 < HG; hg += 2) {
        half_group(hg, std::false_type{}, std::integral_constant<int, 0>{});
        half_group(hg + 1, std::false_type{}, std::integral_constant<int, 1>{});
    }
    half_group(HG - 2, std::false_type{}, std::integral_constant<int, 0>{});
    half_group(HG - 1, std::true_type{}, std::integral_constant<int, 1>{});

    __syncthreads();                                            // every wave is done with the windows and the ring
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc1[i][j][e] = TERMS == 1 ? acc1[i][j][e] * (1.f / A_SCALE) : (acc1[i][j][e] + acc2[i][j][e] * (1.f / LO_SCALE)) * (1.f / A_SCALE);
    conv_epilogue<BM2, BN2, WM, WN, ACT, RES, TM, TN, NT2, true>(p, acc1, reinterpret_cast<float *>(smem), m0, n0, tid);   // whole-tile staging
#ifdef YDS_CLOCK_PROBE
    if (clk_sample) {
        atomicAdd(&yds_clk_win2[0], __builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(&yds_clk_win2[1], wall_clock64() - clk_w0);
    }
#endif
}

int window_rows2(int W) { return (BM2 + 2 * W + 2 + 15) / 16 * 16; }

template <int ACT, int RES, int TERMS = 3> void launch_inst_win2(ConvKernelArgs k, hipStream_t s) {
    const int wrows = window_rows2(k.W);
    const int apw = (wrows / 16 + NW2 - 1) / NW2;
    const size_t smem = std::max((size_t)2 * wrows * ROW2 + (size_t)NSB2 * B_STAGE2 + ROW2, conv_stage_bytes(BM2, BN2));
    static size_t attr_set = 0;
    auto kern = conv3x3_f16x3_win2<ACT, RES, TERMS>;
    if (smem > attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = smem;
    }
    dim3 grid(plan_tile_map(k, BM2, BN2));
    hipLaunchKernelGGL(kern, grid, dim3(NT2), smem, s, k, wrows, apw);
    YDS_HIP(hipGetLastError());
}

}  // namespace

bool conv_win2_applicable(const ConvKernelArgs &k) {
    if (!(k.ksize == 3 && k.stride == 1 && k.pad == 1 && k.fmt_x == FMT_H16 && k.Cin % 32 == 0 && k.H == k.Ho && k.W == k.Wo)) return false;
    const int wrows = window_rows2(k.W);
    if (wrows > MAX_WROWS2) return false;                        // two resident workgroups: <= 80 KB each; <= 6 window pieces per wave and half group
    return (size_t)k.M * ((size_t)k.ldx / 4) < (1ull << 32) && (size_t)k.Cout * (k.Kpad / 4) < (1ull << 32);
}

void conv_win2_clock(unsigned long long *cycles_ticks, bool reset) {
#ifdef YDS_CLOCK_PROBE
    YDS_HIP(hipMemcpyFromSymbol(cycles_ticks, HIP_SYMBOL(yds_clk_win2), 2 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[2] = {};
        YDS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(yds_clk_win2), z, sizeof z));
    }
#else
    cycles_ticks[0] = cycles_ticks[1] = 0;     // product build: no sampling inside the kernel (clock_probe.hip measures beside it)
    (void)reset;
#endif
}

void launch_conv_win2(ConvKernelArgs k, hipStream_t s) {
    if (!conv_win2_applicable(k)) fail("conv: the two-workgroup window-resident kernel needs a 3x3 stride-1 layer with a pre-split input and W <= 127");
    if (k.terms == 1) {
#define YDS_CALL(A, R) launch_inst_win2<A, R, 1>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
    } else {
#define YDS_CALL(A, R) launch_inst_win2<A, R>(k, s)
        YDS_DISPATCH_ACT_RES(k, YDS_CALL)
#undef YDS_CALL
    }
}

}  // namespace yds
