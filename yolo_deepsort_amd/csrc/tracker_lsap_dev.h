// Device-side solvers of the linear assignment problem (included by tracker_lsap.hip: the stand-alone kernels, and by tracker.hip:
// the fused per-frame association kernel calls them in place).
#pragma once
#include "tracker_dev.h"

namespace yds {

// ------------------------------------------------------------------------------------------ LSAP
// scipy.optimize.linear_sum_assignment (rectangular_lsap.cpp, Crouse 2016) on ONE workgroup of four wavefronts.  The
// augmenting-path search is sequential over rows; its column scan is spread over 256 lanes and the sequential tie-break of
// the scalar scan is reproduced exactly:
//   index = last unassigned column (in `remaining` order) among the minimum, else the first minimum
// encoded as a key so that ONE lexicographic (cost, key) reduction per Dijkstra step finds it.  Arithmetic is fp64 in the
// same order as scipy (minVal + c - u[i] - v[j]).  Tall matrices are solved transposed.  Position `it` of `remaining` is
// always scanned - and rewritten - by thread it % 256, the winner's column travels with the reduction, and the per-wave
// partial results are double buffered, so a Dijkstra step costs one barrier.  Solver state lives in LDS (and the cost
// matrix too when it fits); beyond ~3000 rows/columns it moves to a global scratch buffer - no size limit.
// dims_p (optional): device-side {nr, nc}.  row_out/col_out: min(nr,nc) pairs sorted by row, *n_out = that count.
constexpr int LSAP_NT = 256, LSAP_NW = LSAP_NT / 64;
constexpr int LSAP_WAVE_COLS = 64;         // problems up to this many columns go to the single-wavefront kernel (below)
constexpr size_t LSAP_STATE_BYTES = 3 * sizeof(double) + 6 * sizeof(int);      // per row / column
constexpr size_t LSAP_LDS_MAX = 150 * 1024;

// Cross-lane helpers of the LSAP kernels.  Everything is passed as scalars: with the candidate in a struct handled through
// references the compiler kept it in scratch memory (a global-memory round trip per use inside a latency-bound loop).
template <int CTRL> __device__ __forceinline__ int lsap_dpp_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ double lsap_dpp_d(double x) {
    union { double d; int i[2]; } u, w;
    u.d = x;
    w.i[0] = lsap_dpp_i<CTRL>(u.i[0]);
    w.i[1] = lsap_dpp_i<CTRL>(u.i[1]);
    return w.d;
}
__device__ __forceinline__ double lsap_readlane_d(double x, int lane) {
    union { double d; int i[2]; } u, w;
    u.d = x;
    w.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    w.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return w.d;
}
// (v, key) <- lexicographic minimum with (ov, ok)
#define LSAP_TAKE_MIN(v, key, ov, ok)                                      \
    do {                                                                   \
        const double _ov = (ov);                                           \
        const int _ok = (ok);                                              \
        const bool _t = _ov < (v) || (_ov == (v) && _ok < (key));          \
        (v) = _t ? _ov : (v);                                              \
        (key) = _t ? _ok : (key);                                          \
    } while (0)
// lexicographic (cost, key) minimum over the wavefront, result in every lane (uniform).  Two phases instead of one
// 96-bit lexicographic butterfly (round 2: ~80 dependent instructions, 1000+ cycles of every Dijkstra step):
//   1. the minimum VALUE alone: v_min_f64 over four DPP butterflies inside each row of 16 lanes (quad_perm xor 1, xor 2,
//      row_half_mirror, row_mirror), then the four row results through v_readlane;
//   2. lanes holding that value keep their key, the others 0x7fffffff; the minimum KEY with v_min_i32 on DPP operands,
//      rows combined on the scalar ALU.
// Identical result: (min value, smallest key among the lanes that attain it).  Values are never NaN; +inf marks dead lanes.
// (__shfl_xor lowers to ds_bpermute_b32 here: dependent LDS-crossbar round trips.)
__device__ __forceinline__ void lsap_wave_min(double &v, int &key) {
    double m = v;
    m = fmin(m, lsap_dpp_d<0xB1>(m));
    m = fmin(m, lsap_dpp_d<0x4E>(m));
    m = fmin(m, lsap_dpp_d<0x141>(m));
    m = fmin(m, lsap_dpp_d<0x140>(m));
    const double r = fmin(fmin(lsap_readlane_d(m, 0), lsap_readlane_d(m, 16)), fmin(lsap_readlane_d(m, 32), lsap_readlane_d(m, 48)));
    int k = v == r ? key : 0x7fffffff;
    k = min(k, lsap_dpp_i<0xB1>(k));
    k = min(k, lsap_dpp_i<0x4E>(k));
    k = min(k, lsap_dpp_i<0x141>(k));
    k = min(k, lsap_dpp_i<0x140>(k));
    const int rk = min(min(__builtin_amdgcn_readlane(k, 0), __builtin_amdgcn_readlane(k, 16)),
                       min(__builtin_amdgcn_readlane(k, 32), __builtin_amdgcn_readlane(k, 48)));
    v = r;
    key = rk;
}

// GSTATE: solver state in the global scratch buffer (huge problems) instead of LDS - a compile-time choice, so that the LDS
// version addresses its state with ds_read / ds_write (a pointer that may be either makes every access a flat_load)
template <bool GSTATE, bool COST_LDS>
__device__ __forceinline__ void lsap_wg_solve(const float *cost, int nr0, int nc0, int *row_out, int *col_out, int *n_out, char *state_global) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    extern __shared__ __attribute__((aligned(16))) char lsap_smem[];
    __shared__ double red_val[2][LSAP_NW];
    __shared__ int red_key[2][LSAP_NW], red_col[2][LSAP_NW];
    const int n = max(nr, nc);
    double *u, *v, *spc;
    int *path, *col4row, *row4col, *remaining, *SR, *SC;
    auto carve = [&](char *base) {
        u = reinterpret_cast<double *>(base); v = u + n; spc = v + n;
        path = reinterpret_cast<int *>(spc + n); col4row = path + n; row4col = col4row + n; remaining = row4col + n; SR = remaining + n; SC = SR + n;
    };
    if (GSTATE) carve(state_global); else carve(lsap_smem);
    float *cost_lds = reinterpret_cast<float *>(lsap_smem + (GSTATE ? 0 : (size_t)n * LSAP_STATE_BYTES));
    if (COST_LDS) {
        for (int i = tid; i < nr0 * nc0; i += LSAP_NT) cost_lds[i] = cost[i];
    }
    // (two typed accesses, not one pointer that may be LDS or global: that would be a flat_load in the inner loop)
    auto C = [&](int i, int j) -> double {
        const int at = transpose ? j * nc0 + i : i * nc0 + j;
        return (double)(COST_LDS ? cost_lds[at] : cost[at]);
    };
    for (int i = tid; i < nr; i += LSAP_NT) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = tid; j < nc; j += LSAP_NT) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    __syncthreads();
    int parity = 0;
    for (int cur = 0; cur < nr; ++cur) {
        for (int i = tid; i < nr; i += LSAP_NT) SR[i] = 0;
        for (int j = tid; j < nc; j += LSAP_NT) { SC[j] = 0; spc[j] = INFINITY; remaining[j] = nc - j - 1; }   // position it <-> thread it % 256
        __syncthreads();
        double minVal = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            if (tid == 0) SR[i] = 1;
            const double ui = u[i];
            // candidate = lexicographic minimum of (shortest path cost, key): among equal costs the LAST unassigned column in
            // `remaining` order, otherwise the FIRST column: unassigned -> 0x3fffffff - it, assigned -> 0x40000000 + it
            double cv = INFINITY;
            int ckey = 0x7fffffff, cj = -1;
            for (int it = tid; it < num_remaining; it += LSAP_NT) {
                const int j = remaining[it];
                const double r = minVal + C(i, j) - ui - v[j];
                double sv = spc[j];
                if (r < sv) { path[j] = i; spc[j] = r; sv = r; }
                const int key = row4col[j] == -1 ? 0x3fffffff - it : 0x40000000 + it;
                const bool t = sv < cv || (sv == cv && key < ckey);
                cv = t ? sv : cv; ckey = t ? key : ckey; cj = t ? j : cj;
            }
            const int my_key = ckey;
            lsap_wave_min(cv, ckey);
            if (my_key == ckey && ckey != 0x7fffffff) red_col[parity][wave] = cj;      // keys are unique: exactly one lane of the wave
            if (lane == 0) { red_val[parity][wave] = cv; red_key[parity][wave] = ckey; }
            __syncthreads();
            int win = 0;
            cv = red_val[parity][0]; ckey = red_key[parity][0];
#pragma unroll
            for (int w = 1; w < LSAP_NW; ++w) {
                const double ov = red_val[parity][w];
                const int ok = red_key[parity][w];
                const bool t = ov < cv || (ov == cv && ok < ckey);
                cv = t ? ov : cv; ckey = t ? ok : ckey; win = t ? w : win;
            }
            const int j = red_col[parity][win];
            parity ^= 1;
            minVal = cv;
            const int index = ckey < 0x40000000 ? 0x3fffffff - ckey : ckey - 0x40000000;
            const int owner = row4col[j];
            if (owner == -1) sink = j; else i = owner;
            // swap-with-last removal, done by the thread that owns position `index` (the only future reader of it)
            if (tid == (index & (LSAP_NT - 1))) {
                SC[j] = 1;
                remaining[index] = remaining[num_remaining - 1];
            }
            --num_remaining;
        }
        __syncthreads();
        // dual update
        for (int r = tid; r < nr; r += LSAP_NT) {
            if (r == cur) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc[col4row[r]];
        }
        for (int j = tid; j < nc; j += LSAP_NT)
            if (SC[j]) v[j] -= minVal - spc[j];
        __syncthreads();
        if (tid == 0) {
            int j = sink;
            while (true) {
                int r = path[j];
                row4col[j] = r;
                int t = col4row[r]; col4row[r] = j; j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    if (transpose) {
        if (tid == 0) {
            int k = 0;
            for (int r = 0; r < nc; ++r) {          // nc == original row count
                int who = row4col[r];
                if (who >= 0) { row_out[k] = r; col_out[k] = who; ++k; }
            }
        }
    } else {
        for (int r = tid; r < nr; r += LSAP_NT) { row_out[r] = r; col_out[r] = col4row[r]; }
    }
    if (tid == 0 && n_out) *n_out = nr;
}


// ---- register-resident workgroup form for 64 < columns <= 256 (the crowd configuration: 200 tracks x 150 detections).
// Same algorithm, arithmetic and tie-break key as lsap_wg_solve, but position `it` of `remaining` IS thread `it`: the
// column it holds, its shortest-path cost, column dual, owner row, that row's dual and the path predecessor stay in
// registers, so the scan of a Dijkstra step is ONE LDS read (the cost entry) instead of five dependent ones.  Per step:
// scan -> DPP wave minimum -> one 16-byte candidate + the owner's row dual per wave through LDS -> one barrier -> every
// thread picks the winner among four.  The swap-with-last removal hands the state of the last position to the winner's
// position through a double-buffered LDS mailbox written BEFORE the barrier (who is last does not depend on the winner).
// Selected columns leave their registers, so their final path / shortest-path cost (= minVal at selection) go to LDS at
// that moment for the dual update and the augmentation.  ~2.7x fewer cycles per step than the LDS-state form.
struct __attribute__((aligned(16))) LsapCand { double v; int key; unsigned colown; };      // column | (owner row + 1) << 16
struct __attribute__((aligned(16))) LsapMail { double spc, vj, uo; int j, own, pth, pad; };
constexpr int LSAP_REG_COLS = 256;

template <bool COST_LDS>
__device__ __forceinline__ void lsap_reg_solve(const float *cost, int nr0, int nc0, int *row_out, int *col_out, int *n_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;      // nr <= nc <= 256
    extern __shared__ __attribute__((aligned(16))) char lsap_smem[];
    __shared__ LsapCand cand[2][LSAP_NW];
    __shared__ double cand_u[2][LSAP_NW];
    __shared__ LsapMail mail[2];
    constexpr int N = LSAP_REG_COLS;
    double *u = reinterpret_cast<double *>(lsap_smem), *v = u + N, *spc_sel = v + N;
    int *path = reinterpret_cast<int *>(spc_sel + N), *col4row = path + N, *row4col = col4row + N, *SR = row4col + N, *SC = SR + N;
    float *cost_lds = reinterpret_cast<float *>(SC + N);
    if (COST_LDS)
        for (int i = tid; i < nr0 * nc0; i += LSAP_NT) cost_lds[i] = cost[i];
    auto C = [&](int i, int j) -> double {
        const int at = transpose ? j * nc0 + i : i * nc0 + j;
        return (double)(COST_LDS ? cost_lds[at] : cost[at]);
    };
    // SR / SC hold the number (cur + 1) of the row iteration that set them: nothing to clear between iterations
    if (tid < nr) { u[tid] = 0.0; col4row[tid] = -1; SR[tid] = 0; }
    if (tid < nc) { v[tid] = 0.0; row4col[tid] = -1; path[tid] = -1; SC[tid] = 0; }
    __syncthreads();
    int parity = 0;
    const int it = tid;
    for (int cur = 0; cur < nr; ++cur) {
        const int stamp = cur + 1;
        // position it holds column nc - it - 1 (scipy fills `remaining` in reverse order); everything read here was final
        // before the barrier that closed the previous iteration, and the scan below only reads the cost matrix
        int j = nc - it - 1, own = -1, pth = -1;
        double spc = INFINITY, vj = 0.0, uo = 0.0;
        if (it < nc) {
            vj = v[j];
            own = row4col[j];
            uo = own >= 0 ? u[own] : 0.0;
        }
        const int c4r = tid < nr ? col4row[tid] : -1;                 // this row's column BEFORE the augmentation (dual update)
        double minVal = 0.0, ui = u[cur];
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            if (tid == 0) SR[i] = stamp;
            double cv = INFINITY;
            int ckey = 0x7fffffff;
            if (it < num_remaining) {
                const double r = minVal + C(i, j) - ui - vj;
                if (r < spc) { pth = i; spc = r; }
                cv = spc;
                ckey = own == -1 ? 0x3fffffff - it : 0x40000000 + it;
            }
            const int my_key = ckey;
            lsap_wave_min(cv, ckey);
            if (ckey == 0x7fffffff) {                                  // no live position in this wave
                if (lane == 0) { cand[parity][wave].v = INFINITY; cand[parity][wave].key = 0x7fffffff; }
            } else if (my_key == ckey) {                               // keys are unique: exactly one lane of the wave
                LsapCand c;
                c.v = cv; c.key = ckey; c.colown = (unsigned)j | ((unsigned)(own + 1) << 16);
                cand[parity][wave] = c;
                cand_u[parity][wave] = uo;
            }
            if (it == num_remaining - 1) {                             // the state the winner's position inherits
                LsapMail m;
                m.spc = spc; m.vj = vj; m.uo = uo; m.j = j; m.own = own; m.pth = pth; m.pad = 0;
                mail[parity] = m;
            }
            __syncthreads();
            int win = 0;
            LsapCand best = cand[parity][0];
#pragma unroll
            for (int w = 1; w < LSAP_NW; ++w) {
                const LsapCand o = cand[parity][w];
                const bool t = o.v < best.v || (o.v == best.v && o.key < best.key);
                best.v = t ? o.v : best.v; best.key = t ? o.key : best.key; best.colown = t ? o.colown : best.colown; win = t ? w : win;
            }
            const int jw = (int)(best.colown & 0xffffu), owner = (int)(best.colown >> 16) - 1;
            minVal = best.v;
            const int index = best.key < 0x40000000 ? 0x3fffffff - best.key : best.key - 0x40000000;
            if (owner == -1) sink = jw;
            else { i = owner; ui = cand_u[parity][win]; }
            if (it == index) {
                // this thread holds the selected column: its path / cost are final (spc == minVal), then swap-with-last
                path[jw] = pth; spc_sel[jw] = spc; SC[jw] = stamp;
                if (index != num_remaining - 1) {
                    const LsapMail m = mail[parity];
                    spc = m.spc; vj = m.vj; uo = m.uo; j = m.j; own = m.own; pth = m.pth;
                }
            }
            --num_remaining;
            parity ^= 1;
        }
        __syncthreads();
        // dual update (selected columns: spc_sel; the sink's entry equals minVal) - and, concurrently on thread 0, the
        // augmentation: the dual update reads the pre-augmentation columns from registers (c4r), so the two do not interfere
        if (tid < nr) {
            if (tid == cur) u[tid] += minVal;
            else if (SR[tid] == stamp) u[tid] += minVal - spc_sel[c4r];
        }
        if (tid < nc && SC[tid] == stamp) v[tid] -= minVal - spc_sel[tid];
        if (tid == 0) {
            int jj = sink;
            while (true) {
                const int r = path[jj];
                row4col[jj] = r;
                const int t = col4row[r]; col4row[r] = jj; jj = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    if (transpose) {
        if (tid == 0) {
            int k = 0;
            for (int r = 0; r < nc; ++r) {          // nc == original row count
                const int who = row4col[r];
                if (who >= 0) { row_out[k] = r; col_out[k] = who; ++k; }
            }
        }
    } else if (tid < nr) { row_out[tid] = tid; col_out[tid] = col4row[tid]; }
    if (tid == 0 && n_out) *n_out = nr;
}
constexpr size_t LSAP_REG_STATE = (size_t)LSAP_REG_COLS * (3 * sizeof(double) + 5 * sizeof(int));

// ---- single-wavefront form for problems with at most 256 columns (after the tall->wide transposition): every lane keeps
// the scan state of up to four `remaining` positions (column, shortest-path cost, column dual, owner row) in REGISTERS, so a
// Dijkstra step is one LDS cost read per position, the DPP reduction and a register hand-over for the swap-with-last
// removal - no barrier and no LDS round trip on the critical path (the workgroup form above spends ~1 us per step).
// Same arithmetic, same tie-break key, same result.
constexpr int LSAP_WAVE_SLOTS = LSAP_WAVE_COLS / 64;
constexpr size_t LSAP_WAVE_COST_MAX = 128 * 1024;               // cost matrix copied to LDS when it fits

// SLOTS positions per lane (1: up to 64 columns, 4: up to 256).  A single wavefront issues one instruction every few cycles,
// so the step is written for instruction count: only (cost, key) travel through the reduction - the key names the position,
// whose lane then hands out column and owner - and inactive positions are masked with selects instead of branches.
// Synchronisation inside the single-wavefront form: the wave's own LDS traffic only (its LDS operations complete in order once
// the counter is drained) - NOT a workgroup barrier, because the fused association kernel (tracker.hip) runs this form on wave 0
// of a four-wave workgroup whose other waves wait at the barrier behind it.
__device__ __forceinline__ void lsap_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <bool COST_LDS, int SLOTS>
__device__ __forceinline__ void lsap_wave_solve(const float *cost, const float *cost_lds, int nr0, int nc0, char *smem, int *row_out, int *col_out) {
    const int lane = threadIdx.x;
    const bool transpose = nc0 < nr0;
    const int nr = transpose ? nc0 : nr0, nc = transpose ? nr0 : nc0;
    double *u = reinterpret_cast<double *>(smem), *v = u + LSAP_WAVE_COLS, *spc_rm = v + LSAP_WAVE_COLS;
    int *path = reinterpret_cast<int *>(spc_rm + LSAP_WAVE_COLS), *col4row = path + LSAP_WAVE_COLS, *row4col = col4row + LSAP_WAVE_COLS,
        *SR = row4col + LSAP_WAVE_COLS, *SC = SR + LSAP_WAVE_COLS;
    const int sj = transpose ? nc0 : 1, si = transpose ? 1 : nc0;       // cost(i, j) at i * si + j * sj
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    lsap_wave_sync();
    for (int cur = 0; cur < nr; ++cur) {
        int jj[SLOTS], ow[SLOTS], cofs[SLOTS];
        double sp[SLOTS], vv[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int it = lane + 64 * s, j = nc - 1 - it;          // `remaining` starts as nc-1 .. 0
            const bool in = it < nc;
            jj[s] = in ? j : 0; cofs[s] = jj[s] * sj; sp[s] = INFINITY;
            vv[s] = v[jj[s]]; ow[s] = row4col[jj[s]];
        }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        for (int j = lane; j < nc; j += 64) SC[j] = 0;
        lsap_wave_sync();
        double minVal = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            const int row_off = i * si;
            double cv = INFINITY;
            int ckey = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int it = lane + 64 * s;
                const bool active = it < num_remaining;
                const int at = row_off + cofs[s];
                const double cij = (double)(COST_LDS ? cost_lds[at] : cost[at]);
                const double r = minVal + cij - ui - vv[s];            // scipy's order: minVal + cost - u[i] - v[j]
                if (active && r < sp[s]) { path[jj[s]] = i; sp[s] = r; }
                const int key = ow[s] == -1 ? 0x3fffffff - it : 0x40000000 + it;
                LSAP_TAKE_MIN(cv, ckey, active ? sp[s] : (double)INFINITY, active ? key : 0x7fffffff);
            }
            lsap_wave_min(cv, ckey);
            minVal = cv;
            const int index = ckey < 0x40000000 ? 0x3fffffff - ckey : ckey - 0x40000000;
            const int is = SLOTS == 1 ? 0 : index >> 6, il = index & 63;
            // the position's lane hands out its column and owner
            int j = 0, own = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (s == is) { j = __builtin_amdgcn_readlane(jj[s], il); own = __builtin_amdgcn_readlane(ow[s], il); }
            if (own == -1) sink = j; else i = own;
            // swap-with-last removal: position `index` takes over the registers of position num_remaining - 1
            const int last = __builtin_amdgcn_readfirstlane(num_remaining - 1);
            const int ls = SLOTS == 1 ? 0 : last >> 6, ll = last & 63;
            int t_j = 0, t_o = 0, t_c = 0;
            union { double d; int w[2]; } t_sp, t_vv, a;
            t_sp.d = 0.0; t_vv.d = 0.0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (s == ls) {                                       // uniform
                    t_j = __builtin_amdgcn_readlane(jj[s], ll);
                    t_o = __builtin_amdgcn_readlane(ow[s], ll);
                    t_c = __builtin_amdgcn_readlane(cofs[s], ll);
                    a.d = sp[s]; t_sp.w[0] = __builtin_amdgcn_readlane(a.w[0], ll); t_sp.w[1] = __builtin_amdgcn_readlane(a.w[1], ll);
                    a.d = vv[s]; t_vv.w[0] = __builtin_amdgcn_readlane(a.w[0], ll); t_vv.w[1] = __builtin_amdgcn_readlane(a.w[1], ll);
                }
            if (lane == il) { SC[j] = 1; spc_rm[j] = minVal; }       // the removed column keeps its shortest-path cost for the dual update
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const bool here = s == is && lane == il;
                jj[s] = here ? t_j : jj[s]; ow[s] = here ? t_o : ow[s]; cofs[s] = here ? t_c : cofs[s];
                sp[s] = here ? t_sp.d : sp[s]; vv[s] = here ? t_vv.d : vv[s];
            }
            --num_remaining;
        }
        lsap_wave_sync();
        for (int r = lane; r < nr; r += 64) {
            if (r == cur) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc_rm[col4row[r]];
        }
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc_rm[j];
        lsap_wave_sync();
        if (lane == 0) {
            int j = sink;
            while (true) {
                int r = path[j];
                row4col[j] = r;
                int t = col4row[r]; col4row[r] = j; j = t;
                if (r == cur) break;
            }
        }
        lsap_wave_sync();
    }
    if (transpose) {
        if (lane == 0) {
            int k = 0;
            for (int r = 0; r < nc; ++r) {          // nc == original row count
                int who = row4col[r];
                if (who >= 0) { row_out[k] = r; col_out[k] = who; ++k; }
            }
        }
    } else {
        for (int r = lane; r < nr; r += 64) { row_out[r] = r; col_out[r] = col4row[r]; }
    }
}

constexpr size_t LSAP_WAVE_STATE = (size_t)LSAP_WAVE_COLS * (3 * sizeof(double) + 5 * sizeof(int));

// One problem solved by the calling workgroup of LSAP_NT threads, form chosen from the ACTUAL sizes (they are only known on the
// device): <= 64 columns the single-wavefront form on wave 0 (the other waves wait at the closing barrier), <= 256 the register-
// resident form, else the LDS-state form, and with state_global != nullptr the global-state form for problems whose state
// exceeds the LDS.  smem_bytes = dynamic LDS of the launch (lsap_smem).  Every thread must call; *n_out is written by one thread
// and is visible to the workgroup after the closing barrier.
__device__ __forceinline__ void lsap_solve_block(const float *cost, int nr0, int nc0, int *row_out, int *col_out, int *n_out, char *state_global,
                                                 int smem_bytes) {
    extern __shared__ __attribute__((aligned(16))) char lsap_smem[];
    if (nr0 <= 0 || nc0 <= 0) {                                  // linear_assignment.py:48-49 early-out
        if (threadIdx.x == 0) *n_out = 0;
        __syncthreads();
        return;
    }
    const int n = max(nr0, nc0);
    if (n <= LSAP_WAVE_COLS) {
        if (threadIdx.x < 64) {
            float *cost_lds = reinterpret_cast<float *>(lsap_smem + LSAP_WAVE_STATE);
            if (LSAP_WAVE_STATE + (size_t)nr0 * nc0 * sizeof(float) <= (size_t)smem_bytes) {
                for (int i = threadIdx.x; i < nr0 * nc0; i += 64) cost_lds[i] = cost[i];
                lsap_wave_solve<true, 1>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
            } else {
                lsap_wave_solve<false, 1>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
            }
            if (threadIdx.x == 0) *n_out = min(nr0, nc0);
        }
        __syncthreads();
        return;
    }
    if (n <= LSAP_REG_COLS && LSAP_REG_STATE <= (size_t)smem_bytes) {
        if (LSAP_REG_STATE + (size_t)nr0 * nc0 * sizeof(float) <= (size_t)smem_bytes) lsap_reg_solve<true>(cost, nr0, nc0, row_out, col_out, n_out);
        else lsap_reg_solve<false>(cost, nr0, nc0, row_out, col_out, n_out);
        __syncthreads();
        return;
    }
    const size_t state = (size_t)n * LSAP_STATE_BYTES;
    if (state <= (size_t)smem_bytes) {
        if (state + (size_t)nr0 * nc0 * sizeof(float) <= (size_t)smem_bytes) lsap_wg_solve<false, true>(cost, nr0, nc0, row_out, col_out, n_out, nullptr);
        else lsap_wg_solve<false, false>(cost, nr0, nc0, row_out, col_out, n_out, nullptr);
    } else {
        if ((size_t)nr0 * nc0 * sizeof(float) <= (size_t)smem_bytes) lsap_wg_solve<true, true>(cost, nr0, nc0, row_out, col_out, n_out, state_global);
        else lsap_wg_solve<true, false>(cost, nr0, nc0, row_out, col_out, n_out, state_global);
    }
    __syncthreads();
}

}  // namespace yds
