#include "tracker_lsap_dev.h"

namespace yds {

// ------------------------------------------------------------------------------------------ LSAP kernels (solvers: tracker_lsap_dev.h)

// The launch sizes its LDS from host-side upper bounds (inside a batch the live-track count is only bounded by T + sum D);
// whether the cost matrix is copied into LDS is decided HERE from the actual sizes - a 200 x 150 problem launched under a
// bound of 2600 x 150 must not fall back to reading its costs from global memory in the Dijkstra step.
template <bool GSTATE>
__global__ __launch_bounds__(LSAP_NT) void lsap_kernel(const float *cost, int nr0, int nc0, const int *dims_p, int *row_out, int *col_out,
                                                       int *n_out, char *state_global, int smem_bytes) {
    if (dims_p) { nr0 = dims_p[0]; nc0 = dims_p[1]; }
    if (nr0 <= 0 || nc0 <= 0) {                                  // linear_assignment.py:48-49 early-out
        if (threadIdx.x == 0 && n_out) *n_out = 0;
        return;
    }
    if (max(nr0, nc0) <= LSAP_WAVE_COLS) return;                 // solved by lsap_wave_kernel (launched in front of this one)
    if (max(nr0, nc0) <= LSAP_REG_COLS && LSAP_REG_STATE <= (size_t)smem_bytes) {       // register-resident form
        if (LSAP_REG_STATE + (size_t)nr0 * nc0 * sizeof(float) <= (size_t)smem_bytes) lsap_reg_solve<true>(cost, nr0, nc0, row_out, col_out, n_out);
        else lsap_reg_solve<false>(cost, nr0, nc0, row_out, col_out, n_out);
        return;
    }
    const size_t state = GSTATE ? 0 : (size_t)max(nr0, nc0) * LSAP_STATE_BYTES;
    if (state + (size_t)nr0 * nc0 * sizeof(float) <= (size_t)smem_bytes) lsap_wg_solve<GSTATE, true>(cost, nr0, nc0, row_out, col_out, n_out, state_global);
    else lsap_wg_solve<GSTATE, false>(cost, nr0, nc0, row_out, col_out, n_out, state_global);
}

// big = 0: this launch solves problems with <= 256 columns and leaves larger ones to the workgroup kernel (which is launched
// with skip_small = 1 right behind it): the sizes are only known on the device, the host picks nothing.
__global__ __launch_bounds__(64) void lsap_wave_kernel(const float *cost, int nr0, int nc0, const int *dims_p, int *row_out, int *col_out, int *n_out,
                                                      int cost_lds_floats) {
    if (dims_p) { nr0 = dims_p[0]; nc0 = dims_p[1]; }
    extern __shared__ __attribute__((aligned(16))) char lsap_smem[];
    if (nr0 <= 0 || nc0 <= 0) {
        if (threadIdx.x == 0 && n_out) *n_out = 0;
        return;
    }
    if (max(nr0, nc0) > LSAP_WAVE_COLS) return;                 // the workgroup kernel's case
    float *cost_lds = reinterpret_cast<float *>(lsap_smem + LSAP_WAVE_STATE);
    const bool narrow = max(nr0, nc0) <= 64;                     // one position per lane
    if (nr0 * nc0 <= cost_lds_floats) {
        for (int i = threadIdx.x; i < nr0 * nc0; i += 64) cost_lds[i] = cost[i];
        if (narrow) lsap_wave_solve<true, 1>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
        else lsap_wave_solve<true, LSAP_WAVE_SLOTS>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
    } else {
        if (narrow) lsap_wave_solve<false, 1>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
        else lsap_wave_solve<false, LSAP_WAVE_SLOTS>(cost, cost_lds, nr0, nc0, lsap_smem, row_out, col_out);
    }
    if (threadIdx.x == 0 && n_out) *n_out = min(nr0, nc0);
}

// nr_max / nc_max: upper bounds known on the host (they size the LDS / scratch); the real sizes may come from dims_dev
void launch_lsap(const float *cost_dev, int nr_max, int nc_max, const int *dims_dev, int *rows_dev, int *cols_dev, int *n_out_dev,
                 DevBuf<char> &scratch, hipStream_t s) {
    {
        static bool wave_attr = false;
        if (!wave_attr) {
            YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(LSAP_WAVE_STATE + LSAP_WAVE_COST_MAX)));
            wave_attr = true;
        }
        const size_t want = (size_t)std::min(nr_max, LSAP_WAVE_COLS) * std::min(nc_max, LSAP_WAVE_COLS) * sizeof(float);
        const size_t cost_lds = std::min(want, LSAP_WAVE_COST_MAX);
        hipLaunchKernelGGL(lsap_wave_kernel, dim3(1), dim3(64), LSAP_WAVE_STATE + cost_lds, s, cost_dev, nr_max, nc_max, dims_dev, rows_dev, cols_dev,
                           n_out_dev, (int)(cost_lds / sizeof(float)));
        YDS_HIP(hipGetLastError());
        if (std::max(nr_max, nc_max) <= LSAP_WAVE_COLS) return;
    }
    const size_t n = (size_t)std::max(std::max(nr_max, nc_max), 1);
    const size_t state = n * LSAP_STATE_BYTES, cost_bytes = (size_t)nr_max * nc_max * sizeof(float);
    const bool state_lds = state <= LSAP_LDS_MAX;
    // LDS: the state of the largest possible problem, plus the cost matrix if the bounds allow it; when they do not, the whole
    // LDS is requested anyway and the kernel decides from the actual sizes
    // (the register-resident form for <= 256 columns keeps a fixed 11 KB of state: make room for it and its cost copy too)
    const size_t reg_want = LSAP_REG_STATE + (size_t)std::min(nr_max, LSAP_REG_COLS) * std::min(nc_max, LSAP_REG_COLS) * sizeof(float);
    const size_t smem = std::max(state_lds ? std::min(state + cost_bytes, LSAP_LDS_MAX) : std::min(cost_bytes, LSAP_LDS_MAX),
                                 std::min(reg_want, LSAP_LDS_MAX));
    if (!state_lds && scratch.n < state) {
        YDS_HIP(hipStreamSynchronize(s));                        // nothing may still use the old scratch
        scratch.alloc(state);
    }
    static bool attr_set = false;
    if (!attr_set) {
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSAP_LDS_MAX));
        YDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsap_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSAP_LDS_MAX));
        attr_set = true;
    }
    auto kern = state_lds ? lsap_kernel<false> : lsap_kernel<true>;
    hipLaunchKernelGGL(kern, dim3(1), dim3(LSAP_NT), smem, s, cost_dev, nr_max, nc_max, dims_dev, rows_dev, cols_dev, n_out_dev,
                       state_lds ? (char *)nullptr : scratch.p, (int)smem);
    YDS_HIP(hipGetLastError());
}

}  // namespace yds
