// Where does the dispatcher put the workgroups of a launch?  (tuning aid for the phase-offset experiment of conv_win2.hip)
// 256-thread workgroups with 68 KB of LDS (two per CU), each records XCC id, HW_ID (SE / CU) and its start time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void probe(unsigned *out, int spin) {
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)(t0 >> 6);
        lds[0] = 1;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
    if (threadIdx.x == 0) out[blockIdx.x * 4 + 3] = (unsigned)(__builtin_amdgcn_s_memtime() >> 6);
}
int main() {
    const int n = 1444;
    unsigned *d;
    hipMalloc(&d, n * 16);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 68 * 1024, 0, d, 20);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 4);
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned, std::vector<int>> cu;
    unsigned tmin = ~0u;
    for (int b = 0; b < n; ++b) tmin = h[b * 4 + 2] < tmin ? h[b * 4 + 2] : tmin;
    for (int b = 0; b < 600 && b < n; ++b) {
        unsigned hw = h[b * 4], key = ((h[b * 4 + 1] & 0xf) << 16) | (hw & 0xff00);
        cu[key].push_back(b);
    }
    printf("distinct (xcc, se, sh, cu) among the first 600 blocks: %zu\n", cu.size());
    int shown = 0;
    for (auto &kv : cu) {
        if (shown++ >= 12) break;
        printf("xcc %u hw %04x :", kv.first >> 16, kv.first & 0xffff);
        for (int b : kv.second) printf(" b%d(t%u)", b, h[b * 4 + 2] - tmin);
        printf("\n");
    }
    // how many of the first 512 blocks share a CU with block b+256 / b+8 / b+1
    auto key_of = [&](int b) { return ((h[b * 4 + 1] & 0xf) << 16) | (h[b * 4] & 0xff00); };
    int s256 = 0, s8 = 0, s1 = 0, s16 = 0;
    for (int b = 0; b < 256; ++b) { s256 += key_of(b) == key_of(b + 256); s8 += key_of(b) == key_of(b + 8); s1 += key_of(b) == key_of(b + 1); s16 += key_of(b) == key_of(b + 16); }
    printf("of blocks 0..255: same CU as b+256: %d, as b+8: %d, as b+16: %d, as b+1: %d\n", s256, s8, s16, s1);
    return 0;
}
