// Sustained shader clock under load WITHOUT touching the product kernels: one wave of one workgroup, launched on its own stream,
// samples the shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime), then sleeps (s_sleep) until the
// host raises a stop flag (a device word written by a copy on a second stream, polled with an agent-scope atomic load: a plain or
// scalar load would be served from a cache for the kernel's whole life) or a bounded number of ticks has passed, and samples again.  cycles / ticks over
// that interval is the clock the chip held while whatever ran beside the probe had the other 255 CUs (and this CU's other SIMDs).
// Round 6: replaces the sampling that rounds 3-5 compiled INTO the window-resident conv kernels (one s_memtime pair + two atomicAdd
// per 32nd workgroup); that form is now only built with -DYDS_CLOCK_PROBE (tools/tagbuild.sh) for the A/B.
#include "common.h"

namespace yds {
namespace {

__device__ __forceinline__ unsigned long long ticks100mhz() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

__global__ void clock_probe_kernel(int *stop, unsigned long long *out, unsigned long long max_ticks) {
    const unsigned long long t0 = ticks100mhz(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = t0;
    for (;;) {
        for (int i = 0; i < 32; ++i) __builtin_amdgcn_s_sleep(127);             // ~130 us between polls of the host flag
        t1 = ticks100mhz();
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || t1 - t0 > max_ticks) break;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    t1 = ticks100mhz();
    out[0] = c1 - c0;
    out[1] = t1 - t0;
}

struct Probe {
    hipStream_t stream = nullptr, side = nullptr;   // the probe's stream; the stream the stop flag is written on
    int *stop = nullptr;                       // device word
    int *one = nullptr;                        // pinned source of the stop write
    unsigned long long *out = nullptr;         // pinned: the kernel's two counters
    bool running = false;
} g_probe;

}  // namespace

void clock_probe_start(double max_seconds) {
    Probe &p = g_probe;
    if (!p.stream) {
        YDS_HIP(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
        YDS_HIP(hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking));
        YDS_HIP(hipMalloc((void **)&p.stop, sizeof(int)));
        YDS_HIP(hipHostMalloc((void **)&p.one, sizeof(int), hipHostMallocDefault));
        YDS_HIP(hipHostMalloc((void **)&p.out, 2 * sizeof(unsigned long long), hipHostMallocMapped));
        *p.one = 1;
    }
    if (p.running) { double g, m; clock_probe_stop(&g, &m); }
    YDS_HIP(hipMemsetAsync(p.stop, 0, sizeof(int), p.stream));
    p.out[0] = p.out[1] = 0;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(1), 0, p.stream, p.stop, p.out, (unsigned long long)(max_seconds * 1e8));
    YDS_HIP(hipGetLastError());
    p.running = true;
}

void clock_probe_stop(double *ghz, double *ms) {
    Probe &p = g_probe;
    *ghz = 0;
    *ms = 0;
    if (!p.running) return;
    YDS_HIP(hipMemcpyAsync(p.stop, p.one, sizeof(int), hipMemcpyHostToDevice, p.side));
    YDS_HIP(hipStreamSynchronize(p.side));
    YDS_HIP(hipStreamSynchronize(p.stream));
    p.running = false;
    if (p.out[1]) {
        *ghz = (double)p.out[0] / (double)p.out[1] * 0.1;                      // ticks are 10 ns
        *ms = (double)p.out[1] * 1e-5;
    }
}

}  // namespace yds
