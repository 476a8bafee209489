// Runtime plumbing of libydsort: error reporting, device selection, raw HBM buffers.
#include "common.h"

#include <atomic>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

namespace yds {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

// One process drives one GPU (multi-GPU = one process per GPU).  yds_init() binds the process to a device
// once; hipSetDevice is per host thread, so every ABI entry re-selects the bound device for the calling thread.
static std::atomic<int> g_device{-1};
static thread_local int t_device = -1;

int bound_device() { return g_device.load(); }

void bind_thread() {
    const int d = g_device.load();
    if (d >= 0 && t_device != d) {
        if (hipSetDevice(d) == hipSuccess) t_device = d;
    }
}

void fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(buf);
}

// Non-blocking stream for a handle.  latency_role (the tracker's stream, the RCCL stream): a chain of ~200 tiny kernels per
// batch whose completion the host waits for - created at the highest stream priority (YDS_STREAM_PRIO=0 disables), so that its
// workgroups are dispatched ahead of the convolution streams' whenever a slot frees up: the association of a batch took
// 5.2 ms of host time under a saturated GPU against 1.5 ms on an idle one.
// (A CU partition between the convolution streams and the association stream - hipExtStreamCreateWithCUMask with
// complementary masks - was measured in round 2 and removed: masked streams ran the association 1.6x (30 persons) to 5.6x
// (crowd) slower, see DESIGN.md.)
hipStream_t make_stream(bool latency_role) {
    hipStream_t st = nullptr;
    static const bool use_prio = !(getenv("YDS_STREAM_PRIO") && atoi(getenv("YDS_STREAM_PRIO")) == 0);
    if (latency_role && use_prio) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
            YDS_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest));
            return st;
        }
    }
    YDS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
}

}  // namespace yds

extern "C" {

const char *yds_last_error(void) { return yds::g_last_error.c_str(); }

const char *yds_build_info(void) {
    return "libydsort 0.2 (HIP, gfx950; conv: f16x3 split-fp16 MFMA 32x32x16 [default], exact fp32 MFMA, f16 single-term half mode)";
}

int yds_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int yds_init(int device_id) {
    YDS_API_BEGIN
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) yds::fail("no HIP device visible (%s); libydsort has no CPU path", hipGetErrorString(e));
    const int bound = yds::g_device.load();
    if (device_id < 0) device_id = bound >= 0 ? bound : 0;                    /* "whatever is already selected" */
    if (device_id >= n) yds::fail("device %d outside [0,%d)", device_id, n);
    if (bound >= 0 && bound != device_id)
        yds::fail("this process is already bound to device %d; one process drives one GPU (requested %d)", bound, device_id);
    YDS_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    YDS_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        yds::fail("device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
    yds::g_device.store(device_id);
    yds::t_device = device_id;
    YDS_API_END
}

int yds_current_device(void) { return yds::g_device.load(); }

int yds_device_pci_bus_id(char *buf, int len) {
    YDS_API_BEGIN
    const int d = yds::g_device.load();
    if (d < 0) yds::fail("yds_init has not been called");
    YDS_HIP(hipDeviceGetPCIBusId(buf, len, d));
    YDS_API_END
}

void *yds_dev_alloc(size_t nbytes) {
    YDS_API_BEGIN
    void *p = nullptr;
    YDS_HIP(hipMalloc(&p, nbytes ? nbytes : 1));
    return p;
    YDS_API_END_PTR
}

void *yds_host_alloc(size_t nbytes) {
    YDS_API_BEGIN
    void *p = nullptr;
    YDS_HIP(hipHostMalloc(&p, nbytes ? nbytes : 1, hipHostMallocDefault));
    return p;
    YDS_API_END_PTR
}

int yds_host_free(void *host) {
    YDS_API_BEGIN
    if (host) YDS_HIP(hipHostFree(host));
    YDS_API_END
}

int yds_dev_free(void *dev) {
    YDS_API_BEGIN
    if (dev) YDS_HIP(hipFree(dev));
    YDS_API_END
}

int yds_memcpy_h2d(void *dst_dev, const void *src_host, size_t nbytes) {
    YDS_API_BEGIN
    YDS_HIP(hipMemcpy(dst_dev, src_host, nbytes, hipMemcpyHostToDevice));
    YDS_API_END
}

int yds_memcpy_d2h(void *dst_host, const void *src_dev, size_t nbytes) {
    YDS_API_BEGIN
    YDS_HIP(hipMemcpy(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost));
    YDS_API_END
}

int yds_memcpy_d2d(void *dst_dev, const void *src_dev, size_t nbytes) {
    YDS_API_BEGIN
    YDS_HIP(hipMemcpy(dst_dev, src_dev, nbytes, hipMemcpyDeviceToDevice));
    YDS_API_END
}

int yds_device_sync(void) {
    YDS_API_BEGIN
    YDS_HIP(hipDeviceSynchronize());
    YDS_API_END
}

}  // extern "C"
