// Multi-GPU exchange step of the stream-sharded run: one process per GPU, one video stream per rank, and ONE
// collective per frame batch - an all-gather of every rank's result block {int32 count; int32 rows[R][6]} per frame (R chosen by
// the caller, grown when a frame has more rows) so that rank 0 can emit all streams' tracker rows - plus small all-reduces (counters, the max-over-ranks time)
// and a barrier.  RCCL directly (ncclCommInitRank from an id the launcher distributes), on this library's own stream on
// the bound device; no torch types, no torch.distributed on the data path.
//
// librccl is opened lazily with dlopen (RTLD_LOCAL) and only when a communicator is created: single-GPU users never load
// it, and the copy a Python process may already have mapped (torch ships one) is never mixed with another by symbol name.
//
// The reference has no distributed code; the per-stream contract is DeepSort.clone() (deep_sort/deep_sort.py:41-44).
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "ydsort.h"

namespace yds {
namespace {

struct Rccl {
    void *h = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};

Rccl &rccl() {
    static Rccl r;
    if (r.h) return r;
    // The RCCL that belongs to the HIP runtime THIS library runs on: a Python process may hold two ROCm stacks (torch
    // ships its own libamdhip64 / librccl next to /opt/rocm's), and an RCCL resolved by bare name can be the other stack's -
    // its HIP calls then fail on this runtime's streams ("unhandled cuda error" in ncclCommInitRank).  So: the directory
    // of the libamdhip64 that hipGetDeviceCount resolves to, by absolute path (a path with a slash is never matched
    // against an already-loaded library of the same SONAME); bare names only as a fallback.
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash);
            names.push_back(dir + "/librccl.so.1");
            names.push_back(dir + "/librccl.so");
        }
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string &n : names) {
        r.h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.h) { r.path = n; break; }
    }
    if (!r.h) fail("comm: librccl not found (%s); multi-GPU runs need RCCL", dlerror());
    auto sym = [&](const char *n) {
        void *p = dlsym(r.h, n);
        if (!p) fail("comm: librccl has no symbol %s", n);
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    return r;
}

#define YDS_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) ::yds::fail("RCCL: %s failed: %s", #call, ::yds::rccl().GetErrorString(r_)); \
    } while (0)

}  // namespace
}  // namespace yds

struct yds_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    hipStream_t stream = nullptr;
    yds::DevBuf<char> send, recv;         // staging for host-side payloads
    char *pin = nullptr;                  // pinned bounce buffer
    size_t pin_bytes = 0;

    void ensure(size_t send_bytes, size_t recv_bytes) {
        if (send.n < send_bytes) send.alloc(send_bytes * 2);
        if (recv.n < recv_bytes) recv.alloc(recv_bytes * 2);
        if (pin_bytes < recv_bytes + send_bytes) {
            if (pin) (void)hipHostFree(pin);
            pin_bytes = (recv_bytes + send_bytes) * 2;
            YDS_HIP(hipHostMalloc((void **)&pin, pin_bytes));
        }
    }
};

namespace {
yds_comm *live(yds_comm *c) {
    if (!c || !c->comm) yds::fail("comm: null or destroyed communicator handle");
    return c;
}
}  // namespace

extern "C" {

/* local, NON-collective check that a communicator can be attempted: librccl found next to this runtime, its symbols resolved, a
 * device bound by yds_init, a stream created on it.  The ranks vote on it over their host group BEFORE any of them enters
 * ncclCommInitRank (a collective: a rank that fails earlier would leave the others blocked in it). */
int yds_comm_preflight(void) {
    YDS_API_BEGIN
    if (yds::bound_device() < 0) yds::fail("comm: yds_init has not bound a device");
    int v = 0;
    YDS_NCCL(yds::rccl().GetVersion(&v));
    hipStream_t s = yds::make_stream(true);
    YDS_HIP(hipStreamSynchronize(s));
    YDS_HIP(hipStreamDestroy(s));
    YDS_API_END
}

int yds_comm_unique_id(void *id128_out) {
    YDS_API_BEGIN
    static_assert(sizeof(ncclUniqueId) == YDS_COMM_ID_BYTES, "RCCL unique id size");
    ncclUniqueId id;
    YDS_NCCL(yds::rccl().GetUniqueId(&id));
    memcpy(id128_out, &id, sizeof id);
    YDS_API_END
}

yds_comm *yds_comm_create(const void *id128, int world, int rank) {
    YDS_API_BEGIN
    if (yds::bound_device() < 0) yds::fail("comm: yds_init has not bound a device");
    if (world < 1 || rank < 0 || rank >= world) yds::fail("comm: rank %d outside world %d", rank, world);
    std::unique_ptr<yds_comm> c(new yds_comm);
    c->world = world;
    c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    c->stream = yds::make_stream(true);
    try {
        YDS_NCCL(yds::rccl().CommInitRank(&c->comm, world, id, rank));  // collective: every rank of the job calls it
    } catch (...) {
        (void)hipStreamDestroy(c->stream);
        throw;
    }
    return c.release();
    YDS_API_END_PTR
}

void yds_comm_destroy(yds_comm *c) {
    if (!c) return;
    yds::bind_thread();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)yds::rccl().CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->pin) (void)hipHostFree(c->pin);
    delete c;
}

int yds_comm_world(const yds_comm *c) { return c ? c->world : 1; }
int yds_comm_rank(const yds_comm *c) { return c ? c->rank : 0; }

int yds_comm_rccl_version(void) {
    YDS_API_BEGIN
    int v = 0;
    YDS_NCCL(yds::rccl().GetVersion(&v));
    return v;
    }
    catch (const std::exception &e) { yds::set_error(e.what()); return -1; }
}

int yds_comm_allgather_dev(yds_comm *c, const void *send_dev, size_t bytes, void *recv_dev) {
    YDS_API_BEGIN
    live(c);
    YDS_NCCL(yds::rccl().AllGather(send_dev, recv_dev, bytes, ncclChar, c->comm, c->stream));
    YDS_HIP(hipStreamSynchronize(c->stream));
    YDS_API_END
}

int yds_comm_allgather(yds_comm *c, const void *send_host, size_t bytes, void *recv_host) {
    YDS_API_BEGIN
    live(c);
    const size_t total = bytes * (size_t)c->world;
    c->ensure(bytes, total);
    memcpy(c->pin, send_host, bytes);
    YDS_HIP(hipMemcpyAsync(c->send.p, c->pin, bytes, hipMemcpyHostToDevice, c->stream));
    YDS_NCCL(yds::rccl().AllGather(c->send.p, c->recv.p, bytes, ncclChar, c->comm, c->stream));
    YDS_HIP(hipMemcpyAsync(c->pin + bytes, c->recv.p, total, hipMemcpyDeviceToHost, c->stream));
    YDS_HIP(hipStreamSynchronize(c->stream));
    memcpy(recv_host, c->pin + bytes, total);
    YDS_API_END
}

/* the result block of the exchange step: per frame {int32 header; int32 rows[rows_per_block][6]}; header = row count, -1 = the
 * detector returned None, -(2 + n) = the frame has n > rows_per_block rows and was not sent.  *rows_needed = the largest row count
 * any rank announced: when it exceeds rows_per_block the caller repeats the call with a block of at least that many rows (every
 * rank sees the same headers, so every rank takes the same decision). */
int yds_comm_allgather_rows(yds_comm *c, const int32_t *out6_host, int cap, const int32_t *counts_host, int batch, int rows_per_block,
                            int32_t *all_host, int *rows_needed) {
    YDS_API_BEGIN
    live(c);
    // (rows_per_block and batch are protocol constants: the same on every rank, so this check fails on all of them alike)
    if (rows_per_block < 1) yds::fail("comm: rows_per_block must be positive");
    const int BLK = 1 + rows_per_block * 6;
    constexpr int32_t BAD = INT32_MIN;             // header of a frame whose rank found its own input inconsistent
    std::vector<int32_t> blk((size_t)batch * BLK, 0);
    int bad_frame = -1, bad_n = 0;
    for (int b = 0; b < batch; ++b) {
        const int n = counts_host[b];
        // A rank-LOCAL inconsistency must not keep this rank out of the collective (its peers would block in ncclAllGather for
        // ever): it is announced in band, every rank sees it in the gathered headers and all of them fail together.
        if (n > cap) { blk[(size_t)b * BLK] = BAD; if (bad_frame < 0) { bad_frame = b; bad_n = n; } continue; }
        blk[(size_t)b * BLK] = n > rows_per_block ? -(2 + n) : n;
        if (n > 0 && n <= rows_per_block) memcpy(&blk[(size_t)b * BLK + 1], out6_host + (size_t)b * cap * 6, (size_t)n * 6 * sizeof(int32_t));
    }
    if (yds_comm_allgather(c, blk.data(), blk.size() * sizeof(int32_t), all_host) != 0) return -1;
    int need = 0;
    for (size_t f = 0; f < (size_t)c->world * batch; ++f) {
        const int h = all_host[f * BLK];
        if (h == BAD) {
            const int r = (int)(f / batch);
            if (r == c->rank && bad_frame >= 0) yds::fail("comm: frame %d announces %d rows, the caller's buffer holds %d", bad_frame, bad_n, cap);
            yds::fail("comm: rank %d announced an inconsistent result block (frame %d); the exchange step failed on every rank", r, (int)(f % batch));
        }
        need = std::max(need, h <= -2 ? -2 - h : std::max(h, 0));
    }
    if (rows_needed) *rows_needed = need;
    return 0;
    }
    catch (const std::exception &e) { yds::set_error(e.what()); return -1; }
}

int yds_comm_allreduce_f64(yds_comm *c, double *vals_host, int n, int op) {
    YDS_API_BEGIN
    live(c);
    if (op != 0 && op != 1) yds::fail("comm: op must be 0 (sum) or 1 (max)");
    const size_t bytes = (size_t)n * sizeof(double);
    c->ensure(bytes, bytes);
    memcpy(c->pin, vals_host, bytes);
    YDS_HIP(hipMemcpyAsync(c->send.p, c->pin, bytes, hipMemcpyHostToDevice, c->stream));
    YDS_NCCL(yds::rccl().AllReduce(c->send.p, c->recv.p, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
    YDS_HIP(hipMemcpyAsync(c->pin + bytes, c->recv.p, bytes, hipMemcpyDeviceToHost, c->stream));
    YDS_HIP(hipStreamSynchronize(c->stream));
    memcpy(vals_host, c->pin + bytes, bytes);
    YDS_API_END
}

int yds_comm_barrier(yds_comm *c) {
    double one = 1.0;
    return yds_comm_allreduce_f64(c, &one, 1, 0);
}

}  // extern "C"
