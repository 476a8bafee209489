// Detector post-processing on device.
//
// soft_non_max_suppression (reference yolo3/utils/model_build.py:52-137; despite its name it is hard,
// multi-label NMS) + resize_boxes (:12-19).  The greedy step is torchvision.ops.boxes.nms semantics
// (third party; call site model_build.py:119): stable sort by score descending, area (x2-x1)*(y2-y1),
// suppress j when inter/(a_i + a_j - inter) > thr with the threshold held as double.
//
// Pipeline (all sizes stay on device, one host sync at the end):
//   count   one thread per box: obj > thr, number of classes with cls*obj > thr
//   scan    exclusive prefix over boxes -> candidate slot of every (box, class) pair; this reproduces
//           torch.nonzero's row-major order (box ascending, class ascending)
//   emit    rows (x1,y1,x2,y2,score,cls), xywh -> xyxy as x -+ w/2
//   rank    stable descending rank of every candidate (score desc, candidate index asc) and scatter
//   mask    64-bit suppression words over class-offset boxes (box + cls*4096 in fp32, like the reference)
//   sweep   one workgroup walks candidates in score order, keeps <= 300
// Integer/index results are bit-exact against the oracle; the arithmetic is plain fp32 with explicit
// rounding (no FMA contraction) so that threshold decisions match.
#include "engine.h"

#include <string.h>

#include <algorithm>
#include <vector>

namespace yds {

constexpr int MAX_DET = 300;

__global__ void nms_count_kernel(const float *pred_all, size_t pred_stride, int n_boxes, int attrs, float thr, int *box_count_all) {
    const float *pred = pred_all + blockIdx.y * pred_stride;
    int *box_count = box_count_all + (size_t)blockIdx.y * n_boxes;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_boxes) return;
    const float *p = pred + (size_t)i * attrs;
    int cnt = 0;
    float obj = p[4];
    if (obj > thr) {
        for (int j = 5; j < attrs; ++j) cnt += __fmul_rn(p[j], obj) > thr;
    }
    box_count[i] = cnt;
}

// single-workgroup exclusive scan, in place; total -> counts[0]
__global__ void nms_scan_kernel(int *box_count_all, int n_boxes, int *counts_all) {
    int *box_count = box_count_all + (size_t)blockIdx.y * n_boxes;
    int *counts = counts_all + blockIdx.y * 4;
    __shared__ int wave_sum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_boxes; base += blockDim.x) {
        int i = base + tid;
        int v = i < n_boxes ? box_count[i] : 0;
        int incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_sum[w];
        int c = carry;
        if (i < n_boxes) box_count[i] = c + woff + incl - v;
        __syncthreads();
        if (tid == (int)blockDim.x - 1) carry = c + woff + incl;
        __syncthreads();
    }
    if (tid == 0) { counts[0] = carry; counts[1] = 0; }
}

__global__ void nms_emit_kernel(const float *pred_all, size_t pred_stride, int n_boxes, int attrs, float thr, const int *box_off_all, int max_cand,
                                float *cand_all, int corner) {
    const float *pred = pred_all + blockIdx.y * pred_stride;
    const int *box_off = box_off_all + (size_t)blockIdx.y * n_boxes;
    float *cand = cand_all + (size_t)blockIdx.y * max_cand * 6;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_boxes) return;
    const float *p = pred + (size_t)i * attrs;
    float obj = p[4];
    if (!(obj > thr)) return;
    int slot = box_off[i];
    float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];           // is_p1p2=True: boxes arrive in corner form
    if (!corner) {
        float hw = __fdiv_rn(p[2], 2.f), hh = __fdiv_rn(p[3], 2.f);
        x1 = __fsub_rn(p[0], hw); y1 = __fsub_rn(p[1], hh); x2 = __fadd_rn(p[0], hw); y2 = __fadd_rn(p[1], hh);
    }
    for (int j = 5; j < attrs; ++j) {
        float sc = __fmul_rn(p[j], obj);
        if (sc > thr) {
            if (slot < max_cand) {
                float *r = cand + (size_t)slot * 6;
                r[0] = x1; r[1] = y1; r[2] = x2; r[3] = y2; r[4] = sc; r[5] = (float)(j - 5);
            }
            ++slot;
        }
    }
}

__global__ void nms_rank_kernel(const float *cand_all, const int *counts_all, int max_cand, float *sorted_all) {
    const float *cand = cand_all + (size_t)blockIdx.y * max_cand * 6;
    float *sorted = sorted_all + (size_t)blockIdx.y * max_cand * 6;
    const int *counts = counts_all + blockIdx.y * 4;
    __shared__ float tile[256];
    const int n = min(counts[0], max_cand);
    for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        int i = base + threadIdx.x;
        float si = i < n ? cand[(size_t)i * 6 + 4] : 0.f;
        int rank = 0;
        for (int t0 = 0; t0 < n; t0 += 256) {
            int j = t0 + threadIdx.x;
            tile[threadIdx.x] = j < n ? cand[(size_t)j * 6 + 4] : -INFINITY;
            __syncthreads();
            int lim = min(256, n - t0);
            for (int k = 0; k < lim; ++k) {
                float sj = tile[k];
                rank += (sj > si) || (sj == si && (t0 + k) < i);
            }
            __syncthreads();
        }
        if (i < n) {
            const float *s = cand + (size_t)i * 6;
            float *d = sorted + (size_t)rank * 6;
            for (int k = 0; k < 6; ++k) d[k] = s[k];
        }
    }
}

__device__ __forceinline__ void offset_box(const float *r, float b[4]) {
    float c = __fmul_rn(r[5], 4096.f);       // model_build.py:117 class offset, fp32
    b[0] = __fadd_rn(r[0], c); b[1] = __fadd_rn(r[1], c); b[2] = __fadd_rn(r[2], c); b[3] = __fadd_rn(r[3], c);
}

__global__ void nms_mask_kernel(const float *sorted_all, const int *counts_all, int max_cand, double thr, unsigned long long *mask_all, int words_ld) {
    const float *sorted = sorted_all + (size_t)blockIdx.y * max_cand * 6;
    const int *counts = counts_all + blockIdx.y * 4;
    unsigned long long *mask = mask_all + (size_t)blockIdx.y * max_cand * words_ld;
    const int n = min(counts[0], max_cand);
    const int words = (n + 63) / 64;
    const long total = (long)n * words;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int i = idx / words, w = idx - (long)i * words;
        unsigned long long bits = 0;
        int j0 = w * 64;
        if (j0 + 63 > i) {
            float bi[4];
            offset_box(sorted + (size_t)i * 6, bi);
            float ai = __fmul_rn(__fsub_rn(bi[2], bi[0]), __fsub_rn(bi[3], bi[1]));
            for (int b = 0; b < 64; ++b) {
                int j = j0 + b;
                if (j <= i || j >= n) continue;
                float bj[4];
                offset_box(sorted + (size_t)j * 6, bj);
                float aj = __fmul_rn(__fsub_rn(bj[2], bj[0]), __fsub_rn(bj[3], bj[1]));
                float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
                float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
                float ww = fmaxf(0.f, __fsub_rn(xx2, xx1)), hh = fmaxf(0.f, __fsub_rn(yy2, yy1));
                float inter = __fmul_rn(ww, hh);
                float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ai, aj), inter));
                if ((double)ovr > thr) bits |= 1ull << b;
            }
        }
        mask[(size_t)i * words_ld + w] = bits;
    }
}

__global__ void nms_sweep_kernel(const float *sorted_all, const unsigned long long *mask_all, int words_ld, int *counts_all, int max_cand,
                                 float sx, float sy, float *kept_all, int cap) {
    const float *sorted = sorted_all + (size_t)blockIdx.y * max_cand * 6;
    const unsigned long long *mask = mask_all + (size_t)blockIdx.y * max_cand * words_ld;
    int *counts = counts_all + blockIdx.y * 4;
    float *kept = kept_all + (size_t)blockIdx.y * MAX_DET * 6;
    extern __shared__ unsigned long long removed[];
    __shared__ int n_keep;
    const int n = min(counts[0], max_cand);
    const int words = (n + 63) / 64;
    for (int w = threadIdx.x; w < words; w += blockDim.x) removed[w] = 0;
    if (threadIdx.x == 0) n_keep = 0;
    __syncthreads();
    const int lim = min(cap, MAX_DET);
    for (int i = 0; i < n; ++i) {
        bool alive = !((removed[i >> 6] >> (i & 63)) & 1ull);
        int k = n_keep;
        if (k >= lim) break;
        __syncthreads();
        if (alive) {
            for (int w = threadIdx.x; w < words; w += blockDim.x) removed[w] |= mask[(size_t)i * words_ld + w];
            if (threadIdx.x < 6) {
                float v = sorted[(size_t)i * 6 + threadIdx.x];
                if (threadIdx.x == 0 || threadIdx.x == 2) v = __fmul_rn(v, sx);     // resize_boxes :12-19
                if (threadIdx.x == 1 || threadIdx.x == 3) v = __fmul_rn(v, sy);
                kept[(size_t)k * 6 + threadIdx.x] = v;
            }
            if (threadIdx.x == 0) n_keep = k + 1;
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[1] = n_keep;
}

NmsWorkspace::NmsWorkspace(int max_candidates, int frames) : max_cand(0), frames(0) { resize(max_candidates, frames); }

// The reference has no candidate limit (model_build.py:93-121 works on whatever passes the threshold); the workspace
// grows instead of failing.  The suppression bit matrix is max_cand^2 / 8 bytes per frame, which bounds what is
// practical: 262144 candidates = 8.6 GB (the reference's O(n^2) greedy loop would need hours there).
void NmsWorkspace::resize(int max_candidates, int n_frames) {
    int m = 1024;
    while (m < max_candidates) m *= 2;
    if (m > kMaxCandidates) fail("nms: %d candidates per image exceed the supported %d (raise conf_thres)", max_candidates, kMaxCandidates);
    if (m == max_cand && n_frames == frames) return;
    max_cand = m; frames = n_frames;
    cand.alloc((size_t)frames * max_cand * 6);
    sorted.alloc((size_t)frames * max_cand * 6);
    counts.alloc((size_t)frames * 4);
    mask.alloc((size_t)frames * max_cand * (max_cand / 64));
    kept.alloc((size_t)frames * MAX_DET * 6);
    if (h_counts) (void)hipHostFree(h_counts);
    if (h_kept) (void)hipHostFree(h_kept);
    YDS_HIP(hipHostMalloc((void **)&h_counts, (size_t)frames * 4 * sizeof(int)));
    YDS_HIP(hipHostMalloc((void **)&h_kept, (size_t)frames * MAX_DET * 6 * sizeof(float)));
}

int NmsWorkspace::needed(int n_frames) const {
    int m = 0;
    for (int f = 0; f < n_frames; ++f) m = std::max(m, h_counts[f * 4]);
    return m;
}

NmsWorkspace::~NmsWorkspace() {
    if (h_counts) (void)hipHostFree(h_counts);
    if (h_kept) (void)hipHostFree(h_kept);
}

// Results go to the host through a KERNEL that stores into the pinned buffers (host-coherent memory the device writes in place),
// not through copy commands: a device-to-host hipMemcpyAsync is a copy-engine job and queues behind a frame upload that is in
// flight (200 MB, 3.6 ms) - the host then learns the detections late, which the pipeline's serialized schedule cannot hide
// (measured: +1.2 ms per step with host frames).  One workgroup per image: the four counters, then the kept rows.
__global__ __launch_bounds__(256) void nms_publish_kernel(const int *counts, const float *kept, int *h_counts, float *h_kept, int cap) {
    const int f = blockIdx.x;
    if (threadIdx.x < 4) h_counts[f * 4 + threadIdx.x] = counts[f * 4 + threadIdx.x];
    const int n = min(counts[f * 4 + 1], cap) * 6;
    const float *src = kept + (size_t)f * MAX_DET * 6;
    float *dst = h_kept + (size_t)f * MAX_DET * 6;
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// All `n_frames` images go through each stage in ONE launch (blockIdx.y = image).
void NmsWorkspace::launch(const float *pred_dev, size_t pred_stride, int n_frames, int n_boxes, int attrs, float conf_thres, float iou_thres,
                          float sx, float sy, int cap, hipStream_t s) {
    if (attrs < 6) fail("nms: predictions need at least one class");
    if (n_frames < 1) fail("nms: no frames");
    if (n_frames > frames) resize(max_cand, n_frames);
    box_count.ensure((size_t)frames * n_boxes);
    const int nb = (n_boxes + 255) / 256;
    hipLaunchKernelGGL(nms_count_kernel, dim3(nb, n_frames), dim3(256), 0, s, pred_dev, pred_stride, n_boxes, attrs, conf_thres, box_count.p);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1, n_frames), dim3(1024), 0, s, box_count.p, n_boxes, counts.p);
    hipLaunchKernelGGL(nms_emit_kernel, dim3(nb, n_frames), dim3(256), 0, s, pred_dev, pred_stride, n_boxes, attrs, conf_thres, box_count.p,
                       max_cand, cand.p, corner ? 1 : 0);
    hipLaunchKernelGGL(nms_rank_kernel, dim3(16, n_frames), dim3(256), 0, s, cand.p, counts.p, max_cand, sorted.p);
    const int words_ld = max_cand / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(64, n_frames), dim3(256), 0, s, sorted.p, counts.p, max_cand, (double)iou_thres, mask.p, words_ld);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1, n_frames), dim3(256), words_ld * sizeof(unsigned long long), s, sorted.p, mask.p, words_ld,
                       counts.p, max_cand, sx, sy, kept.p, cap);
    // results land in pinned host memory; the caller synchronises the stream (or an event) before collect()
    hipLaunchKernelGGL(nms_publish_kernel, dim3(n_frames), dim3(256), 0, s, counts.p, kept.p, h_counts, h_kept, std::min(cap, (int)MAX_DET));
    YDS_HIP(hipGetLastError());
}

int NmsWorkspace::collect(int frame, float *out6_host, int cap) {
    const int *c = h_counts + frame * 4;
    if (c[0] > max_cand) fail("nms: %d candidates exceed the workspace capacity %d (caller must resize and relaunch)", c[0], max_cand);
    int n = c[1] < cap ? c[1] : cap;
    if (n > 0) memcpy(out6_host, h_kept + (size_t)frame * MAX_DET * 6, (size_t)n * 6 * sizeof(float));
    return n;
}

int NmsWorkspace::run(const float *pred_dev, int n_boxes, int attrs, float conf_thres, float iou_thres, float sx, float sy,
                      float *out6_host, int cap, hipStream_t s) {
    launch(pred_dev, 0, 1, n_boxes, attrs, conf_thres, iou_thres, sx, sy, cap, s);
    YDS_HIP(hipStreamSynchronize(s));
    if (needed(1) > max_cand) {                                  // more candidates than the workspace holds: grow, run again
        resize(needed(1), frames);
        launch(pred_dev, 0, 1, n_boxes, attrs, conf_thres, iou_thres, sx, sy, cap, s);
        YDS_HIP(hipStreamSynchronize(s));
    }
    return collect(0, out6_host, cap);
}

// soft_non_max_suppression(merge=True, is_p1p2=True) as the reference behaves (model_build.py:122-131, see
// oracle/nms.py soft_non_max_suppression_merge): `bbox_iou(boxes[i], boxes)` is elementwise, so it only broadcasts when
// the number of kept boxes k is 1 or equals the number of candidates n.  In those two cases every kept box is replaced
// by one weighted-mean box before the following line raises into the bare except; otherwise the plain result stands.
int NmsWorkspace::run_merge(const float *pred_dev, int n_boxes, int attrs, float conf_thres, float iou_thres, float *out6_host, int cap,
                            hipStream_t s) {
    corner = true;
    try {
        launch(pred_dev, 0, 1, n_boxes, attrs, conf_thres, iou_thres, 1.f, 1.f, MAX_DET, s);
        YDS_HIP(hipStreamSynchronize(s));
        if (needed(1) > max_cand) {
            resize(needed(1), frames);
            launch(pred_dev, 0, 1, n_boxes, attrs, conf_thres, iou_thres, 1.f, 1.f, MAX_DET, s);
            YDS_HIP(hipStreamSynchronize(s));
        }
    } catch (...) { corner = false; throw; }
    corner = false;
    const int n = h_counts[0], k = h_counts[1];
    if (n > 1 && n < 3000 && (k == n || k == 1)) {
        std::vector<float> c((size_t)n * 6);
        YDS_HIP(hipMemcpy(c.data(), cand.p, c.size() * sizeof(float), hipMemcpyDeviceToHost));
        // kept order = stable descending score order over the candidates (all of them, or just the first)
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return c[(size_t)a * 6 + 4] > c[(size_t)b * 6 + 4]; });
        auto off = [&](int i, float b[4]) {
            const float o = c[(size_t)i * 6 + 5] * 4096.f;
            for (int d = 0; d < 4; ++d) b[d] = c[(size_t)i * 6 + d] + o;
        };
        double acc[4] = {0, 0, 0, 0};
        float wsum = 0.f;
        for (int j = 0; j < n; ++j) {
            float a[4], b[4];
            off(order[k == 1 ? 0 : j], a);
            off(j, b);
            // bbox_iou, +1 pixel convention (model_build.py:354-381), fp32
            float iw = fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + 1.f, ih = fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + 1.f;
            iw = iw > 0.f ? iw : 0.f; ih = ih > 0.f ? ih : 0.f;
            const float inter = iw * ih;
            const float a1 = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f), a2 = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
            const float iou = inter / ((a1 + a2) - inter + 1e-16f);
            const float wgt = iou > iou_thres ? c[(size_t)j * 6 + 4] : 0.f;
            wsum += wgt;
            for (int d = 0; d < 4; ++d) acc[d] += (double)wgt * c[(size_t)j * 6 + d];
        }
        for (int r = 0; r < k; ++r)
            for (int d = 0; d < 4; ++d) h_kept[(size_t)r * 6 + d] = (float)acc[d] / wsum;      // 0/0 -> NaN like the reference
    }
    const int rows = k < cap ? k : cap;
    if (rows > 0) memcpy(out6_host, h_kept, (size_t)rows * 6 * sizeof(float));
    return rows;
}

}  // namespace yds

// ============================================================================================ C ABI

namespace {
yds::NmsWorkspace &workspace() {
    static thread_local std::unique_ptr<yds::NmsWorkspace> ws;
    if (!ws) ws.reset(new yds::NmsWorkspace());
    return *ws;
}
}  // namespace

extern "C" {

int yds_nms(yds_net *n, int image, float conf_thres, float iou_thres, int frame_h, int frame_w, float *out6_host, int cap, int *n_out) {
    YDS_API_BEGIN
    yds::Darknet *d = n->d;
    if (image < 0 || image >= d->batch_max) yds::fail("nms: image %d outside batch", image);
    // resize_boxes (model_build.py:12-19): python-double ratio rounded to fp32, fp32 multiply
    float sx = frame_h > 0 ? (float)((double)frame_w / d->img_w) : 1.f;
    float sy = frame_h > 0 ? (float)((double)frame_h / d->img_h) : 1.f;
    const float *pred = d->out.p + (size_t)image * d->total_boxes * d->attrs;
    *n_out = workspace().run(pred, d->total_boxes, d->attrs, conf_thres, iou_thres, sx, sy, out6_host, cap, d->stream);
    YDS_API_END
}

int yds_detect_tiled(yds_net *n, const uint8_t *rgb_hwc_host, int h, int w, const int32_t *tiles_xyhw, int n_tiles, float conf_thres,
                     float iou_thres, float *out6_host, int cap, int *n_out) {
    YDS_API_BEGIN
    yds::Darknet *d = n->d;
    d->forward_tiles_host(rgb_hwc_host, h, w, tiles_xyhw, n_tiles);
    *n_out = workspace().run_merge(d->tiled_pred.p, n_tiles * d->total_boxes, d->attrs, conf_thres, iou_thres, out6_host, cap, d->stream);
    YDS_API_END
}

int yds_nms_merge_pred(const float *pred_host, int n_boxes, int attrs, float conf_thres, float iou_thres, float *out6_host, int cap, int *n_out) {
    YDS_API_BEGIN
    yds::DevBuf<float> pred;
    pred.upload(pred_host, (size_t)n_boxes * attrs);
    YDS_HIP(hipStreamSynchronize(nullptr));
    *n_out = workspace().run_merge(pred.p, n_boxes, attrs, conf_thres, iou_thres, out6_host, cap, nullptr);
    YDS_API_END
}

int yds_nms_pred(const float *pred_host, int n_boxes, int attrs, float conf_thres, float iou_thres, float *out6_host, int cap, int *n_out) {
    YDS_API_BEGIN
    yds::DevBuf<float> pred;
    pred.upload(pred_host, (size_t)n_boxes * attrs);
    YDS_HIP(hipStreamSynchronize(nullptr));
    *n_out = workspace().run(pred.p, n_boxes, attrs, conf_thres, iou_thres, 1.f, 1.f, out6_host, cap, nullptr);
    YDS_API_END
}

}  // extern "C"
