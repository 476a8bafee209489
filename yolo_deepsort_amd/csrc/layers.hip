// HBM-bound layers of the detector / ReID graphs, NHWC fp32 with a per-pixel stride (ld) so that
// producers can write straight into channel slices of concatenated buffers.
//   maxpool  <- nn.MaxPool2d (+ ZeroPad2d for size=2,stride=1)   yolo3/models/models.py:58-64
//   upsample <- UpsampleExpand (nearest)                           models.py:118-133
//   copy     <- torch.cat / chunk of route layers                  models.py:300-303
//   add      <- shortcut                                           models.py:304-306
//   yolo     <- YOLOLayer.forward inference branch                 models.py:185-224
//   resize   <- cv2.resize(INTER_LINEAR) + /255                    yolo3/detect/img_detect.py:70-72
//   crop     <- frame[y1:y2,x1:x2] + cv2.resize + mean/std         deep_sort/deep_sort.py:138-141,
//                                                                  deep_sort/deep/feature_extractor.py:34-51
//   avgpool  <- AvgPool2d((8,4)) + x/||x||                         deep_sort/deep/model.py:70,88-91
// All kernels are one-element(-vector)-per-thread streaming kernels with 16-byte accesses where the
// layout allows; grids are capped and grid-strided.
#include "common.h"

#include <algorithm>
#include "h16.h"

namespace yds {

static inline int grid_for(size_t work, int block = 256) {
    size_t g = (work + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------ maxpool
__global__ void maxpool_kernel(const float *x, float *y, int N, int H, int W, int C, int ldx, int Ho, int Wo, int ldy,
                               int k, int stride, int pad, int zero_br, int fmt_in, int fmt_out) {
    fp16_saturate_on();
    const int C4 = C >> 2;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int c4 = idx % C4;
        size_t pix = idx / C4;
        int ox = pix % Wo;
        size_t t = pix / Wo;
        int oy = t % Ho;
        int n = t / Ho;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = 0; dy < k; ++dy) {
            int iy = oy * stride - pad + dy;
            for (int dx = 0; dx < k; ++dx) {
                int ix = ox * stride - pad + dx;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    load4(x + ((size_t)(n * H + iy) * W + ix) * ldx, c4 * 4, fmt_in, v);
                else if (!zero_br)
                    continue;
                m.x = fmaxf(m.x, v[0]); m.y = fmaxf(m.y, v[1]); m.z = fmaxf(m.z, v[2]); m.w = fmaxf(m.w, v[3]);
            }
        }
        const float o[4] = {m.x, m.y, m.z, m.w};
        store4(y + pix * ldy, c4 * 4, fmt_out, o);
    }
}

void launch_maxpool(const View &x, const View &y, int k, int stride, int pad, bool zero_pad_br, hipStream_t s) {
    if (x.c % 4 || x.ld % 4 || y.ld % 4) fail("maxpool: channels must be a multiple of 4");
    size_t total = y.pixels() * (x.c / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, s, x.p, y.p, x.n, x.h, x.w, x.c, x.ld, y.h, y.w,
                       y.ld, k, stride, pad, zero_pad_br ? 1 : 0, x.fmt, y.fmt);
    YDS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------ upsample
__global__ void upsample_kernel(const float *x, float *y, int N, int H, int W, int C, int ldx, int ldy, int s) {
    const int C4 = C >> 2, Ho = H * s, Wo = W * s;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int c4 = idx % C4;
        size_t pix = idx / C4;
        int ox = pix % Wo;
        size_t t = pix / Wo;
        int oy = t % Ho;
        int n = t / Ho;
        *reinterpret_cast<float4 *>(y + pix * ldy + c4 * 4) =
            *reinterpret_cast<const float4 *>(x + ((size_t)(n * H + oy / s) * W + ox / s) * ldx + c4 * 4);
    }
}

void launch_upsample(const View &x, const View &y, int stride, hipStream_t s) {
    if (x.c % 4 || x.ld % 4 || y.ld % 4) fail("upsample: channels must be a multiple of 4");
    if (x.fmt != y.fmt) fail("upsample: source and destination formats differ");
    const int slots = fmt_slots(x.fmt, x.c);                    // byte-wise copy: an F16 pixel holds c / 2 float slots
    if (slots % 4) fail("upsample: channels must fill whole 16-byte pieces");
    size_t total = y.pixels() * (slots / 4);
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for(total)), dim3(256), 0, s, x.p, y.p, x.n, x.h, x.w, slots, x.ld, y.ld, stride);
    YDS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------ copy / add
__global__ void copy_kernel(const float *x, float *y, size_t pixels, int C, int ldx, int ldy) {
    const int C4 = C >> 2;
    const size_t total = pixels * C4;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int c4 = idx % C4;
        size_t pix = idx / C4;
        *reinterpret_cast<float4 *>(y + pix * ldy + c4 * 4) = *reinterpret_cast<const float4 *>(x + pix * ldx + c4 * 4);
    }
}
__global__ void copy_scalar_kernel(const float *x, float *y, size_t pixels, int C, int ldx, int ldy) {
    const size_t total = pixels * C;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int c = idx % C;
        size_t pix = idx / C;
        y[pix * ldy + c] = x[pix * ldx + c];
    }
}

void launch_copy(const View &x, const View &y, hipStream_t s) {
    if (x.fmt != y.fmt) fail("copy: source and destination formats differ");
    const int slots = fmt_slots(x.fmt, x.c);                    // byte-wise copy: an F16 pixel holds c / 2 float slots
    bool vec = !(slots % 4 || x.ld % 4 || y.ld % 4 || ((uintptr_t)x.p & 15) || ((uintptr_t)y.p & 15));
    if (vec)
        hipLaunchKernelGGL(copy_kernel, dim3(grid_for(x.pixels() * (slots / 4))), dim3(256), 0, s, x.p, y.p, x.pixels(), slots, x.ld, y.ld);
    else
        hipLaunchKernelGGL(copy_scalar_kernel, dim3(grid_for(x.pixels() * slots)), dim3(256), 0, s, x.p, y.p, x.pixels(), slots, x.ld, y.ld);
    YDS_HIP(hipGetLastError());
}

__global__ void add_kernel(const float *a, const float *b, float *y, size_t pixels, int C, int lda, int ldb, int ldy, int fa, int fb, int fy) {
    fp16_saturate_on();
    const int C4 = C >> 2;
    const size_t total = pixels * C4;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int c4 = idx % C4;
        size_t pix = idx / C4;
        float u[4], v[4];
        load4(a + pix * lda, c4 * 4, fa, u);
        load4(b + pix * ldb, c4 * 4, fb, v);
        const float o[4] = {u[0] + v[0], u[1] + v[1], u[2] + v[2], u[3] + v[3]};
        store4(y + pix * ldy, c4 * 4, fy, o);
    }
}

void launch_add(const View &a, const View &b, const View &y, hipStream_t s) {
    if (a.c % 4 || a.ld % 4 || b.ld % 4 || y.ld % 4) fail("add: channels must be a multiple of 4");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(a.pixels() * (a.c / 4))), dim3(256), 0, s, a.p, b.p, y.p, a.pixels(), a.c, a.ld, b.ld, y.ld,
                       a.fmt, b.fmt, y.fmt);
    YDS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------ layout
__global__ void nchw_to_nhwc_kernel(const float *src, float *y, int N, int C, int H, int W, int ldy, int Cdst) {
    const size_t total = (size_t)N * H * W * Cdst;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int c = idx % Cdst;
        size_t pix = idx / Cdst;
        int xw = pix % W;
        size_t t = pix / W;
        int yh = t % H;
        int n = t / H;
        y[pix * ldy + c] = c < C ? src[((size_t)(n * C + c) * H + yh) * W + xw] : 0.f;
    }
}
void launch_nchw_to_nhwc(const float *src_nchw, const View &y, int c_src, hipStream_t s) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(y.pixels() * y.c)), dim3(256), 0, s, src_nchw, y.p, y.n, c_src, y.h, y.w, y.ld, y.c);
    YDS_HIP(hipGetLastError());
}

__global__ void nhwc_to_nchw_kernel(const float *x, float *dst, int N, int C, int H, int W, int ldx, int fmt) {
    const size_t total = (size_t)N * C * H * W;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int xw = idx % W;
        size_t t = idx / W;
        int yh = t % H;
        t /= H;
        int c = t % C;
        int n = t / C;
        const float *pixel = x + ((size_t)(n * H + yh) * W + xw) * ldx;
        if (fmt == FMT_H16 || fmt == FMT_F16) {
            float v[4];
            load4(pixel, c & ~3, fmt, v);
            dst[idx] = v[c & 3];
        } else {
            dst[idx] = pixel[c];
        }
    }
}
__global__ void pack_h16_kernel(const float *src, float *y, size_t pixels, int C, int ldy) {
    fp16_saturate_on();
    const int C4 = C >> 2;
    const size_t total = pixels * C4;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int c4 = idx % C4;
        size_t pix = idx / C4;
        float4 t = *reinterpret_cast<const float4 *>(src + pix * C + c4 * 4);
        const float v[4] = {t.x, t.y, t.z, t.w};
        h16_store4(y + pix * ldy, c4 * 4, v);
    }
}
void launch_pack_h16(const float *src_f32, const View &y, hipStream_t s) {
    if (y.c % 32 || y.ld % 32) fail("pack_h16: channels must be a multiple of 32");
    hipLaunchKernelGGL(pack_h16_kernel, dim3(grid_for(y.pixels() * (y.c / 4))), dim3(256), 0, s, src_f32, y.p, y.pixels(), y.c, y.ld);
    YDS_HIP(hipGetLastError());
}

void launch_nhwc_to_nchw(const View &x, float *dst_nchw, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(x.pixels() * x.c)), dim3(256), 0, s, x.p, dst_nchw, x.n, x.c, x.h, x.w, x.ld, x.fmt);
    YDS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------ yolo decode
struct YoloParams {
    float aw[8], ah[8];   // anchors / scale, fp32 like the reference's scaled_anchors
    float s0, s1;         // (img_h / H, img_w / W); quirk: x uses s0, y uses s1 (models.py:169-172,216)
};

// One wavefront per grid cell: the cell -> (row, column) arithmetic is done once per cell on the scalar side instead of three
// integer divisions per value (round 3: the per-value form was ALU bound, 175 us per 16-frame pass; the values themselves are
// computed by the same expressions).  The A x (5 + C) values of a cell are contiguous in the head tensor: lanes cover them 64 at
// a time (255 values = four full sweeps), every anchor's row of the output is written contiguously.
__global__ void yolo_decode_kernel(const float *head, float *out, int N, int H, int W, int ld, int A, int attrs, int total_boxes,
                                   int box_off, YoloParams yp) {
    const int HW = H * W, n = blockIdx.y, per_cell = A * attrs;
    (void)N;
    const int lane = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    for (int cell = wave0; cell < HW; cell += nwaves) {
        const int gy = cell / W, gx = cell - gy * W;
        const float *src = head + ((size_t)(n * H + gy) * W + gx) * ld;
        float *dst = out + ((size_t)n * total_boxes + box_off + cell) * attrs;
        for (int i = lane; i < per_cell; i += 64) {
            int a = 0;
#pragma unroll
            for (int k = 1; k < 8; ++k) a += (k < A && i >= k * attrs) ? 1 : 0;
            const int t = i - a * attrs;
            const float v = src[i];
            float r;
            if (t == 0) r = __fmul_rn(__fadd_rn(1.f / (1.f + expf(-v)), (float)gx), yp.s0);
            else if (t == 1) r = __fmul_rn(__fadd_rn(1.f / (1.f + expf(-v)), (float)gy), yp.s1);
            else if (t == 2) r = __fmul_rn(__fmul_rn(expf(v), yp.aw[a]), yp.s0);
            else if (t == 3) r = __fmul_rn(__fmul_rn(expf(v), yp.ah[a]), yp.s1);
            else r = 1.f / (1.f + expf(-v));
            dst[(size_t)a * HW * attrs + t] = r;                    // box = a * HW + cell
        }
    }
}

void launch_yolo_decode(const View &head, float *out, int total_boxes, int box_off, int num_classes, const float *anchors_wh,
                        int A, int img_h, int img_w, hipStream_t s) {
    if (A > 8) fail("yolo: at most 8 anchors per head");
    YoloParams yp;
    yp.s0 = (float)((double)img_h / head.h);
    yp.s1 = (float)((double)img_w / head.w);
    for (int a = 0; a < A; ++a) {
        yp.aw[a] = anchors_wh[2 * a] / yp.s0;
        yp.ah[a] = anchors_wh[2 * a + 1] / yp.s1;
    }
    int attrs = num_classes + 5;
    if (head.c != A * attrs) fail("yolo: head has %d channels, expected %d", head.c, A * attrs);
    const size_t per_img = (size_t)A * head.h * head.w * attrs;
    if (per_img >= (1ull << 31)) fail("yolo: head too large for 32-bit indexing");
    const unsigned cells = (unsigned)head.h * head.w;                   // four wavefronts per workgroup, a few cells per wavefront
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(std::min((cells + 3) / 4, 2048u), head.n), dim3(256), 0, s, head.p, out, head.n, head.h, head.w,
                       head.ld, A, attrs, total_boxes, box_off, yp);
    YDS_HIP(hipGetLastError());
}

// bench-only logit injection: objectness := -logit everywhere, then scripted cells
__global__ void inject_clear_kernel(float *head, int image, int H, int W, int ld, int A, int attrs, float logit) {
    int total = H * W * A;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int a = idx % A, cell = idx / A;
        head[((size_t)image * H * W + cell) * ld + a * attrs + 4] = -logit;
    }
}
// Two scripted boxes can land on the same (head, anchor, cell).  The rows are applied "in order, the last one wins" (what a
// sequential loop does, oracle/pipeline.py make_injector): a row is skipped when a later row of the table targets its cell,
// so that the parallel writes never race.
__device__ __forceinline__ bool inject_row_superseded(const float *rows, int r, int n) {
    const float *row = rows + (size_t)r * 9;
    for (int j = r + 1; j < n; ++j) {
        const float *o = rows + (size_t)j * 9;
        if (o[0] == row[0] && o[1] == row[1] && o[2] == row[2] && o[3] == row[3]) return true;
    }
    return false;
}
__global__ void inject_rows_kernel(float *head, int image, int H, int W, int ld, int attrs, const float *rows, int n, int head_index,
                                   float logit) {
    int r = blockIdx.x;
    if (r >= n) return;
    const float *row = rows + r * 9;
    if ((int)row[0] != head_index) return;
    if (inject_row_superseded(rows, r, n)) return;
    int a = (int)row[1], gy = (int)row[2], gx = (int)row[3], cls = (int)row[8];
    float *cell = head + ((size_t)(image * H + gy) * W + gx) * ld + a * attrs;
    for (int t = threadIdx.x; t < attrs; t += blockDim.x) {
        float v;
        if (t < 4) v = row[4 + t];
        else if (t == 4) v = logit;
        else v = (t - 5 == cls) ? logit : -logit;
        cell[t] = v;
    }
}
void launch_inject(const View &head, int image, const float *rows_dev, int n, int head_index, int num_classes, float logit,
                   hipStream_t s) {
    int attrs = num_classes + 5, A = head.c / attrs;
    hipLaunchKernelGGL(inject_clear_kernel, dim3(grid_for((size_t)head.h * head.w * A)), dim3(256), 0, s, head.p, image, head.h, head.w,
                       head.ld, A, attrs, logit);
    if (n > 0)
        hipLaunchKernelGGL(inject_rows_kernel, dim3(n), dim3(128), 0, s, head.p, image, head.h, head.w, head.ld, attrs, rows_dev, n,
                           head_index, logit);
    YDS_HIP(hipGetLastError());
}

// the same for a whole batch in two launches: image = blockIdx.y, its rows are table[offsets[image] .. offsets[image + 1])
__global__ void inject_clear_batch_kernel(float *head, int H, int W, int ld, int A, int attrs, float logit) {
    const int total = H * W * A, image = blockIdx.y;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int a = idx % A, cell = idx / A;
        head[((size_t)image * H * W + cell) * ld + a * attrs + 4] = -logit;
    }
}
__global__ void inject_rows_batch_kernel(float *head, int H, int W, int ld, int attrs, const float *table, const int *offsets, int head_index,
                                         float logit) {
    const int image = blockIdx.y, o0 = offsets[image], n = offsets[image + 1] - o0, r = blockIdx.x;
    if (r >= n) return;
    const float *row = table + (size_t)(o0 + r) * 9;
    if ((int)row[0] != head_index) return;
    if (inject_row_superseded(table + (size_t)o0 * 9, r, n)) return;
    int a = (int)row[1], gy = (int)row[2], gx = (int)row[3], cls = (int)row[8];
    float *cell = head + ((size_t)(image * H + gy) * W + gx) * ld + a * attrs;
    for (int t = threadIdx.x; t < attrs; t += blockDim.x) {
        float v;
        if (t < 4) v = row[4 + t];
        else if (t == 4) v = logit;
        else v = (t - 5 == cls) ? logit : -logit;
        cell[t] = v;
    }
}
void launch_inject_batch(const View &head, int batch, const float *table_dev, const int *offsets_dev, int max_rows, int head_index,
                         int num_classes, float logit, hipStream_t s) {
    int attrs = num_classes + 5, A = head.c / attrs;
    hipLaunchKernelGGL(inject_clear_batch_kernel, dim3(min(grid_for((size_t)head.h * head.w * A), 64), batch), dim3(256), 0, s, head.p, head.h,
                       head.w, head.ld, A, attrs, logit);
    if (max_rows > 0)
        hipLaunchKernelGGL(inject_rows_batch_kernel, dim3(max_rows, batch), dim3(128), 0, s, head.p, head.h, head.w, head.ld, attrs, table_dev,
                           offsets_dev, head_index, logit);
    YDS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------ bilinear u8
// cv2.resize(INTER_LINEAR) on 8-bit images, bit for bit (OpenCV modules/imgproc/src/resize.cpp; spec and citations in
// oracle/resize.py): sample position in double, one rounding to float, 11-bit fixed-point weights (round half to even),
// integer horizontal pass, vertical pass (((b0*(h0>>4))>>16) + ((b1*(h1>>4))>>16) + 2) >> 2; exact 2x down-scaling takes
// INTER_AREA's (a+b+c+d+2)>>2, equal sizes copy.
struct Tap { int i0, i1, a0, a1; };
__device__ __forceinline__ double cv_scale(int dst, int src) { return __ddiv_rn(1.0, __ddiv_rn((double)dst, (double)src)); }
template <bool X_AXIS> __device__ __forceinline__ Tap axis_tap(int d, double scale, int src) {
    float f = __double2float_rn(__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5));
    const float fl = floorf(f);
    int s = (int)fl;
    f = __fsub_rn(f, fl);
    if (X_AXIS) {                                               // the x axis zeroes the weight when it clamps ...
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
    }
    Tap t;
    t.a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    t.a1 = (int)rintf(__fmul_rn(f, 2048.f));
    t.i0 = min(max(s, 0), src - 1);                             // ... the y axis clips the two row indices and keeps its weights
    t.i1 = min(max(s + 1, 0), src - 1);
    return t;
}
enum { RESIZE_LINEAR = 0, RESIZE_AREA2 = 1, RESIZE_COPY = 2 };
__device__ __forceinline__ int resize_mode(int src_h, int src_w, int dst_h, int dst_w) {
    if (src_h == dst_h && src_w == dst_w) return RESIZE_COPY;
    if (src_h == 2 * dst_h && src_w == 2 * dst_w) return RESIZE_AREA2;
    return RESIZE_LINEAR;
}
// one output pixel (3 channels) of the region whose top-left source pixel is `base` (row stride `row_bytes`)
__device__ __forceinline__ void resize_px(const uint8_t *base, size_t row_bytes, int src_h, int src_w, int dst_h, int dst_w, int oy, int ox,
                                          float o[3]) {
    const int mode = resize_mode(src_h, src_w, dst_h, dst_w);
    if (mode == RESIZE_COPY) {
        const uint8_t *p = base + (size_t)oy * row_bytes + ox * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        return;
    }
    if (mode == RESIZE_AREA2) {
        const uint8_t *p = base + (size_t)(2 * oy) * row_bytes + 2 * ox * 3, *q = p + row_bytes;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (float)(((int)p[c] + (int)p[3 + c] + (int)q[c] + (int)q[3 + c] + 2) >> 2);
        return;
    }
    const Tap tx = axis_tap<true>(ox, cv_scale(dst_w, src_w), src_w), ty = axis_tap<false>(oy, cv_scale(dst_h, src_h), src_h);
    const uint8_t *r0 = base + (size_t)ty.i0 * row_bytes, *r1 = base + (size_t)ty.i1 * row_bytes;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = (int)r0[tx.i0 * 3 + c] * tx.a0 + (int)r0[tx.i1 * 3 + c] * tx.a1;
        const int h1 = (int)r1[tx.i0 * 3 + c] * tx.a0 + (int)r1[tx.i1 * 3 + c] * tx.a1;
        o[c] = (float)(((((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff);
    }
}

// bgr: the frames hold B, G, R bytes (a decoder's order): output channel c reads source byte 2 - c - the same integers as resizing the
// channel-reversed frame (video_detect.py:33-36 reverses on the host first), without the reversed copy
__global__ void resize_u8_kernel(const uint8_t *frames, int N, int H, int W, float *y, int Ho, int Wo, int ldy, int bgr) {
    const size_t total = (size_t)N * Ho * Wo;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int ox = idx % Wo;
        size_t t = idx / Wo;
        int oy = t % Ho;
        int n = t / Ho;
        float o[3];
        resize_px(frames + (size_t)n * H * W * 3, (size_t)W * 3, H, W, Ho, Wo, oy, ox, o);
        if (bgr) { const float t = o[0]; o[0] = o[2]; o[2] = t; }
        *reinterpret_cast<float4 *>(y + idx * ldy) = make_float4(__fdiv_rn(o[0], 255.f), __fdiv_rn(o[1], 255.f), __fdiv_rn(o[2], 255.f), 0.f);
    }
}

void launch_resize_u8(const uint8_t *frames, int n, int h, int w, const View &y, hipStream_t s, bool bgr) {
    if (y.c != 4 || y.ld != 4) fail("resize: destination must be NHWC4");
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid_for((size_t)n * y.h * y.w)), dim3(256), 0, s, frames, n, h, w, y.p, y.h, y.w, y.ld, bgr ? 1 : 0);
    YDS_HIP(hipGetLastError());
}

// Sliding-window front end (img_detect.py:103-111): window `b` of the frame (x, y, th, tw) is stretched to the model
// size into batch slot b, same bilinear arithmetic as resize_u8_kernel.
__global__ void tile_resize_kernel(const uint8_t *frame, int W, const int *tiles, int n_tiles, float *y, int Ho, int Wo) {
    const size_t total = (size_t)n_tiles * Ho * Wo;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int ox = idx % Wo;
        size_t t = idx / Wo;
        int oy = t % Ho;
        int n = t / Ho;
        const int x0 = tiles[n * 4 + 0], y0 = tiles[n * 4 + 1], th = tiles[n * 4 + 2], tw = tiles[n * 4 + 3];
        float o[3];
        resize_px(frame + ((size_t)y0 * W + x0) * 3, (size_t)W * 3, th, tw, Ho, Wo, oy, ox, o);
        *reinterpret_cast<float4 *>(y + idx * 4) = make_float4(__fdiv_rn(o[0], 255.f), __fdiv_rn(o[1], 255.f), __fdiv_rn(o[2], 255.f), 0.f);
    }
}

void launch_tile_resize(const uint8_t *frame, int w, const int *tiles_dev, int n_tiles, const View &y, hipStream_t s) {
    if (y.c != 4 || y.ld != 4) fail("resize: destination must be NHWC4");
    hipLaunchKernelGGL(tile_resize_kernel, dim3(grid_for((size_t)n_tiles * y.h * y.w)), dim3(256), 0, s, frame, w, tiles_dev, n_tiles, y.p, y.h, y.w);
    YDS_HIP(hipGetLastError());
}

// img_detect.py:131-139: centre form -> corner form (x -+ w/2), resize_boxes to the window's own size (python-double
// ratio rounded to fp32, passed in `scale`), shift by the window origin; the other attributes are copied.
__global__ void tile_boxes_kernel(const float *pred, int n_boxes, int attrs, const int *tiles, const float *scale, int n_tiles, float *dst) {
    const size_t total = (size_t)n_tiles * n_boxes;
    for (size_t row = blockIdx.x * (size_t)(blockDim.x / 32) + threadIdx.x / 32; row < total; row += (size_t)gridDim.x * (blockDim.x / 32)) {
        const int b = row / n_boxes, lane = threadIdx.x % 32;
        const float *p = pred + row * attrs;
        float *d = dst + row * attrs;
        for (int j = 4 + lane; j < attrs; j += 32) d[j] = p[j];
        if (lane < 4) {
            const float half = __fdiv_rn(p[2 + (lane & 1)], 2.f);
            float v = lane < 2 ? __fsub_rn(p[lane & 1], half) : __fadd_rn(p[lane & 1], half);
            v = __fmul_rn(v, scale[b * 2 + (lane & 1)]);
            d[lane] = __fadd_rn(v, (float)tiles[b * 4 + (lane & 1)]);
        }
    }
}

void launch_tile_boxes(const float *pred, int n_boxes, int attrs, const int *tiles_dev, const float *scale_dev, int n_tiles, float *dst, hipStream_t s) {
    hipLaunchKernelGGL(tile_boxes_kernel, dim3(grid_for((size_t)n_tiles * n_boxes * 32)), dim3(256), 0, s, pred, n_boxes, attrs, tiles_dev, scale_dev,
                       n_tiles, dst);
    YDS_HIP(hipGetLastError());
}

__global__ void crop_resize_kernel(const uint8_t *frames, int H, int W, const int *boxes, int D, float *y, int Ho, int Wo, int bgr) {
    const size_t total = (size_t)D * Ho * Wo;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int ox = idx % Wo;
        size_t t = idx / Wo;
        int oy = t % Ho;
        int d = t / Ho;
        int x1 = boxes[d * 5], y1 = boxes[d * 5 + 1], cw = boxes[d * 5 + 2] - x1, ch = boxes[d * 5 + 3] - y1;
        const uint8_t *frame = frames + (size_t)boxes[d * 5 + 4] * H * W * 3;
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        float o[3];
        resize_px(frame + ((size_t)y1 * W + x1) * 3, (size_t)W * 3, ch, cw, Ho, Wo, oy, ox, o);
        if (bgr) { const float t = o[0]; o[0] = o[2]; o[2] = t; }       // (BGR frames: see resize_u8_kernel)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = __fdiv_rn(__fsub_rn(__fdiv_rn(o[c], 255.f), mean[c]), stdv[c]);
        *reinterpret_cast<float4 *>(y + idx * 4) = make_float4(o[0], o[1], o[2], 0.f);
    }
}

void launch_crop_resize(const uint8_t *frame, int h, int w, const int *boxes_xyxy_dev, int D, const View &y, hipStream_t s, bool bgr) {   // boxes: [D,5]
    if (D == 0) return;
    hipLaunchKernelGGL(crop_resize_kernel, dim3(grid_for((size_t)D * y.h * y.w)), dim3(256), 0, s, frame, h, w, boxes_xyxy_dev, D, y.p,
                       y.h, y.w, bgr ? 1 : 0);
    YDS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------ ReID tail
__global__ void avgpool_l2norm_kernel(const float *x, float *out, int P, int C, int ld, int fmt) {
    // one workgroup per crop; C == blockDim.x * 2
    __shared__ float red[8];
    const int d = blockIdx.x;
    float v[2], ss = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int c = threadIdx.x + j * blockDim.x;
        float sum = 0.f;
        for (int p = 0; p < P; ++p) {
            const float *pixel = x + ((size_t)d * P + p) * ld;
            if (fmt == FMT_H16) {
                float q[4];
                h16_load4(pixel, c & ~3, q);
                sum += q[c & 3];
            } else {
                sum += pixel[c];
            }
        }
        v[j] = sum / (float)P;
        ss += v[j] * v[j];
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
    float nrm = sqrtf(tot);
#pragma unroll
    for (int j = 0; j < 2; ++j) out[(size_t)d * C + threadIdx.x + j * blockDim.x] = v[j] / nrm;
}

void launch_avgpool_l2norm(const View &x, float *out, hipStream_t s) {
    if (x.n == 0) return;
    if (x.c != 512) fail("avgpool: expected 512 channels");
    hipLaunchKernelGGL(avgpool_l2norm_kernel, dim3(x.n), dim3(256), 0, s, x.p, out, x.h * x.w, x.c, x.ld, x.fmt);
    YDS_HIP(hipGetLastError());
}

}  // namespace yds
