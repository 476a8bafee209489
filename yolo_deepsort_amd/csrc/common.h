// Shared declarations for libydsort (host side + kernel launchers).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <stdexcept>

namespace yds {

// ---- error plumbing ---------------------------------------------------------------------
void set_error(const std::string &msg);
struct Error : std::runtime_error {
    explicit Error(const std::string &m) : std::runtime_error(m) {}
};
[[noreturn]] void fail(const char *fmt, ...);
int bound_device();     // device yds_init bound this process to, -1 before
void bind_thread();     // selects that device for the calling host thread (hipSetDevice is per thread)
hipStream_t make_stream(bool latency_role);   // non-blocking stream; optional CU partition between conv and association work

#define YDS_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            ::yds::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// Runs `body`, converts C++ exceptions into the C ABI's status code.
#define YDS_API_BEGIN try { ::yds::bind_thread();
#define YDS_API_END                                                                            \
    }                                                                                          \
    catch (const std::exception &e) {                                                          \
        ::yds::set_error(e.what());                                                            \
        return -1;                                                                             \
    }                                                                                          \
    return 0;
#define YDS_API_END_PTR                                                                        \
    }                                                                                          \
    catch (const std::exception &e) {                                                          \
        ::yds::set_error(e.what());                                                            \
        return nullptr;                                                                        \
    }

// ---- device buffer ------------------------------------------------------------------------
template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        if (count == 0) return;
        YDS_HIP(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) { if (count > n) alloc(count); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    void upload(const T *src, size_t count, hipStream_t s = nullptr) {
        ensure(count);
        if (count) YDS_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
};

// ---- NHWC tensor view ----------------------------------------------------------------------
// fp32, pixel-major: element (n,y,x,c) at p[((n*h + y)*w + x)*ld + c].  ld >= c lets a layer write
// straight into a channel slice of a wider (concatenated) buffer.
struct View {
    float *p = nullptr;
    int n = 0, h = 0, w = 0, c = 0, ld = 0;
    int fmt = 0;             // FMT_F32, FMT_H16 or FMT_F16 (h16.h); H16 needs c, ld and the channel offset to be multiples of 32; F16 (half mode): multiples of 64 channels, ld and offsets in float slots = channels / 2
    size_t pixels() const { return (size_t)n * h * w; }
};

enum Act { ACT_LINEAR = 0, ACT_LEAKY = 1, ACT_MISH = 2, ACT_RELU = 3 };
enum ResMode { RES_NONE = 0, RES_AFTER_ACT = 1, RES_BEFORE_ACT = 2 };

// Implicit-GEMM convolution, fused bias (folded BN) + activation (+ residual).
// w: [cout][kpad] fp32, k index = (kh*ksize + kw)*cin + ci, rows zero padded to kpad (multiple of 32).
struct ConvArgs {
    View x, y;
    const float *w = nullptr, *bias = nullptr;
    const void *w16 = nullptr;   // same weights pre-split for the f16x3 kernel: [cout][kpad/32][32 hi | 32 lo] fp16
    View res;              // optional residual (same n,h,w,c as y)
    int ksize = 1, stride = 1, pad = 0, kpad = 0;
    int act = ACT_LINEAR, res_mode = RES_NONE;
    int terms = 3;         // f16x3 kernels: 3 = split-fp16 (fp32-class), 1 = single-term fp16 operands ("half" mode of a detector)
    // merged launch of two convolutions that read the same tensor (CSP split of yolov4): y.c = both filter counts, filters
    // [n_split, y.c) are written to y2 (same pixels, own stride / channel offset); w / w16 / bias hold both filter sets
    View y2;
    int n_split = 0;
    // order of the tiles inside an XCD's rectangle (conv_common.h tile_of_block): filter tiles walked together; 0 = one.  Placement only:
    // results do not depend on it.  launch_conv takes it from bits 8.. of its `variant` argument (what conv_autotune returns).
    int tile_gn = 0;
};
// returns the tile-variant id that was launched (see conv_variant_name)
int launch_conv(const ConvArgs &a, hipStream_t s, int variant = -1);   // variant < 0: built-in default choice; bits 8..: tile order (conv_autotune's packed result); returns the plain id
constexpr int kVariantMask = 0xff, kTileGnShift = 8;
int conv_default_variant(const ConvArgs &a);
bool conv_presplit_input(const ConvArgs &a);   // H16, or F16 in half mode: what the LDS-DMA / window kernels fetch as opaque chunks
// 3x3 RGB stem + MaxPool2d(3, 2, 1) in one kernel (ReID); a.y is the pooled view.  Returns false when the layer does not qualify.
bool launch_conv_maxpool3s2(const ConvArgs &a, hipStream_t s);
int conv_autotune(const ConvArgs &a, hipStream_t s, float *best_us);   // measured fastest variant
constexpr int kF32Variants = 7, kDirectVariant = 23, kConvVariants = 24;   // ids 0-6: fp32 MFMA tiles, 7-22: f16x3 tiles, 23: direct RGB 3x3
enum ConvMath { MATH_F32 = 0, MATH_F16X3 = 1 };
int conv_math();                 // process-wide arithmetic mode (env YDS_CONV_MATH=f32|f16x3, default f16x3)
void set_conv_math(int m);
void pack_weights_f16x3(const float *w, int cout, int kpad, std::vector<uint16_t> &out);
const char *conv_variant_name(int v);
double conv_flops(const ConvArgs &a);
// algorithmic HBM bytes of one conv launch: input read once, weights once, output written once, residual read once
double conv_bytes_io(const View &v);
double conv_bytes(const ConvArgs &a);

// ---- simple layers (layers.hip) --------------------------------------------------------------
void launch_maxpool(const View &x, const View &y, int k, int stride, int pad, bool zero_pad_br, hipStream_t s);
void launch_upsample(const View &x, const View &y, int stride, hipStream_t s);
void launch_copy(const View &x, const View &y, hipStream_t s);                 // y[..., :c] = x[..., :c]
void launch_add(const View &a, const View &b, const View &y, hipStream_t s);
void launch_nchw_to_nhwc(const float *src_nchw, const View &y, int c_src, hipStream_t s);   // pads channels with 0
void launch_nhwc_to_nchw(const View &x, float *dst_nchw, hipStream_t s);   // decodes H16 views
void launch_pack_h16(const float *src_f32, const View &y, hipStream_t s);   // fp32 NHWC (ld = c) -> H16 view (tests / tools)
// yolo decode: head NHWC [n,h,w,A*(5+C)] -> out[n, box_off + a*h*w + y*w + x, 5+C]
void launch_yolo_decode(const View &head, float *out, int total_boxes, int box_off, int num_classes,
                        const float *anchors_wh /*host, A pairs*/, int A, int img_h, int img_w, hipStream_t s);
void launch_inject(const View &head, int image, const float *rows_dev, int n, int head_index, int num_classes,
                   float logit, hipStream_t s);
void launch_inject_batch(const View &head, int batch, const float *table_dev, const int *offsets_dev, int max_rows, int head_index,
                         int num_classes, float logit, hipStream_t s);
// stretch-resize uint8 HWC frames to NHWC4 fp32 in [0,1] (4th channel 0)
void launch_resize_u8(const uint8_t *frames, int n, int h, int w, const View &y, hipStream_t s, bool bgr = false);   // bgr: frames in a decoder's B, G, R byte order
void launch_tile_resize(const uint8_t *frame, int w, const int *tiles_dev, int n_tiles, const View &y, hipStream_t s);
void launch_tile_boxes(const float *pred, int n_boxes, int attrs, const int *tiles_dev, const float *scale_dev, int n_tiles, float *dst,
                       hipStream_t s);
// ReID: crop + resize to 64x128 + /255 + mean/std -> NHWC4
// boxes: [D,5] = x1,y1,x2,y2,frame index (frames are h*w*3 bytes apart)
void launch_crop_resize(const uint8_t *frames, int h, int w, const int *boxes5_dev, int D, const View &y,
                        hipStream_t s, bool bgr = false);
void launch_avgpool_l2norm(const View &x, float *out, hipStream_t s);         // [D,8,4,512] -> [D,512]

}  // namespace yds
