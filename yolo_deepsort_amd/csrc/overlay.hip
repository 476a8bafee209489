// Output stage of the video generator on the device: the overlay of LabelDrawer.draw_labels_by_trackers - box outline,
// filled label plate above its top-left corner, black label text (reference yolo3/utils/label_draw.py:17-60,171-191) - the
// generator's RGB -> BGR conversion and its FPS text (yolo3/detect/video_detect.py:161-186), for a whole batch of frames that are
// already resident in HBM (they were uploaded for the detector).  The host form (yolo_deepsort_amd/label_draw.py, numpy) caps the
// generator at ~100 frames/s on 1080p frames (two 6 MB strided passes per frame); this form costs one read and one write of the
// frame on the device plus the D2H copy of the result.
//
// Pixel semantics are those of label_draw.py in this package (cv2.rectangle geometry, a 5 x 7 bitmap font at an integer pixel
// scale standing in for cv2's Hershey face), bit for bit: tests/test_gpu_video_detect.py compares the two.
// Order: boxes are drawn in list order, a later box over an earlier one, exactly like the sequential host loop - one workgroup
// owns a frame and separates the steps with barriers.
#include "common.h"

#include <mutex>
#include <vector>

#include "ydsort.h"

namespace yds {
namespace {

// out[i] = channel-reversed copy of frame src_slot[i]; four pixels = three dwords per thread when everything is dword aligned
__global__ __launch_bounds__(256) void overlay_copy_kernel(const uint8_t *frames, const int *src_slot, uint8_t *out, size_t frame_bytes, int aligned) {
    const int i = blockIdx.y;
    const uint8_t *src = frames + (size_t)src_slot[i] * frame_bytes;
    uint8_t *dst = out + (size_t)i * frame_bytes;
    if (aligned) {                                   // (the launcher sets it only when frame_bytes is a multiple of 12)
        const size_t quads = frame_bytes / 12;
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
        uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
        for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
            const uint32_t w0 = s4[3 * q], w1 = s4[3 * q + 1], w2 = s4[3 * q + 2];
            // bytes: w0 = c0 c1 c2 d0 | w1 = d1 d2 e0 e1 | w2 = e2 f0 f1 f2   (pixels c d e f, channel 0..2)
            const uint32_t o0 = ((w0 >> 16) & 0xff) | (w0 & 0xff00) | ((w0 & 0xff) << 16) | ((w1 & 0xff00) << 16);           // c2 c1 c0 d2
            const uint32_t o1 = (w1 & 0xff) | ((w0 >> 24) << 8) | ((w2 & 0xff) << 16) | (w1 & 0xff000000);                    // d1 d0 e2 e1
            const uint32_t o2 = ((w1 >> 16) & 0xff) | ((w2 >> 24) << 8) | (w2 & 0xff0000) | ((w2 & 0xff00) << 16);            // e0 f2 f1 f0
            d4[3 * q] = o0; d4[3 * q + 1] = o1; d4[3 * q + 2] = o2;
        }
    } else {
        const size_t pixels = frame_bytes / 3;
        for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < pixels; p += (size_t)gridDim.x * blockDim.x) {
            dst[3 * p] = src[3 * p + 2]; dst[3 * p + 1] = src[3 * p + 1]; dst[3 * p + 2] = src[3 * p];
        }
    }
}

// in-place channel swap of packed 3-byte pixels (BGR frames of a decoder -> the RGB the detector reads, video_detect.py:33-36)
__global__ __launch_bounds__(256) void swap_rb_kernel(uint8_t *frames, size_t pixels) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < pixels; p += (size_t)gridDim.x * blockDim.x) {
        const uint8_t a = frames[3 * p], c = frames[3 * p + 2];
        frames[3 * p] = c; frames[3 * p + 2] = a;
    }
}

struct DrawArgs {
    const int *boxes;        // [total][8]: x1, y1, x2, y2, colour (c0 | c1 << 8 | c2 << 16 in the RGB image's channel order), label offset, label length, 0
    const int *box_ptr;      // [n + 1]
    const uint8_t *text;     // glyph codes
    const int *fps;          // [n][2]: offset, length of the frame's FPS text (length 0: none)
    const uint8_t *font;     // [glyphs][7] rows, bit 4 = left column
    int H, W, thickness, scale;
};

// label_draw.py rectangle(), thickness < 0: rows [max(y1,0), min(y2+1,H)), columns [max(x1,0), min(x2+1,W))
__device__ __forceinline__ void fill(uint8_t *img, int W, int ya, int yb, int xa, int xb, uint8_t b0, uint8_t b1, uint8_t b2) {
    if (yb <= ya || xb <= xa) return;
    const int cols = xb - xa, total = (yb - ya) * cols;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        uint8_t *px = img + ((size_t)(ya + idx / cols) * W + xa + idx % cols) * 3;
        px[0] = b0; px[1] = b1; px[2] = b2;
    }
}
__device__ __forceinline__ void fill_clipped(uint8_t *img, int H, int W, int ya, int yb, int xa, int xb, uint8_t b0, uint8_t b1, uint8_t b2) {
    fill(img, W, max(ya, 0), min(yb, H), max(xa, 0), min(xb, W), b0, b1, b2);
}
// put_text(): bottom-left corner at (ox, oy), glyph i at x = ox + 6 * scale * i, rows oy - 7 * scale ..; set dots only
__device__ __forceinline__ void put_text(uint8_t *img, int H, int W, const uint8_t *codes, int len, const uint8_t *font, int ox, int oy, int scale,
                                         uint8_t b0, uint8_t b1, uint8_t b2) {
    const int gw = 5 * scale, gh = 7 * scale, per = gw * gh, y0 = oy - gh;
    for (int idx = threadIdx.x; idx < len * per; idx += blockDim.x) {
        const int ch = idx / per, r = idx % per, gy = r / gw, gx = r % gw;
        const int bit = (font[codes[ch] * 7 + gy / scale] >> (4 - gx / scale)) & 1;
        const int y = y0 + gy, x = ox + 6 * scale * ch + gx;
        if (bit && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            uint8_t *px = img + ((size_t)y * W + x) * 3;
            px[0] = b0; px[1] = b1; px[2] = b2;
        }
    }
}

// slots != NULL: draw IN PLACE on frame slots[i] of `out` (frames that are already in the result's channel order)
__global__ __launch_bounds__(256) void overlay_draw_kernel(uint8_t *out, size_t frame_bytes, DrawArgs a, const int *slots) {
    const int i = blockIdx.x, H = a.H, W = a.W;
    uint8_t *img = out + (size_t)(slots ? slots[i] : i) * frame_bytes;
    const int lo = a.thickness / 2, hi = (a.thickness - 1) / 2, fh = 7 * a.scale;
    for (int k = a.box_ptr[i]; k < a.box_ptr[i + 1]; ++k) {
        const int *b = a.boxes + (size_t)k * 8;
        const int c1x = b[0], c1y = b[1], c2x = b[2], c2y = b[3], col = b[4], loff = b[5], llen = b[6];
        // the image is written channel-reversed: a colour (c0, c1, c2) of the RGB image lands as (c2, c1, c0)
        const uint8_t p0 = (col >> 16) & 0xff, p1 = (col >> 8) & 0xff, p2 = col & 0xff;
        const int x1 = min(c1x, c2x), x2 = max(c1x, c2x), y1 = min(c1y, c2y), y2 = max(c1y, c2y);
        if (a.thickness < 0) {
            fill(img, W, max(y1, 0), max(min(y2 + 1, H), 0), max(x1, 0), max(min(x2 + 1, W), 0), p0, p1, p2);
        } else {                                     // outline centred on the edges (cv2.rectangle)
            fill_clipped(img, H, W, y1 - lo, y1 + hi + 1, x1 - lo, x2 + hi + 1, p0, p1, p2);
            fill_clipped(img, H, W, y2 - lo, y2 + hi + 1, x1 - lo, x2 + hi + 1, p0, p1, p2);
            fill_clipped(img, H, W, y1 - lo, y2 + hi + 1, x1 - lo, x1 + hi + 1, p0, p1, p2);
            fill_clipped(img, H, W, y1 - lo, y2 + hi + 1, x2 - lo, x2 + hi + 1, p0, p1, p2);
        }
        if (llen >= 0) {                             // only_rect = False: plate + text (label_draw.py draw_rects_and_labels)
            const int fw = 6 * llen * a.scale;
            const int pya = max(0, c1y - 3 - fh), pyb = max(c1y, 3 + fh), pxa = c1x, pxb = c1x + fw;
            const int qy1 = min(pya, pyb), qy2 = max(pya, pyb), qx1 = min(pxa, pxb), qx2 = max(pxa, pxb);
            fill(img, W, max(qy1, 0), max(min(qy2 + 1, H), 0), max(qx1, 0), max(min(qx2 + 1, W), 0), p0, p1, p2);
            __syncthreads();                         // the text goes over the plate (and over this box's outline)
            put_text(img, H, W, a.text + loff, llen, a.font, c1x, max(c1y - 3, fh), a.scale, 0, 0, 0);
        }
        __syncthreads();                             // the next box goes over this one
    }
    if (a.fps[2 * i + 1] > 0)                        // video_detect.py:183-186: FPS text on the BGR result, colour (255, 0, 0) as given
        put_text(img, H, W, a.text + a.fps[2 * i], a.fps[2 * i + 1], a.font, 3, 15, 2, 255, 0, 0);
}

// Two host threads use this file concurrently in the generator (detect.py): the reader thread swaps the channels of the batch it has
// just uploaded, the consumer thread renders the previous one.  Each gets its own stream and its own scratch, created once
// (function-local statics: initialisation is thread safe) - a swap never queues behind a 200 MB result copy.
struct OverlayState {
    std::mutex mu;                       // one render at a time: the scratch below is shared by every caller of this process
    hipStream_t stream = nullptr;
    DevBuf<int> boxes, box_ptr, fps, slots;
    DevBuf<uint8_t> text, font;
    OverlayState() { YDS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); }
};
OverlayState &state() {
    static OverlayState s;
    return s;
}
struct SwapState {
    hipStream_t stream = nullptr;
    SwapState() { YDS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); }
};
SwapState &swap_state() {
    static SwapState s;
    return s;
}

}  // namespace
}  // namespace yds

extern "C" {

int yds_swap_rb(uint8_t *frames_dev, size_t pixels) {
    YDS_API_BEGIN
    if (!frames_dev) yds::fail("swap_rb: NULL frames");
    yds::SwapState &s = yds::swap_state();
    if (pixels) {
        const unsigned blocks = (unsigned)std::min<size_t>((pixels + 255) / 256, 65535u * 4);
        yds::swap_rb_kernel<<<blocks, 256, 0, s.stream>>>(frames_dev, pixels);
        YDS_HIP(hipGetLastError());
        YDS_HIP(hipStreamSynchronize(s.stream));
    }
    YDS_API_END
}

namespace {
// validation + upload of the draw tables shared by both entries; returns the kernel's argument block
yds::DrawArgs overlay_tables(yds::OverlayState &s, const int32_t *src_slot_host, int n_src, int n_out, int h, int w, const int32_t *boxes_host,
                             const int32_t *box_ptr_host, const uint8_t *text_host, int n_text, const int32_t *fps_host, const uint8_t *font_host,
                             int n_glyphs, int thickness, int scale) {
    if (!src_slot_host || !box_ptr_host || !fps_host || !font_host) yds::fail("overlay: NULL argument");
    if (h < 1 || w < 1 || scale < 1 || n_glyphs < 1) yds::fail("overlay: bad geometry (h %d, w %d, scale %d, glyphs %d)", h, w, scale, n_glyphs);
    const int total = box_ptr_host[n_out];
    if (total < 0 || (total > 0 && !boxes_host)) yds::fail("overlay: bad box table");
    for (int k = 0; k < total; ++k) {
        const int32_t *b = boxes_host + (size_t)k * 8;
        if (b[6] > 0 && (b[5] < 0 || b[5] + b[6] > n_text)) yds::fail("overlay: label of box %d outside the text buffer", k);
    }
    for (int i = 0; i < n_out; ++i) {
        if (fps_host[2 * i + 1] > 0 && (fps_host[2 * i] < 0 || fps_host[2 * i] + fps_host[2 * i + 1] > n_text)) yds::fail("overlay: FPS text of frame %d outside the text buffer", i);
        if (src_slot_host[i] < 0 || (n_src > 0 && src_slot_host[i] >= n_src)) yds::fail("overlay: frame %d reads slot %d of %d staged frames", i, src_slot_host[i], n_src);
    }
    for (int k = 0; k < n_text; ++k)
        if (text_host[k] >= n_glyphs) yds::fail("overlay: glyph code %d outside the font (%d glyphs)", (int)text_host[k], n_glyphs);
    s.boxes.upload(boxes_host, (size_t)std::max(total, 1) * 8 * (total > 0), s.stream);
    s.box_ptr.upload(box_ptr_host, (size_t)n_out + 1, s.stream);
    s.fps.upload(fps_host, (size_t)n_out * 2, s.stream);
    s.slots.upload(src_slot_host, (size_t)n_out, s.stream);
    s.text.upload(text_host, (size_t)n_text, s.stream);
    s.font.upload(font_host, (size_t)n_glyphs * 7, s.stream);
    return yds::DrawArgs{s.boxes.p, s.box_ptr.p, s.text.p, s.fps.p, s.font.p, h, w, thickness, scale};
}
}  // namespace

int yds_overlay_tracks(const uint8_t *frames_dev, const int32_t *src_slot_host, int n_out, int h, int w, const int32_t *boxes_host,
                       const int32_t *box_ptr_host, const uint8_t *text_host, int n_text, const int32_t *fps_host, const uint8_t *font_host,
                       int n_glyphs, int thickness, int scale, uint8_t *out_dev, uint8_t *out_host) {
    YDS_API_BEGIN
    if (n_out <= 0) return 0;
    if (!frames_dev || !out_dev) yds::fail("overlay: NULL argument");
    yds::OverlayState &s = yds::state();
    std::lock_guard<std::mutex> lock(s.mu);                      // held until the stream has drained: the scratch is shared
    const yds::DrawArgs a = overlay_tables(s, src_slot_host, 0, n_out, h, w, boxes_host, box_ptr_host, text_host, n_text, fps_host, font_host, n_glyphs, thickness, scale);
    const size_t frame_bytes = (size_t)h * w * 3;
    const int aligned = frame_bytes % 12 == 0 && ((uintptr_t)frames_dev % 4) == 0 && ((uintptr_t)out_dev % 4) == 0;
    const unsigned bx = (unsigned)std::min<size_t>((frame_bytes / 12 + 255) / 256 + 1, 2048);
    yds::overlay_copy_kernel<<<dim3(bx, n_out), 256, 0, s.stream>>>(frames_dev, s.slots.p, out_dev, frame_bytes, aligned);
    YDS_HIP(hipGetLastError());
    yds::overlay_draw_kernel<<<n_out, 256, 0, s.stream>>>(out_dev, frame_bytes, a, nullptr);
    YDS_HIP(hipGetLastError());
    if (out_host) YDS_HIP(hipMemcpyAsync(out_host, out_dev, frame_bytes * n_out, hipMemcpyDeviceToHost, s.stream));
    YDS_HIP(hipStreamSynchronize(s.stream));
    YDS_API_END
}

int yds_overlay_tracks_bgr(uint8_t *frames_bgr_dev, int n_src, const int32_t *src_slot_host, int n_out, int h, int w, const int32_t *boxes_host,
                           const int32_t *box_ptr_host, const uint8_t *text_host, int n_text, const int32_t *fps_host, const uint8_t *font_host,
                           int n_glyphs, int thickness, int scale, uint8_t *out_host) {
    YDS_API_BEGIN
    if (n_out <= 0) return 0;
    if (!frames_bgr_dev || !out_host || n_src < 1) yds::fail("overlay: NULL argument");
    yds::OverlayState &s = yds::state();
    std::lock_guard<std::mutex> lock(s.mu);
    const yds::DrawArgs a = overlay_tables(s, src_slot_host, n_src, n_out, h, w, boxes_host, box_ptr_host, text_host, n_text, fps_host, font_host, n_glyphs, thickness, scale);
    for (int i = 0; i < n_out; ++i)
        for (int j = 0; j < i; ++j)
            if (src_slot_host[i] == src_slot_host[j]) yds::fail("overlay: frames %d and %d would be drawn on the same staged slot %d", j, i, src_slot_host[i]);
    const size_t frame_bytes = (size_t)h * w * 3;
    yds::overlay_draw_kernel<<<n_out, 256, 0, s.stream>>>(frames_bgr_dev, frame_bytes, a, s.slots.p);
    YDS_HIP(hipGetLastError());
    // results leave in the caller's frame order: runs of consecutive slots go out as one copy (all of them when every frame was processed)
    for (int i = 0; i < n_out;) {
        int j = i + 1;
        while (j < n_out && src_slot_host[j] == src_slot_host[j - 1] + 1) ++j;
        YDS_HIP(hipMemcpyAsync(out_host + (size_t)i * frame_bytes, frames_bgr_dev + (size_t)src_slot_host[i] * frame_bytes, frame_bytes * (size_t)(j - i),
                               hipMemcpyDeviceToHost, s.stream));
        i = j;
    }
    YDS_HIP(hipStreamSynchronize(s.stream));
    YDS_API_END
}

}  // extern "C"
