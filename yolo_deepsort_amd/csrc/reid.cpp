// DeepSORT appearance extractor on the HIP conv engine.
//
// Net(reid=True).forward, reference deep_sort/deep/model.py:48-95 (BasicBlock :5-37): stem
// conv3x3(3->64, bias)+BN+ReLU+MaxPool(3,2,1), four stages of two BasicBlocks (64,128,256,512;
// stride-2 first block with a 1x1 stride-2 conv+BN shortcut in stages 2-4), AvgPool(8,4), x/||x||.
// BN is folded into the conv at finalize(); ReLU and the residual add run in the conv epilogue, so a
// crop batch costs 20 conv launches + 1 maxpool + 1 pool/normalise launch.
// Crop boxes follow DeepSort._s_tlwh_to_xyxy (deep_sort/deep_sort.py:116-122): python int()
// truncation of fp32 sums, clipped to [0, W-1] / [0, H-1], end-exclusive slices.
#include "engine.h"
#include "h16.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace yds {

namespace {
struct StageDef { const char *name; int cin, cout; bool down; };
const StageDef kStages[4] = {{"layer1", 64, 64, false}, {"layer2", 64, 128, true}, {"layer3", 128, 256, true}, {"layer4", 256, 512, true}};
constexpr int CROP_H = 128, CROP_W = 64, EMB = 512;
}  // namespace

ReidNet::ReidNet(int max_crops) : max_crops(max_crops) {
    if (max_crops < 1) fail("reid: max_crops must be positive");
    stream = make_stream(false);
}
ReidNet::~ReidNet() {
    if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
    for (int *b : boxes_pin) if (b) (void)hipHostFree(b);
}

void ReidNet::load_tensor(const std::string &name, const float *data, const int64_t *shape, int ndim) {
    size_t n = 1;
    std::vector<int64_t> sh;
    for (int i = 0; i < ndim; ++i) { n *= (size_t)shape[i]; sh.push_back(shape[i]); }
    raw[name].assign(data, data + n);
    raw_shape[name] = sh;
    ready = false;
}

int64_t ReidNet::flops_per_crop() {
    int64_t f = 2ll * 128 * 64 * 64 * 27;                       // stem
    int h = 64, w = 32;
    for (const StageDef &s : kStages) {
        int ho = s.down ? h / 2 : h, wo = s.down ? w / 2 : w;
        f += 2ll * ho * wo * s.cout * 9 * s.cin;                // block0.conv1
        f += 2ll * ho * wo * s.cout * 9 * s.cout * 3;           // block0.conv2, block1.conv1, block1.conv2
        if (s.down) f += 2ll * ho * wo * s.cout * s.cin;        // 1x1 shortcut
        h = ho; w = wo;
    }
    return f;
}

void ReidNet::finalize() {
    auto need = [&](const std::string &k) -> const std::vector<float> & {
        auto it = raw.find(k);
        if (it == raw.end()) fail("reid: state dict lacks '%s'", k.c_str());
        return it->second;
    };
    convs.clear();
    auto add_conv = [&](const std::string &conv, const std::string &bn, int cin_file, int cout, int k, int stride, bool has_bias) {
        const auto &w = need(conv + ".weight");
        if (w.size() != (size_t)cout * cin_file * k * k) fail("reid: %s.weight has %zu elements, expected %d", conv.c_str(), w.size(), cout * cin_file * k * k);
        const auto &g = need(bn + ".weight"), &b = need(bn + ".bias"), &m = need(bn + ".running_mean"), &v = need(bn + ".running_var");
        const std::vector<float> *cb = has_bias ? &need(conv + ".bias") : nullptr;
        ConvW c;
        c.cin_file = cin_file;
        c.cin = (cin_file + 3) / 4 * 4;
        c.cout = cout; c.k = k; c.stride = stride; c.pad = k == 3 ? 1 : 0;
        c.kpad = (k * k * c.cin + 31) / 32 * 32;
        std::vector<float> packed((size_t)cout * c.kpad, 0.f), bias(cout);
        for (int o = 0; o < cout; ++o) {
            double scale = (double)g[o] / sqrt((double)v[o] + 1e-5);     // nn.BatchNorm2d default eps
            double b0 = cb ? (double)(*cb)[o] : 0.0;
            bias[o] = (float)((b0 - (double)m[o]) * scale + (double)b[o]);
            for (int ci = 0; ci < cin_file; ++ci)
                for (int kh = 0; kh < k; ++kh)
                    for (int kw = 0; kw < k; ++kw)
                        packed[(size_t)o * c.kpad + (kh * k + kw) * c.cin + ci] = (float)((double)w[(((size_t)o * cin_file + ci) * k + kh) * k + kw] * scale);
        }
        c.wt.upload(packed.data(), packed.size(), stream);
        {
            std::vector<uint16_t> split;
            pack_weights_f16x3(packed.data(), cout, c.kpad, split);
            c.wt16.upload(split.data(), split.size(), stream);
            YDS_HIP(hipStreamSynchronize(stream));
        }
        c.bias.upload(bias.data(), bias.size(), stream);
        YDS_HIP(hipStreamSynchronize(stream));
        convs.push_back(std::move(c));
    };
    add_conv("conv.0", "conv.1", 3, 64, 3, 1, true);
    for (const StageDef &s : kStages) {
        for (int b = 0; b < 2; ++b) {
            std::string p = std::string(s.name) + "." + std::to_string(b);
            int ci = b == 0 ? s.cin : s.cout;
            bool down = b == 0 && s.down;
            add_conv(p + ".conv1", p + ".bn1", ci, s.cout, 3, down ? 2 : 1, false);
            add_conv(p + ".conv2", p + ".bn2", s.cout, s.cout, 3, 1, false);
            if (down) add_conv(p + ".downsample.0", p + ".downsample.1", ci, s.cout, 1, 2, false);
        }
    }
    allocate_buffers();
    ready = true;
}

// activation buffers: input, stem, pooled, then (y1, out, [shortcut]) per block - everything sized by max_crops
void ReidNet::allocate_buffers() {
    in.alloc((size_t)max_crops * CROP_H * CROP_W * 4);
    feat.alloc((size_t)max_crops * EMB);
    bufs.clear();
    bufs.emplace_back((size_t)max_crops * CROP_H * CROP_W * 64);           // 0 stem
    bufs.emplace_back((size_t)max_crops * 64 * 32 * 64);                   // 1 pooled
    int h = 64, w = 32;
    for (const StageDef &s : kStages) {
        if (s.down) { h /= 2; w /= 2; }
        for (int k = 0; k < 5; ++k) bufs.emplace_back((size_t)max_crops * h * w * s.cout);   // y1a, outa, sc, y1b, outb
    }
}

// The reference's extractor takes any number of crops (feature_extractor.py:53-58); the buffers grow on demand.
// Nothing of an earlier pass may still be in flight on this stream's buffers when they are replaced.
void ReidNet::reserve(int D) {
    if (D <= max_crops) return;
    YDS_HIP(hipStreamSynchronize(stream));
    int m = max_crops;
    while (m < D) m *= 2;
    max_crops = m;
    if (ready) allocate_buffers();
    else in.alloc((size_t)max_crops * CROP_H * CROP_W * 4);
}

void ReidNet::forward(int D) {
    if (!ready) fail("reid: weights not loaded (yds_reid_finalize)");
    if (D < 1) fail("reid: no crops");
    reserve(D);
    const bool f16 = conv_math() == MATH_F16X3;
    auto mk = [&](DevBuf<float> &b, int h, int w, int c) {
        View v; v.p = b.p; v.n = D; v.h = h; v.w = w; v.c = c; v.ld = c;
        v.fmt = (f16 && c % 32 == 0) ? FMT_H16 : FMT_F32;      // activations between the convs stay pre-split
        return v;
    };
    conv_flops_last = 0;
    auto run = [&](int ci, const View &x, const View &y, int act, const View *res, int res_mode) {
        const ConvW &c = convs[ci];
        ConvArgs a;
        a.x = x; a.y = y; a.w = c.wt.p; a.w16 = c.wt16.p; a.bias = c.bias.p;
        a.ksize = c.k; a.stride = c.stride; a.pad = c.pad; a.kpad = c.kpad; a.act = act;
        if (res) { a.res = *res; a.res_mode = res_mode; }
        // measured tile choice per layer; the crop count varies from call to call, so a measurement is reused
        // while D stays within a factor of two of the D it was taken at
        if (tuned_math != conv_math()) { tuned.clear(); tuned_math = conv_math(); }
        auto it = tuned.find(ci);
        if (it == tuned.end() || D > 2 * it->second.first || 2 * D < it->second.first) {
            int v = getenv("YDS_NO_AUTOTUNE") ? -1 : conv_autotune(a, stream, nullptr);
            tuned[ci] = std::make_pair(D, v);
            it = tuned.find(ci);
        }
        (void)launch_conv(a, stream, it->second.second);
        conv_flops_last += conv_flops(a);
    };
    View x0 = mk(in, CROP_H, CROP_W, 4);
    View cur = mk(bufs[1], 64, 32, 64);
    {
        // stem conv + BN + ReLU + MaxPool2d(3, 2, 1) (model.py:52-60) as one kernel; the unfused pair stays as fallback
        const ConvW &c = convs[0];
        ConvArgs a;
        a.x = x0; a.y = cur; a.w = c.wt.p; a.w16 = c.wt16.p; a.bias = c.bias.p;
        a.ksize = c.k; a.stride = c.stride; a.pad = c.pad; a.kpad = c.kpad; a.act = ACT_RELU;
        if (!getenv("YDS_REID_UNFUSED") && launch_conv_maxpool3s2(a, stream)) {
            conv_flops_last += 2.0 * D * CROP_H * CROP_W * 64 * 9 * 4;
        } else {
            View stem = mk(bufs[0], CROP_H, CROP_W, 64);
            run(0, x0, stem, ACT_RELU, nullptr, RES_NONE);
            launch_maxpool(stem, cur, 3, 2, 1, false, stream);
        }
    }
    int ci = 1, bi = 2, h = 64, w = 32;
    for (const StageDef &s : kStages) {
        if (s.down) { h /= 2; w /= 2; }
        for (int b = 0; b < 2; ++b) {
            bool down = b == 0 && s.down;
            View y1 = mk(bufs[bi + (b == 0 ? 0 : 3)], h, w, s.cout);
            View out = mk(bufs[bi + (b == 0 ? 1 : 4)], h, w, s.cout);
            run(ci, cur, y1, ACT_RELU, nullptr, RES_NONE);                        // conv1+bn1+relu
            View res = cur;
            if (down) {
                res = mk(bufs[bi + 2], h, w, s.cout);
                run(ci + 2, cur, res, ACT_LINEAR, nullptr, RES_NONE);             // 1x1 s2 conv + bn
            }
            run(ci + 1, y1, out, ACT_RELU, &res, RES_BEFORE_ACT);                 // relu(x + bn2(conv2))
            ci += down ? 3 : 2;
            cur = out;
        }
        bi += 5;
    }
    launch_avgpool_l2norm(cur, feat.p, stream);
}

static void crop_boxes_host(const float *tlwh, int D, int H, int W, std::vector<int> &out, const int *frame_of = nullptr) {
    out.resize((size_t)D * 5);
    for (int d = 0; d < D; ++d) {
        float x = tlwh[d * 4], y = tlwh[d * 4 + 1], w = tlwh[d * 4 + 2], h = tlwh[d * 4 + 3];
        float xe = x + w, ye = y + h;                         // fp32 sums like the reference's tensor ops
        int x1 = (int)x > 0 ? (int)x : 0;
        int y1 = (int)y > 0 ? (int)y : 0;
        int x2 = (int)xe < W - 1 ? (int)xe : W - 1;
        int y2 = (int)ye < H - 1 ? (int)ye : H - 1;
        if (x2 <= x1 || y2 <= y1) fail("reid: detection %d yields an empty crop (%d,%d,%d,%d); cv2.resize raises in the reference", d, x1, y1, x2, y2);
        out[d * 5] = x1; out[d * 5 + 1] = y1; out[d * 5 + 2] = x2; out[d * 5 + 3] = y2; out[d * 5 + 4] = frame_of ? frame_of[d] : 0;
    }
}

void ReidNet::embed_dev(const uint8_t *frame_dev, int h, int w, const float *tlwh_host, int D, float *out_host) {
    if (D == 0) return;
    if (!ready) fail("reid: weights not loaded (yds_reid_finalize)");
    reserve(D);
    std::vector<int> boxes;
    crop_boxes_host(tlwh_host, D, h, w, boxes);
    boxes_dev.upload(boxes.data(), boxes.size(), stream);
    YDS_HIP(hipStreamSynchronize(stream));                      // `boxes` is a stack temporary
    View x0; x0.p = in.p; x0.n = D; x0.h = CROP_H; x0.w = CROP_W; x0.c = 4; x0.ld = 4;
    launch_crop_resize(frame_dev, h, w, boxes_dev.p, D, x0, stream);
    forward(D);
    if (out_host) YDS_HIP(hipMemcpyAsync(out_host, feat.p, (size_t)D * EMB * sizeof(float), hipMemcpyDeviceToHost, stream));
    // always complete before returning: the caller hands `feat` to the tracker, which runs on its own non-blocking
    // stream and would otherwise read features of a pass that is still in flight
    YDS_HIP(hipStreamSynchronize(stream));
}

void ReidNet::embed_multi_dev(const uint8_t *frames_dev, int h, int w, const float *tlwh_host, const int *frame_of, int D, bool bgr) {
    if (D == 0) return;
    if (!ready) fail("reid: weights not loaded (yds_reid_finalize)");
    reserve(D);
    crop_boxes_host(tlwh_host, D, h, w, boxes_host, frame_of);
    const int t = boxes_pin_turn ^= 1;
    if (boxes_pin_cap[t] < boxes_host.size()) {
        if (boxes_pin[t]) {
            // the previous tenants of this list may still be read by crop kernels: on this stream, and - when the pipeline moved
            // the pass to another stream since - on the one the earlier pass ran on
            YDS_HIP(hipStreamSynchronize(stream));
            if (sync_before_regrow) YDS_HIP(hipStreamSynchronize(sync_before_regrow));
            (void)hipHostFree(boxes_pin[t]);
            boxes_pin[t] = nullptr;
        }
        boxes_pin_cap[t] = std::max<size_t>(boxes_host.size() * 2, 4096);
        YDS_HIP(hipHostMalloc((void **)&boxes_pin[t], boxes_pin_cap[t] * sizeof(int), hipHostMallocDefault));
    }
    memcpy(boxes_pin[t], boxes_host.data(), boxes_host.size() * sizeof(int));
    View x0; x0.p = in.p; x0.n = D; x0.h = CROP_H; x0.w = CROP_W; x0.c = 4; x0.ld = 4;
    launch_crop_resize(frames_dev, h, w, boxes_pin[t], D, x0, stream, bgr);
    forward(D);
}

void ReidNet::embed_host(const uint8_t *frame_host, int h, int w, const float *tlwh_host, int D, float *out_host) {
    size_t n = (size_t)h * w * 3;
    stage_u8.ensure(n);
    YDS_HIP(hipMemcpyAsync(stage_u8.p, frame_host, n, hipMemcpyHostToDevice, stream));
    embed_dev(stage_u8.p, h, w, tlwh_host, D, out_host);
    YDS_HIP(hipStreamSynchronize(stream));
}

void ReidNet::preprocess_host(const uint8_t *frame_host, int h, int w, const float *tlwh_host, int D, float *nchw_host) {
    if (D == 0) return;
    reserve(D);
    if (!in.p) in.alloc((size_t)max_crops * CROP_H * CROP_W * 4);
    size_t n = (size_t)h * w * 3;
    stage_u8.ensure(n);
    YDS_HIP(hipMemcpyAsync(stage_u8.p, frame_host, n, hipMemcpyHostToDevice, stream));
    std::vector<int> boxes;
    crop_boxes_host(tlwh_host, D, h, w, boxes);
    boxes_dev.upload(boxes.data(), boxes.size(), stream);
    YDS_HIP(hipStreamSynchronize(stream));
    View x0; x0.p = in.p; x0.n = D; x0.h = CROP_H; x0.w = CROP_W; x0.c = 4; x0.ld = 4;
    launch_crop_resize(stage_u8.p, h, w, boxes_dev.p, D, x0, stream);
    View x3 = x0; x3.c = 3;
    DevBuf<float> tmp((size_t)D * 3 * CROP_H * CROP_W);
    launch_nhwc_to_nchw(x3, tmp.p, stream);
    YDS_HIP(hipMemcpyAsync(nchw_host, tmp.p, tmp.n * sizeof(float), hipMemcpyDeviceToHost, stream));
    YDS_HIP(hipStreamSynchronize(stream));
}

void ReidNet::forward_f32_host(const float *nchw, int D, float *out_host) {
    if (D == 0) return;
    if (!ready) fail("reid: weights not loaded (yds_reid_finalize)");
    reserve(D);
    size_t n = (size_t)D * 3 * CROP_H * CROP_W;
    stage_f32.ensure(n);
    YDS_HIP(hipMemcpyAsync(stage_f32.p, nchw, n * sizeof(float), hipMemcpyHostToDevice, stream));
    View x0; x0.p = in.p; x0.n = D; x0.h = CROP_H; x0.w = CROP_W; x0.c = 4; x0.ld = 4;
    launch_nchw_to_nhwc(stage_f32.p, x0, 3, stream);
    forward(D);
    YDS_HIP(hipMemcpyAsync(out_host, feat.p, (size_t)D * EMB * sizeof(float), hipMemcpyDeviceToHost, stream));
    YDS_HIP(hipStreamSynchronize(stream));
}

}  // namespace yds

// ============================================================================================ C ABI

extern "C" {

yds_reid *yds_reid_create(int max_crops) {
    YDS_API_BEGIN
    return new yds_reid{new yds::ReidNet(max_crops)};
    YDS_API_END_PTR
}
void yds_reid_destroy(yds_reid *r) {
    if (r) { delete r->r; delete r; }
}
int yds_reid_load_tensor(yds_reid *r, const char *name, const float *data_host, const int64_t *shape, int ndim) {
    YDS_API_BEGIN
    r->r->load_tensor(name, data_host, shape, ndim);
    YDS_API_END
}
int yds_reid_finalize(yds_reid *r) {
    YDS_API_BEGIN
    r->r->finalize();
    YDS_API_END
}
int64_t yds_reid_flops_per_crop(void) { return yds::ReidNet::flops_per_crop(); }
int yds_reid_embed(yds_reid *r, const uint8_t *frame, int h, int w, const float *tlwh, int D, float *out) {
    YDS_API_BEGIN
    r->r->embed_host(frame, h, w, tlwh, D, out);
    YDS_API_END
}
int yds_reid_embed_dev(yds_reid *r, const uint8_t *frame_dev, int h, int w, const float *tlwh, int D, float *out) {
    YDS_API_BEGIN
    r->r->embed_dev(frame_dev, h, w, tlwh, D, out);
    YDS_API_END
}
const float *yds_reid_features_dev(yds_reid *r) { return r->r->feat.p; }
int yds_reid_preprocess(yds_reid *r, const uint8_t *frame, int h, int w, const float *tlwh, int D, float *nchw) {
    YDS_API_BEGIN
    r->r->preprocess_host(frame, h, w, tlwh, D, nchw);
    YDS_API_END
}
int yds_reid_forward_f32(yds_reid *r, const float *nchw, int D, float *out) {
    YDS_API_BEGIN
    r->r->forward_f32_host(nchw, D, out);
    YDS_API_END
}

}  // extern "C"
