// Darknet detector engine: cfg -> static execution plan over NHWC buffers -> HIP kernels.
//
// Mirrors reference yolo3/models/models.py: create_modules :25-102 (layer semantics),
// Darknet.forward :292-313 (graph walk), load_darknet_weights :315-366 (file layout) and
// yolo3/utils/parse_config.py:1-19 (cfg syntax), but is planned once at create time:
//   * every conv is one fused implicit-GEMM launch (BN folded into weights/bias at load time,
//     activation and the following shortcut add in the epilogue);
//   * single-source routes and grouped routes are views (no copy); multi-source routes make their
//     producers write straight into channel slices of the concatenated buffer when possible;
//   * every layer keeps its own buffer for batch_max images (288 GB of HBM: no reuse games).
#include "engine.h"
#include "h16.h"
#include "conv_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <sstream>

namespace yds {

// ------------------------------------------------------------------------------------------ cfg
std::vector<CfgBlock> parse_cfg(const std::string &text) {
    // parse_config.py:1-19: drop empty and '#' lines, strip, '[type]' opens a block, key=value otherwise
    std::vector<CfgBlock> blocks;
    std::istringstream in(text);
    std::string line;
    auto strip = [](std::string s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    };
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        line = strip(line);
        if (line.empty()) fail("cfg: whitespace-only line (the reference parser rejects it too)");
        if (line[0] == '[') {
            CfgBlock b;
            b.type = strip(line.substr(1, line.size() - 2));
            if (b.type == "convolutional") b.kv["batch_normalize"] = "0";
            blocks.push_back(b);
        } else {
            size_t eq = line.find('=');
            if (eq == std::string::npos || line.find('=', eq + 1) != std::string::npos) fail("cfg: bad line '%s'", line.c_str());
            if (blocks.empty()) fail("cfg: key before first section");
            blocks.back().kv[strip(line.substr(0, eq))] = strip(line.substr(eq + 1));
        }
    }
    return blocks;
}

static int geti(const CfgBlock &b, const char *k) {
    auto it = b.kv.find(k);
    if (it == b.kv.end()) fail("cfg: [%s] lacks '%s'", b.type.c_str(), k);
    return atoi(it->second.c_str());
}
static std::vector<int> get_list(const CfgBlock &b, const char *k) {
    auto it = b.kv.find(k);
    if (it == b.kv.end()) fail("cfg: [%s] lacks '%s'", b.type.c_str(), k);
    std::vector<int> v;
    std::stringstream ss(it->second);
    std::string tok;
    while (std::getline(ss, tok, ',')) v.push_back(atoi(tok.c_str()));
    return v;
}

// ------------------------------------------------------------------------------------------ plan
Darknet::Darknet(const std::string &cfg_text, int img_h, int img_w, int batch_max) : img_h(img_h), img_w(img_w), batch_max(batch_max) {
    stream = make_stream(false);
    auto blocks = parse_cfg(cfg_text);
    if (blocks.empty() || blocks[0].type != "net") fail("cfg: first section must be [net]");
    in_channels = geti(blocks[0], "channels");
    if (in_channels > 4) fail("cfg: more than 4 input channels is not supported");
    blocks.erase(blocks.begin());
    const int L = (int)blocks.size();
    layers.resize(L);
    auto resolve = [&](int ref, int i) {
        int j = ref < 0 ? i + ref : ref;
        if (j < 0 || j >= i) fail("cfg: layer %d references layer %d", i, ref);
        return j;
    };
    // ---- pass 1: shapes, operators, references
    int pc = 4, ph = img_h, pw = img_w;   // previous layer's logical shape (input is channel-padded to 4)
    int prev_c_logical = in_channels;
    (void)prev_c_logical;
    for (int i = 0; i < L; ++i) {
        Layer &l = layers[i];
        const CfgBlock &b = blocks[i];
        l.type = b.type;
        l.root = i;
        if (b.type == "convolutional") {
            l.bn = geti(b, "batch_normalize");
            l.ksize = geti(b, "size");
            l.stride = geti(b, "stride");
            l.pad = (l.ksize - 1) / 2;                       // models.py:39, cfg 'pad' ignored
            l.cin = pc;
            l.cin_file = i == 0 ? in_channels : pc;
            l.c = geti(b, "filters");
            l.h = (ph + 2 * l.pad - l.ksize) / l.stride + 1;
            l.w = (pw + 2 * l.pad - l.ksize) / l.stride + 1;
            std::string act = b.kv.count("activation") ? b.kv.at("activation") : "linear";
            l.act = act == "leaky" ? ACT_LEAKY : act == "mish" ? ACT_MISH : ACT_LINEAR;   // models.py:53-56
            l.src = i - 1;
        } else if (b.type == "maxpool") {
            l.ksize = geti(b, "size");
            l.stride = geti(b, "stride");
            l.c = pc;
            if (l.ksize == 2 && l.stride == 1) {             // models.py:61-63 ZeroPad2d((0,1,0,1)) + MaxPool2d(2,1,pad 0)
                l.zero_br = true;
                l.pad = 0;
                l.h = ph;
                l.w = pw;
            } else {
                l.pad = (l.ksize - 1) / 2;
                l.h = (ph + 2 * l.pad - l.ksize) / l.stride + 1;
                l.w = (pw + 2 * l.pad - l.ksize) / l.stride + 1;
            }
            l.src = i - 1;
        } else if (b.type == "upsample") {
            l.stride = geti(b, "stride");
            l.c = pc;
            l.h = ph * l.stride;
            l.w = pw * l.stride;
            l.src = i - 1;
        } else if (b.type == "route") {
            for (int r : get_list(b, "layers")) l.refs.push_back(resolve(r, i));
            l.c = 0;
            for (int j : l.refs) {
                l.c += layers[j].c;
                if (layers[j].h != layers[l.refs[0]].h || layers[j].w != layers[l.refs[0]].w) fail("cfg: route %d joins different sizes", i);
            }
            l.h = layers[l.refs[0]].h;
            l.w = layers[l.refs[0]].w;
            if (b.kv.count("groups")) {
                l.groups = geti(b, "groups");
                l.group_id = geti(b, "group_id");
                if (l.c % l.groups) fail("cfg: route %d groups do not divide channels", i);
                l.c /= l.groups;
            }
        } else if (b.type == "shortcut") {
            l.refs = {i - 1, resolve(geti(b, "from"), i)};
            l.c = layers[l.refs[1]].c;
            l.h = ph;
            l.w = pw;
            if (layers[l.refs[0]].c != l.c || layers[l.refs[1]].h != ph) fail("cfg: shortcut %d shape mismatch", i);
        } else if (b.type == "yolo") {
            auto mask = get_list(b, "mask");
            auto an = get_list(b, "anchors");
            for (int m : mask) {
                if (2 * m + 1 >= (int)an.size()) fail("cfg: yolo %d mask out of range", i);
                l.anchors.push_back((float)an[2 * m]);
                l.anchors.push_back((float)an[2 * m + 1]);
            }
            l.classes = geti(b, "classes");
            l.c = pc; l.h = ph; l.w = pw;
            l.src = i - 1;
            if (pc != (int)mask.size() * (l.classes + 5)) fail("cfg: yolo %d expects %d input channels, has %d", i, (int)mask.size() * (l.classes + 5), pc);
            if (attrs == 0) attrs = l.classes + 5;
            if (attrs != l.classes + 5) fail("cfg: yolo layers disagree on classes");
            l.box_off = total_boxes;
            total_boxes += (int)mask.size() * ph * pw;
            yolo_layers.push_back(i);
        } else {
            fail("cfg: unsupported section [%s]", b.type.c_str());
        }
        pc = l.c; ph = l.h; pw = l.w;
    }
    if (yolo_layers.empty()) fail("cfg: no [yolo] layer");

    // ---- pass 2: who reads whom (a conv can absorb the following shortcut only if nobody else
    //      needs its pre-add output)
    std::vector<int> readers(L, 0);
    for (int i = 0; i < L; ++i) {
        const Layer &l = layers[i];
        if (l.src >= 0) readers[l.src]++;
        for (int j : l.refs) readers[j]++;
    }
    for (int i = 1; i < L; ++i) {
        Layer &l = layers[i];
        if (l.type == "shortcut" && layers[i - 1].type == "convolutional" && readers[i - 1] == 1 && l.refs[1] != i - 1) {
            layers[i - 1].fused_res = l.refs[1];
            l.fused = true;
            l.root = i - 1;          // view of the conv's (post-add) buffer
            l.coff = 0;
        }
    }
    // the first two convolutions can run as one kernel (conv_stem2.hip) when only layer 1 reads layer 0
    stem_fusable = L >= 2 && layers[0].type == "convolutional" && layers[1].type == "convolutional" && readers[0] == 1 &&
                   layers[1].src == 0 && layers[0].fused_res < 0 && layers[1].fused_res < 0;
    // ... and so can the first residual block (conv_block1.hip): conv a (read only by conv b), conv b absorbing the
    // shortcut back to conv a's input
    for (int i = 1; i + 1 < L && block1_at < 0; ++i)
        if (layers[i].type == "convolutional" && layers[i + 1].type == "convolutional" && readers[i] == 1 && layers[i + 1].src == i &&
            layers[i].fused_res < 0 && layers[i + 1].fused_res >= 0 && layers[i + 1].fused_res == layers[i].src && layers[i].ksize == 1 &&
            layers[i + 1].ksize == 3)
            block1_at = i;
    // single-source routes are views
    for (int i = 0; i < L; ++i) {
        Layer &l = layers[i];
        if (l.type == "route" && l.refs.size() == 1) {
            const Layer &s = layers[l.refs[0]];
            l.root = s.root;
            l.coff = s.coff + (l.groups ? l.group_id * l.c : 0);
            l.is_view = true;
        }
    }
    // ---- pass 3: storage.  Producers own [batch_max, h, w, ld] buffers; a multi-source route owns the
    //      concatenated buffer and redirects producers into its channel slices when they are free.
    storage.resize(L);
    auto is_producer = [&](int j) {
        const Layer &s = layers[j];
        return s.root == j && (s.type == "convolutional" || s.type == "maxpool" || s.type == "upsample" || (s.type == "shortcut" && !s.fused));
    };
    for (int i = 0; i < L; ++i) {
        Layer &l = layers[i];
        if (!(l.type == "route" && l.refs.size() > 1)) continue;
        int ctot = 0;
        for (int j : l.refs) ctot += layers[j].c;
        if (ctot % 4) continue;                               // keep float4 alignment; falls back to a copy
        storage[i].ld = ctot;
        int off = 0;
        for (int j : l.refs) {
            const Layer &s = layers[j];
            int r = s.root;
            // redirect only whole-buffer sources that nobody redirected before
            bool whole = r >= 0 && is_producer(r) && s.coff == 0 && s.c == layers[r].c && !storage[r].redirected && off % 4 == 0;
            if (whole) {
                storage[r].redirected = true;
                storage[r].into = i;
                storage[r].coff = off;
            } else {
                l.copies.push_back({j, off});
            }
            off += s.c;
        }
        if (l.groups) fail("cfg: grouped multi-source routes are not supported");
    }
    // ---- pass 4: tensor formats.  In f16x3 mode a buffer is kept pre-split (H16, h16.h) when everything that
    //      lives in it sits at 32-channel granularity; the image and the 255-channel heads stay fp32.
    math = conv_math();
    std::vector<char> h16_ok(L, math == MATH_F16X3 ? 1 : 0);
    for (int j = 0; j < L; ++j) {
        if (layers[j].type == "yolo") continue;
        int off = 0, o = owner_of(j, off);
        if (off % 32 || layers[j].c % 32) h16_ok[o] = 0;
    }
    // a concatenation fed by copies must share the copied tensors' format: demote both sides to fp32 on mismatch
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < L; ++i)
            for (auto &cp : layers[i].copies) {
                int off = 0, o = owner_of(cp.first, off);
                if (o >= 0 && h16_ok[o] != h16_ok[i]) h16_ok[o] = h16_ok[i] = 0;
            }
    for (int i = 0; i < L; ++i) {
        Layer &l = layers[i];
        bool owns = (is_producer(i) && !storage[i].redirected) || (l.type == "route" && l.refs.size() > 1);
        if (!owns) continue;
        if (storage[i].ld == 0) storage[i].ld = (l.c + 3) / 4 * 4;
        storage[i].fmt = (h16_ok[i] && storage[i].ld % 32 == 0) ? FMT_H16 : FMT_F32;
        storage[i].owns = true;
    }
    // ---- pass 5: CSP splits.  conv a (1x1) / route to a's input / conv b (1x1): both read the same tensor; one launch
    //      with the filters of both (a's first) writes a's channels to a's view and b's to b's view.
    const bool no_merge = getenv("YDS_NO_CSP_MERGE") != nullptr;     // (read at plan time: tests build both forms in one process)
    for (int i = 1; i + 2 < L && !no_merge; ++i) {
        Layer &a = layers[i];
        const Layer &r = layers[i + 1];
        Layer &b = layers[i + 2];
        if (a.type != "convolutional" || b.type != "convolutional" || r.type != "route" || r.refs.size() != 1 || r.groups) continue;
        if (a.src < 0 || r.refs[0] != a.src || b.src != i + 1) continue;
        if (a.ksize != 1 || b.ksize != 1 || a.stride != 1 || b.stride != 1 || a.act != b.act || a.fused_res >= 0 || b.fused_res >= 0) continue;
        if (a.merged_into >= 0 || a.merge_next >= 0 || a.cin != b.cin) continue;
        if ((int)i == block1_at || (int)i + 2 == block1_at || i + 2 == (int)block1_at + 1) continue;      // (the fused first block keeps its own kernel)
        int oa = 0, ob = 0;
        const int sa = owner_of(i, oa), sb = owner_of(i + 2, ob);
        if (storage[sa].fmt != storage[sb].fmt || a.c % 4 || b.c % 4) continue;
        if (storage[sa].fmt == FMT_H16 && (a.c % 32 || b.c % 32)) continue;
        a.merge_next = i + 2;
        b.merged_into = i;
    }
    plan_half_formats();
    allocate_buffers();
}

int Darknet::owner_of(int j, int &off) const {
    const Layer &l = layers[j];
    if (l.type == "route" && l.refs.size() > 1) { off = 0; return j; }
    int r = l.root;
    if (storage[r].redirected) { off = storage[r].coff + l.coff; return storage[r].into; }
    off = l.coff;
    return r;
}

// Half mode with 2-byte activations (round 4; reference: model.half() converts every activation to fp16, img_detect.py:49-50).
// The half-mode kernels read the hi halves only, so a buffer can hold just those (FMT_F16, h16.h: half the HBM bytes of the H16
// record) when everything that lives in it sits at 64-channel granularity (the 64-channel K steps of the LDS-DMA and window kernels
// fetch 128 contiguous bytes per pixel) and every kernel that touches it takes the format:
//   * any convolution writes it (shared epilogue), the LDS-DMA and window kernels read it; maxpool / shortcut-add go through the
//     generic accessors, upsample and route copies are byte-wise;
//   * the fused stem writes it through the same epilogue; the fused first residual block reads an F16 input patch into the hi
//     chunks of its LDS rows (lo = 0) and takes input, residual and output in ONE format (tied below);
//   * a convolution and its fused shortcut source share one format, so do the two outputs of a merged CSP launch and both sides
//     of a route copy - disagreeing pairs fall back to H16 together.
// Buffers keep their (H16-sized) allocation; an F16 view addresses it in float slots (ld / 2, channel offset / 2).
void Darknet::plan_half_formats() {
    const int L = (int)layers.size();
    for (Storage &st : storage) st.fmt_half = st.fmt;
    const bool off = getenv("YDS_HALF_H16") != nullptr;          // tuning aid, read at plan time: round 3's half mode (H16 tensors, hi halves read)
    if (!half_mode || off || math != MATH_F16X3) return;
    std::vector<char> ok(L, 0);
    for (int i = 0; i < L; ++i) ok[i] = storage[i].owns && storage[i].fmt == FMT_H16 && storage[i].ld % 64 == 0;
    auto own = [&](int j) { int off = 0; return owner_of(j, off); };
    for (int j = 0; j < L; ++j) {
        if (layers[j].type == "yolo") continue;
        int off = 0, o = owner_of(j, off);
        if (o >= 0 && (off % 64 || layers[j].c % 64)) ok[o] = 0;
    }
    if (block1_at >= 0) ok[own(block1_at)] = 0;                  // (32 channels: never written by the fused block anyway)
    for (int pass = 0; pass < L; ++pass) {
        bool changed = false;
        auto tie = [&](int a, int b) {
            if (a >= 0 && b >= 0 && ok[a] != ok[b]) { ok[a] = ok[b] = 0; changed = true; }
        };
        for (int i = 0; i < L; ++i) {
            const Layer &l = layers[i];
            for (auto &cp : l.copies) tie(own(cp.first), i);
            if (l.type == "convolutional" && l.fused_res >= 0) tie(own(i), own(l.fused_res));
            if (l.type == "convolutional" && l.merge_next >= 0) tie(own(i), own(l.merge_next));
            if (l.type == "upsample") tie(own(i), own(l.src));
            if (i == block1_at && l.src >= 0) tie(own(l.src), own(i + 1));
        }
        if (!changed) break;
    }
    for (int i = 0; i < L; ++i)
        if (ok[i]) storage[i].fmt_half = FMT_F16;
}

void Darknet::set_half(bool on) {
    if (on == half_mode) return;
    YDS_HIP(hipStreamSynchronize(stream));
    half_mode = on;
    plan_half_formats();
    stem_checked_reset();
}

// Everything whose size depends on batch_max.  set_batch_max() re-runs it inside the SAME object, so handles held by
// others (a pipeline, the Python wrapper) stay valid when a caller later sends a larger batch.
void Darknet::allocate_buffers() {
    size_t total_floats = 0;
    for (size_t i = 0; i < layers.size(); ++i) {
        if (!storage[i].owns) continue;
        const Layer &l = layers[i];
        size_t n = (size_t)batch_max * l.h * l.w * storage[i].ld;
        storage[i].buf.alloc(n);
        YDS_HIP(hipMemsetAsync(storage[i].buf.p, 0, n * sizeof(float), stream));
        total_floats += n;
    }
    input.alloc((size_t)batch_max * img_h * img_w * 4);
    out.alloc((size_t)batch_max * total_boxes * attrs);
    stage_f32.alloc((size_t)batch_max * img_h * img_w * 4);
    activation_bytes = total_floats * sizeof(float);
    YDS_HIP(hipStreamSynchronize(stream));
    inject_rows.clear();
    inject_rows.resize(batch_max);
    inject_n.assign(batch_max, 0);
    inject_active = false;                                       // injection tables are laid out per batch slot
    inject_offsets.clear();
    inject_set = -1;
    stem_checked = block1_checked = -1;
    stage_n = 0;
}

void Darknet::set_batch_max(int b) {
    if (b < 1) fail("set_batch_max: %d", b);
    if (b == batch_max) return;
    YDS_HIP(hipStreamSynchronize(stream));
    batch_max = b;
    allocate_buffers();
}

Darknet::~Darknet() {
    for (const ConvTimeRec &r : conv_pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
}

View Darknet::view(int i, int batch) const {
    const Layer &l = layers[i];
    int r = l.root;
    View v;
    v.n = batch; v.h = l.h; v.w = l.w; v.c = l.c;
    // (strides and channel offsets of an F16 buffer count float slots: half the channels, h16.h)
    if (l.type == "route" && l.refs.size() > 1) {
        v.fmt = half_mode ? storage[i].fmt_half : storage[i].fmt;
        v.p = storage[i].buf.p; v.ld = fmt_slots(v.fmt, storage[i].ld);
        v.p += (size_t)lane_img0 * l.h * l.w * v.ld;
        return v;
    }
    const Storage &st = storage[r];
    if (st.redirected) {
        const Storage &dst = storage[st.into];
        v.fmt = half_mode ? dst.fmt_half : dst.fmt;
        v.p = dst.buf.p + fmt_slots(v.fmt, st.coff + l.coff);
        v.ld = fmt_slots(v.fmt, dst.ld);
    } else {
        v.fmt = half_mode ? st.fmt_half : st.fmt;
        v.p = st.buf.p + fmt_slots(v.fmt, l.coff);
        v.ld = fmt_slots(v.fmt, st.ld);
    }
    v.p += (size_t)lane_img0 * l.h * l.w * v.ld;                // image range of the lane being enqueued (run_graph)
    return v;
}

View Darknet::input_view(int batch) const {
    View v;
    v.p = input.p + (size_t)lane_img0 * img_h * img_w * 4; v.n = batch; v.h = img_h; v.w = img_w; v.c = 4; v.ld = 4;
    return v;
}

// --------------------------------------------------------------------------------------- weights
size_t Darknet::weight_floats() const {
    size_t n = 0;
    for (const Layer &l : layers)
        if (l.type == "convolutional") n += (size_t)(l.bn ? 4 : 1) * l.c + (size_t)l.c * l.cin_file * l.ksize * l.ksize;
    return n;
}

void Darknet::load_weights(const void *blob, size_t nbytes, int cutoff) {
    // models.py:315-366: 5 x int32 header, then per conv: [beta, gamma, mean, var] | [bias], then W[cout][cin][k][k]
    if (nbytes < 20) fail("weights: blob shorter than the 20-byte header");
    const float *w = reinterpret_cast<const float *>(static_cast<const char *>(blob) + 20);
    size_t avail = (nbytes - 20) / 4, ptr = 0;
    memcpy(header, blob, 20);
    for (int i = 0; i < (int)layers.size(); ++i) {
        if (cutoff >= 0 && i == cutoff) break;
        Layer &l = layers[i];
        if (l.type != "convolutional") continue;
        const int co = l.c, ci = l.cin_file, k = l.ksize;
        size_t need = (size_t)(l.bn ? 4 : 1) * co + (size_t)co * ci * k * k;
        if (ptr + need > avail) fail("weights: file too short at layer %d (need %zu more floats, have %zu)", i, need, avail - ptr);
        std::vector<double> scale(co, 1.0);
        std::vector<float> bias(co);
        if (l.bn) {
            const float *beta = w + ptr, *gamma = beta + co, *mean = gamma + co, *var = mean + co;
            ptr += 4 * (size_t)co;
            for (int o = 0; o < co; ++o) {
                scale[o] = (double)gamma[o] / sqrt((double)var[o] + 1e-5);      // BatchNorm2d eps=1e-5, models.py:52
                bias[o] = (float)((double)beta[o] - (double)mean[o] * scale[o]);
            }
        } else {
            for (int o = 0; o < co; ++o) bias[o] = w[ptr + o];
            ptr += co;
        }
        const int cin_p = l.cin;                             // channel-padded input (3 -> 4 for the image)
        const int K = k * k * cin_p;
        l.kpad = (K + 31) / 32 * 32;
        std::vector<float> packed((size_t)co * l.kpad, 0.f);
        const float *src = w + ptr;
        for (int o = 0; o < co; ++o)
            for (int c = 0; c < ci; ++c)
                for (int kh = 0; kh < k; ++kh)
                    for (int kw = 0; kw < k; ++kw)
                        packed[(size_t)o * l.kpad + (kh * k + kw) * cin_p + c] =
                            (float)((double)src[(((size_t)o * ci + c) * k + kh) * k + kw] * scale[o]);
        ptr += (size_t)co * ci * k * k;
        l.wt.upload(packed.data(), packed.size(), stream);
        {
            std::vector<uint16_t> split;
            pack_weights_f16x3(packed.data(), co, l.kpad, split);
            l.wt16.upload(split.data(), split.size(), stream);
            YDS_HIP(hipStreamSynchronize(stream));
        }
        l.bias.upload(bias.data(), bias.size(), stream);
        YDS_HIP(hipStreamSynchronize(stream));
        l.loaded = true;
    }
    // concatenated filter sets of the CSP splits (device-to-device: [a's filters ; b's filters], same K)
    for (int i = 0; i < (int)layers.size(); ++i) {
        Layer &a = layers[i];
        if (a.merge_next < 0) continue;
        Layer &b = layers[a.merge_next];
        if (!a.loaded || !b.loaded || a.kpad != b.kpad) { a.merge_next = -1; b.merged_into = -1; continue; }
        const size_t na = (size_t)a.c * a.kpad, nb = (size_t)b.c * b.kpad;
        a.wt_m.alloc(na + nb); a.wt16_m.alloc(2 * (na + nb)); a.bias_m.alloc((size_t)a.c + b.c);
        YDS_HIP(hipMemcpyAsync(a.wt_m.p, a.wt.p, na * 4, hipMemcpyDeviceToDevice, stream));
        YDS_HIP(hipMemcpyAsync(a.wt_m.p + na, b.wt.p, nb * 4, hipMemcpyDeviceToDevice, stream));
        YDS_HIP(hipMemcpyAsync(a.wt16_m.p, a.wt16.p, na * 4, hipMemcpyDeviceToDevice, stream));
        YDS_HIP(hipMemcpyAsync(a.wt16_m.p + 2 * na, b.wt16.p, nb * 4, hipMemcpyDeviceToDevice, stream));
        YDS_HIP(hipMemcpyAsync(a.bias_m.p, a.bias.p, (size_t)a.c * 4, hipMemcpyDeviceToDevice, stream));
        YDS_HIP(hipMemcpyAsync(a.bias_m.p + a.c, b.bias.p, (size_t)b.c * 4, hipMemcpyDeviceToDevice, stream));
        YDS_HIP(hipStreamSynchronize(stream));
    }
    weights_loaded = true;
}

// --------------------------------------------------------------------------------------- forward
ConvArgs Darknet::conv_args(int i, int batch) const {
    const Layer &l = layers[i];
    ConvArgs a;
    a.x = l.src < 0 ? input_view(batch) : view(l.src, batch);
    a.y = view(i, batch);
    a.w = l.wt.p; a.bias = l.bias.p; a.w16 = l.wt16.p;
    a.ksize = l.ksize; a.stride = l.stride; a.pad = l.pad; a.kpad = l.kpad;
    a.act = l.act;
    if (l.fused_res >= 0) { a.res = view(l.fused_res, batch); a.res_mode = RES_AFTER_ACT; }
    a.terms = half_mode ? 1 : 3;
    return a;
}

ConvArgs Darknet::merged_conv_args(int i, int batch) const {
    const Layer &l = layers[i];
    ConvArgs a = conv_args(i, batch);
    const ConvArgs b = conv_args(l.merge_next, batch);
    a.y2 = b.y;
    a.n_split = a.y.c;
    a.y.c = a.y.c + b.y.c;                                       // (pointer, stride and format stay those of the first output)
    a.w = l.wt_m.p; a.w16 = l.wt16_m.p; a.bias = l.bias_m.p;
    return a;
}

void Darknet::autotune(int batch) {
    static const bool off = getenv("YDS_NO_AUTOTUNE") != nullptr;
    for (int i = 0; i < (int)layers.size(); ++i) {
        Layer &l = layers[i];
        const int mode = conv_math() + (half_mode ? 10 : 0);
        if (l.type != "convolutional" || !l.loaded || (l.tuned_batch == batch && l.tuned_math == mode)) continue;
        if (l.merged_into >= 0 && layers[l.merged_into].wt_m.p) { l.tuned_batch = batch; l.tuned_math = mode; continue; }   // launched by its partner
        l.variant = off ? -1 : conv_autotune(l.merge_next >= 0 && l.wt_m.p ? merged_conv_args(i, batch) : conv_args(i, batch), stream, nullptr);
        l.tuned_batch = batch;
        l.tuned_math = mode;
    }
}

// The detector is stateless per image, so a batch is enqueued as LANES independent image ranges on their own streams:
// while one lane sits in the tail of a layer (a partial last round of workgroups, a burst of epilogue stores) the other
// lane's kernels could fill the idle CUs.  Measured (cfg2, 16 frames): 1 lane 1275 frames/s, 2 lanes 1087, 4 lanes 964 -
// lanes that start together run the same layers in lock-step and contend instead of complementing each other (two
// desynchronised PROCESSES with half the batch each did gain 11 %), so the default is one lane and YDS_DET_LANES opts in.
// Results are unchanged by the split (every image sees exactly the same kernels).
void Darknet::run_graph(int batch) {
    if (batch < 1 || batch > batch_max) fail("forward: batch %d outside [1,%d]", batch, batch_max);
    if (math != conv_math()) fail("forward: this network was planned for conv math %d, current mode is %d (re-create it)", math, conv_math());
    int lanes = getenv("YDS_DET_LANES") ? atoi(getenv("YDS_DET_LANES")) : 1;     // opt-in: see the note above
    while (lanes > 1 && batch / lanes < 4) --lanes;             // at least four images per lane
    if (lanes < 1) lanes = 1;
    if (lanes == 1) { run_lane(0, batch, stream); return; }
    while ((int)lane_streams.size() < lanes - 1) {
        hipStream_t st; hipEvent_t ev;
        YDS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        YDS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        lane_streams.push_back(st); lane_done.push_back(ev);
    }
    if (!lane_fork) YDS_HIP(hipEventCreateWithFlags(&lane_fork, hipEventDisableTiming));
    // every lane count is tuned before anything is enqueued (tuning launches on the main stream)
    const int per = (batch + lanes - 1) / lanes;
    YDS_HIP(hipEventRecord(lane_fork, stream));                 // input resized, previous consumers of `out` enqueued
    for (int k = 0; k < lanes; ++k) {
        const int first = k * per, cnt = std::min(per, batch - first);
        if (cnt <= 0) break;
        // with per-launch event timing (roofline pass) the lanes run one after the other on the main stream
        hipStream_t st = (k == 0 || time_convs) ? stream : lane_streams[k - 1];
        if (st != stream) YDS_HIP(hipStreamWaitEvent(st, lane_fork, 0));
        run_lane(first, cnt, st);
        if (st != stream) {
            YDS_HIP(hipEventRecord(lane_done[k - 1], st));
            YDS_HIP(hipStreamWaitEvent(stream, lane_done[k - 1], 0));
        }
    }
}

// Layers the pipeline enqueues as the first piece of a pass: up to the first convolution past ~6 % of the network's
// arithmetic (yolov3 / yolov4 at 608 x 608: the stem and the first residual block, ~0.04 ms per image) - enough stream-ordered
// work to cover the host's share of the hand-over (NMS results -> crop list -> ReID launches) without delaying that ReID pass.
int Darknet::head_layers() const {
    double total = 0, acc = 0;
    for (int i = 0; i < (int)layers.size(); ++i)
        if (layers[i].type == "convolutional") total += conv_flops(conv_args(i, 1));
    for (int i = 0; i < (int)layers.size(); ++i) {
        if (layers[i].type != "convolutional") continue;
        acc += conv_flops(conv_args(i, 1));
        if (acc >= 0.06 * total) return i + 1;
    }
    return (int)layers.size();
}

bool Darknet::forward_resized_part(int batch, int part) {
    if (batch < 1 || batch > batch_max) fail("forward: batch %d outside [1,%d]", batch, batch_max);
    if (math != conv_math()) fail("forward: this network was planned for conv math %d, current mode is %d (re-create it)", math, conv_math());
    if (getenv("YDS_DET_LANES") && atoi(getenv("YDS_DET_LANES")) > 1) return false;
    const int cut = head_layers();
    if (part == 0) run_lane(0, batch, stream, 0, cut);
    else run_lane(0, batch, stream, cut, -1);
    return true;
}

void Darknet::run_lane(int first, int batch, hipStream_t stream, int l0, int l1) {
    autotune(batch);                                            // (tuning launches use the main stream; cached per image count)
    lane_img0 = first;
    struct Reset { int &v; ~Reset() { v = 0; } } reset{lane_img0};
    if (l1 < 0 || l1 > (int)layers.size()) l1 = (int)layers.size();
    for (int i = l0; i < l1; ++i) {
        Layer &l = layers[i];
        if (l.type == "convolutional") {
            if (!l.loaded) fail("forward: layer %d has no weights (call load_darknet_weights)", i);
            if (l.merged_into >= 0 && layers[l.merged_into].wt_m.p) continue;   // computed by the first convolution of the CSP split
            ConvArgs a = l.merge_next >= 0 && l.wt_m.p ? merged_conv_args(i, batch) : conv_args(i, batch);
            if (i == 0 && stem_fused(batch)) continue;              // computed inside layer 1's launch
            if (i == block1_at && block1_fused(batch)) continue;    // computed inside the next layer's launch
            ConvTimeRec rec;
            if (time_convs) { rec.e0 = timing_event(); rec.e1 = timing_event(); YDS_HIP(hipEventRecord(rec.e0, stream)); }
            int variant;
            double extra_flops = 0, extra_bytes = 0;
            if (i == 1 && stem_fused(batch)) {
                ConvArgs a0 = conv_args(0, batch);
                ConvKernelArgs k0 = make_conv_args(a0), k1 = make_conv_args(a);
                k1.w = reinterpret_cast<const float *>(a.w16);
                launch_conv_stem2(k0, k1, stream);
                variant = kDirectVariant;                           // accounted with the direct first-layer kernel
                extra_flops = conv_flops(a0);
                extra_bytes = conv_bytes(a0) - conv_bytes_io(a0.y) - conv_bytes_io(a.x);     // the intermediate tensor never reaches HBM
            } else if (i == block1_at + 1 && block1_at >= 0 && block1_fused(batch)) {
                ConvArgs a2 = conv_args(block1_at, batch);
                ConvKernelArgs k2 = make_conv_args(a2), k3 = make_conv_args(a);
                k2.w = reinterpret_cast<const float *>(a2.w16);
                k3.w = reinterpret_cast<const float *>(a.w16);
                launch_conv_block1(k2, k3, stream);
                variant = kF32Variants + 8;                         // accounted with the window-resident 3x3 kernel
                extra_flops = conv_flops(a2);
                // the block input is read once (it is also the residual), the 32-channel intermediate never reaches HBM
                extra_bytes = conv_bytes(a2) - conv_bytes_io(a2.y) - conv_bytes_io(a.x) - (a.res.p ? conv_bytes_io(a.res) : 0.0);
            } else {
                variant = launch_conv(a, stream, l.variant);
            }
            if (time_convs) {
                // no host synchronisation here: the pair is resolved when the counters are read, so the pass runs exactly as it
                // does untimed (other streams live, the stream fed a pass ahead)
                YDS_HIP(hipEventRecord(rec.e1, stream));
                rec.variant = variant;
                rec.flops = conv_flops(a) + extra_flops;
                rec.bytes = conv_bytes(a) + extra_bytes;
                // bound of the arithmetic the launch really used: f16x3 = three fp16 MFMAs per product block; the window kernel's
                const double peak = conv_math() == MATH_F32 ? 157.3e12 : (a.terms == 1 ? 2500e12 : 2500e12 / 3);
                rec.attain_us = std::max(rec.flops / peak, rec.bytes / 6.29e12) * 1e6;
                conv_pending.push_back(rec);
            }
        } else if (l.type == "maxpool") {
            // SPP (yolov4: 5 / 9 / 13, stride 1, all on one tensor): a k x k max with -inf padding is a 5 x 5 max of the
            // (k-4) x (k-4) max, so a pool whose input already has a (k-4)-pool of the same tensor reads THAT instead -
            // 25 loads per value instead of 81 / 169, same maxima
            int cascade = -1;
            if (l.stride == 1 && !l.zero_br && l.ksize >= 9 && l.pad == (l.ksize - 1) / 2)
                for (int j = i - 1; j >= 0 && cascade < 0; --j) {
                    const Layer &q = layers[j];
                    if (q.type == "maxpool" && q.stride == 1 && !q.zero_br && q.ksize == l.ksize - 4 && q.pad == (q.ksize - 1) / 2 && q.c == l.c &&
                        layers[q.src].root == layers[l.src].root && layers[q.src].coff == layers[l.src].coff)
                        cascade = j;
                }
            if (cascade >= 0) launch_maxpool(view(cascade, batch), view(i, batch), 5, 1, 2, false, stream);
            else launch_maxpool(view(l.src, batch), view(i, batch), l.ksize, l.stride, l.pad, l.zero_br, stream);
        } else if (l.type == "upsample") {
            launch_upsample(view(l.src, batch), view(i, batch), l.stride, stream);
        } else if (l.type == "route") {
            if (l.refs.size() > 1) {
                View dst = view(i, batch);
                for (auto &cp : l.copies) {
                    View d = dst;
                    d.p += fmt_slots(dst.fmt, cp.second);
                    launch_copy(view(cp.first, batch), d, stream);
                }
            }
        } else if (l.type == "shortcut") {
            if (!l.fused) launch_add(view(l.refs[0], batch), view(l.refs[1], batch), view(i, batch), stream);
        } else if (l.type == "yolo") {
            View head = view(l.src, batch);
            int hidx = 0;
            for (int y : yolo_layers) { if (y == i) break; ++hidx; }
            // `out` is about to be overwritten: wait for whoever still reads the previous pass's predictions (pipeline NMS
            // running on its own stream); the first decode sits ~3/4 into the pass, so this wait is free in practice
            if (hidx == 0 && out_guard) YDS_HIP(hipStreamWaitEvent(stream, out_guard, 0));
            if (inject_active && inject_set >= 0) {
                launch_inject_batch(head, batch, inject_table.p, inject_offsets_dev.p + (size_t)inject_set * batch_max + first, inject_max_rows, hidx,
                                    l.classes, inject_logit, stream);
            } else {
                for (int b = 0; b < batch && inject_active; ++b)
                    launch_inject(head, b, inject_rows[first + b].p, inject_n[first + b], hidx, l.classes, inject_logit, stream);
            }
            launch_yolo_decode(head, out.p + (size_t)first * total_boxes * attrs, total_boxes, l.box_off, l.classes, l.anchors.data(), (int)l.anchors.size() / 2, img_h, img_w, stream);
        }
    }
}

void Darknet::forward_f32_host(const float *nchw, int batch, float *out_host) {
    size_t n = (size_t)batch * in_channels * img_h * img_w;
    YDS_HIP(hipMemcpyAsync(stage_f32.p, nchw, n * sizeof(float), hipMemcpyHostToDevice, stream));
    launch_nchw_to_nhwc(stage_f32.p, input_view(batch), in_channels, stream);
    run_graph(batch);
    if (out_host) YDS_HIP(hipMemcpyAsync(out_host, out.p, (size_t)batch * total_boxes * attrs * sizeof(float), hipMemcpyDeviceToHost, stream));
    YDS_HIP(hipStreamSynchronize(stream));
}

void Darknet::forward_u8_dev(const uint8_t *frames_dev, int h, int w, int batch) {
    if (in_channels != 3) fail("forward_u8: network expects %d channels", in_channels);
    if (batch < 1 || batch > batch_max) fail("forward: batch %d outside [1,%d]", batch, batch_max);
    launch_resize_u8(frames_dev, batch, h, w, input_view(batch), stream);
    run_graph(batch);
}

void Darknet::forward_u8_host(const uint8_t *frames, int h, int w, int batch, float *out_host) {
    size_t n = (size_t)batch * h * w * 3;
    stage_u8.ensure(n);
    YDS_HIP(hipMemcpyAsync(stage_u8.p, frames, n, hipMemcpyHostToDevice, stream));
    stage_h = h; stage_w = w; stage_n = batch;
    forward_u8_dev(stage_u8.p, h, w, batch);
    if (out_host) YDS_HIP(hipMemcpyAsync(out_host, out.p, (size_t)batch * total_boxes * attrs * sizeof(float), hipMemcpyDeviceToHost, stream));
    YDS_HIP(hipStreamSynchronize(stream));
}

void Darknet::forward_tiles_host(const uint8_t *frame, int h, int w, const int *tiles, int n_tiles) {
    if (in_channels != 3) fail("forward_tiles: network expects %d channels", in_channels);
    if (n_tiles < 1) fail("forward_tiles: no windows");
    std::vector<float> scale((size_t)n_tiles * 2);
    for (int t = 0; t < n_tiles; ++t) {
        const int x = tiles[t * 4], y = tiles[t * 4 + 1], th = tiles[t * 4 + 2], tw = tiles[t * 4 + 3];
        if (x < 0 || y < 0 || th < 1 || tw < 1 || x + tw > w || y + th > h) fail("forward_tiles: window %d (%d,%d,%d,%d) outside the %dx%d frame", t, x, y, th, tw, w, h);
        scale[t * 2] = (float)((double)tw / img_w);            // resize_boxes: python-double ratio, fp32 multiply
        scale[t * 2 + 1] = (float)((double)th / img_h);
    }
    const size_t nbytes = (size_t)h * w * 3;
    stage_u8.ensure(nbytes);
    tile_rects.ensure((size_t)n_tiles * 4);
    tile_scale.ensure((size_t)n_tiles * 2);
    tiled_pred.ensure((size_t)n_tiles * total_boxes * attrs);
    YDS_HIP(hipMemcpyAsync(stage_u8.p, frame, nbytes, hipMemcpyHostToDevice, stream));
    stage_h = h; stage_w = w; stage_n = 1;                      // stage_u8 now holds exactly this frame
    YDS_HIP(hipMemcpyAsync(tile_rects.p, tiles, (size_t)n_tiles * 4 * sizeof(int), hipMemcpyHostToDevice, stream));
    YDS_HIP(hipMemcpyAsync(tile_scale.p, scale.data(), scale.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    for (int t0 = 0; t0 < n_tiles; t0 += batch_max) {
        const int nb = std::min(batch_max, n_tiles - t0);
        launch_tile_resize(stage_u8.p, w, tile_rects.p + (size_t)t0 * 4, nb, input_view(nb), stream);
        run_graph(nb);
        launch_tile_boxes(out.p, total_boxes, attrs, tile_rects.p + (size_t)t0 * 4, tile_scale.p + (size_t)t0 * 2, nb,
                          tiled_pred.p + (size_t)t0 * total_boxes * attrs, stream);
    }
    YDS_HIP(hipStreamSynchronize(stream));          // `scale` and the caller's buffers may go away
}

bool Darknet::stem_fused(int batch) {
    static const bool off = getenv("YDS_NO_STEM_FUSE") != nullptr;
    // (half mode keeps the fused kernels: they compute the first layers in the default arithmetic, which half mode does anyway)
    if (off || !stem_fusable || conv_math() != MATH_F16X3 || !layers[0].loaded || !layers[1].loaded) return false;
    if (stem_checked != batch) {
        ConvArgs a0 = conv_args(0, batch), a1 = conv_args(1, batch);
        stem_ok = a1.w16 && a1.y.fmt != FMT_F32 && conv_stem2_applicable(make_conv_args(a0), make_conv_args(a1));
        stem_checked = batch;
    }
    return stem_ok;
}

bool Darknet::block1_fused(int batch) {
    static const bool off = getenv("YDS_NO_BLOCK_FUSE") != nullptr;
    if (off || block1_at < 0 || conv_math() != MATH_F16X3 || !layers[block1_at].loaded || !layers[block1_at + 1].loaded) return false;
    if (block1_checked != batch) {
        ConvArgs a2 = conv_args(block1_at, batch), a3 = conv_args(block1_at + 1, batch);
        block1_ok = a2.w16 && a3.w16 && conv_block1_applicable(make_conv_args(a2), make_conv_args(a3));
        block1_checked = batch;
    }
    return block1_ok;
}

void Darknet::layer_output_host(int i, int batch, float *nchw) {
    if (i < 0 || i >= (int)layers.size()) fail("layer_output: no layer %d", i);
    const Layer &l = layers[i];
    if (i == block1_at && block1_fused(batch)) {                  // never written by the fused block: produce it on demand
        ConvArgs a2 = conv_args(i, batch);
        (void)launch_conv(a2, stream, layers[i].variant);
    }
    if (i == 0 && stem_fused(batch)) {                            // the fused stem never writes layer 0: produce it on demand
        ConvArgs a0 = conv_args(0, batch);
        (void)launch_conv(a0, stream, layers[0].variant);
    }
    if (l.type == "yolo") fail("layer_output: yolo layers are read through the forward output");
    if (l.fused_res >= 0) fail("layer_output: layer %d is fused with the following shortcut", i);
    View v = view(i, batch);
    DevBuf<float> tmp(v.pixels() * v.c);
    launch_nhwc_to_nchw(v, tmp.p, stream);
    YDS_HIP(hipMemcpyAsync(nchw, tmp.p, tmp.n * sizeof(float), hipMemcpyDeviceToHost, stream));
    YDS_HIP(hipStreamSynchronize(stream));
}

void Darknet::get_input_host(int batch, float *nchw) {
    View v = input_view(batch);
    v.c = in_channels;
    DevBuf<float> tmp(v.pixels() * v.c);
    launch_nhwc_to_nchw(v, tmp.p, stream);
    YDS_HIP(hipMemcpyAsync(nchw, tmp.p, tmp.n * sizeof(float), hipMemcpyDeviceToHost, stream));
    YDS_HIP(hipStreamSynchronize(stream));
}

void Darknet::set_injection(int image, const float *rows, int n, float logit) {
    if (image < 0 || image >= batch_max) fail("inject: image %d outside batch", image);
    inject_n[image] = n;
    if (n > 0) inject_rows[image].upload(rows, (size_t)n * 9, stream);
    YDS_HIP(hipStreamSynchronize(stream));
    inject_logit = logit;
    inject_active = false;
    for (int b = 0; b < batch_max; ++b) inject_active |= inject_n[b] > 0;
}

void Darknet::load_injection_sets(const float *rows, const int *offsets, int n_sets, float logit) {
    if (n_sets <= 0) { inject_set = -1; inject_active = false; inject_offsets.clear(); return; }
    inject_offsets.assign(offsets, offsets + (size_t)n_sets * batch_max + 1);
    inject_table.ensure((size_t)inject_offsets.back() * 9 + 9);
    inject_table.upload(rows, (size_t)inject_offsets.back() * 9, stream);
    inject_offsets_dev.ensure(inject_offsets.size());
    inject_offsets_dev.upload(inject_offsets.data(), inject_offsets.size(), stream);
    inject_max_rows = 0;
    for (size_t i = 0; i + 1 < inject_offsets.size(); ++i) inject_max_rows = std::max(inject_max_rows, inject_offsets[i + 1] - inject_offsets[i]);
    YDS_HIP(hipStreamSynchronize(stream));
    inject_logit = logit;
    inject_set = 0;
    inject_active = true;
}

void Darknet::select_injection_set(int set) {
    if (inject_offsets.empty()) fail("inject: no sets loaded");
    int n_sets = (int)(inject_offsets.size() - 1) / batch_max;
    if (set < 0 || set >= n_sets) fail("inject: set %d outside [0,%d)", set, n_sets);
    inject_set = set;
}

int64_t Darknet::flops_per_image() const {
    int64_t f = 0;
    for (const Layer &l : layers)
        if (l.type == "convolutional") f += 2ll * l.h * l.w * l.c * l.ksize * l.ksize * l.cin_file;
    return f;
}

void Darknet::enable_conv_timing(bool on) {
    if (!on) resolve_conv_timing();
    time_convs = on;
}

hipEvent_t Darknet::timing_event() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e;
    YDS_HIP(hipEventCreate(&e));
    return e;
}

// Waits for the stream, turns every recorded (start, stop) pair into per-variant totals and recycles the events.
void Darknet::resolve_conv_timing() {
    if (conv_pending.empty()) return;
    YDS_HIP(hipStreamSynchronize(stream));
    for (const ConvTimeRec &r : conv_pending) {
        float ms = 0;
        YDS_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        conv_us[r.variant] += ms * 1e3;
        conv_launches[r.variant]++;
        conv_flops_acc[r.variant] += r.flops;
        conv_bytes_acc[r.variant] += r.bytes;
        conv_attain_us[r.variant] += r.attain_us;
        ev_pool.push_back(r.e0);
        ev_pool.push_back(r.e1);
    }
    conv_pending.clear();
}

}  // namespace yds

// ============================================================================================ C ABI
using yds::Darknet;

extern "C" {

yds_net *yds_darknet_create(const char *cfg_text, int img_h, int img_w, int batch_max) {
    YDS_API_BEGIN
    if (!cfg_text) yds::fail("cfg_text is NULL");
    if (img_h <= 0 || img_w <= 0 || batch_max <= 0) yds::fail("bad geometry %dx%d batch %d", img_h, img_w, batch_max);
    auto *n = new yds_net{new Darknet(cfg_text, img_h, img_w, batch_max)};
    return n;
    YDS_API_END_PTR
}
void yds_darknet_destroy(yds_net *n) {
    if (n) { delete n->d; delete n; }
}
int yds_darknet_load_weights(yds_net *n, const void *blob, size_t nbytes, int cutoff) {
    YDS_API_BEGIN
    n->d->load_weights(blob, nbytes, cutoff);
    YDS_API_END
}
int yds_darknet_set_batch_max(yds_net *n, int batch_max) {
    YDS_API_BEGIN
    n->d->set_batch_max(batch_max);
    YDS_API_END
}
int yds_darknet_batch_max(const yds_net *n) { return n->d->batch_max; }
int yds_darknet_set_half(yds_net *n, int on) {
    YDS_API_BEGIN
    if (on && n->d->math != yds::MATH_F16X3) yds::fail("half mode needs the split-fp16 tensor formats (conv math f16x3)");
    n->d->set_half(on != 0);
    YDS_API_END
}
int yds_darknet_layer_format(const yds_net *n, int layer) {
    if (!n || layer < 0 || layer >= (int)n->d->layers.size() || n->d->layers[layer].type == "yolo") return -1;
    return n->d->view(layer, 1).fmt;
}
int yds_darknet_num_boxes(const yds_net *n) { return n->d->total_boxes; }
int yds_darknet_num_attrs(const yds_net *n) { return n->d->attrs; }
int yds_darknet_num_layers(const yds_net *n) { return (int)n->d->layers.size(); }
int yds_darknet_layer_shape(const yds_net *n, int layer, int *c, int *h, int *w) {
    YDS_API_BEGIN
    if (layer < 0 || layer >= (int)n->d->layers.size()) yds::fail("no layer %d", layer);
    const auto &l = n->d->layers[layer];
    *c = l.c; *h = l.h; *w = l.w;
    YDS_API_END
}
int64_t yds_darknet_conv_flops(const yds_net *n) { return n->d->flops_per_image(); }
int yds_darknet_forward_f32(yds_net *n, const float *nchw_host, int batch, float *out_host) {
    YDS_API_BEGIN
    n->d->forward_f32_host(nchw_host, batch, out_host);
    YDS_API_END
}
int yds_darknet_forward_u8(yds_net *n, const uint8_t *rgb, int h, int w, int batch, float *out_host) {
    YDS_API_BEGIN
    n->d->forward_u8_host(rgb, h, w, batch, out_host);
    YDS_API_END
}
const uint8_t *yds_darknet_last_frames_dev(yds_net *n, int *h, int *w, int *batch) {
    if (!n || !n->d->stage_n) return nullptr;
    *h = n->d->stage_h; *w = n->d->stage_w; *batch = n->d->stage_n;
    return n->d->stage_u8.p;
}
int yds_darknet_forward_u8_dev(yds_net *n, const uint8_t *rgb_dev, int h, int w, int batch) {
    YDS_API_BEGIN
    n->d->forward_u8_dev(rgb_dev, h, w, batch);
    YDS_API_END
}
int yds_darknet_layer_output(yds_net *n, int layer, int batch, float *nchw_host) {
    YDS_API_BEGIN
    n->d->layer_output_host(layer, batch, nchw_host);
    YDS_API_END
}
int yds_darknet_get_input(yds_net *n, int batch, float *nchw_host) {
    YDS_API_BEGIN
    n->d->get_input_host(batch, nchw_host);
    YDS_API_END
}
int yds_darknet_set_injection(yds_net *n, int image, const float *rows, int cnt, float logit) {
    YDS_API_BEGIN
    n->d->set_injection(image, rows, cnt, logit);
    YDS_API_END
}
int yds_conv_timing_ex(yds_net *n, int mode, double *total_us, int64_t *launches, double *flops, double *bytes, double *attainable_us) {
    YDS_API_BEGIN
    Darknet *d = n->d;
    if (mode == 2) d->enable_conv_timing(false);           // resolves the pending event pairs (one stream synchronisation)
    else d->resolve_conv_timing();
    for (int v = 0; v < yds::kConvVariants; ++v) {
        if (total_us) total_us[v] = d->conv_us[v];
        if (launches) launches[v] = d->conv_launches[v];
        if (flops) flops[v] = d->conv_flops_acc[v];
        if (bytes) bytes[v] = d->conv_bytes_acc[v];
        if (attainable_us) attainable_us[v] = d->conv_attain_us[v];
    }
    if (mode == 1) {
        for (int v = 0; v < yds::kConvVariants; ++v) {
            d->conv_us[v] = 0; d->conv_launches[v] = 0; d->conv_flops_acc[v] = 0; d->conv_bytes_acc[v] = 0; d->conv_attain_us[v] = 0;
        }
        d->enable_conv_timing(true);
    }
    YDS_API_END
}
int yds_conv_timing(yds_net *n, int mode, double *total_us4, int64_t *launches4, double *flops4) {
    return yds_conv_timing_ex(n, mode, total_us4, launches4, flops4, nullptr, nullptr);
}
const char *yds_conv_variant_name(int v) { return yds::conv_variant_name(v); }
int yds_conv_num_variants(void) { return yds::kConvVariants; }
int yds_set_conv_math(int mode) { yds::set_conv_math(mode); return 0; }
int yds_get_conv_math(void) { return yds::conv_math(); }
int yds_conv_bench(int n, int h, int w, int cin, int cout, int ksize, int stride, int act, int with_residual, int iters, double *avg_us,
                   int *variant) {
    YDS_API_BEGIN
    using namespace yds;
    const int pad = (ksize - 1) / 2, ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    const int kpad = (ksize * ksize * cin + 31) / 32 * 32, ldy = (cout + 3) / 4 * 4;
    std::vector<float> hx((size_t)n * h * w * cin), hw((size_t)cout * kpad), hb(cout);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
    // YDS_BENCH_DATA=zero | const: power experiment (the chip is power limited: operands that do not toggle run at a higher clock)
    const char *dk = getenv("YDS_BENCH_DATA");
    const int data_kind = !dk ? 0 : (!strcmp(dk, "zero") ? 1 : (!strcmp(dk, "const") ? 2 : 0));
    for (auto &v : hx) v = data_kind == 1 ? 0.f : data_kind == 2 ? 0.5f : rnd();
    for (auto &v : hw) v = data_kind == 1 ? 0.f : data_kind == 2 ? 0.03125f : rnd() * 0.05f;
    for (auto &v : hb) v = rnd();
    DevBuf<float> x, wt, b, y((size_t)n * ho * wo * ldy), r((size_t)n * ho * wo * ldy);
    DevBuf<uint16_t> wt16;
    x.upload(hx.data(), hx.size()); wt.upload(hw.data(), hw.size()); b.upload(hb.data(), hb.size());
    {
        std::vector<uint16_t> split;
        pack_weights_f16x3(hw.data(), cout, kpad, split);
        wt16.upload(split.data(), split.size());
        YDS_HIP(hipDeviceSynchronize());
    }
    YDS_HIP(hipMemset(r.p, 0, r.n * sizeof(float)));
    ConvArgs a;
    const bool f16 = conv_math() == MATH_F16X3;
    a.x = View{x.p, n, h, w, cin, cin, (f16 && cin % 32 == 0) ? FMT_H16 : FMT_F32};
    a.y = View{y.p, n, ho, wo, cout, ldy, (f16 && cout % 32 == 0) ? FMT_H16 : FMT_F32};
    if (a.x.fmt == FMT_H16) {
        DevBuf<float> raw;
        raw.upload(hx.data(), hx.size());
        launch_pack_h16(raw.p, a.x, nullptr);
        YDS_HIP(hipDeviceSynchronize());
    }
    a.w = wt.p; a.w16 = wt16.p; a.bias = b.p; a.ksize = ksize; a.stride = stride; a.pad = pad; a.kpad = kpad; a.act = act;
    if (with_residual) { a.res = View{r.p, n, ho, wo, cout, ldy, a.y.fmt}; a.res_mode = RES_AFTER_ACT; }
    if (const char *t = getenv("YDS_BENCH_TERMS")) a.terms = atoi(t) == 1 ? 1 : 3;      // tuning aid: the half-mode kernels on single layers
    hipStream_t st;
    YDS_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    YDS_HIP(hipEventCreate(&e0)); YDS_HIP(hipEventCreate(&e1));
    int tuned = getenv("YDS_NO_AUTOTUNE") ? -1 : conv_autotune(a, st, nullptr);
    for (int i = 0; i < 3; ++i) *variant = launch_conv(a, st, tuned);
    YDS_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch_conv(a, st, tuned);
    YDS_HIP(hipEventRecord(e1, st));
    YDS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    YDS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1e3 / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    YDS_API_END
}
int yds_conv_clock(double *ghz, double *sampled_ms, int reset) {
    YDS_API_BEGIN
    double g = 0, ms = 0;
#ifdef YDS_CLOCK_PROBE
    // tools/ builds only (-DYDS_CLOCK_PROBE): the rounds 3-5 form, sampled inside the window-resident kernels
    YDS_HIP(hipDeviceSynchronize());
    unsigned long long a[2], b[2], c[2];
    yds::conv_win_clock(a, reset != 0);
    yds::conv_win2_clock(b, reset != 0);
    yds::conv_win16_clock(c, reset != 0);
    const double cycles = (double)a[0] + (double)b[0] + (double)c[0], ticks = (double)a[1] + (double)b[1] + (double)c[1];
    g = ticks > 0 ? cycles / ticks * 0.1 : 0.0;                          // ticks are 10 ns
    ms = ticks * 1e-5;
#else
    // product build: nothing is sampled inside the kernels (round 6).  A probe wave of its own on a side stream was tried and dropped:
    // the s_memtime / s_memrealtime ratio of a wave that mostly sleeps reads 2.39-2.41 GHz whether the chip idles or runs the window
    // kernel back to back, where waves of that kernel read 1.86-1.89 on the same box (profiles/r06_clock_probe_negative.txt) - it does
    // not see the clock the loaded CUs run at.  bench.py samples the driver's sclk (sysfs pp_dpm_sclk) beside its diagnostic leg.
    (void)reset;
#endif
    if (ghz) *ghz = g;
    if (sampled_ms) *sampled_ms = ms;
    YDS_API_END
}
int yds_conv_run(int variant, int n, int h, int w, int cin, int cout, int ksize, int stride, int act, int res_mode, const float *x_nhwc,
                 const float *w_okkc, const float *bias, const float *res_nhwc, float *y_nchw) {
    YDS_API_BEGIN
    using namespace yds;
    const int pad = (ksize - 1) / 2, ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    const int K = ksize * ksize * cin, kpad = (K + 31) / 32 * 32, ldy = (cout + 3) / 4 * 4;
    if (cin % 4) fail("conv_run: input channels must be a multiple of 4");
    std::vector<float> hw((size_t)cout * kpad, 0.f);
    for (int o = 0; o < cout; ++o) memcpy(&hw[(size_t)o * kpad], w_okkc + (size_t)o * K, (size_t)K * sizeof(float));
    const size_t npix_in = (size_t)n * h * w, npix_out = (size_t)n * ho * wo;
    DevBuf<float> x, raw, wt, b, y(npix_out * ldy), r, rraw, out(npix_out * cout);
    DevBuf<uint16_t> wt16;
    wt.upload(hw.data(), hw.size()); b.upload(bias, cout);
    std::vector<uint16_t> split;
    pack_weights_f16x3(hw.data(), cout, kpad, split);
    wt16.upload(split.data(), split.size());
    YDS_HIP(hipDeviceSynchronize());
    const bool f16 = conv_math() == MATH_F16X3;
    ConvArgs a;
    a.x = View{nullptr, n, h, w, cin, cin, (f16 && cin % 32 == 0) ? FMT_H16 : FMT_F32};
    a.y = View{y.p, n, ho, wo, cout, ldy, (f16 && cout % 32 == 0) ? FMT_H16 : FMT_F32};
    raw.upload(x_nhwc, npix_in * cin);
    if (a.x.fmt == FMT_H16) { x.alloc(npix_in * cin); a.x.p = x.p; launch_pack_h16(raw.p, a.x, nullptr); }
    else a.x.p = raw.p;
    a.w = wt.p; a.w16 = wt16.p; a.bias = b.p; a.ksize = ksize; a.stride = stride; a.pad = pad; a.kpad = kpad; a.act = act;
    if (res_mode) {
        if (!res_nhwc) fail("conv_run: residual mode %d without a residual tensor", res_mode);
        a.res = View{nullptr, n, ho, wo, cout, cout, a.y.fmt};
        rraw.upload(res_nhwc, npix_out * cout);
        if (a.res.fmt == FMT_H16) { r.alloc(npix_out * cout); a.res.p = r.p; launch_pack_h16(rraw.p, a.res, nullptr); }
        else a.res.p = rraw.p;
        a.res_mode = res_mode;
    }
    YDS_HIP(hipDeviceSynchronize());
    launch_conv(a, nullptr, variant);
    launch_nhwc_to_nchw(a.y, out.p, nullptr);
    YDS_HIP(hipMemcpy(y_nchw, out.p, npix_out * cout * sizeof(float), hipMemcpyDeviceToHost));
    YDS_API_END
}
int yds_darknet_load_injection_sets(yds_net *n, const float *rows_host, const int32_t *offsets_host, int n_sets, float logit) {
    YDS_API_BEGIN
    n->d->load_injection_sets(rows_host, offsets_host, n_sets, logit);
    YDS_API_END
}
int yds_darknet_select_injection_set(yds_net *n, int set) {
    YDS_API_BEGIN
    n->d->select_injection_set(set);
    YDS_API_END
}

}  // extern "C"
