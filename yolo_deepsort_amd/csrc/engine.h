// Host-side engine classes of libydsort (detector, ReID, tracker, pipeline).
#pragma once
#include "common.h"

#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace yds {

struct CfgBlock {
    std::string type;
    std::map<std::string, std::string> kv;
};
std::vector<CfgBlock> parse_cfg(const std::string &text);

struct Layer {
    std::string type;
    int c = 0, h = 0, w = 0;              // logical output shape
    int root = -1, coff = 0;              // storage owner + channel offset (views)
    bool is_view = false;
    int src = -1;                         // single data input (conv/pool/upsample/yolo)
    std::vector<int> refs;                // route sources / shortcut operands
    std::vector<std::pair<int, int>> copies;   // (source layer, channel offset) needing a copy kernel
    // conv
    int bn = 0, ksize = 1, stride = 1, pad = 0, cin = 0, cin_file = 0, kpad = 0, act = ACT_LINEAR;
    int fused_res = -1;                   // residual layer absorbed from the following shortcut
    int variant = -1, tuned_batch = 0, tuned_math = -1;   // measured conv tile variant, the batch and math mode it was measured at
    bool loaded = false;
    DevBuf<float> wt, bias;
    DevBuf<uint16_t> wt16;                // split-fp16 copy of wt for the f16x3 kernel
    // CSP split (yolov4: conv 1x1, route -2, conv 1x1 - two convolutions of the same tensor): the first one launches both
    // (merge_next = the second conv, which is skipped: merged_into = the first) from concatenated filters, so the shared
    // input is read from HBM once
    int merge_next = -1, merged_into = -1;
    DevBuf<float> wt_m, bias_m;
    DevBuf<uint16_t> wt16_m;
    // shortcut / route / pool
    bool fused = false, zero_br = false;
    int groups = 0, group_id = 0;
    // yolo
    std::vector<float> anchors;
    int classes = 0, box_off = 0;
};

struct Storage {
    DevBuf<float> buf;
    int ld = 0;
    int fmt = 0;                          // TensorFmt of the owned buffer (H16 only in f16x3 mode, 32-channel granularity)
    int fmt_half = 0;                     // ... while the network is in half mode: FMT_F16 where every user of the buffer can take it (plan_half_formats)
    bool owns = false;                    // this layer owns `buf` (sized by batch_max)
    bool redirected = false;              // producer writes into a slice of storage[into]
    int into = -1, coff = 0;
};

class Darknet {
public:
    Darknet(const std::string &cfg_text, int img_h, int img_w, int batch_max);
    ~Darknet();
    void load_weights(const void *blob, size_t nbytes, int cutoff);
    void set_batch_max(int b);               // re-sizes the activation buffers in place (weights, plan and handle stay)
    void forward_f32_host(const float *nchw, int batch, float *out_host);
    void forward_u8_host(const uint8_t *frames, int h, int w, int batch, float *out_host);
    void forward_u8_dev(const uint8_t *frames_dev, int h, int w, int batch);
    void forward_resized(int batch) { run_graph(batch); }      // input buffer already filled
    // the same pass enqueued in two pieces (layers [0, head_layers()) and the rest): the pipeline puts another stream-ordered
    // job between them (pipeline.cpp, serialized schedule); false = this configuration runs in lanes and cannot be split
    bool forward_resized_part(int batch, int part);
    int head_layers() const;
    // sliding-window front end (img_detect.py:97-139): windows (x, y, th, tw) of one host frame -> corner-form, window-
    // shifted predictions [n_tiles * total_boxes, attrs] in tiled_pred (windows run in chunks of batch_max)
    void forward_tiles_host(const uint8_t *frame, int h, int w, const int *tiles_xyhw, int n_tiles);
    void layer_output_host(int layer, int batch, float *nchw);
    void get_input_host(int batch, float *nchw);
    void set_injection(int image, const float *rows, int n, float logit);
    void load_injection_sets(const float *rows, const int *offsets, int n_sets, float logit);
    void select_injection_set(int set);
    void enable_conv_timing(bool on);
    void autotune(int batch);                // measure the fastest conv tile per layer at this batch size
    int64_t flops_per_image() const;
    size_t weight_floats() const;
    View view(int layer, int batch) const;
    ConvArgs conv_args(int layer, int batch) const;
    ConvArgs merged_conv_args(int layer, int batch) const;     // both convolutions of a CSP split in one launch
    View input_view(int batch) const;

    int img_h, img_w, batch_max, in_channels = 3;
    int math = 0;                            // conv arithmetic the plan (tensor formats) was built for
    bool half_mode = false;                  // Darknet.half(): single-term fp16 operands in the LDS-DMA / window kernels
    void stem_checked_reset() { stem_checked = block1_checked = -1; }
    void set_half(bool on);                  // half mode on / off: re-plans the activation formats (2-byte tensors while it is on)
    void plan_half_formats();
    int owner_of(int layer, int &off) const; // layer -> the storage its view lives in and the channel offset of that view
    int total_boxes = 0, attrs = 0;
    std::vector<Layer> layers;
    std::vector<Storage> storage;
    std::vector<int> yolo_layers;
    DevBuf<float> input, out, stage_f32;
    DevBuf<uint8_t> stage_u8;
    int stage_h = 0, stage_w = 0, stage_n = 0;               // frames last uploaded by forward_u8_host (device copy in stage_u8)
    DevBuf<float> tiled_pred, tile_scale;
    DevBuf<int> tile_rects;
    hipStream_t stream = nullptr;
    int32_t header[5] = {0, 0, 0, 0, 0};
    bool weights_loaded = false;
    size_t activation_bytes = 0;
    // bench-only logit injection
    std::vector<DevBuf<float>> inject_rows;
    std::vector<int> inject_n;
    bool inject_active = false;
    DevBuf<float> inject_table;              // preloaded sets: rows of all (set, image) pairs
    std::vector<int> inject_offsets;         // n_sets * batch_max + 1 row offsets into inject_table
    DevBuf<int> inject_offsets_dev;
    int inject_max_rows = 0;
    int inject_set = -1;
    float inject_logit = 6.f;
    // conv timing: a HIP event pair around every conv launch on this stream, resolved when the counters are read (no host
    // synchronisation inside the pass)
    struct ConvTimeRec { hipEvent_t e0 = nullptr, e1 = nullptr; int variant = 0; double flops = 0, bytes = 0, attain_us = 0; };
    bool time_convs = false;
    std::vector<ConvTimeRec> conv_pending;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t timing_event();
    void resolve_conv_timing();
    double conv_us[kConvVariants] = {}, conv_flops_acc[kConvVariants] = {}, conv_bytes_acc[kConvVariants] = {}, conv_attain_us[kConvVariants] = {};
    int64_t conv_launches[kConvVariants] = {};

private:
    void allocate_buffers();
    void run_graph(int batch);
    // layers [l0, l1) over images [first, first + batch) on one stream (l1 < 0: to the end)
    void run_lane(int first, int batch, hipStream_t st, int l0 = 0, int l1 = -1);
    bool stem_fusable = false, stem_ok = false;               // layers 0+1 as one kernel (conv_stem2.hip)
    int stem_checked = -1;
    bool stem_fused(int batch);
    int block1_at = -1, block1_checked = -1;                  // first conv of the fused residual block (conv_block1.hip), -1: none
    bool block1_ok = false;
    bool block1_fused(int batch);
    hipEvent_t out_guard = nullptr;                           // optional: event the decode waits for before it overwrites `out`
    int lane_img0 = 0;                                        // image offset applied by view() / input_view() while a lane is enqueued
    std::vector<hipStream_t> lane_streams;
    std::vector<hipEvent_t> lane_done;
    hipEvent_t lane_fork = nullptr;
};

// --------------------------------------------------------------------------------------------- NMS
// Device multi-label NMS over decoded predictions [n_boxes, attrs] (nms.hip).
class NmsWorkspace {
public:
    explicit NmsWorkspace(int max_candidates = 16384, int frames = 1);
    ~NmsWorkspace();
    NmsWorkspace(const NmsWorkspace &) = delete;
    // asynchronous form: launch() enqueues the kernels and the copies into pinned host memory, collect() reads
    // them after the caller synchronised the stream
    void launch(const float *pred_dev, size_t pred_stride, int n_frames, int n_boxes, int attrs, float conf_thres, float iou_thres,
                float sx, float sy, int cap, hipStream_t s);
    int collect(int frame, float *out6_host, int cap);
    void resize(int max_candidates, int n_frames);       // (re)allocates; contents are lost
    int needed(int n_frames) const;                       // largest candidate count of the last launch (after the caller's sync)
    static constexpr int kMaxCandidates = 1 << 18;
    // returns number of rows written to out6_host (<= cap); rows sorted by score, boxes in model pixels
    // scaled by (sx, sy) when scale is requested (resize_boxes).
    int run(const float *pred_dev, int n_boxes, int attrs, float conf_thres, float iou_thres, float sx, float sy,
            float *out6_host, int cap, hipStream_t s);
    // merge=True / is_p1p2=True form used by the sliding-window detector: corner-form boxes, the reference's merge branch
    int run_merge(const float *pred_dev, int n_boxes, int attrs, float conf_thres, float iou_thres, float *out6_host, int cap, hipStream_t s);
    int max_cand, frames;
    bool corner = false;         // predictions already hold x1,y1,x2,y2
    DevBuf<float> cand;          // [max_cand, 6]   x1,y1,x2,y2,score,cls in candidate order
    DevBuf<float> sorted;        // [max_cand, 6]   score order
    DevBuf<int> counts;          // [0] n candidates, [1] n kept
    DevBuf<int> box_count;       // per box candidate count / offsets
    DevBuf<unsigned long long> mask;   // [max_cand, max_cand/64] suppression bits
    DevBuf<float> kept;          // [300, 6]
    DevBuf<int> order;
    int *h_counts = nullptr;     // pinned
    float *h_kept = nullptr;     // pinned [300,6]
};

// --------------------------------------------------------------------------------------------- ReID
class ReidNet {
public:
    explicit ReidNet(int max_crops);
    ~ReidNet();
    void load_tensor(const std::string &name, const float *data, const int64_t *shape, int ndim);
    void finalize();
    void embed_dev(const uint8_t *frame_dev, int h, int w, const float *tlwh_host, int D, float *out_host);
    // crops of several frames in one batch: frame_of[d] selects frames_dev + frame_of[d]*h*w*3; asynchronous
    void embed_multi_dev(const uint8_t *frames_dev, int h, int w, const float *tlwh_host, const int *frame_of, int D, bool bgr = false);
    void embed_host(const uint8_t *frame_host, int h, int w, const float *tlwh_host, int D, float *out_host);
    void preprocess_host(const uint8_t *frame_host, int h, int w, const float *tlwh_host, int D, float *nchw_host);
    void forward_f32_host(const float *nchw, int D, float *out_host);
    void forward(int D);                     // input already in `in` (NHWC4)
    void reserve(int D);                     // grows the activation buffers (and `feat`) to hold D crops
    void allocate_buffers();
    static int64_t flops_per_crop();

    struct ConvW {
        int cin = 0, cin_file = 0, cout = 0, k = 0, stride = 1, pad = 0, kpad = 0;
        DevBuf<float> wt, bias;
        DevBuf<uint16_t> wt16;
    };
    int max_crops;
    std::map<std::string, std::vector<float>> raw;
    std::map<std::string, std::vector<int64_t>> raw_shape;
    bool ready = false;
    std::vector<ConvW> convs;                // in execution order
    std::vector<DevBuf<float>> bufs;
    DevBuf<float> in, feat, stage_f32;
    DevBuf<uint8_t> stage_u8;
    DevBuf<int> boxes_dev;
    std::vector<int> boxes_host;
    // crop rectangles of the batched pass, in PINNED host memory the crop kernel reads in place (19 KB for 960 crops): no
    // host-to-device copy command on the stream - a pageable hipMemcpyAsync stalls the host behind everything queued on that
    // stream and shares the copy engine with the frame uploads.  Two buffers alternate (a pass may be enqueued while the crop
    // kernel of the previous one has not run yet).
    int *boxes_pin[2] = {nullptr, nullptr};
    size_t boxes_pin_cap[2] = {0, 0};
    int boxes_pin_turn = 0;
    hipStream_t sync_before_regrow = nullptr;   // set by the pipeline while a pass runs on another stream than the previous one
    std::map<int, std::pair<int, int>> tuned;   // conv index -> (D at measurement, measured tile variant)
    int tuned_math = -1;
    hipStream_t stream = nullptr;
    double conv_flops_last = 0;
};

// --------------------------------------------------------------------------------------------- tracker
// Implemented in tracker.hip; the pipeline drives it through this interface.
struct TrackerIface {
    virtual ~TrackerIface() {}
    // feats: [D,512] on device when feats_on_device, else host.  Returns rows written to out6 (int32 [m,6]).
    virtual int step(const float *tlwh_host, const float *feats, bool feats_on_device, const float *payload_host, int D,
                     int32_t *out6_host, int cap) = 0;
    // the frames of one batch in order with ONE host synchronisation: detections of frame b are rows [first[b], first[b+1])
    // of tlwh / feats_dev / payload; skip[b] != 0: tracker not called for that frame (counts[b] = -1)
    virtual void step_batch(int n, const float *tlwh_host, const int *first, const float *feats_dev, const float *payload_host, const char *skip,
                            int32_t *out6_host, int cap, int32_t *counts) = 0;
    virtual int num_tracks() const = 0;
    // the next step / step_batch starts behind `ev` on the tracker's own stream (features produced on another stream: no host wait)
    virtual void wait_for(hipEvent_t ev) = 0;
};

}  // namespace yds

// C handles (shared by the translation units that implement the ABI)
struct yds_net { yds::Darknet *d; };
struct yds_reid { yds::ReidNet *r; };
struct yds_trk { yds::TrackerIface *t; };
