// Per-frame hot glue: detect -> NMS -> class mask -> p1p2Toxywh -> ReID -> tracker, in frame order.
//
// Mirrors the loop body of VideoDetector.detect, reference yolo3/detect/video_detect.py:134-157:
//   * ImageDetector.detect (img_detect.py:61-95): resize + /255 + Darknet + NMS + resize_boxes;
//   * the tracker is NOT called when the detector returns None (video_detect.py:137), but IS called
//     with D = 0 when the class mask empties the list (:141-149);
//   * boxes go to the tracker as (x1, y1, x2-x1, y2-y1) fp32 (p1p2Toxywh, model_build.py:326-332),
//     payload = class id.
// Scheduling (results are identical to the frame-by-frame loop, only the order of independent work
// changes):  the detector is stateless, so it runs over a whole batch of frames; NMS for all frames of
// the batch is enqueued behind it; the ReID CNN embeds the crops of ALL frames of the batch in one
// launch sequence; only the association consumes frames strictly in order.  While ReID + association of
// batch i run on their streams, the detector of batch i+1 (if the caller already has those frames) is
// enqueued on the detector stream, so the small latency-bound tracker kernels and their host syncs hide
// under MFMA work.
#include "engine.h"

#include <chrono>

namespace yds {

class Pipeline {
public:
    Pipeline(Darknet *net, ReidNet *reid, TrackerIface *trk, float conf, float nms_iou, const int32_t *mask, int n_mask)
        : net(net), reid(reid), trk(trk), conf(conf), nms_thres(nms_iou), class_mask(mask, mask + n_mask) {
        for (hipEvent_t *e : {&e0, &e1, &e2, &e_nms}) YDS_HIP(hipEventCreate(e));
        nms.reset(new NmsWorkspace(4096, net->batch_max));
    }
    ~Pipeline() {
        for (hipEvent_t e : {e0, e1, e2, e_nms}) (void)hipEventDestroy(e);
    }

    void launch_detector(const uint8_t *frames_dev, int h, int w, int batch) {
        YDS_HIP(hipEventRecord(e0, net->stream));
        launch_resize_u8(frames_dev, batch, h, w, net->input_view(batch), net->stream);
        YDS_HIP(hipEventRecord(e1, net->stream));
        net->forward_resized(batch);
        YDS_HIP(hipEventRecord(e2, net->stream));
        in_flight = frames_dev;
        in_flight_batch = batch;
    }

    void step(const uint8_t *frames_dev, const uint8_t *next_frames_dev, int next_inject_set, int h, int w, int batch, int32_t *out6,
              int cap, int32_t *counts) {
        using clk = std::chrono::steady_clock;
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<float, std::micro>(b - a).count(); };
        if (batch < 1 || batch > net->batch_max) fail("pipeline: batch %d outside [1,%d]", batch, net->batch_max);
        auto t_begin = clk::now();
        if (in_flight != frames_dev || in_flight_batch != batch) launch_detector(frames_dev, h, w, batch);
        in_flight = nullptr;
        // NMS of every frame behind the detector, results into pinned host buffers
        const float sx = (float)((double)w / net->img_w), sy = (float)((double)h / net->img_h);
        nms->launch(net->out.p, (size_t)net->total_boxes * net->attrs, batch, net->total_boxes, net->attrs, conf, nms_thres, sx, sy, 300,
                    net->stream);
        YDS_HIP(hipEventRecord(e_nms, net->stream));
        YDS_HIP(hipEventSynchronize(e_nms));
        float ms01 = 0, ms12 = 0;
        YDS_HIP(hipEventElapsedTime(&ms01, e0, e1));
        YDS_HIP(hipEventElapsedTime(&ms12, e1, e2));
        auto t_nms = clk::now();
        // detector of the next batch goes in flight now: it overwrites `out` only after the NMS kernels above
        if (next_frames_dev) {
            if (next_inject_set >= 0) net->select_injection_set(next_inject_set);      // bench-only logit injection
            launch_detector(next_frames_dev, h, w, batch);
        }
        // host glue: class mask + p1p2Toxywh for all frames
        std::vector<float> det(300 * 6);
        tlwh.clear(); payload.clear(); frame_of.clear();
        std::vector<int> first(batch + 1, 0), n_det(batch, 0);
        for (int b = 0; b < batch; ++b) {
            n_det[b] = nms->collect(b, det.data(), 300);
            for (int i = 0; i < n_det[b]; ++i) {
                const float *r = &det[i * 6];
                bool keep = class_mask.empty();
                for (int m : class_mask) keep |= (r[5] == (float)m);
                if (!keep) continue;
                tlwh.push_back(r[0]); tlwh.push_back(r[1]); tlwh.push_back(r[2] - r[0]); tlwh.push_back(r[3] - r[1]);
                payload.push_back(r[5]);
                frame_of.push_back(b);
            }
            first[b + 1] = (int)payload.size();
        }
        // one ReID pass over the crops of the whole batch (chunked by the extractor's capacity)
        const int D_all = (int)payload.size();
        if (D_all > reid->max_crops) fail("pipeline: %d crops in one batch exceed the extractor capacity %d", D_all, reid->max_crops);
        if (D_all) {
            reid->embed_multi_dev(frames_dev, h, w, tlwh.data(), frame_of.data(), D_all);
            YDS_HIP(hipStreamSynchronize(reid->stream));
        }
        auto t_reid = clk::now();
        for (int b = 0; b < batch; ++b) {
            if (n_det[b] == 0) { counts[b] = -1; continue; }       // detector returned None: tracker not called
            const int D = first[b + 1] - first[b];
            counts[b] = trk->step(tlwh.data() + (size_t)first[b] * 4, reid->feat.p + (size_t)first[b] * 512, true,
                                  payload.data() + first[b], D, out6 + (size_t)b * cap * 6, cap);
        }
        auto t_end = clk::now();
        stage_us[0] = ms01 * 1e3f; stage_us[1] = ms12 * 1e3f;
        stage_us[2] = us(t_begin, t_nms); stage_us[3] = us(t_nms, t_reid); stage_us[4] = us(t_reid, t_end);
    }

    Darknet *net;
    ReidNet *reid;
    TrackerIface *trk;
    float conf, nms_thres;
    std::vector<int32_t> class_mask;
    std::unique_ptr<NmsWorkspace> nms;
    std::vector<float> tlwh, payload;
    std::vector<int> frame_of;
    int next_inject_set = -1;      // bench-only: injection set of the prefetched detector pass
    const uint8_t *in_flight = nullptr;
    int in_flight_batch = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e_nms = nullptr;
    float stage_us[5] = {0, 0, 0, 0, 0};
};

}  // namespace yds

struct yds_pipe { yds::Pipeline *p; };

extern "C" {

yds_pipe *yds_pipeline_create(yds_net *n, yds_reid *r, yds_trk *t, float conf_thres, float nms_thres, const int32_t *class_mask, int n_mask) {
    YDS_API_BEGIN
    if (!n || !r || !t) yds::fail("pipeline: NULL handle");
    return new yds_pipe{new yds::Pipeline(n->d, r->r, t->t, conf_thres, nms_thres, class_mask, class_mask ? n_mask : 0)};
    YDS_API_END_PTR
}
void yds_pipeline_destroy(yds_pipe *p) {
    if (p) { delete p->p; delete p; }
}
int yds_pipeline_step(yds_pipe *p, const uint8_t *frames_dev, const uint8_t *next_frames_dev, int h, int w, int batch, int32_t *out6_host,
                      int cap, int32_t *counts_host) {
    YDS_API_BEGIN
    p->p->step(frames_dev, next_frames_dev, p->p->next_inject_set, h, w, batch, out6_host, cap, counts_host);
    p->p->next_inject_set = -1;
    YDS_API_END
}
int yds_pipeline_set_next_injection(yds_pipe *p, int set) {
    YDS_API_BEGIN
    p->p->next_inject_set = set;
    YDS_API_END
}
int yds_pipeline_stage_us(yds_pipe *p, float *us5) {
    YDS_API_BEGIN
    for (int i = 0; i < 5; ++i) us5[i] = p->p->stage_us[i];
    YDS_API_END
}

}  // extern "C"
