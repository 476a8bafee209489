// Per-frame hot glue: detect -> NMS -> class mask -> p1p2Toxywh -> ReID -> tracker, in frame order.
//
// Mirrors the loop body of VideoDetector.detect, reference yolo3/detect/video_detect.py:134-157:
//   * ImageDetector.detect (img_detect.py:61-95): resize + /255 + Darknet + NMS + resize_boxes;
//   * the tracker is NOT called when the detector returns None (video_detect.py:137), but IS called
//     with D = 0 when the class mask empties the list (:141-149);
//   * boxes go to the tracker as (x1, y1, x2-x1, y2-y1) fp32 (p1p2Toxywh, model_build.py:326-332),
//     payload = class id.
// The detector runs over the whole batch of frames first (it is stateless); association then consumes
// the frames strictly in order.
#include "engine.h"

#include <chrono>

namespace yds {

class Pipeline {
public:
    Pipeline(Darknet *net, ReidNet *reid, TrackerIface *trk, float conf, float nms, const int32_t *mask, int n_mask)
        : net(net), reid(reid), trk(trk), conf(conf), nms_thres(nms), class_mask(mask, mask + n_mask) {
        YDS_HIP(hipEventCreate(&e0));
        YDS_HIP(hipEventCreate(&e1));
        YDS_HIP(hipEventCreate(&e2));
    }
    ~Pipeline() {
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    }

    void step(const uint8_t *frames_dev, int h, int w, int batch, int32_t *out6, int cap, int32_t *counts) {
        using clk = std::chrono::steady_clock;
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<float, std::micro>(b - a).count(); };
        YDS_HIP(hipEventRecord(e0, net->stream));
        launch_resize_u8(frames_dev, batch, h, w, net->input_view(batch), net->stream);
        YDS_HIP(hipEventRecord(e1, net->stream));
        net->forward_resized(batch);
        YDS_HIP(hipEventRecord(e2, net->stream));
        float t_nms = 0, t_reid = 0, t_assoc = 0;
        std::vector<float> det(300 * 6), tlwh, payload;
        for (int b = 0; b < batch; ++b) {
            auto t0 = clk::now();
            float sx = (float)((double)w / net->img_w), sy = (float)((double)h / net->img_h);
            const float *pred = net->out.p + (size_t)b * net->total_boxes * net->attrs;
            int n = nms.run(pred, net->total_boxes, net->attrs, conf, nms_thres, sx, sy, det.data(), 300, net->stream);
            auto t1 = clk::now();
            t_nms += us(t0, t1);
            if (n == 0) { counts[b] = -1; continue; }
            tlwh.clear(); payload.clear();
            for (int i = 0; i < n; ++i) {
                const float *r = &det[i * 6];
                bool keep = class_mask.empty();
                for (int m : class_mask) keep |= (r[5] == (float)m);
                if (!keep) continue;
                tlwh.push_back(r[0]); tlwh.push_back(r[1]); tlwh.push_back(r[2] - r[0]); tlwh.push_back(r[3] - r[1]);
                payload.push_back(r[5]);
            }
            const int D = (int)payload.size();
            const uint8_t *frame = frames_dev + (size_t)b * h * w * 3;
            if (D) {
                reid->embed_dev(frame, h, w, tlwh.data(), D, nullptr);
                YDS_HIP(hipStreamSynchronize(reid->stream));
            }
            auto t2 = clk::now();
            t_reid += us(t1, t2);
            counts[b] = trk->step(tlwh.data(), reid->feat.p, true, payload.data(), D, out6 + (size_t)b * cap * 6, cap);
            t_assoc += us(t2, clk::now());
        }
        float ms01 = 0, ms12 = 0;
        YDS_HIP(hipEventElapsedTime(&ms01, e0, e1));
        YDS_HIP(hipEventElapsedTime(&ms12, e1, e2));
        stage_us[0] = ms01 * 1e3f; stage_us[1] = ms12 * 1e3f; stage_us[2] = t_nms; stage_us[3] = t_reid; stage_us[4] = t_assoc;
    }

    Darknet *net;
    ReidNet *reid;
    TrackerIface *trk;
    float conf, nms_thres;
    std::vector<int32_t> class_mask;
    NmsWorkspace nms;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
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
int yds_pipeline_step(yds_pipe *p, const uint8_t *frames_dev, int h, int w, int batch, int32_t *out6_host, int cap, int32_t *counts_host) {
    YDS_API_BEGIN
    p->p->step(frames_dev, h, w, batch, out6_host, cap, counts_host);
    YDS_API_END
}
int yds_pipeline_stage_us(yds_pipe *p, float *us5) {
    YDS_API_BEGIN
    for (int i = 0; i < 5; ++i) us5[i] = p->p->stage_us[i];
    YDS_API_END
}

}  // extern "C"
