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

#include <limits.h>
#include <stdlib.h>

#include <chrono>

namespace yds {

class Pipeline {
public:
    Pipeline(Darknet *net, ReidNet *reid, TrackerIface *trk, float conf, float nms_iou, const int32_t *mask, int n_mask)
        : net(net), reid(reid), trk(trk), conf(conf), nms_thres(nms_iou), class_mask(mask, mask + n_mask) {
        for (int k = 0; k < 2; ++k) {
            for (hipEvent_t *e : {&e0[k], &e1[k], &e2[k], &e_nms[k]}) YDS_HIP(hipEventCreate(e));
            nms[k].reset(new NmsWorkspace(4096, net->batch_max));
        }
        for (int k = 0; k < NSTAGE; ++k)
            for (hipEvent_t *e : {&up_done[k], &rd_det[k], &rd_reid[k]}) YDS_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        YDS_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        YDS_HIP(hipEventCreateWithFlags(&ev_feat, hipEventDisableTiming));
        YDS_HIP(hipEventCreateWithFlags(&ev_reid_done, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k)
            for (hipEvent_t *e : {&e_r0[k], &e_r1[k]}) YDS_HIP(hipEventCreate(e));
    }
    ~Pipeline() {
        for (int k = 0; k < 2; ++k) {
            for (hipEvent_t e : {e0[k], e1[k], e2[k], e_nms[k]}) (void)hipEventDestroy(e);
        }
        for (int k = 0; k < NSTAGE; ++k)
            for (hipEvent_t e : {up_done[k], rd_det[k], rd_reid[k]}) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(copy_stream);
        (void)hipEventDestroy(ev_feat);
        (void)hipEventDestroy(ev_reid_done);
        for (int k = 0; k < 2; ++k)
            for (hipEvent_t e : {e_r0[k], e_r1[k]}) (void)hipEventDestroy(e);
    }

    // ---- frames handed over as HOST memory (img_detect.py:70-71 starts from a host frame) ------------------------------
    // Three device staging buffers take turns; the copy runs on its own stream (SDMA engine), so uploads overlap the detector /
    // ReID / association of earlier batches.  A batch is matched by its host pointer for a BOUNDED time: the slot is forgotten
    // when the step that consumed it returns, and an announced batch that is never consumed is forgotten too (`next` must be the
    // following call's `frames`, a prefetch_host batch must be consumed within two calls: stage_ttl) - so a caller may reuse a
    // host buffer for new frames and a later buffer at the same address can never match stale device frames.  A slot is only
    // overwritten after the kernels that read it (the detector's resize, the extractor's crops) have passed: the copy stream
    // waits on their events.  Pinned source memory (yds_host_alloc) makes the copy asynchronous and full speed.
    // Depth: step_host(frames, next) starts the upload of `next` when it is called - but the detector stream runs a whole pass
    // ahead of the host chain (NMS -> ReID -> association), so it wants `next` at that very moment and would idle for the
    // 1.7 ms of a 100 MB copy (1352 vs 1447 frames/s).  prefetch_host(frames of the step after next) starts that copy one
    // step earlier; step_host then finds both of its batches resident.
    int staged(const uint8_t *host) const {
        for (int k = 0; k < NSTAGE; ++k)
            if (host && stage_host[k] == host) return k;
        return -1;
    }
    int upload(const uint8_t *host, size_t bytes, int ttl, int keep_a = -1, int keep_b = -1) {
        int k = -1;
        for (int pass = 0; pass < 2 && k < 0; ++pass)               // next slot in turn: an empty one first, else any not in use
            for (int t = 1; t <= NSTAGE; ++t) {
                const int c = (stage_turn + t) % NSTAGE;
                if (c == keep_a || c == keep_b || (pass == 0 && stage_host[c])) continue;
                k = c;
                break;
            }
        if (k < 0) fail("pipeline: no staging buffer free");
        stage_turn = k;
        // the previous tenant's readers (a prefetched detector pass, an early ReID pass) may still be running on their streams
        if (rd_det_set[k]) { YDS_HIP(hipStreamWaitEvent(copy_stream, rd_det[k], 0)); rd_det_set[k] = false; }
        if (rd_reid_set[k]) { YDS_HIP(hipStreamWaitEvent(copy_stream, rd_reid[k], 0)); rd_reid_set[k] = false; }
        if (stage[k].n < bytes) {                                    // (re)allocation frees the old buffer: its readers must be done
            YDS_HIP(hipStreamSynchronize(net->stream));
            YDS_HIP(hipStreamSynchronize(reid->stream));
        }
        stage[k].ensure(bytes);
        stage_ttl[k] = ttl;
        YDS_HIP(hipMemcpyAsync(stage[k].p, host, bytes, hipMemcpyHostToDevice, copy_stream));
        YDS_HIP(hipEventRecord(up_done[k], copy_stream));
        stage_host[k] = host;
        up_pending[k] = true;
        return k;
    }
    void prefetch_host(const uint8_t *frames_host, int h, int w, int batch) {
        if (staged(frames_host) >= 0) return;
        upload(frames_host, (size_t)batch * h * w * 3, 3, cur_k, next_k);      // survives this call's step and the next: consumed by the one after
    }
    void step_host(const uint8_t *frames_host, const uint8_t *next_host, int next_inject_set, int h, int w, int batch, int32_t *out6, int cap,
                   int32_t *counts) {
        const size_t bytes = (size_t)batch * h * w * 3;
        // batches handed over earlier (as `next` of the previous call, or through prefetch_host) are already resident or on their way
        cur_k = staged(frames_host);
        if (cur_k < 0) cur_k = upload(frames_host, bytes, 1, staged(next_host));
        next_k = -1;
        if (next_host) {
            next_k = staged(next_host);
            if (next_k < 0) next_k = upload(next_host, bytes, 2, cur_k);              // survives this step: the next call's `frames`
        }
        step(stage[cur_k].p, next_k >= 0 ? stage[next_k].p : nullptr, next_inject_set, h, w, batch, out6, cap, counts, true);
        // every host buffer handed over so far may be reused by the caller when this returns; the consumed batch is forgotten
        for (int k = 0; k < NSTAGE; ++k)
            if (up_pending[k]) { YDS_HIP(hipEventSynchronize(up_done[k])); up_pending[k] = false; }
        stage_host[cur_k] = nullptr;
        for (int k = 0; k < NSTAGE; ++k)                            // announced but never consumed: forget it (its address may be reused)
            if (k != cur_k && stage_host[k] && --stage_ttl[k] <= 0) stage_host[k] = nullptr;
        cur_k = -1;
    }
    int slot_of(const uint8_t *frames_dev) const {
        for (int b = 0; b < NSTAGE; ++b)
            if (stage[b].p && frames_dev == stage[b].p) return b;
        return -1;
    }

    // One detector pass over a batch AND its NMS, all asynchronous on the detector stream.  Two NMS workspaces (and
    // their pinned result buffers) alternate, so that the pass of batch i+1 can be enqueued before the host has waited
    // for and read the results of batch i: the detector stream never drains between passes.
    void launch_detector(const uint8_t *frames_dev, int h, int w, int batch, int slot = -1) {
        launch_detector_head(frames_dev, h, w, batch, slot, false);
        launch_detector_tail(frames_dev, h, w, batch);
    }
    // The pass in two pieces (serialized schedule, see step()): head = upload wait + resize + the first layers
    // (Darknet::head_layers), tail = the remaining layers + NMS.  split = false (or a network that cannot be split) puts the whole
    // network into the tail.
    void launch_detector_head(const uint8_t *frames_dev, int h, int w, int batch, int slot, bool split) {
        const int k = slot >= 0 ? slot : (in_flight_slot ^= 1);
        head_slot = k;
        reid_in_pass[k] = false;                                    // (a head re-recorded for this slot: no ReID interval of an earlier pass belongs to it)
        for (int b = 0; b < NSTAGE; ++b)                            // frames uploaded by step_host: wait for the copy engine
            if (stage[b].p && frames_dev == stage[b].p && hipEventQuery(up_done[b]) != hipSuccess)
                YDS_HIP(hipStreamWaitEvent(net->stream, up_done[b], 0));   // (only while the copy is still running: see step())
        YDS_HIP(hipEventRecord(e0[k], net->stream));
        launch_resize_u8(frames_dev, batch, h, w, net->input_view(batch), net->stream, frames_bgr);
        YDS_HIP(hipEventRecord(e1[k], net->stream));
        if (const int sl = slot_of(frames_dev); sl >= 0) { YDS_HIP(hipEventRecord(rd_det[sl], net->stream)); rd_det_set[sl] = true; }
        head_split = split && net->forward_resized_part(batch, 0);
    }
    void launch_detector_tail(const uint8_t *frames_dev, int h, int w, int batch) {
        const int k = head_slot;
        if (head_split) (void)net->forward_resized_part(batch, 1);
        else net->forward_resized(batch);
        YDS_HIP(hipEventRecord(e2[k], net->stream));
        const float sx = (float)((double)w / net->img_w), sy = (float)((double)h / net->img_h);
        nms[k]->launch(net->out.p, (size_t)net->total_boxes * net->attrs, batch, net->total_boxes, net->attrs, conf, nms_thres, sx, sy, 300,
                       net->stream);
        YDS_HIP(hipEventRecord(e_nms[k], net->stream));
        in_flight = frames_dev;
        in_flight_batch = batch;
    }

    // detections of one batch after NMS + class mask + p1p2Toxywh, ready for the extractor and the tracker
    struct Dets {
        std::vector<float> tlwh, payload;
        std::vector<int> frame_of, first, n_det;
        const uint8_t *frames = nullptr;
        int batch = 0;
        bool reid_in_flight = false;
        hipStream_t reid_on = nullptr;            // stream the ReID pass of this batch was enqueued on
    };

    // wait for the detector pass + NMS enqueued in slot k, build the detection lists
    void finish_detector(Dets &d, int k, const uint8_t *frames_dev, int batch) {
        YDS_HIP(hipEventSynchronize(e_nms[k]));
        if (nms[k]->needed(batch) > nms[k]->max_cand) {
            // More candidates than the workspace holds (the reference has no limit): grow it and redo this batch.  The
            // prefetched pass of the next batch may already have overwritten the predictions, so the detector runs again
            // after that pass has drained (its NMS results sit in the other slot's pinned buffers and stay valid).
            // Rare slow path; bench-only logit injection is not re-selected for it.
            YDS_HIP(hipStreamSynchronize(net->stream));
            nms[k]->resize(nms[k]->needed(batch), nms[k]->frames);
            const uint8_t *keep = in_flight;
            const int keep_batch = in_flight_batch, keep_head = head_slot;
            const bool keep_split = head_split;
            launch_detector(frames_dev, last_h, last_w, batch, k);
            in_flight = keep; in_flight_batch = keep_batch;      // the prefetched pass (if any) is still the one in flight
            head_slot = keep_head; head_split = keep_split;
            head_stale = true;                                   // ... but a head enqueued for the next pass has been overwritten
            YDS_HIP(hipEventSynchronize(e_nms[k]));
        }
        float ms01 = 0, ms12 = 0;
        YDS_HIP(hipEventElapsedTime(&ms01, e0[k], e1[k]));
        YDS_HIP(hipEventElapsedTime(&ms12, e1[k], e2[k]));
        if (reid_in_pass[k]) {
            // serialized schedule: the ReID pass of the PREVIOUS batch ran on this stream between the head and the tail of this
            // pass (between e1 and e2); its own event pair takes it out of the detector's figure again, so that stage_us[1]
            // means the same under both schedules
            float msr = 0;
            YDS_HIP(hipEventElapsedTime(&msr, e_r0[k], e_r1[k]));
            ms12 -= msr;
            reid_in_pass[k] = false;
        }
        stage_us[0] = ms01 * 1e3f; stage_us[1] = ms12 * 1e3f;
        NmsWorkspace *nmsw = nms[k].get();
        std::vector<float> det(300 * 6);
        d.tlwh.clear(); d.payload.clear(); d.frame_of.clear();
        d.first.assign(batch + 1, 0); d.n_det.assign(batch, 0);
        d.frames = frames_dev; d.batch = batch; d.reid_in_flight = false;
        for (int b = 0; b < batch; ++b) {
            d.n_det[b] = nmsw->collect(b, det.data(), 300);
            for (int i = 0; i < d.n_det[b]; ++i) {
                const float *r = &det[i * 6];
                bool keep = class_mask.empty();
                for (int m : class_mask) keep |= (r[5] == (float)m);
                if (!keep) continue;
                d.tlwh.push_back(r[0]); d.tlwh.push_back(r[1]); d.tlwh.push_back(r[2] - r[0]); d.tlwh.push_back(r[3] - r[1]);
                d.payload.push_back(r[5]);
                d.frame_of.push_back(b);
            }
            d.first[b + 1] = (int)d.payload.size();
        }
    }
    // one ReID pass over the crops of the whole batch, asynchronous on the extractor's stream
    // `on` = the detector's stream: the pass is SERIALIZED with the detector passes (stream order) instead of sharing the CUs
    // with them from the extractor's own stream
    // inside_pass >= 0: the pass sits between the head and the tail of the detector pass in NMS slot `inside_pass` (serialized
    // schedule) and gets an event pair of its own (see finish_detector).
    void launch_reid(Dets &d, int h, int w, hipStream_t on = nullptr, int inside_pass = -1) {
        d.reid_on = on ? on : reid->stream;
        if (!d.payload.empty()) {
            struct Swap { hipStream_t &s; hipStream_t keep; ~Swap() { s = keep; } } swap{reid->stream, reid->stream};
            // Both streams' passes use the extractor's ONE set of buffers (input, activations, features, pinned crop list).  A pass
            // on another stream than the previous one (the schedule changed, or an announced batch was abandoned) is ordered
            // behind it explicitly; on the same stream the stream order does it.
            if (reid_last_on && reid_last_on != d.reid_on) YDS_HIP(hipStreamWaitEvent(d.reid_on, ev_reid_done, 0));
            reid->stream = d.reid_on;
            if (reid_last_on && reid_last_on != d.reid_on) reid->sync_before_regrow = reid_last_on;
            if (inside_pass >= 0) YDS_HIP(hipEventRecord(e_r0[inside_pass], d.reid_on));
            reid->embed_multi_dev(d.frames, h, w, d.tlwh.data(), d.frame_of.data(), (int)d.payload.size(), frames_bgr);
            reid->sync_before_regrow = nullptr;
            if (inside_pass >= 0) { YDS_HIP(hipEventRecord(e_r1[inside_pass], d.reid_on)); reid_in_pass[inside_pass] = true; }
            YDS_HIP(hipEventRecord(ev_reid_done, d.reid_on));
            reid_last_on = d.reid_on;
            if (const int sl = slot_of(d.frames); sl >= 0) { YDS_HIP(hipEventRecord(rd_reid[sl], d.reid_on)); rd_reid_set[sl] = true; }
        }
        d.reid_in_flight = true;
    }

    // ---- stream schedule by measurement (round 5) ----------------------------------------------------------------------
    // Which schedule is faster is a property of the box (round 4: serialized +1.2 % on one, two-stream +1.3-3.1 % on four others),
    // so the pipeline times both on the caller's own steps, like conv_autotune times tile variants: steady-state steps (a next
    // batch handed over, >= 256 crops) run in groups of TRIAL_SKIP + TRIAL_STEPS - serialized, two-stream, serialized, two-stream; the first
    // step of a group absorbs the transition - the wall time of the three measured steps is taken between the returns of step(), and the
    // schedule whose better group is shorter is kept (20 steps in all; round 6: see trial_step_done).
    // Results do not depend on the schedule (parity tests run both), so the trial is invisible to the caller.  One decision per
    // entry (frames resident in HBM / uploaded inside the step): their balance differs.
    static constexpr int TRIAL_STEPS = 3, TRIAL_GROUPS = 4;      // groups alternate serialized / two-stream: S T S T
    // unmeasured steps at the head of a group (round 6: two - with ONE the step after a switch still ran short on work the other
    // schedule had left in flight, and serialized measured 7 % faster in the trial where the steady rates were equal)
    static constexpr int TRIAL_SKIP = 2, TRIAL_LEN = TRIAL_SKIP + TRIAL_STEPS;
    struct Trial {
        int n = 0;                      // steady-state steps seen
        double t0 = 0, t_serial = 0, t_two = 0;   // t0: start of the running group; t_*: the better group of each schedule (3 measured steps)
        int decided = 0;                // 0 = measuring, 1 = serialized, -1 = two-stream
    };
    // Schedule of the NEXT ReID pass for an entry while its trial runs.  A group = TRIAL_SKIP transition steps + TRIAL_STEPS measured steps;
    // the groups alternate (serialized first) and each schedule is measured twice, once earlier and once later in the run, so that
    // the clock / temperature drift of the first second under load (the first group ran 7 % faster than steady state on one box)
    // does not decide the comparison.
    bool trial_wants_serial(const Trial &t) const { return t.decided ? t.decided > 0 : (t.n / TRIAL_LEN) % 2 == 0; }
    // A group is timed by the wall clock between the RETURNS of step() - the steady-state period, which is what the schedules differ in
    // (round 6 first timed the seconds spent INSIDE step(): that is not the period - the serialized schedule returns earlier relative to
    // the device's work - and it preferred serialized by 8 % where the frame rates were equal; profiles/r06_bench_cfg3.json of that tree).
    // Per schedule the BETTER of its two groups counts (round 6): one hiccup of the caller inside a three-step group - a slow
    // decoder, a consumer rendering - no longer fixes the decision; a steadily slow caller stretches both schedules alike.
    void trial_step_done(Trial &t) {
        using clk = std::chrono::steady_clock;
        if (t.decided) return;
        const double now = std::chrono::duration<double>(clk::now().time_since_epoch()).count();
        const int group = t.n / TRIAL_LEN, k = t.n % TRIAL_LEN;
        if (k == TRIAL_SKIP - 1) t.t0 = now;                      // the group's transition steps have returned
        if (k == TRIAL_LEN - 1) {
            double &best = group % 2 == 0 ? t.t_serial : t.t_two;
            best = best > 0 ? std::min(best, now - t.t0) : now - t.t0;
        }
        ++t.n;
        if (t.n == TRIAL_GROUPS * TRIAL_LEN) t.decided = t.t_serial <= t.t_two ? 1 : -1;
    }

    void step(const uint8_t *frames_dev, const uint8_t *next_frames_dev, int next_inject_set, int h, int w, int batch, int32_t *out6,
              int cap, int32_t *counts, bool uploaded = false) {
        using clk = std::chrono::steady_clock;
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<float, std::micro>(b - a).count(); };
        if (batch < 1 || batch > net->batch_max) fail("pipeline: batch %d outside [1,%d]", batch, net->batch_max);
        auto t_begin = clk::now();
        last_h = h; last_w = w;
        const bool resumed = ahead.reid_in_flight && ahead.frames == frames_dev && ahead.batch == batch;
        // Serialized schedule (round 4).  The ReID pass of batch i and the detector pass of batch i+1 are both chip-filling
        // sequences of matrix-core kernels; from two streams they time-share the CUs, every launch stretched by the other
        // stream's work (1.3x on the detector's kernels at cfg2) for the same total.  With >= serial_min crops in the batch the
        // ReID pass is enqueued on the DETECTOR's stream instead, between the head of the next pass (resize + first layers,
        // already queued while the host waited for this batch's NMS and built the crop list: the stream never drains) and its
        // tail: every conv kernel then has the chip to itself, a launch takes its isolated time, and only the association's
        // small kernels (own high-priority stream) run beside them.  Smaller batches (the frame-by-frame API) keep the two-stream
        // form - a 30-crop ReID pass cannot fill the chip and gains from running beside the detector.
        // Measured on the final tree (tools/ab_serial.sh, tools/ab_upload.sh, profiles/r04_serial_schedule_ab.txt; alternating runs on
        // one box), two-stream -> serialized, 32 frames per step, frames resident in HBM:
        //   cfg2 yolov3 1597-1604 -> 1614-1626 frames/s (+1.2 %), window kernel 423 -> 303 us per launch in the pipeline (isolated: 316)
        //   cfg3 yolov4 1499-1501 -> 1515 (+1.0 %), cfg5 yolov4 crowd 682-693 -> 689-690 (equal), exact-fp32 cfg2 573-575 -> 565-566 (-1.5 %)
        // (before the association ran as three launches per frame - round 4 - its ~450 small launches per batch gained from the
        //  two-stream form on yolov4: 1456 -> 1415 then.)  With the frames uploaded inside the step (yds_pipeline_step_host) the
        // serialized form LOSES 4 % (1469-1508 -> 1426-1436: +1.3 ms per step on the detector's stream that neither the start time of
        // the copy nor the result read-back explains - open), so that entry keeps two streams.
        // A detector in half mode keeps two streams as well: its pass is half as long, the (default-arithmetic) ReID pass is 40 % of
        // the step, and its kernels are no longer power bound - running beside the ReID network's gains 3.5 % there (cfg2 --half,
        // alternating runs on one box: 2210-2214 serialized, 2287-2295 two-stream).
        // Policy: serialize from 256 crops per batch when the frames are already in HBM and the detector runs the default arithmetic.
        // yds_pipeline_set_schedule / YDS_PIPE_SERIAL=<crops> force a threshold for either entry, -1 = never.
        // Policy (round 5): yds_pipeline_set_schedule / YDS_PIPE_SERIAL=<crops> force a threshold (-1 = never serialize); otherwise a
        // ReID pass of >= 256 crops takes the schedule the trial measured faster on THIS box (serialized while none has been
        // decided: see Trial), smaller passes keep two streams.
        const bool forced = schedule_min_crops != INT_MIN || getenv("YDS_PIPE_SERIAL");
        Trial &trial = trials[uploaded ? 1 : 0];
        const int serial_min = schedule_min_crops != INT_MIN ? schedule_min_crops
                               : getenv("YDS_PIPE_SERIAL")  ? atoi(getenv("YDS_PIPE_SERIAL"))
                               : (trial_wants_serial(trial) ? 256 : -1);
        int next_slot = -1;
        bool next_head_only = false;
        auto launch_next = [&](bool head_only) {                    // detector (+ NMS) of the next batch goes in flight
            if (!next_frames_dev) return;
            if (next_inject_set >= 0) net->select_injection_set(next_inject_set);      // bench-only logit injection
            if (head_only) {
                head_stale = false;
                launch_detector_head(next_frames_dev, h, w, batch, -1, true);
                next_head_only = true;
            } else {
                launch_detector(next_frames_dev, h, w, batch);
            }
            next_slot = in_flight_slot;
        };
        auto launch_next_tail = [&]() {
            if (!next_head_only) return;
            if (head_stale) launch_detector_head(next_frames_dev, h, w, batch, head_slot, true);    // (overwritten by a redone pass)
            launch_detector_tail(next_frames_dev, h, w, batch);
            next_head_only = false;
        };
        auto copy_feats = [&]() {
            // the tracker reads its own copy so that the extractor can start on the next batch during the association
            if (cur.payload.empty()) return;
            feat_cur.ensure((size_t)reid->max_crops * 512);
            YDS_HIP(hipMemcpyAsync(feat_cur.p, reid->feat.p, cur.payload.size() * 512 * sizeof(float), hipMemcpyDeviceToDevice, cur.reid_on));
            YDS_HIP(hipEventRecord(ev_feat, cur.reid_on));
        };
        bool serial = false;
        if (resumed) {
            std::swap(cur, ahead);                                  // NMS done and ReID already running since the previous call
            ahead.reid_in_flight = false;
            serial = cur.reid_on == net->stream;
            copy_feats();                                           // (behind that ReID pass, ahead of the next detector pass)
            launch_next(false);
        } else {
            ahead.reid_in_flight = false;
            if (in_flight != frames_dev || in_flight_batch != batch) launch_detector(frames_dev, h, w, batch);
            const int slot = in_flight_slot;
            in_flight = nullptr;
            launch_next(serial_min >= 0);                           // enqueued BEFORE the host waits for this batch's NMS
            finish_detector(cur, slot, frames_dev, batch);
            serial = serial_min >= 0 && (int)cur.payload.size() >= std::max(serial_min, 1);
            if (serial) {
                launch_reid(cur, h, w, net->stream, next_head_only ? head_slot : -1);
                copy_feats();
                launch_next_tail();
            } else {
                launch_next_tail();
                launch_reid(cur, h, w);
                copy_feats();
            }
        }
        last_serial = serial;
        auto t_nms = clk::now();
        const int D_all = (int)cur.payload.size();
        // One frame per step (the frame-by-frame API): the association is ordered behind the features ON THE DEVICE - its stream waits
        // for the event - instead of by a host wake-up between the two (round 6: one round trip less on the latency path; stage_us[3]
        // then holds the enqueue only and stage_us[4] the ReID pass + association).  Batches keep the host wait: it is what lets the
        // host start the next batch's work in the right order below.
        if (D_all) { if (batch == 1 && !next_frames_dev) trk->wait_for(ev_feat); else YDS_HIP(hipEventSynchronize(ev_feat)); }
        auto t_reid = clk::now();
        // Crowded scenes (the association of a batch takes long and is all small latency-bound kernels and host syncs):
        // before associating, finish the next batch's detector + NMS and start its ReID pass, so that the matrix
        // cores stay busy underneath.  Sparse scenes keep the simpler order (the detector alone covers the association).
        static const int deep_min = getenv("YDS_PIPE_DEEP_MIN") ? atoi(getenv("YDS_PIPE_DEEP_MIN")) : 64;    // detections per frame
        if (next_frames_dev && D_all >= deep_min * batch) {
            finish_detector(ahead, next_slot, next_frames_dev, batch);
            in_flight = nullptr;
            launch_reid(ahead, h, w, serial ? net->stream : nullptr);
        }
        // association of the whole batch, frame after frame on the tracker's stream, one host synchronisation
        std::vector<char> skip(batch, 0);
        for (int b = 0; b < batch; ++b) skip[b] = cur.n_det[b] == 0;  // detector returned None: tracker not called (video_detect.py:137)
        trk->step_batch(batch, cur.tlwh.data(), cur.first.data(), feat_cur.p, cur.payload.data(), skip.data(), out6, cap, counts);
        auto t_end = clk::now();
        stage_us[2] = us(t_begin, t_nms); stage_us[3] = us(t_nms, t_reid); stage_us[4] = us(t_reid, t_end);
        // a steady-state step of a chip-filling ReID pass counts towards the schedule trial of its entry
        if (!forced && next_frames_dev && D_all >= 256) trial_step_done(trial);
    }

    Darknet *net;
    ReidNet *reid;
    TrackerIface *trk;
    float conf, nms_thres;
    std::vector<int32_t> class_mask;
    std::unique_ptr<NmsWorkspace> nms[2];
    int in_flight_slot = 0;
    int last_h = 0, last_w = 0;
    static constexpr int NSTAGE = 3;
    DevBuf<uint8_t> stage[NSTAGE];            // device copies of host frames (step_host / prefetch_host)
    const uint8_t *stage_host[NSTAGE] = {nullptr, nullptr, nullptr};
    bool up_pending[NSTAGE] = {false, false, false};
    int stage_ttl[NSTAGE] = {0, 0, 0};        // step_host calls an unconsumed slot may still survive
    hipEvent_t rd_det[NSTAGE] = {}, rd_reid[NSTAGE] = {};     // recorded behind the last kernels that read a slot
    bool rd_det_set[NSTAGE] = {false, false, false}, rd_reid_set[NSTAGE] = {false, false, false};
    int cur_k = -1, next_k = -1;
    hipEvent_t up_done[NSTAGE] = {};
    hipStream_t copy_stream = nullptr;
    int stage_turn = 0;
    Dets cur, ahead;                // this batch; the next batch when its ReID pass was started early
    DevBuf<float> feat_cur;
    int next_inject_set = -1;      // bench-only: injection set of the prefetched detector pass
    const uint8_t *in_flight = nullptr;
    int in_flight_batch = 0;
    hipEvent_t e0[2] = {}, e1[2] = {}, e2[2] = {}, e_nms[2] = {};
    hipEvent_t ev_feat = nullptr;      // this batch's embeddings have been copied for the tracker
    int schedule_min_crops = INT_MIN;  // yds_pipeline_set_schedule: crops per batch from which the ReID pass is serialized (INT_MIN: policy)
    Trial trials[2];                   // schedule trial per entry: [0] frames resident in HBM, [1] uploaded inside the step
    hipEvent_t ev_reid_done = nullptr; // behind the last ReID pass, on the stream it ran on
    hipStream_t reid_last_on = nullptr;
    hipEvent_t e_r0[2] = {}, e_r1[2] = {};     // around a ReID pass enqueued inside the detector pass of NMS slot k
    bool reid_in_pass[2] = {false, false};
    bool last_serial = false;          // schedule of the last step
    bool frames_bgr = false;           // yds_pipeline_set_frame_order: the frames handed over hold B, G, R bytes (a decoder's order)
    int head_slot = 0;                 // NMS slot of the pass whose head was enqueued last
    bool head_split = false, head_stale = false;
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
int yds_pipeline_step_host(yds_pipe *p, const uint8_t *frames_host, const uint8_t *next_frames_host, int h, int w, int batch,
                           int32_t *out6_host, int cap, int32_t *counts_host) {
    YDS_API_BEGIN
    p->p->step_host(frames_host, next_frames_host, p->p->next_inject_set, h, w, batch, out6_host, cap, counts_host);
    p->p->next_inject_set = -1;
    YDS_API_END
}
int yds_pipeline_prefetch_host(yds_pipe *p, const uint8_t *frames_host, int h, int w, int batch) {
    YDS_API_BEGIN
    p->p->prefetch_host(frames_host, h, w, batch);
    YDS_API_END
}
int yds_pipeline_set_next_injection(yds_pipe *p, int set) {
    YDS_API_BEGIN
    p->p->next_inject_set = set;
    YDS_API_END
}
int yds_pipeline_set_frame_order(yds_pipe *p, int bgr) {
    YDS_API_BEGIN
    p->p->frames_bgr = bgr != 0;
    YDS_API_END
}
int yds_pipeline_set_schedule(yds_pipe *p, int min_crops) {
    YDS_API_BEGIN
    p->p->schedule_min_crops = min_crops < -1 ? INT_MIN : min_crops;
    YDS_API_END
}
int yds_pipeline_last_schedule(yds_pipe *p) {
    return p && p->p->last_serial ? 1 : 0;
}
int yds_pipeline_schedule_trial(yds_pipe *p, int uploaded, int *decided, double *serialized_s, double *two_stream_s) {
    YDS_API_BEGIN
    const yds::Pipeline::Trial &t = p->p->trials[uploaded ? 1 : 0];
    if (decided) *decided = t.decided;
    if (serialized_s) *serialized_s = t.t_serial;
    if (two_stream_s) *two_stream_s = t.t_two;
    YDS_API_END
}
int yds_pipeline_stage_us(yds_pipe *p, float *us5) {
    YDS_API_BEGIN
    for (int i = 0; i < 5; ++i) us5[i] = p->p->stage_us[i];
    YDS_API_END
}

}  // extern "C"
