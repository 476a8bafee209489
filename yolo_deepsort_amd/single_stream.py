"""ONE video stream on N GPUs (optional mode): the detector and the ReID network are stateless per frame
(reference yolo3/detect/video_detect.py:134-149: `image_detector.detect(frame)` and the extractor inside
`tracker.update`), the association is strictly sequential (ids, galleries, Kalman state).  So frame f is detected and
embedded on rank f % N, every round the ranks all-gather one block per frame

    {count, tlwh[cap][4], payload[cap], feats[cap][512]}        (cap x (16 + 4 + 2048) B; cap starts at 64 = 132 KB and grows)

- the only place of this library where RCCL carries data-path bytes - and rank 0 runs the tracker over the N frames of the
round in frame order.  count = -1 marks "the detector returned None" (the tracker is not called for that frame, like the
reference's loop); count = -(2 + d) announces a frame whose d detections exceed the block: every rank sees it in the gathered
headers, grows its block to hold d and the round's exchange is repeated (like every other capacity of the library).

Round 4: a round carries `frames_per_rank` consecutive frames per rank (one exchange per N * B frames), and with a live RCCL
communicator the embeddings never leave HBM: every rank keeps its frames' feature rows in a device staging block, the small
header block {count, tlwh, payload} travels as before, the feature blocks are all-gathered device to device
(`yds_comm_allgather_dev`) and rank 0's tracker reads each frame's rows straight from the gathered buffer
(`yds_tracker_step_sel(..., feats_on_device = 1)`)."""

from __future__ import annotations

from functools import reduce

import numpy as np

MIN_DET = 64
EMB = 512


def cap_for(d):
    """Block capacity (detections per frame) that holds d detections: a multiple of MIN_DET."""
    return max(MIN_DET, (int(d) + MIN_DET - 1) // MIN_DET * MIN_DET)


def block_floats(cap, emb=EMB):
    return 1 + cap * (4 + 1 + emb)


def pack_frame(tlwh, payload, feats, cap=MIN_DET, emb=EMB):
    """None (no detections object) or ([d,4], [d], [d,512]) -> float32 [block_floats(cap)]; emb = 0: header block only (the
    feature rows travel device to device)"""
    blk = np.zeros(block_floats(cap, emb), np.float32)
    if tlwh is None:
        blk[0] = -1.0
        return blk
    d = int(len(tlwh))
    if d > cap:
        blk[0] = -(2.0 + d)                  # does not fit: the receivers grow the block and the round is exchanged again
        return blk
    blk[0] = d
    if d:
        blk[1:1 + 4 * d] = np.asarray(tlwh, np.float32).reshape(-1)
        o = 1 + 4 * cap
        blk[o:o + d] = np.asarray(payload, np.float32).reshape(-1)
        o += cap
        if emb:
            blk[o:o + emb * d] = np.asarray(feats, np.float32).reshape(-1)
    return blk


def dets_needed(blks):
    """Largest detection count any header of the gathered blocks announces."""
    h = np.asarray(blks)[..., 0]
    return int(np.where(h <= -2, -2 - h, np.maximum(h, 0)).max(initial=0))


def unpack_frame(blk, emb=EMB):
    cap = (blk.shape[-1] - 1) // (4 + 1 + emb)
    d = int(blk[0])
    if d <= -2:
        raise ValueError(f"exchange block of {cap} detections cannot hold a frame of {-2 - d}")
    if d < 0:
        return None
    o1 = 1 + 4 * cap
    o2 = o1 + cap
    return blk[1:1 + 4 * d].reshape(d, 4).copy(), blk[o1:o1 + d].copy(), (blk[o2:o2 + emb * d].reshape(d, emb).copy() if emb else None)


def _dev_block(nbytes):
    """Raw HBM block owned by Python (freed with its owner)."""
    from . import _lib
    return _lib.DeviceBuffer(max(int(nbytes), 16))


class SingleStream:
    """detect(frame) -> None or (tlwh [d,4], payload [d], feats [d,512]) runs on the rank that owns the frame;
    track(tlwh, payload, feats) -> rows runs on rank 0 only, in frame order.
    Device form (round 4): detect_dev(frame) -> None or (tlwh, payload, device pointer of d feature rows, valid until the next
    call), track_dev(tlwh, payload, device pointer) - used when both are given and the ranks hold an RCCL communicator."""

    def __init__(self, ranks, detect, track, frames_per_rank=1, detect_dev=None, track_dev=None):
        self.ranks, self.detect, self.track = ranks, detect, track
        self.detect_dev, self.track_dev = detect_dev, track_dev
        self.B = max(1, int(frames_per_rank))
        self.cap = MIN_DET                       # detections per frame the exchange block holds (same value on every rank)
        self.on_device = detect_dev is not None and track_dev is not None and getattr(ranks, "comm", None) is not None
        self._stage = self._all = None           # device blocks: [B][cap][512] of this rank, [world][B][cap][512] gathered
        self._stage_cap = 0                      # rows per slot the device blocks are laid out for
        self._held = [0] * self.B                # feature rows held per slot in the current round
        self.exchanges = 0                       # rounds exchanged (a repeated round counts once per exchange)

    @classmethod
    def from_components(cls, ranks, image_detector, deepsort, class_mask=None, frames_per_rank=1, device=True):
        """The reference's per-frame glue split at the tracker boundary (video_detect.py:134-149)."""
        from .detect import p1p2Toxywh
        if getattr(deepsort, "nms_max_overlap", 1) != 1:
            raise ValueError("SingleStream: the tracker-side NMS (nms_max_overlap != 1) is part of DeepSort.update, not of this split")
        model = image_detector.model

        def boxes_of(frame):
            det = image_detector.detect(frame)
            if det is None:
                return None
            det = det.numpy() if hasattr(det, "numpy") else det
            boxs, class_ids = p1p2Toxywh(det[:, :4]).astype(np.float32), det[:, -1]
            if class_mask is not None:
                mask = reduce(lambda a, b: a | b, [class_ids == m for m in class_mask])
                boxs, class_ids = boxs[mask], class_ids[mask]
            return boxs, class_ids.astype(np.float32), (model.last_frame_dev(frame) if hasattr(model, "last_frame_dev") else None)

        def detect(frame):
            got = boxes_of(frame)
            if got is None:
                return None
            boxs, cls_f, dev = got
            feats = deepsort.extractor.embed(frame, boxs, to_host=True, frame_dev=dev) if len(boxs) else np.zeros((0, EMB), np.float32)
            return boxs, cls_f, feats

        def detect_dev(frame):
            got = boxes_of(frame)
            if got is None:
                return None
            boxs, cls_f, dev = got
            if len(boxs):
                deepsort.extractor.embed(frame, boxs, to_host=False, frame_dev=dev)     # rows stay in the extractor's device buffer
            return boxs, cls_f, (deepsort.extractor.features_dev() if len(boxs) else None)

        def track(tlwh, payload, feats):
            rows = deepsort.tracker.step(tlwh, feats, payload)
            return rows if len(rows) else []

        def track_dev(tlwh, payload, feats_ptr):
            rows = deepsort.tracker.step(tlwh, None, payload, feats_dev=feats_ptr if len(tlwh) else None)
            return rows if len(rows) else []

        return cls(ranks, detect, track, frames_per_rank, detect_dev if device else None, track_dev if device else None)

    # ---- device staging -------------------------------------------------------------------------------------------------------
    # One persistent block [B][cap][512] per rank: a frame's feature rows go from the extractor's buffer (complete when detect_dev
    # returns: yds_reid_embed_dev synchronises its stream) STRAIGHT into their slot - no per-frame allocation, no device-wide
    # synchronisation.  The block is re-laid out only when the rows per slot grow (a frame with more detections than
    # any before, or the cap the ranks agree on in the header exchange).
    def _relayout(self, cap, upto):
        """Make the staging block hold `cap` rows per slot, keeping the rows already held in slots < upto."""
        from . import _lib
        row = EMB * 4
        if self._stage is not None and self._stage_cap == cap:
            return
        old, old_cap = self._stage, self._stage_cap
        self._stage = _dev_block(self.B * cap * row)
        self._all = _dev_block(self.ranks.world * self.B * cap * row)
        self._stage_cap = cap
        if old is not None:
            lib = _lib.load()
            for s in range(upto):
                if self._held[s]:
                    _lib.check(lib.yds_memcpy_d2d(self._stage.offset(s * cap * row), old.offset(s * old_cap * row), self._held[s] * row))

    def _hold(self, b, ptr, d):
        """Keep the d feature rows at device pointer `ptr` (the extractor's buffer, overwritten by the next frame) in slot b."""
        from . import _lib
        row = EMB * 4
        cap = max(self._stage_cap, self.cap)
        self._relayout(cap if d <= cap else cap_for(d), upto=b)
        _lib.check(_lib.load().yds_memcpy_d2d(self._stage.offset(b * self._stage_cap * row), ptr, d * row))
        self._held[b] = d

    def run(self, frames):
        """frames: a sequence every rank can index (rank r reads frames base + r * B .. + B of every round of N * B frames).  Returns
        on rank 0 the per-frame results in order (None where the detector returned None), on the other ranks an empty list."""
        from . import _lib
        n, world, rank, B = len(frames), self.ranks.world, self.ranks.rank, self.B
        row = EMB * 4
        out = []
        for base in range(0, n, world * B):
            mine = []
            self._held = [0] * B
            for b in range(B):
                f = base + rank * B + b
                if f >= n:
                    mine.append(None)
                elif self.on_device:
                    got = self.detect_dev(frames[f])
                    if got is not None and got[2] is not None and len(got[0]):
                        self._hold(b, got[2], len(got[0]))          # the extractor's buffer is overwritten by the next frame
                        got = (got[0], got[1], None)
                    mine.append(got)
                else:
                    mine.append(self.detect(frames[f]))
            while True:
                if self.on_device:
                    hdr = [pack_frame(*((m[0], m[1], np.zeros((len(m[0]), 0), np.float32)) if m is not None else (None, None, None)), cap=self.cap, emb=0)
                           for m in mine]
                else:
                    hdr = [pack_frame(*(m if m is not None else (None, None, None)), cap=self.cap) for m in mine]
                allb = self.ranks.gather_array(np.stack(hdr, 0))       # [world, B, block]: the round's frames in frame order
                self.exchanges += 1
                need = dets_needed(allb)
                if need <= self.cap:
                    break
                self.cap = cap_for(need)                           # every rank sees the same headers and grows alike
            if self.on_device:
                # the agreed cap is the largest any rank ever needed, so it is never below this rank's own layout; one layout everywhere
                assert self.cap >= self._stage_cap
                self._relayout(self.cap, upto=B)
                _lib.check(_lib.load().yds_comm_allgather_dev(self.ranks.comm, self._stage.offset(0), self.B * self.cap * row, self._all.offset(0)))
            if rank == 0:
                for k in range(min(world * B, n - base)):
                    r, b = divmod(k, B)
                    if self.on_device:
                        got = unpack_frame(allb[r, b], emb=0)
                        out.append(None if got is None else self.track_dev(got[0], got[1], self._all.offset((r * B + b) * self.cap * row)))
                    else:
                        got = unpack_frame(allb[r, b])
                        out.append(None if got is None else self.track(*got))
        return out
