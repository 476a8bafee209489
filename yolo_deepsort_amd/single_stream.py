"""ONE video stream on N GPUs (optional mode): the detector and the ReID network are stateless per frame
(reference yolo3/detect/video_detect.py:134-149: `image_detector.detect(frame)` and the extractor inside
`tracker.update`), the association is strictly sequential (ids, galleries, Kalman state).  So frame f is detected and
embedded on rank f % N, every round the ranks all-gather one block per frame

    {count, tlwh[cap][4], payload[cap], feats[cap][512]}        (cap x (16 + 4 + 2048) B; cap starts at 64 = 132 KB and grows)

- the only place of this library where RCCL carries data-path bytes - and rank 0 runs the tracker over the N frames of the
round in frame order.  count = -1 marks "the detector returned None" (the tracker is not called for that frame, like the
reference's loop); count = -(2 + d) announces a frame whose d detections exceed the block: every rank sees it in the gathered
headers, grows its block to hold d and the round's exchange is repeated (like every other capacity of the library)."""

from __future__ import annotations

from functools import reduce

import numpy as np

MIN_DET = 64
EMB = 512


def cap_for(d):
    """Block capacity (detections per frame) that holds d detections: a multiple of MIN_DET."""
    return max(MIN_DET, (int(d) + MIN_DET - 1) // MIN_DET * MIN_DET)


def block_floats(cap):
    return 1 + cap * (4 + 1 + EMB)


def pack_frame(tlwh, payload, feats, cap=MIN_DET):
    """None (no detections object) or ([d,4], [d], [d,512]) -> float32 [block_floats(cap)]"""
    blk = np.zeros(block_floats(cap), np.float32)
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
        blk[o:o + EMB * d] = np.asarray(feats, np.float32).reshape(-1)
    return blk


def dets_needed(blks):
    """Largest detection count any header of the gathered blocks announces."""
    h = np.asarray(blks)[..., 0]
    return int(np.where(h <= -2, -2 - h, np.maximum(h, 0)).max(initial=0))


def unpack_frame(blk):
    cap = (blk.shape[-1] - 1) // (4 + 1 + EMB)
    d = int(blk[0])
    if d <= -2:
        raise ValueError(f"exchange block of {cap} detections cannot hold a frame of {-2 - d}")
    if d < 0:
        return None
    o1 = 1 + 4 * cap
    o2 = o1 + cap
    return blk[1:1 + 4 * d].reshape(d, 4).copy(), blk[o1:o1 + d].copy(), blk[o2:o2 + EMB * d].reshape(d, EMB).copy()


class SingleStream:
    """detect(frame) -> None or (tlwh [d,4], payload [d], feats [d,512]) runs on the rank that owns the frame;
    track(tlwh, payload, feats) -> rows runs on rank 0 only, in frame order."""

    def __init__(self, ranks, detect, track):
        self.ranks, self.detect, self.track = ranks, detect, track
        self.cap = MIN_DET                       # detections per frame the exchange block holds (same value on every rank)

    @classmethod
    def from_components(cls, ranks, image_detector, deepsort, class_mask=None):
        """The reference's per-frame glue split at the tracker boundary (video_detect.py:134-149)."""
        from .detect import p1p2Toxywh
        if getattr(deepsort, "nms_max_overlap", 1) != 1:
            raise ValueError("SingleStream: the tracker-side NMS (nms_max_overlap != 1) is part of DeepSort.update, not of this split")
        model = image_detector.model

        def detect(frame):
            det = image_detector.detect(frame)
            if det is None:
                return None
            det = det.numpy() if hasattr(det, "numpy") else det
            boxs, class_ids = p1p2Toxywh(det[:, :4]).astype(np.float32), det[:, -1]
            if class_mask is not None:
                mask = reduce(lambda a, b: a | b, [class_ids == m for m in class_mask])
                boxs, class_ids = boxs[mask], class_ids[mask]
            dev = model.last_frame_dev(frame) if hasattr(model, "last_frame_dev") else None
            feats = deepsort.extractor.embed(frame, boxs, to_host=True, frame_dev=dev) if len(boxs) else np.zeros((0, EMB), np.float32)
            return boxs, class_ids.astype(np.float32), feats

        def track(tlwh, payload, feats):
            rows = deepsort.tracker.step(tlwh, feats, payload)
            return rows if len(rows) else []

        return cls(ranks, detect, track)

    def run(self, frames):
        """frames: a sequence every rank can index (rank r reads frames r, r + N, ...).  Returns on rank 0 the per-frame results
        in order (None where the detector returned None), on the other ranks an empty list."""
        n, world, rank = len(frames), self.ranks.world, self.ranks.rank
        out = []
        for base in range(0, n, world):
            f = base + rank
            mine = self.detect(frames[f]) if f < n else None
            while True:
                blk = pack_frame(*(mine if mine is not None else (None, None, None)), cap=self.cap)
                allb = self.ranks.gather_array(blk)                # [world, block_floats(cap)]: the round's frames in frame order
                need = dets_needed(allb)
                if need <= self.cap:
                    break
                self.cap = cap_for(need)                           # every rank sees the same headers and grows alike
            if rank == 0:
                for k in range(min(world, n - base)):
                    got = unpack_frame(allb[k])
                    out.append(None if got is None else self.track(*got))
        return out
