"""ONE video stream on N GPUs (optional mode): the detector and the ReID network are stateless per frame
(reference yolo3/detect/video_detect.py:134-149: `image_detector.detect(frame)` and the extractor inside
`tracker.update`), the association is strictly sequential (ids, galleries, Kalman state).  So frame f is detected and
embedded on rank f % N, every round the ranks all-gather one fixed-size block per frame

    {count, tlwh[MAX_DET][4], payload[MAX_DET], feats[MAX_DET][512]}        (<= 150 x (16 + 4 + 2048) B = 310 KB)

- the only place of this library where RCCL carries data-path bytes - and rank 0 runs the tracker over the N frames of the
round in frame order.  count = -1 marks "the detector returned None" (the tracker is not called for that frame, like the
reference's loop)."""

from __future__ import annotations

from functools import reduce

import numpy as np

MAX_DET = 150
EMB = 512
BLOCK_FLOATS = 1 + MAX_DET * (4 + 1 + EMB)


def pack_frame(tlwh, payload, feats):
    """None (no detections object) or ([d,4], [d], [d,512]) -> float32 [BLOCK_FLOATS]"""
    blk = np.zeros(BLOCK_FLOATS, np.float32)
    if tlwh is None:
        blk[0] = -1.0
        return blk
    d = int(len(tlwh))
    if d > MAX_DET:
        raise ValueError(f"{d} detections in one frame exceed the exchange block ({MAX_DET})")
    blk[0] = d
    if d:
        blk[1:1 + 4 * d] = np.asarray(tlwh, np.float32).reshape(-1)
        o = 1 + 4 * MAX_DET
        blk[o:o + d] = np.asarray(payload, np.float32).reshape(-1)
        o += MAX_DET
        blk[o:o + EMB * d] = np.asarray(feats, np.float32).reshape(-1)
    return blk


def unpack_frame(blk):
    d = int(blk[0])
    if d < 0:
        return None
    o1 = 1 + 4 * MAX_DET
    o2 = o1 + MAX_DET
    return blk[1:1 + 4 * d].reshape(d, 4).copy(), blk[o1:o1 + d].copy(), blk[o2:o2 + EMB * d].reshape(d, EMB).copy()


class SingleStream:
    """detect(frame) -> None or (tlwh [d,4], payload [d], feats [d,512]) runs on the rank that owns the frame;
    track(tlwh, payload, feats) -> rows runs on rank 0 only, in frame order."""

    def __init__(self, ranks, detect, track):
        self.ranks, self.detect, self.track = ranks, detect, track

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
            blk = pack_frame(*(mine if mine is not None else (None, None, None)))
            allb = self.ranks.gather_array(blk)                    # [world, BLOCK_FLOATS]: the round's frames in frame order
            if rank == 0:
                for k in range(min(world, n - base)):
                    got = unpack_frame(allb[k])
                    out.append(None if got is None else self.track(*got))
        return out
