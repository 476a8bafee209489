"""Host-side overlay: ``LabelDrawer`` (reference yolo3/utils/label_draw.py:118-191, helpers :17-60) without cv2.

Same constructor, colour table (``np.random.seed(1)`` stream, :140-146), label strings (``"<track id>:<class>"`` for
tracker rows :176-183, ``"<class> (<conf %>)"`` for raw detections :86-95) and box / label-plate geometry; rectangles and
text are rasterised with numpy (a built-in 5x7 bitmap font stands in for cv2's Hershey face, so glyph pixels differ from
cv2's - this stage is outside the measured path)."""

from __future__ import annotations

import numpy as np

# 5x7 glyphs, one string of 7 rows per character (bit 4 = left column)
_FONT = {
    "0": "0E11131519110E", "1": "040C0404040E00"[:14], "2": "0E11010204081F", "3": "1F02040201110E", "4": "02060A121F0202",
    "5": "1F101E0101110E", "6": "0608101E11110E", "7": "1F010204080808", "8": "0E11110E11110E", "9": "0E11110F01020C",
    ":": "000C0C000C0C00", ".": "0000000000 0C0C".replace(" ", ""), "%": "18190204081303", "(": "02040808080402", ")": "08040202020408",
    " ": "00000000000000", "-": "0000001F000000", "_": "0000000000001F",
    "a": "00000E010F110F", "b": "10101619111 11E".replace(" ", ""), "c": "00000E1010110E", "d": "01010D1311110F", "e": "00000E111F100E",
    "f": "06090 81C080808".replace(" ", ""), "g": "000F110F01110E"[:14], "h": "10101619111111", "i": "0400 0C0404040E".replace(" ", ""),
    "j": "0200060202120C", "k": "10101214181412", "l": "0C04040404040E", "m": "00001A15151111", "n": "00001619111111",
    "o": "00000E1111110E", "p": "00001E111E1010", "q": "00000D130F0101", "r": "00001619101010", "s": "00000E100E011E",
    "t": "08081C0808090 6".replace(" ", ""), "u": "00001111111 30D".replace(" ", ""), "v": "0000111111 0A04".replace(" ", ""),
    "w": "00001111151 50A".replace(" ", ""), "x": "0000110A040A11", "y": "000011110F010E", "z": "00001F0204081F",
}


def _glyph(ch):
    code = _FONT.get(ch.lower(), _FONT["_"])
    rows = [int(code[2 * r:2 * r + 2], 16) for r in range(7)]
    return np.array([[(row >> (4 - c)) & 1 for c in range(5)] for row in rows], dtype=bool)


def text_size(text, scale):
    """(width, height) in pixels of `text` at integer pixel `scale` per font dot (the cv2.getTextSize role)."""
    return (6 * len(text) * scale, 7 * scale)


def put_text(img, text, org, scale, color):
    """Draws `text` with its bottom-left corner at `org` (x, y), clipped to the image."""
    x0, y0 = int(org[0]), int(org[1]) - 7 * scale
    for i, ch in enumerate(text):
        g = np.kron(_glyph(ch), np.ones((scale, scale), dtype=bool))
        gx, gy = x0 + 6 * scale * i, y0
        h, w = g.shape
        ys, xs = max(gy, 0), max(gx, 0)
        ye, xe = min(gy + h, img.shape[0]), min(gx + w, img.shape[1])
        if ye > ys and xe > xs:
            sub = g[ys - gy:ye - gy, xs - gx:xe - gx]
            img[ys:ye, xs:xe][sub] = color
    return img


def rectangle(img, c1, c2, color, thickness):
    """cv2.rectangle semantics: outline centred on the edges for thickness > 0, filled for thickness < 0; clipped."""
    H, W = img.shape[:2]
    x1, x2 = sorted((int(c1[0]), int(c2[0])))
    y1, y2 = sorted((int(c1[1]), int(c2[1])))
    if thickness < 0:
        img[max(y1, 0):max(min(y2 + 1, H), 0), max(x1, 0):max(min(x2 + 1, W), 0)] = color
        return img
    lo, hi = thickness // 2, (thickness - 1) // 2

    def fill(ya, yb, xa, xb):
        ya, yb, xa, xb = max(ya, 0), min(yb, H), max(xa, 0), min(xb, W)
        if yb > ya and xb > xa:
            img[ya:yb, xa:xb] = color
    fill(y1 - lo, y1 + hi + 1, x1 - lo, x2 + hi + 1)
    fill(y2 - lo, y2 + hi + 1, x1 - lo, x2 + hi + 1)
    fill(y1 - lo, y2 + hi + 1, x1 - lo, x1 + hi + 1)
    fill(y1 - lo, y2 + hi + 1, x2 - lo, x2 + hi + 1)
    return img


def draw_rects(img, dets, colors, thickness):
    """label_draw.py:17-28"""
    for det in dets:
        cls = int(det[-1])
        rectangle(img, (int(det[0]), int(det[1])), (int(det[2]), int(det[3])), colors[cls % len(colors)], thickness)
    return img


def draw_rects_and_labels(img, dets, colors, labels, thickness, font_size):
    """label_draw.py:31-60 (the cv2 fallback-font branch): box, filled plate above its top-left corner, black text."""
    scale = max(1, int(round(2 * font_size)))
    for det, label in zip(dets, labels):
        cls = int(det[-1])
        color = colors[cls % len(colors)]
        c1, c2 = (int(det[0]), int(det[1])), (int(det[2]), int(det[3]))
        rectangle(img, c1, c2, color, thickness)
        fw, fh = text_size(label, scale)
        rectangle(img, (c1[0], max(0, int(c1[1] - 3 - fh))), (c1[0] + fw, max(c1[1], int(3 + fh))), color, -1)
        put_text(img, label, (c1[0], max(c1[1] - 3, fh)), scale, (0, 0, 0))
    return img


class LabelDrawer:
    def __init__(self, classes, font_path, font_size, thickness, img_size, statistic=False, id2label=None):
        self.thickness = thickness
        self.statistic = statistic
        self.classes = classes
        self.img_size = img_size
        self.font_size = font_size
        self.id2label = id2label
        self.font_path = font_path
        self.font = None                                     # cv2.freetype faces are not available without cv2
        num_classes = len(self.classes)
        rng = np.random.RandomState(1)                       # the np.random.seed(1) stream of label_draw.py:140-146, global state untouched
        colors = (rng.rand(min(999, num_classes), 3) * 255).astype(int)
        self.colors = [(int(c[0]), int(c[1]), int(c[2])) for c in colors]

    def clone(self):
        return LabelDrawer(self.classes, self.font_path, self.font_size, self.thickness, self.img_size, self.statistic, None)

    def draw_labels(self, img, detections, only_rect, scaled=True):
        """Raw detector rows [n,6] or [n,7] (label_draw.py:63-106)."""
        if detections is None:
            return img, None, None
        det = detections.cpu().float().numpy() if hasattr(detections, "cpu") else np.asarray(detections, dtype=np.float32)
        if only_rect:
            draw_rects(img, det, self.colors, self.thickness)
        else:
            labels = []
            for d in det:
                conf = d[-3] * d[-2] if len(d) == 7 else d[-2]
                labels.append(self.classes[int(d[-1])] + " (" + str(round(float(conf) * 100, 2)) + "%)")
            draw_rects_and_labels(img, det, self.colors, labels, self.thickness, img.shape[0] / 1000.)
        return img, None, None

    def draw_labels_by_trackers(self, img, detections, only_rect):
        """Tracker rows int32 [m,6] = x1,y1,x2,y2,track id,class (label_draw.py:171-191)."""
        if only_rect:
            draw_rects(img, detections, self.colors, self.thickness)
        else:
            labels = []
            for d in detections:
                key = str(int(d[4]))
                name = self.id2label[key] if self.id2label is not None and key in self.id2label else self.classes[int(d[-1])]
                labels.append(key + ":" + name)
            draw_rects_and_labels(img, detections, self.colors, labels, self.thickness, img.shape[0] / 1000.)
        return img, None, None


# ---- device form of draw_labels_by_trackers + RGB -> BGR + FPS text (csrc/overlay.hip, yds_overlay_tracks) ---------------------------
_GLYPH_ORDER = sorted(_FONT)
_GLYPH_CODE = {ch: i for i, ch in enumerate(_GLYPH_ORDER)}


def font_table():
    """uint8 [n_glyphs, 7]: the row bytes of the 5 x 7 glyphs in code order (bit 4 = left column)."""
    return np.array([[int(_FONT[ch][2 * r:2 * r + 2], 16) for r in range(7)] for ch in _GLYPH_ORDER], dtype=np.uint8)


def encode_text(text):
    """Glyph codes of `text` as put_text draws it (lower-cased, unknown characters as '_')."""
    return np.array([_GLYPH_CODE.get(ch.lower(), _GLYPH_CODE["_"]) for ch in text], dtype=np.uint8)


class _PinnedBlock:
    """A pinned host block (yds_host_alloc) whose lifetime follows the numpy arrays made of it: np.asarray(block) and every
    view of that keep the block alive; it is handed back to the runtime when the last one is gone."""

    def __init__(self, nbytes):
        from . import _lib
        self.nbytes = int(nbytes)
        self.ptr = _lib.check_ptr(_lib.load().yds_host_alloc(self.nbytes))
        self.__array_interface__ = dict(shape=(self.nbytes,), typestr="|u1", data=(self.ptr, False), version=3)

    def __del__(self):
        try:
            from . import _lib
            if self.ptr:
                _lib.load().yds_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class DeviceOverlay:
    """LabelDrawer.draw_labels_by_trackers for a whole batch of frames that are resident in HBM, with the generator's RGB -> BGR
    conversion and FPS text (video_detect.py:161-186) in the same pass.  Pixel-identical to this module's host functions."""

    def __init__(self, drawer):
        self.drawer = drawer
        self.font = font_table()
        self._labels = {}                # (track id, class) -> glyph codes
        self._out_dev = None
        self._pool = []                  # pinned result blocks; one is reused once no array handed out still refers to it

    def _pinned(self, nbytes):
        import sys
        for blk in self._pool:
            if blk.nbytes >= nbytes and sys.getrefcount(blk) <= 3:          # the list, the loop variable, getrefcount's argument
                return blk
        if len(self._pool) >= 4:                                            # keep the pool small; blocks still referenced live on with their arrays
            self._pool = [b for b in self._pool if sys.getrefcount(b) > 3][-3:]
        blk = _PinnedBlock(nbytes)
        self._pool.append(blk)
        return blk

    def _label(self, tid, cls):
        # id2label is consulted on EVERY frame like the host path (label_draw.py:174-176 of the reference): a name given to a track
        # after it was first drawn shows from the next frame on; the cache only saves the glyph encoding of an unchanged string
        d = self.drawer
        name = d.id2label[str(tid)] if d.id2label is not None and str(tid) in d.id2label else d.classes[cls]
        key = (tid, name)
        codes = self._labels.get(key)
        if codes is None:
            codes = self._labels[key] = encode_text(str(tid) + ":" + name)
            if len(self._labels) > 100000:
                self._labels.clear()
        return codes

    def render(self, frames_dev, src_slots, h, w, holds, fps_texts=None, only_rect=False, bgr_frames=0):
        """frames_dev: device pointer of uint8 RGB frames, h*w*3 bytes apart; output i shows frame src_slots[i] with the tracker
        rows holds[i] (int32 [m,6], or None / empty: nothing drawn).  Returns a list of BGR uint8 [h,w,3] arrays (views of one
        pinned block that stays alive as long as any of them does).
        bgr_frames = n > 0: the n staged frames are ALREADY BGR (a decoder's order): they are drawn on in place and copied out - no
        reversed device copy (yds_overlay_tracks_bgr); the staged frames are consumed."""
        from . import _lib
        n = len(holds)
        if n == 0:
            return []
        d = self.drawer
        colors, ncol = d.colors, len(d.colors)
        boxes, ptr, texts, toff = [], [0], [], 0
        for hold in holds:
            if hold is not None and len(hold):
                for r in np.asarray(hold, dtype=np.int64).reshape(-1, 6).tolist():
                    c = colors[r[5] % ncol]
                    if only_rect:
                        off, ln = 0, -1
                    else:
                        codes = self._label(r[4], r[5])
                        off, ln = toff, len(codes)
                        texts.append(codes)
                        toff += ln
                    boxes.append((r[0], r[1], r[2], r[3], c[0] | (c[1] << 8) | (c[2] << 16), off, ln, 0))
            ptr.append(len(boxes))
        fps = np.zeros((n, 2), np.int32)
        if fps_texts is not None:
            for i, t in enumerate(fps_texts):
                if t:
                    codes = encode_text(t)
                    fps[i] = (toff, len(codes))
                    texts.append(codes)
                    toff += len(codes)
        text = np.ascontiguousarray(np.concatenate(texts) if texts else np.zeros(1, np.uint8))
        boxes_a = np.ascontiguousarray(np.array(boxes, dtype=np.int32).reshape(-1, 8)) if boxes else np.zeros((1, 8), np.int32)
        ptr_a = np.ascontiguousarray(ptr, dtype=np.int32)
        slots = np.ascontiguousarray(src_slots, dtype=np.int32)
        nbytes = n * h * w * 3
        blk = self._pinned(nbytes)
        scale = max(1, int(round(2 * (h / 1000.))))                      # draw_rects_and_labels' font_size = img.shape[0] / 1000.
        if bgr_frames:
            _lib.check(_lib.load().yds_overlay_tracks_bgr(frames_dev, int(bgr_frames), _lib.ptr(slots), n, h, w, _lib.ptr(boxes_a), _lib.ptr(ptr_a),
                                                          _lib.ptr(text), toff, _lib.ptr(fps), _lib.ptr(self.font), len(self.font),
                                                          int(d.thickness), scale, blk.ptr))
            out = np.asarray(blk)[:nbytes].reshape(n, h, w, 3)
            del blk
            return [out[i] for i in range(n)]
        if self._out_dev is None or self._out_dev.nbytes < nbytes:
            self._out_dev = _lib.DeviceBuffer(nbytes)
        _lib.check(_lib.load().yds_overlay_tracks(frames_dev, _lib.ptr(slots), n, h, w, _lib.ptr(boxes_a), _lib.ptr(ptr_a), _lib.ptr(text), toff,
                                                  _lib.ptr(fps), _lib.ptr(self.font), len(self.font), int(d.thickness), scale,
                                                  self._out_dev.ptr, blk.ptr))
        out = np.asarray(blk)[:nbytes].reshape(n, h, w, 3)
        del blk
        return [out[i] for i in range(n)]
