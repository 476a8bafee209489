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
