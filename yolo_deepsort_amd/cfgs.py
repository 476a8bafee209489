"""Programmatic Darknet ``.cfg`` text for the four detector graphs the hot path runs.

The GPU box has no copy of the reference tree, so bench/smoke/tests need the
network descriptions from somewhere.  Rather than carrying cfg files around,
the graphs are described here as code (residual stages, CSP stages, SPP, PAN
necks) and rendered to standard Darknet INI text that any Darknet-cfg parser
accepts, including ``parse_model_config`` (reference
``yolo3/utils/parse_config.py:1-19``).  ``tests/test_oracle_darknet.py`` checks the rendered graphs against
``tests/golden/cfg_parse.json`` - the reference's ``parse_model_config`` output for
``config/{yolov3,yolov3-tiny,yolov4,yolov4-tiny}.cfg`` on every key the
reference reads (``yolo3/models/models.py:29-97``).
"""

from __future__ import annotations

V3_ANCHORS = "10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326"
V3T_ANCHORS = "10,14,  23,27,  37,58,  81,82,  135,169,  344,319"
V4_ANCHORS = "12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401"


class _Cfg:
    def __init__(self, width, height, extra_net=()):
        self.blocks = []
        net = [("batch", "1"), ("subdivisions", "1"), ("width", str(width)),
               ("height", str(height)), ("channels", "3")]
        net += list(extra_net)
        self.net = net

    # each helper appends one block and returns its layer index
    def conv(self, filters, size, stride=1, act="leaky", bn=True):
        kv = []
        if bn:
            kv.append(("batch_normalize", "1"))
        kv += [("filters", str(filters)), ("size", str(size)), ("stride", str(stride)),
               ("pad", "1"), ("activation", act)]
        return self._add("convolutional", kv)

    def shortcut(self, frm=-3):
        return self._add("shortcut", [("from", str(frm)), ("activation", "linear")])

    def route(self, layers, groups=None, group_id=None):
        kv = [("layers", layers)]
        if groups is not None:
            kv += [("groups", str(groups)), ("group_id", str(group_id))]
        return self._add("route", kv)

    def maxpool(self, size, stride):
        return self._add("maxpool", [("size", str(size)), ("stride", str(stride))])

    def upsample(self, stride=2):
        return self._add("upsample", [("stride", str(stride))])

    def yolo(self, mask, anchors, num, extra=()):
        kv = [("mask", mask), ("anchors", anchors), ("classes", "80"), ("num", str(num))]
        kv += list(extra)
        return self._add("yolo", kv)

    def _add(self, typ, kv):
        self.blocks.append((typ, kv))
        return len(self.blocks) - 1

    def render(self):
        out = ["[net]"] + [f"{k}={v}" for k, v in self.net] + [""]
        for typ, kv in self.blocks:
            out.append(f"[{typ}]")
            out += [f"{k}={v}" for k, v in kv]
            out.append("")
        return "\n".join(out) + "\n"


def yolov3_cfg(width=416, height=416):
    c = _Cfg(width, height)
    c.conv(32, 3)
    # Darknet-53: a stride-2 3x3 followed by n bottleneck residual units
    for filters, n in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        c.conv(filters, 3, 2)
        for _ in range(n):
            c.conv(filters // 2, 1)
            c.conv(filters, 3)
            c.shortcut(-3)

    def head(filters, mask):
        for _ in range(3):
            c.conv(filters, 1)
            c.conv(filters * 2, 3)
        c.conv(255, 1, act="linear", bn=False)
        c.yolo(mask, V3_ANCHORS, 9)

    head(512, "6,7,8")
    c.route("-4")
    c.conv(256, 1)
    c.upsample(2)
    c.route("-1, 61")
    head(256, "3,4,5")
    c.route("-4")
    c.conv(128, 1)
    c.upsample(2)
    c.route("-1, 36")
    head(128, "0,1,2")
    return c.render()


def yolov3_tiny_cfg(width=416, height=416):
    c = _Cfg(width, height)
    for f in (16, 32, 64, 128, 256):
        c.conv(f, 3)
        c.maxpool(2, 2)
    c.conv(512, 3)
    c.maxpool(2, 1)
    c.conv(1024, 3)
    c.conv(256, 1)
    c.conv(512, 3)
    c.conv(255, 1, act="linear", bn=False)
    c.yolo("3,4,5", V3T_ANCHORS, 6)
    c.route("-4")
    c.conv(128, 1)
    c.upsample(2)
    c.route("-1, 8")
    c.conv(256, 3)
    c.conv(255, 1, act="linear", bn=False)
    c.yolo("1,2,3", V3T_ANCHORS, 6)
    return c.render()


def yolov4_tiny_cfg(width=416, height=416):
    c = _Cfg(width, height)
    c.conv(32, 3, 2)
    c.conv(64, 3, 2)
    for f in (64, 128, 256):
        c.conv(f, 3)
        c.route("-1", groups=2, group_id=1)
        c.conv(f // 2, 3)
        c.conv(f // 2, 3)
        c.route("-1,-2")
        c.conv(f, 1)
        c.route("-6,-1")
        c.maxpool(2, 2)
    c.conv(512, 3)
    c.conv(256, 1)
    c.conv(512, 3)
    c.conv(255, 1, act="linear", bn=False)
    c.yolo("3,4,5", V3T_ANCHORS, 6, [("scale_x_y", "1.05")])
    c.route("-4")
    c.conv(128, 1)
    c.upsample(2)
    c.route("-1, 23")
    c.conv(256, 3)
    c.conv(255, 1, act="linear", bn=False)
    c.yolo("1,2,3", V3T_ANCHORS, 6, [("scale_x_y", "1.05")])
    return c.render()


def yolov4_cfg(width=608, height=608):
    c = _Cfg(width, height)
    c.conv(32, 3, act="mish")

    # CSPDarknet-53 stage: downsample, split, n residual units, transition, merge
    def csp(filters, n, first=False):
        c.conv(filters, 3, 2, "mish")
        half = filters if first else filters // 2
        c.conv(half, 1, act="mish")
        c.route("-2")
        c.conv(half, 1, act="mish")
        for _ in range(n):
            c.conv(filters // 2 if first else half, 1, act="mish")
            c.conv(half, 3, act="mish")
            c.shortcut(-3)
        c.conv(half, 1, act="mish")
        c.route(f"-1,-{4 + 3 * n}")
        c.conv(filters, 1, act="mish")

    csp(64, 1, first=True)
    csp(128, 2)
    csp(256, 8)
    csp(512, 8)
    csp(1024, 4)

    # SPP neck
    c.conv(512, 1)
    c.conv(1024, 3)
    c.conv(512, 1)
    c.maxpool(5, 1)
    c.route("-2")
    c.maxpool(9, 1)
    c.route("-4")
    c.maxpool(13, 1)
    c.route("-1,-3,-5,-6")
    c.conv(512, 1)
    c.conv(1024, 3)
    c.conv(512, 1)

    # PAN top-down
    def topdown(filters, lateral):
        c.conv(filters, 1)
        c.upsample(2)
        c.route(str(lateral))
        c.conv(filters, 1)
        c.route("-1, -3")
        c.conv(filters, 1)
        c.conv(filters * 2, 3)
        c.conv(filters, 1)
        c.conv(filters * 2, 3)
        c.conv(filters, 1)

    topdown(256, 85)
    topdown(128, 54)

    def head(filters, mask, sxy):
        c.conv(filters * 2, 3)
        c.conv(255, 1, act="linear", bn=False)
        c.yolo(mask, V4_ANCHORS, 9, [("scale_x_y", sxy)])

    def bottomup(filters, skip):
        c.route("-4")
        c.conv(filters, 3, 2)
        c.route(f"-1, {skip}")
        c.conv(filters, 1)
        c.conv(filters * 2, 3)
        c.conv(filters, 1)
        c.conv(filters * 2, 3)
        c.conv(filters, 1)

    head(128, "0,1,2", "1.2")
    bottomup(256, -16)
    head(256, "3,4,5", "1.1")
    bottomup(512, -37)
    head(512, "6,7,8", "1.05")
    return c.render()


CFG_BUILDERS = {
    "yolov3": yolov3_cfg,
    "yolov3-tiny": yolov3_tiny_cfg,
    "yolov4": yolov4_cfg,
    "yolov4-tiny": yolov4_tiny_cfg,
}


def cfg_text(name, width=None, height=None):
    """Return Darknet cfg text for ``name`` (``yolov3``, ``yolov3-tiny``, ``yolov4``, ``yolov4-tiny``)."""
    fn = CFG_BUILDERS[name]
    if width is None:
        return fn()
    return fn(width, height if height is not None else width)


COCO_NAMES = (
    "person bicycle car motorbike aeroplane bus train truck boat traffic_light fire_hydrant "
    "stop_sign parking_meter bench bird cat dog horse sheep cow elephant bear zebra giraffe "
    "backpack umbrella handbag tie suitcase frisbee skis snowboard sports_ball kite "
    "baseball_bat baseball_glove skateboard surfboard tennis_racket bottle wine_glass cup fork "
    "knife spoon bowl banana apple sandwich orange broccoli carrot hot_dog pizza donut cake "
    "chair sofa pottedplant bed diningtable toilet tvmonitor laptop mouse remote keyboard "
    "cell_phone microwave oven toaster sink refrigerator book clock vase scissors teddy_bear "
    "hair_drier toothbrush"
).split()


def coco_names_text():
    """80 COCO class labels, one per line, newline-terminated (``load_classes`` drops the
    last split element, reference ``yolo3/utils/helper.py:8-14``)."""
    return "\n".join(n.replace("_", " ") for n in COCO_NAMES) + "\n"
