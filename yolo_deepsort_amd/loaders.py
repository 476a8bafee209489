"""Host-side loaders kept drop-in with the reference.

parse_model_config: yolo3/utils/parse_config.py:1-19
load_classes:       yolo3/utils/helper.py:8-14
load_reid_checkpoint: Extractor.__init__, deep_sort/deep/feature_extractor.py:13-17
"""

from __future__ import annotations

import numpy as np


def parse_model_config(path, text=None):
    """Parses a Darknet cfg into a list of dicts, exactly like the reference: blank and
    '#' lines dropped, fringe whitespace stripped, values stay strings and
    ``convolutional`` blocks default to ``batch_normalize = 0`` (int)."""
    if text is None:
        with open(path, "r") as f:
            text = f.read()
    module_defs = []
    for line in text.split("\n"):
        if not line or line.startswith("#"):
            continue
        line = line.strip()
        if line.startswith("["):
            block = {"type": line[1:-1].rstrip()}
            if block["type"] == "convolutional":
                block["batch_normalize"] = 0
            module_defs.append(block)
        else:
            key, value = line.split("=")
            module_defs[-1][key.rstrip()] = value.strip()
    return module_defs


def load_classes(path):
    """Class labels, one per line; the file must end with a newline (last split element dropped)."""
    with open(path, "r", encoding="utf-8") as fp:
        return fp.read().split("\n")[:-1]


def load_reid_checkpoint(path):
    """``torch.load(path)['net_dict']`` as {name: float32 ndarray}.  torch is used purely as a
    file-format reader here (both the legacy and the zip checkpoint formats)."""
    import torch
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    except TypeError:
        ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["net_dict"]
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}
