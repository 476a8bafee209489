"""Oracle (numpy restatement) vs golden vectors produced by the real reference."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, golden, has_reference
from oracle.cfg import parse_model_config_text
from oracle.darknet import DarknetOracle, conv_flops
from oracle import nms as onms
from yolo_deepsort_amd import cfgs, synth

F32 = np.float32
RTOL, ATOL = 1e-3, 1e-3      # north_star: bbox / embedding tensors within 1e-3 fp32


def _oracle_net(cfg_text, size, seed, obj_bias=-4.0):
    net = DarknetOracle(cfg_text, size, is_text=True)
    blob = synth.darknet_weights_blob(cfg_text, seed, obj_bias)
    w = np.frombuffer(blob, dtype=F32, offset=20)
    used = net.load_weights_array(w)
    assert used == w.size == net.n_weight_floats()
    return net


def test_cfg_generators_match_reference_parse():
    ref = json.load(open(os.path.join(GOLD, "cfg_parse.json")))
    for name, ref_defs in ref.items():
        mine = parse_model_config_text(cfgs.cfg_text(name))
        assert len(mine) == len(ref_defs), name
        for i, (a, b) in enumerate(zip(mine, ref_defs)):
            for k, v in b.items():
                if k in ("height", "width"):
                    continue            # resolution is a generator argument
                av = a.get(k)
                if k in ("anchors", "layers", "mask"):
                    av, v = av.replace(" ", ""), v.replace(" ", "")
                assert str(av) == str(v), (name, i, k, av, v)


def test_weight_counts_match_darknet_files():
    # canonical float counts of the published .weights files
    for name, n in (("yolov3-tiny", 8858734), ("yolov3", 62001757), ("yolov4", 64429405), ("yolov4-tiny", 6062814)):
        assert DarknetOracle(cfgs.cfg_text(name), 416, is_text=True).n_weight_floats() == n


def test_conv_flops_match_survey():
    assert abs(conv_flops(DarknetOracle(cfgs.cfg_text("yolov3"), 608, is_text=True), 608, 608) / 1e9 - 140.692) < 0.01
    assert abs(conv_flops(DarknetOracle(cfgs.cfg_text("yolov4"), 608, is_text=True), 608, 608) / 1e9 - 128.389) < 0.01


def test_mini_darknet_every_layer():
    from oracle.gen_golden import MINI_CFG
    g = golden("mini_darknet")
    net = _oracle_net(MINI_CFG, (32, 32), 3, -1.0)
    out = net.forward(g["x"], keep_layers=True)
    for i, lo in enumerate(net.layer_outputs):
        key = f"layer{i}"
        if key in g.files:
            np.testing.assert_allclose(lo, g[key], rtol=1e-4, atol=1e-5, err_msg=key)
    np.testing.assert_allclose(out, g["out"], rtol=RTOL, atol=ATOL)


def test_tiny416_end_to_end():
    g = golden("darknet_tiny416_seed0")
    net = _oracle_net(cfgs.cfg_text("yolov3-tiny"), 416, 0)
    x = np.random.RandomState(0).rand(1, 3, 416, 416).astype(F32)
    out = net(x)
    assert out.shape == (1, 2535, 85)
    np.testing.assert_allclose(out, g["out"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("name", ["yolov3", "yolov4"])
def test_full_608_sampled(name):
    g = golden(f"darknet_{name}_608_seed0")
    net = _oracle_net(cfgs.cfg_text(name, 608, 608), 608, 0)
    x = np.random.RandomState(1).rand(1, 3, 608, 608).astype(F32)
    out = net(x)
    assert tuple(out.shape) == tuple(g["shape"]) == (1, 22743, 85)
    np.testing.assert_allclose(out.reshape(-1)[g["idx"]], g["val"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out[0, :, 4], g["obj"], rtol=RTOL, atol=ATOL)


def test_nms_cases_bit_exact():
    g = golden("nms_cases")
    names = sorted({k[:-5] for k in g.files if k.endswith("_pred")})
    assert len(names) >= 7
    for n in names:
        ct, it = g[n + "_thr"]
        out = onms.soft_non_max_suppression(g[n + "_pred"], ct, it)[0]
        ref = g[n + "_out"]
        if ref.shape[0] == 0:
            assert out is None
        else:
            assert out.dtype == np.float32 and np.array_equal(out, ref), n


def test_detect_plumbing_cfg1():
    """BASELINE cfg1: yolov3-tiny 416, tracker=None, one 640x480 frame."""
    from oracle.resize import resize_bilinear_u8
    g = golden("detect_plumbing_640x480")
    net = _oracle_net(cfgs.cfg_text("yolov3-tiny"), 416, 0, float(g["obj_bias"]))
    frame = np.random.RandomState(0).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    img = resize_bilinear_u8(frame, (416, 416)).astype(F32).transpose(2, 0, 1)[None] / F32(255.)
    det = onms.soft_non_max_suppression(net(img), 0.5, 0.4)[0]
    det = onms.resize_boxes(det, (416, 416), (480, 640))
    ref = g["out"]
    assert det.shape == ref.shape and ref.shape[0] > 0
    assert np.array_equal(det[:, 5], ref[:, 5])
    np.testing.assert_allclose(det, ref, rtol=RTOL, atol=ATOL)


def test_rectangular_input_and_the_scale_pairing_quirk():
    """a5 on non-square inputs (reference models.py:169-172,216: x and w go with the HEIGHT ratio): yolov3-tiny at
    (416, 608) raw + through detect on a 640x480 frame, and the 44x64 net whose two ratios differ (8.8 / 8.0)."""
    from oracle.gen_golden import QUIRK_CFG
    from oracle.resize import resize_bilinear_u8
    g = golden("darknet_rect")
    net = _oracle_net(cfgs.cfg_text("yolov3-tiny", 608, 416), (416, 608), 0, -1.0)
    x = np.random.RandomState(2).rand(1, 3, 416, 608).astype(F32)
    y = net(x)
    assert y.shape == tuple(g["tiny_shape"]) == (1, 3705, 85)
    np.testing.assert_allclose(y[0, :, :5], g["tiny_box"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(y.reshape(-1)[g["tiny_idx"]], g["tiny_val"], rtol=RTOL, atol=ATOL)
    frame = np.random.RandomState(0).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    img = resize_bilinear_u8(frame, (608, 416)).astype(F32).transpose(2, 0, 1)[None] / F32(255.)
    assert img.shape == (1, 3, 416, 608)
    det = onms.resize_boxes(onms.soft_non_max_suppression(net(img), 0.5, 0.4)[0], (416, 608), (480, 640))
    ref = g["tiny_det"]
    assert det.shape == ref.shape and ref.shape[0] > 0 and np.array_equal(det[:, 5], ref[:, 5])
    np.testing.assert_allclose(det, ref, rtol=RTOL, atol=ATOL)
    q = _oracle_net(QUIRK_CFG, (44, 64), 5, -1.0)
    xq = np.random.RandomState(6).rand(2, 3, 44, 64).astype(F32)
    yq = q(xq)
    assert tuple(g["quirk_grid"]) == (5, 8) and np.allclose(g["quirk_scale"].ravel(), [8.8, 8.0])
    assert yq.shape == g["quirk_out"].shape == (2, 120, 7)
    np.testing.assert_allclose(yq, g["quirk_out"], rtol=RTOL, atol=ATOL)
    # the pairing matters on this net: the un-quirked scaling (x by the width ratio) is far outside the tolerance
    assert np.abs(yq[..., 0] * (8.0 / 8.8) - g["quirk_out"][..., 0]).max() > 1.0


@pytest.mark.ref
@pytest.mark.skipif(not has_reference(), reason="reference tree not present")
def test_real_cfg_files_parse_identically():
    ref = json.load(open(os.path.join(GOLD, "cfg_parse.json")))
    for name in ref:
        text = open(f"/root/reference/config/{name}.cfg").read()
        mine = parse_model_config_text(text)
        red = [{k: v for k, v in d.items() if k in ref[name][i]} for i, d in enumerate(mine)]
        assert red == ref[name]


def test_tiled_detection_and_merge_quirk():
    """8f row 1: sliding-window detection; the merge=True branch collapses boxes exactly like the reference."""
    from oracle.tiled import detect_tiled
    g = golden("tiled_detect")
    for name in ("all_kept", "one_kept", "plain", "single"):
        out = onms.soft_non_max_suppression_merge(g[name + "_pred"], 0.5, 0.4, is_p1p2=True)[0]
        ref = g[name + "_out"]
        assert out.shape == ref.shape, name
        assert np.array_equal(out[:, 4:], ref[:, 4:]), name
        np.testing.assert_allclose(out[:, :4], ref[:, :4], rtol=1e-6, atol=1e-4, err_msg=name)
    assert np.ptp(g["all_kept_out"][:, 0]) == 0            # the collapse really happened in the reference
    frame = np.random.RandomState(0).randint(0, 256, (480, 640, 3)).astype(np.uint8)
    for tag in ("a", "b"):
        net = _oracle_net(cfgs.cfg_text("yolov3-tiny"), 416, 0, float(g["obj_bias_" + tag]))
        det = detect_tiled(net, frame, (416, 416), 0.15, 0.5, 0.4)
        ref = g["tiled_out_" + tag]
        assert det.shape == ref.shape and ref.shape[0] > 3
        assert np.array_equal(det[:, 5], ref[:, 5])
        np.testing.assert_allclose(det, ref, rtol=RTOL, atol=ATOL, equal_nan=True)
    assert np.isnan(g["tiled_out_b"][:, :4]).all() and not np.isnan(g["tiled_out_a"]).any()


def test_wide_range_fixture_vs_oracle_restatement():
    """The oracle on the wide-dynamic-range fixture (BatchNorm statistics over decades, black / saturated inputs): every sampled conv
    block and the decoded heads of the 12-conv residual net (tests/golden/wide_range.npz, generated from the imported reference)."""
    from oracle.gen_golden import WIDE_RES_CFG, WIDE_SEED, wide_inputs
    g = golden("wide_range")
    net = DarknetOracle(WIDE_RES_CFG, (64, 64), is_text=True)
    blob = synth.darknet_weights_blob(WIDE_RES_CFG, WIDE_SEED, -1.0, profile="wide")
    w = np.frombuffer(blob, dtype=F32, offset=20)
    assert net.load_weights_array(w) == w.size
    # the profile does what it says: variances over >= 4 decades, exact zeros among the gammas, |beta| up to 3
    n0 = 32
    beta, gamma, mean, var = (w[k * n0:(k + 1) * n0] for k in range(4))
    assert (gamma == 0).sum() >= 1 and gamma.max() > 1.5 and np.abs(beta).max() > 2.5
    allvar = []
    off = 0
    for _, cin, cout, k, bn, is_head, _ in synth.conv_shapes(WIDE_RES_CFG):
        if bn:
            allvar.append(w[off + 3 * cout:off + 4 * cout])
            off += 4 * cout
        else:
            off += cout
        off += cout * cin * k * k
    allvar = np.concatenate(allvar)
    assert allvar.min() < 3e-3 and allvar.max() > 30
    x = wide_inputs(64)
    out = net.forward(x, keep_layers=True)
    n = 0
    for key in [k for k in g.files if k.startswith("res_L") and k.endswith("_idx")]:
        i = int(key[5:-4])
        np.testing.assert_allclose(net.layer_outputs[i].reshape(-1)[g[key]], g[f"res_L{i}_val"], rtol=1e-4, atol=1e-4, err_msg=key)
        n += 1
    assert n == 12
    np.testing.assert_allclose(out, g["res_out"], rtol=RTOL, atol=ATOL)


def test_wide_range_mish_csp_fixture_vs_oracle_restatement():
    """The yolov4 constructs (Mish, CSP split, shortcut, two-source and grouped routes) under the wide BatchNorm statistics:
    tests/golden/wide_range_mish.npz, computed by the imported reference."""
    from oracle.gen_golden import WIDE_CSP_CFG, WIDE_SEED, wide_inputs
    g = golden("wide_range_mish")
    net = DarknetOracle(WIDE_CSP_CFG, (64, 64), is_text=True)
    blob = synth.darknet_weights_blob(WIDE_CSP_CFG, WIDE_SEED + 2, -1.0, profile="wide")
    w = np.frombuffer(blob, dtype=F32, offset=20)
    assert net.load_weights_array(w) == w.size
    out = net.forward(wide_inputs(64, seed=3), keep_layers=True)
    n = 0
    for key in [k for k in g.files if k.startswith("L") and k.endswith("_idx")]:
        i = int(key[1:-4])
        np.testing.assert_allclose(net.layer_outputs[i].reshape(-1)[g[key]], g[f"L{i}_val"], rtol=1e-4, atol=1e-4, err_msg=key)
        n += 1
    assert n == 11
    np.testing.assert_allclose(out, g["out"], rtol=RTOL, atol=ATOL)
