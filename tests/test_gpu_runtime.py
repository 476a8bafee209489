"""Device binding on the GPU box: one process drives one GPU."""
import json
import os
import subprocess
import sys
import threading

import pytest

from yolo_deepsort_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bound_device_survives_constructors_and_threads():
    from yolo_deepsort_amd import cfgs, synth
    from yolo_deepsort_amd.models import Darknet
    from yolo_deepsort_amd.deep_sort import DeepSort
    dev = _lib.init()
    lib = _lib.load()
    assert lib.yds_current_device() == dev == _lib.current_device()
    net = Darknet(None, img_size=(96, 96), cfg_text=cfgs.cfg_text("yolov3-tiny", 96, 96))
    ds = DeepSort(synth.reid_state_dict(0), use_cuda=True)
    clone = ds.clone()
    assert lib.yds_current_device() == dev and net.device == f"cuda:{dev}"
    assert clone.extractor is ds.extractor and clone.tracker is not ds.tracker       # deep_sort.py:41-44
    assert len(_lib.pci_bus_id()) >= 7
    with pytest.raises(_lib.YdsError):
        _lib.init(dev + 1)
    seen = {}

    def worker():            # hipSetDevice is per host thread: ABI entries re-select the bound device
        buf = _lib.DeviceBuffer(1024)
        seen["dev"] = lib.yds_current_device()
        buf.free()
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen["dev"] == dev


def test_rank_beyond_device_count_fails_loudly():
    """A rank whose LOCAL_RANK has no GPU must raise - never fall back to GPU 0."""
    n = _lib.load().yds_device_count()
    code = ("import sys; sys.path.insert(0, %r)\nfrom yolo_deepsort_amd import _lib\n"
            "try:\n    _lib.init()\nexcept _lib.YdsError as e:\n    print('ERR', e)\nelse:\n    print('BOUND', _lib.current_device())\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LOCAL_RANK=str(n)), capture_output=True, text=True, timeout=300)
    assert "ERR" in out.stdout and "outside" in out.stdout, out.stdout + out.stderr


def test_two_ranks_use_two_gpus():
    """bench.py --gpus 2 launches itself under torch.distributed.run; each rank must sit on its own GPU."""
    if _lib.load().yds_device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's 8-GPU node runs the scaling bench)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-frames", "0",
                          "--no-roofline"], capture_output=True, text=True, timeout=1200)
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and len(set(line["config"]["rank_pci_bus_ids"])) == 2, line


def test_two_rank_bench_path_on_one_gpu():
    """The N > 1 code path end to end on the 1-GPU box: two processes under torch.distributed.run, one stream per rank
    (seeds 0 and 1), barriers + max-over-ranks timing + gathered device ids - with both ranks bound to GPU 0 and gloo for
    the rendezvous (RCCL refuses two ranks on one device; the real run uses nccl and one GPU per rank)."""
    env = dict(os.environ, YDS_DEVICE="0", YDS_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                          "--cpu-frames", "0", "--no-extras", "--no-roofline"], capture_output=True, text=True, timeout=1500, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-500:] + out.stderr[-1500:]          # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["streams"] == 2 and line["scaling"] == "weak"
    assert line["config"]["rank_devices"] == [0, 0] and line["value"] > 0
    assert line["config"]["tracker_rows_out"] > 0                                                   # both streams produced rows
