"""CPU oracle for the detect -> ReID -> association hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy / C restatement of the
reference algorithms (each function cites the reference file:line it follows).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.  The
product (``yolo_deepsort_amd``) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * Everything that lives in the reference tree (Darknet forward, YOLO decode,
    multi-label NMS front end, ReID net, Kalman filter, gating, cosine metric,
    IOU cost, min_cost_matching bookkeeping, track lifecycle, DeepSort output
    stage) is pinned by golden vectors under ``tests/golden/`` that were produced
    by importing ``/root/reference`` itself (``oracle/gen_golden.py``).
  * ``scipy.optimize.linear_sum_assignment`` (third party, not in the tree) is
    pinned against real scipy (installed in the image) on tie-heavy fixtures.
  * ``torchvision.ops.nms`` and ``cv2.resize`` are third party and NOT installed
    here: their arithmetic is restated from their documented behaviour; parity
    for those two steps is "unpinned" by anything but that restatement.
"""
