"""CPU baseline leg of bench.py (TEST ORACLE, never the product): the oracle pipeline - resize, Darknet, NMS, class mask,
crops + ReID, DeepSORT association - over the first N frames of the SAME synthetic stream bench.py runs on the GPU, on
the host cores (C + OpenMP kernels of oracle/csrc/fastconv.c, BLAS sgemm, numpy for the small stages).

Runs as its own process so that the thread pools can be configured before any library loads
(OPENBLAS_THREAD_TIMEOUT / OMP_WAIT_POLICY: BLAS workers that spin after a GEMM would otherwise steal the cores from
the OpenMP loops that follow).  Prints one JSON object.

    python -m oracle.cpu_baseline --config cfg2 --frames 16
"""
import os
import sys

os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import argparse       # noqa: E402
import json           # noqa: E402
import time           # noqa: E402

import numpy as np    # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {"cfg2": ("yolov3", 30, None), "cfg3": ("yolov4", 30, None), "cfg5": ("yolov4", 200, 150)}
DS_PARAMS = dict(max_dist=0.3, nn_budget=30, n_init=3, max_iou_distance=0.7, max_age=30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--plain", action="store_true", help="the single-threaded-gather numpy oracle instead of the OpenMP kernels")
    args = ap.parse_args()
    from oracle.darknet import DarknetOracle, yolo_heads
    from oracle.fast import DarknetFast, ReidFast, threads
    from oracle.pipeline import run_stream
    from yolo_deepsort_amd import cfgs, synth
    net_name, persons, visible = CONFIGS[args.config]
    S = 608
    cfg_text = cfgs.cfg_text(net_name, S, S)
    net = DarknetOracle(cfg_text, S, is_text=True)
    net.load_weights_array(np.frombuffer(synth.darknet_weights_blob(cfg_text, seed=0), dtype=np.float32, offset=20))
    sd = synth.reid_state_dict(0)
    scene = synth.PersonScene(persons, seed=args.seed, n_visible=visible)
    heads = yolo_heads(net, S, S)
    frames = [scene.frame(t) for t in range(args.frames + 1)]
    inj = [synth.head_injection(scene.boxes(t)[1], (scene.H, scene.W), (S, S), heads, cls=0) for t in range(args.frames + 1)]
    run_net, reid_fn = (net, None) if args.plain else (DarknetFast(net), ReidFast(sd))
    run_stream(run_net, sd, DS_PARAMS, frames[:1], inj[:1], reid_fn=reid_fn)              # warm-up: page in, thread pools up
    t0 = time.perf_counter()
    outs = run_stream(run_net, sd, DS_PARAMS, frames[:args.frames], inj[:args.frames], reid_fn=reid_fn)
    dt = time.perf_counter() - t0
    rows = sum(len(o) for o in outs if o is not None)
    blas = None
    try:
        from threadpoolctl import threadpool_info
        blas = max((p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"), default=None)
    except Exception:
        pass
    n_threads = os.cpu_count() if args.plain else threads()
    # `cores` = the threads the dominant stage (the BLAS GEMMs of the convolutions) really ran on: this numpy's BLAS caps its pool
    # (64 on the 256-thread bench host), the OpenMP stages around it use `openmp_threads`
    print(json.dumps(dict(value=round(args.frames / dt, 4), unit="frames/s", cores=int(blas or n_threads), kind="port",
                          sample=f"first {args.frames} frames of the same stream through oracle/ "
                                 f"({'numpy' if args.plain else 'C + OpenMP gather/epilogue/pool'} on {n_threads} threads, BLAS sgemm on {blas} "
                                 f"of {os.cpu_count()} host threads; {dt:.1f} s)",
                          seconds=round(dt, 2), tracker_rows=int(rows), host_cpus=os.cpu_count(), openmp_threads=int(n_threads),
                          blas_threads=blas)))


if __name__ == "__main__":
    main()
