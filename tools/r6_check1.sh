#!/bin/bash
# round 6, first GPU call: the new bench line, the un-injected leg on real-format files, the touched tests, clock-probe A/B
mkdir -p gpurun_out/r6a
python bench.py --steps 20 --warmup 5 > gpurun_out/r6a/bench_cfg2.json 2> gpurun_out/r6a/bench_cfg2.err
tail -c 600 gpurun_out/r6a/bench_cfg2.err
python tools/make_real_files.py /tmp/realw yolov3 > /dev/null
python bench.py --steps 10 --warmup 3 --no-extras --no-roofline --latency-steps 0 --cpu-frames 0 --weights /tmp/realw/yolov3.weights --ckpt /tmp/realw/ckpt.t7 > gpurun_out/r6a/bench_realw.json 2> gpurun_out/r6a/bench_realw.err
tail -c 300 gpurun_out/r6a/bench_realw.err
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_video_detect.py tests/test_gpu_runtime.py -x -q 2>&1 | tail -5
TAGS="probe" tools/ab_tags.sh > gpurun_out/r6a/clock_probe_ab.txt 2>&1
cat gpurun_out/r6a/clock_probe_ab.txt
