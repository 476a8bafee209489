#!/bin/bash
mkdir -p gpurun_out/r6b
python tools/clock_probe_check.py > gpurun_out/r6b/clock_probe_side.txt 2>&1
YDS_BUILD_TAG=inkernel python tools/clock_probe_check.py > gpurun_out/r6b/clock_probe_inkernel.txt 2>&1
cat gpurun_out/r6b/clock_probe_side.txt gpurun_out/r6b/clock_probe_inkernel.txt
timeout 1200 python -m pytest tests/test_gpu_wide_range.py -x -q -s 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_gpu_runtime.py -x -q -k "rank" 2>&1 | tail -5
python tools/make_real_files.py /tmp/realw yolov3 > /dev/null
python bench.py --steps 10 --warmup 3 --no-extras --no-roofline --latency-steps 0 --cpu-frames 0 --weights /tmp/realw/yolov3.weights --ckpt /tmp/realw/ckpt.t7 > gpurun_out/r6b/bench_realw.json 2> gpurun_out/r6b/bench_realw.err
tail -c 400 gpurun_out/r6b/bench_realw.err; python -c "
import json;d=json.loads([l for l in open('gpurun_out/r6b/bench_realw.json') if l.startswith('{')][-1]);print(d['value'],d.get('value_real_weights'),d.get('value_real_weights_error'))"
