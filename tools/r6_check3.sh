#!/bin/bash
mkdir -p gpurun_out/r6c
python tools/clock_probe_check.py > gpurun_out/r6c/clock_probe_side.txt 2>&1; cat gpurun_out/r6c/clock_probe_side.txt
timeout 1500 python -m pytest tests/test_gpu_video_detect.py tests/test_gpu_dropin.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 > gpurun_out/r6c/bench_cfg2.json 2> gpurun_out/r6c/bench_cfg2.err; tail -c 300 gpurun_out/r6c/bench_cfg2.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r6c/bench_cfg2.json') if l.startswith('{')][-1])
print(d['value'], 'vd', d.get('value_video_detector'), d.get('video_detector'), 'fbf', d.get('value_frame_by_frame'))
r=d['roofline']; print({k:r.get(k) for k in ('frac','sustained_clock_ghz','frac_at_sustained_clock')}, r.get('all_conv_kernels'))
P
python tools/make_real_files.py /tmp/realw yolov3 > /dev/null
python bench.py --steps 10 --warmup 3 --no-extras --no-roofline --latency-steps 0 --cpu-frames 0 --weights /tmp/realw/yolov3.weights --ckpt /tmp/realw/ckpt.t7 > gpurun_out/r6c/bench_realw.json 2> gpurun_out/r6c/bench_realw.err
tail -c 300 gpurun_out/r6c/bench_realw.err; python -c "
import json;d=json.loads([l for l in open('gpurun_out/r6c/bench_realw.json') if l.startswith('{')][-1]);print(d['value'],d.get('value_real_weights'),d.get('value_real_weights_error'))"
python tools/conv_bench.py --batch 32 --net yolov3 > gpurun_out/r6c/convbench_yolov3_b32.txt 2>&1; tail -40 gpurun_out/r6c/convbench_yolov3_b32.txt
