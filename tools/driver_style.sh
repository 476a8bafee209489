#!/bin/bash
mkdir -p gpurun_out/r6j
s=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6j/bench_driver_style.json 2> gpurun_out/r6j/bench_driver_style.err
e=$(date +%s); echo "driver-style wall seconds: $((e-s)) rc=$?"
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r6j/bench_driver_style.json') if l.startswith('{')][-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], d['config']['frames_per_step'], r['frac'], r['all_conv_kernels']['frac'], r['mfma_busy'], r.get('mfma_busy_note','')[-40:], r['sustained_clock_ghz'], d['config']['schedule_trial']['per_rank'])
print(d['cpu_baseline']['value'], d['value_video_detector'], d['value_frame_by_frame'])
P
