#!/bin/bash
mkdir -p gpurun_out/r6f
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r6f/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r6f/bench_cfg2.json 2> gpurun_out/r6f/bench_cfg2.err; tail -c 300 gpurun_out/r6f/bench_cfg2.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r6f/bench_cfg2.json') if l.startswith('{')][-1])
print(d['value'], d['config']['frames_per_step'], 'upload', d.get('value_with_upload'), 'vd', d.get('value_video_detector'), 'fbf', d.get('value_frame_by_frame'), d.get('value_frame_by_frame_lookahead1'), d.get('stage_us_frame_by_frame'))
r=d['roofline']; print({k:r.get(k) for k in ('kernel','frac','sustained_clock_ghz','sustained_clock_source','frac_at_sustained_clock')}, r.get('all_conv_kernels'))
print('f32', d.get('value_f32_math'), d.get('roofline_f32',{}).get('frac'), 'half', d.get('value_half_mode'))
P
