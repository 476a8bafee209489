#!/bin/bash
# the bench lines of a round on one box (run on the GPU box): cfg2 / cfg3 / cfg5 + the sustained cfg2 run -> gpurun_out/lines/
O=gpurun_out/lines; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --config cfg3 --steps 20 --warmup 5 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --steps 600 --warmup 5 --no-extras --latency-steps 0 --cpu-frames 0 > $O/bench_cfg2_sustained.json 2> $O/bench_cfg2_sustained.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/lines/bench_cfg*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}; t=d['config'].get('schedule_trial') or {}
        print(f.split('/')[-1], d['value'], d['config']['schedule'], 'other', d.get('value_other_schedule'), 'trial', t.get('per_rank'), 'upl', d.get('value_with_upload'), 'vd', d.get('value_video_detector'), 'fbf', d.get('value_frame_by_frame'),
              'frac', r.get('frac'), 'clk', r.get('sustained_clock_ghz'), 'all', (r.get('all_conv_kernels') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
P
timeout 900 python -m pytest tests/test_gpu_long_stream.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
