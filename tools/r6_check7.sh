#!/bin/bash
mkdir -p gpurun_out/r6g
timeout 2400 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_long_stream.py tests/test_gpu_pipeline.py tests/test_gpu_runtime.py tests/test_gpu_video_detect.py tests/test_gpu_wide_range.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r6g/gpu_tests.txt
run() { python bench.py --config $1 --batch $2 --steps $3 --warmup 4 --no-extras --no-roofline --latency-steps 0 --cpu-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 batch', d['config']['frames_per_step'], 'value', d['value'], d['config']['schedule'], 'ms/step', d['ms_per_step'])"; }
{ run cfg2 64 12; run cfg2 68 12; run cfg2 64 12; run cfg2 68 12; run cfg2 136 6; run cfg3 64 12; run cfg3 68 12; run cfg5 32 12; run cfg5 34 12; run cfg5 32 12; run cfg5 34 12; } 2>&1 | tee gpurun_out/r6g/batch_sweep2.txt
