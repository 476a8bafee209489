#!/bin/bash
# A/B (GPU box): the product library (tile order measured per layer by the autotuner) against libydsort_old.so (HEAD before it), same box
mkdir -p gpurun_out/r6m
run() { YDS_BUILD_TAG=$1 python bench.py --config $2 --steps 20 --warmup 5 --no-extras --latency-steps 0 --cpu-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('[$1] $2 value', d['value'], 'dominant', r['frac'], 'all-conv', r['all_conv_kernels']['frac'], 'of attainable', r['all_conv_kernels']['frac_of_attainable'], 'conv us/frame', r['all_conv_kernels']['us_per_frame'])"; }
{ for rep in 1 2; do run old cfg2; run "" cfg2; done; run old cfg3; run "" cfg3; run old cfg3; run "" cfg3; } 2>&1 | tee gpurun_out/r6m/tile_order_ab.txt
timeout 1500 python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_detector.py tests/test_gpu_bench_shape.py -x -q 2>&1 | tail -4 | tee gpurun_out/r6m/tests.txt
