# A/B of the serialized schedule on one box
B="python bench.py --no-extras --cpu-frames 0 --latency-steps 0 --steps 30 --warmup 4"
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_bench_shape.py tests/test_gpu_video_detect.py -x -q > gpurun_out/serial_tests.log 2>&1; tail -3 gpurun_out/serial_tests.log
for cfg in cfg2 cfg5 cfg3; do
  for m in -1 256 -1 256; do
    YDS_PIPE_SERIAL=$m $B --config $cfg > gpurun_out/serial_${cfg}_$m.json 2>gpurun_out/serial_err.log
    python - <<P
import json
d=json.loads(open('gpurun_out/serial_${cfg}_$m.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$cfg', '$m', d['value'], 'frac', r['frac'], 'iso', r['frac_isolated'], 'us', r['avg_launch_us'], r['avg_launch_us_isolated'], 'pipe', r['pipeline_conv_frac'], d.get('stage_us_last_step'))
P
  done
done
YDS_PIPE_SERIAL=-1 $B --math f32 --steps 10 > gpurun_out/serial_f32_-1.json 2>>gpurun_out/serial_err.log
YDS_PIPE_SERIAL=256 $B --math f32 --steps 10 > gpurun_out/serial_f32_256.json 2>>gpurun_out/serial_err.log
python - <<P
import json
for m in ('-1','256'):
    d=json.loads(open('gpurun_out/serial_f32_%s.json'%m).read().strip().splitlines()[-1])
    r=d['roofline']
    print('f32', m, d['value'], 'frac', r['frac'], 'iso', r['frac_isolated'], 'us', r['avg_launch_us'], r['avg_launch_us_isolated'], 'pipe', r['pipeline_conv_frac'])
P
