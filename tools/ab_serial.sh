# A/B of the stream schedule on one box: value under both schedules, alternating, per configuration (+ the exact-fp32 mode)
B="python bench.py --no-extras --no-roofline --cpu-frames 0 --latency-steps 0 --steps 40 --warmup 4"
for cfg in cfg3 cfg5 cfg2; do
  for m in two-stream serialized two-stream serialized; do
    $B --config $cfg --schedule $m > gpurun_out/ab.json 2>gpurun_out/ab_err.log
    python - <<P
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print('$cfg', '$m', d['value'], d['config']['schedule'], d['stage_us_last_step'])
P
  done
done
for m in two-stream serialized two-stream serialized; do
  $B --config cfg2 --math f32 --steps 10 --schedule $m > gpurun_out/ab.json 2>gpurun_out/ab_err.log
  python - <<P
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print('cfg2 f32', '$m', d['value'], d['config']['schedule'])
P
done
