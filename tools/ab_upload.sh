B="python bench.py --cpu-frames 0 --latency-steps 0 --no-roofline --steps 30 --warmup 4"
for m in serialized two-stream serialized two-stream; do
  $B --schedule $m > gpurun_out/up_$m.json 2>gpurun_out/up_err.log
  python - <<P
import json
d=json.loads(open('gpurun_out/up_$m.json').read().strip().splitlines()[-1])
print('$m', d['value'], 'upload', d['value_with_upload'], 'other', d['value_other_schedule'], 'f32', d['value_f32_math'])
P
done
