# the non-default configurations' bench lines (profiles/r04_bench_cfg3.json, r04_bench_cfg5.json)
for c in cfg3 cfg5; do
  python bench.py --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  python - <<P
import json
d=json.loads(open('gpurun_out/bench_$c.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$c', d['value'], d['config']['schedule'], d['value_other_schedule'], 'upload', d['value_with_upload'], 'f32', d['value_f32_math'], 'fbf', d['value_frame_by_frame'], d['value_frame_by_frame_lookahead1'])
print('   frac', r['frac'], 'iso', r['frac_isolated'], 'pipe', r['pipeline_conv_frac'], 'allconv', r['all_conv_kernels']['frac'], r['all_conv_kernels']['frac_isolated'], d['stage_us_last_step'], d['stage_us_frame_by_frame'])
P
done
