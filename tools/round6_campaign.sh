#!/bin/bash
# round 6 measurement campaign (run on the GPU box: gpurun -- tools/round6_campaign.sh): rocprofv3 summaries + PMC passes first, then the
# bench lines that cite them (profiles/r06_*)
O=gpurun_out/r6q; mkdir -p $O profiles
copy() {
  for f in bench_$1_kernel_stats.csv bench_$1_under_rocprof.json; do [ -f $O/$f ] && cp $O/$f profiles/r06_$f; done
  for f in $1_conv_by_tile.txt $1_conv_by_grid.txt $1_timeline.txt; do [ -f $O/$f ] && cp $O/$f profiles/r06_$f; done
  for f in traffic_$1.json mfma_busy_$1.json; do [ -f $O/$f ] && cp $O/$f profiles/r06_$f; done
}
PMC=1 MFMA=1 TIMELINE=1 tools/profile_bench.sh cfg2 $O > $O/log_cfg2.txt 2>&1; copy cfg2
PMC=1 MFMA=1 tools/profile_bench.sh cfg3 $O > $O/log_cfg3.txt 2>&1; copy cfg3
MFMA=1 TIMELINE=1 tools/profile_bench.sh cfg5 $O > $O/log_cfg5.txt 2>&1; copy cfg5
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cp $O/bench_cfg2.json profiles/r06_bench_cfg2.json
python bench.py --config cfg3 --steps 20 --warmup 5 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cp $O/bench_cfg3.json profiles/r06_bench_cfg3.json
python bench.py --config cfg5 --steps 20 --warmup 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; cp $O/bench_cfg5.json profiles/r06_bench_cfg5.json
python bench.py --steps 600 --warmup 5 --no-extras --latency-steps 0 --cpu-frames 0 > $O/bench_cfg2_sustained.json 2> $O/bench_cfg2_sustained.err; cp $O/bench_cfg2_sustained.json profiles/r06_bench_cfg2_sustained.json
mkdir -p $O/profiles_r06; cp profiles/r06_* $O/profiles_r06/
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6q/bench_cfg*.json')):
    if 'plain' in f or 'rocprof' in f or 'timeline' in f: continue
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['config']['frames_per_step'], d['config']['schedule'], 'upl', d.get('value_with_upload'), 'f32', d.get('value_f32_math'), 'half', d.get('value_half_mode'), 'fbf', d.get('value_frame_by_frame'), 'vd', d.get('value_video_detector'),
              'frac', r.get('frac'), 'clk', r.get('sustained_clock_ghz'), 'traffic', r.get('traffic'), 'busy', r.get('mfma_busy'), 'all', (r.get('all_conv_kernels') or {}).get('frac'), (r.get('all_conv_kernels') or {}).get('frac_of_attainable'))
    except Exception as e: print(f, 'ERR', e)
P
