#!/bin/bash
mkdir -p gpurun_out/r6d
python tools/clock_probe_check.py > gpurun_out/r6d/clock_probe_side.txt 2>&1; cat gpurun_out/r6d/clock_probe_side.txt
export DET_LOOP_B=32 PASSES=20
for rep in 1 2; do
for tag in "" ep1; do
  echo "== tag [$tag]"; YDS_BUILD_TAG=$tag tools/det_kstats.sh conv_stem2 conv_block1; grep "detector pass" /tmp/det_out.txt
done; done 2>&1 | tee gpurun_out/r6d/epilogue_ab.txt
for b in 32 64 32 64; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-extras --no-roofline --latency-steps 0 --cpu-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch', d['config']['frames_per_step'], 'value', d['value'], d['config']['schedule'])"
done 2>&1 | tee gpurun_out/r6d/batch_ab.txt
