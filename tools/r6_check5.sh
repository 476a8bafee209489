#!/bin/bash
mkdir -p gpurun_out/r6e
ls /sys/bus/pci/devices/*/pp_dpm_sclk 2>&1 | head -3; for f in /sys/bus/pci/devices/*/pp_dpm_sclk; do echo $f; cat $f; done 2>&1 | head -12
ls /sys/class/drm/ 2>&1 | head; ls /sys/class/drm/card*/device/ 2>/dev/null | grep -i -E "pp_|gpu_metrics|power" | head
run() { python bench.py --config $1 --batch $2 --steps $3 --warmup 4 --no-extras --no-roofline --latency-steps 0 --cpu-frames 0 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 batch', d['config']['frames_per_step'], 'value', d['value'], d['config']['schedule'], 'ms/step', d['ms_per_step'])"; }
{ run cfg2 32 24; run cfg2 48 16; run cfg2 64 12; run cfg2 96 8; run cfg2 128 6; run cfg2 64 12; run cfg2 128 6;
  run cfg3 32 24; run cfg3 64 12; run cfg5 32 12; run cfg5 64 6; run cfg3 128 6; } 2>&1 | tee gpurun_out/r6e/batch_sweep.txt
