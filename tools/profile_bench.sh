# rocprofv3 kernel statistics of the default bench command (run on the GPU box): [MATH=f32] [HALF=1] [PMC=1] [MFMA=1] [SCHED=serialized] tools/profile_bench.sh <cfg> <outdir>
# SCHED: stream schedule of the profiled steps (default serialized = the schedule of bench.py's roofline leg, under which a conv
# launch has the chip to itself; "policy" profiles what the product picks on the box).  MFMA=1 adds the matrix-pipe utilisation pass.
# pass 1 fills the conv tune cache so that the profiled pass holds no autotune launches.  MATH=f32 profiles the exact-fp32 leg
# (bench.py --math f32: the arithmetic of value_f32_math / roofline_f32); output names then carry the suffix _f32.
export TMPDIR=/tmp
R=$PWD; cfg=${1:-cfg2}; out=$R/${2:-gpurun_out/prof}; mkdir -p $out
math=${MATH:-f16x3}; sfx=""; [ "$math" = "f32" ] && sfx="_f32"
half=""; [ "${HALF:-0}" = "1" ] && { half="--half"; sfx="_half"; }      # Darknet.half(): the 2-byte activation mode
steps=${STEPS:-20}; [ "$math" = "f32" ] && steps=${STEPS:-8}
export YDS_TUNE_CACHE=/tmp/yds_tune_$cfg$sfx.txt
export YDS_PROFILE_BATCH=$(python -c "from yolo_deepsort_amd.workload import DEFAULT_BATCH; print(DEFAULT_BATCH['$cfg'])")   # frames per step the passes run at (recorded in traffic_*.json)
common="--config $cfg --math $math $half --no-extras --cpu-frames 0 --latency-steps 0 --schedule ${SCHED:-serialized}"
python bench.py $common --steps 5 --warmup 2 > $out/bench_${cfg}${sfx}_plain.json 2>$out/err1.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench_$cfg$sfx -- python $R/bench.py $common --steps $steps --warmup 3 > $out/bench_${cfg}${sfx}_under_rocprof.json 2>$out/err2.log
cd $R
python tools/rocprof_agg.py $out/bench_${cfg}${sfx}_kernel_stats.csv > $out/${cfg}${sfx}_conv_by_tile.txt
python tools/rocprof_by_grid.py $out/bench_${cfg}${sfx}_kernel_trace.csv > $out/${cfg}${sfx}_conv_by_grid.txt
# what runs beside what (is the association on the critical path?): a trace of timed steps only - no roofline leg, no side legs
if [ "${TIMELINE:-0}" = "1" ]; then
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $out -o tl_$cfg$sfx -- python $R/bench.py $common --steps 60 --warmup 3 --no-roofline > $out/bench_${cfg}${sfx}_timeline_run.json 2>$out/err_tl.log
  cd $R
  python tools/timeline_overlap.py $out/tl_${cfg}${sfx}_kernel_trace.csv 0.35 0.98 > $out/${cfg}${sfx}_timeline.txt
  cat $out/${cfg}${sfx}_timeline.txt
fi
ls $out
# HBM traffic of the conv kernels: PMC passes of their own (kernel trace only), FETCH_SIZE and WRITE_SIZE separately
if [ "${PMC:-0}" = "1" ]; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o pmc_${cfg}${sfx}_$c -- python $R/bench.py $common --steps 3 --warmup 1 --no-roofline > /dev/null 2>$out/err_pmc_$c.log
  done
  cd $R
  python tools/traffic_from_pmc.py $out/pmc_${cfg}${sfx}_FETCH_SIZE_counter_collection.csv $out/pmc_${cfg}${sfx}_WRITE_SIZE_counter_collection.csv > $out/traffic_$cfg$sfx.json
  cat $out/traffic_$cfg$sfx.json
fi
# matrix-pipe utilisation of the conv kernels (north_star: "MFMA utilisation"): one PMC pass of its own, kernel trace only
if [ "${MFMA:-0}" = "1" ]; then
  cd /tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out -o pmc_${cfg}${sfx}_mfma -- python $R/bench.py $common --steps 3 --warmup 1 --no-roofline > /dev/null 2>$out/err_pmc_mfma.log
  cd $R
  python tools/mfma_busy_from_pmc.py $out/pmc_${cfg}${sfx}_mfma_counter_collection.csv $out/pmc_${cfg}${sfx}_mfma_kernel_trace.csv > $out/mfma_busy_$cfg$sfx.json
  cat $out/mfma_busy_$cfg$sfx.json
fi
# the per-dispatch traces are tens of MB: keep the summaries only (gpurun copies at most 64 MiB back)
rm -f $out/*_kernel_trace.csv $out/pmc_${cfg}${sfx}_*_counter_collection.csv $out/pmc_${cfg}${sfx}_*_kernel_trace.csv $out/*_agent_info.csv
