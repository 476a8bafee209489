# rocprofv3 kernel statistics of the default bench command (run on the GPU box): tools/profile_bench.sh <cfg> <outdir>
# pass 1 fills the conv tune cache so that the profiled pass holds no autotune launches
export TMPDIR=/tmp
R=$PWD; cfg=${1:-cfg2}; out=$R/${2:-gpurun_out/prof}; mkdir -p $out
export YDS_TUNE_CACHE=/tmp/yds_tune_$cfg.txt
python bench.py --config $cfg --steps 5 --warmup 2 --no-extras --cpu-frames 0 --latency-steps 0 > $out/bench_${cfg}_plain.json 2>$out/err1.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench_$cfg -- python $R/bench.py --config $cfg --steps 20 --warmup 3 --no-extras --cpu-frames 0 --latency-steps 0 > $out/bench_${cfg}_under_rocprof.json 2>$out/err2.log
cd $R
ls $out
# HBM traffic of the conv kernels: PMC passes of their own (kernel trace only), FETCH_SIZE and WRITE_SIZE separately
if [ "${PMC:-0}" = "1" ]; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o pmc_${cfg}_$c -- python $R/bench.py --config $cfg --steps 3 --warmup 1 --no-extras --cpu-frames 0 --no-roofline --latency-steps 0 > /dev/null 2>$out/err_pmc_$c.log
  done
  cd $R
  python tools/traffic_from_pmc.py $out/pmc_${cfg}_FETCH_SIZE_counter_collection.csv $out/pmc_${cfg}_WRITE_SIZE_counter_collection.csv > $out/traffic_$cfg.json
  cat $out/traffic_$cfg.json
fi
# the per-dispatch traces are tens of MB: keep the summaries only (gpurun copies at most 64 MiB back)
rm -f $out/*_kernel_trace.csv $out/pmc_${cfg}_*_counter_collection.csv $out/*_agent_info.csv
