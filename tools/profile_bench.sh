# rocprofv3 kernel statistics of the default bench command (run on the GPU box): tools/profile_bench.sh <cfg> <outdir>
# pass 1 fills the conv tune cache so that the profiled pass holds no autotune launches
export TMPDIR=/tmp
R=$PWD; cfg=${1:-cfg2}; out=$R/${2:-gpurun_out/prof}; mkdir -p $out
export YDS_TUNE_CACHE=/tmp/yds_tune_$cfg.txt
python bench.py --config $cfg --steps 5 --warmup 2 > $out/bench_${cfg}_plain.json 2>$out/err1.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench_$cfg -- python $R/bench.py --config $cfg --steps 20 --warmup 3 > $out/bench_${cfg}_under_rocprof.json 2>$out/err2.log
cd $R
ls $out
