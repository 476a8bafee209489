# GPU-box verification: the whole -m gpu suite, the bench lines of every configuration, rocprofv3 summaries + PMC traffic of the
# default command (both maths), the half-mode line and profile, the upload-inclusive A/B of the schedules
python -m pytest tests -m gpu -x -q > gpurun_out/verify_tests.log 2>&1; grep -E "passed|failed" gpurun_out/verify_tests.log | tail -2
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 200 gpurun_out/bench_default.json; echo
for c in cfg3 cfg5; do python bench.py --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; done
python bench.py --half --no-extras --cpu-frames 0 --latency-steps 0 > gpurun_out/bench_half.json 2> gpurun_out/bench_half.err
YDS_HALF_H16=1 python bench.py --half --no-extras --cpu-frames 0 --latency-steps 0 > gpurun_out/bench_half_h16.json 2> gpurun_out/bench_half_h16.err
python bench.py --half --config cfg3 --no-extras --cpu-frames 0 --latency-steps 0 > gpurun_out/bench_half_cfg3.json 2> gpurun_out/bench_half_cfg3.err
PMC=1 bash tools/profile_bench.sh cfg2 gpurun_out/prof_cfg2 > gpurun_out/prof_cfg2.log 2>&1
MATH=f32 PMC=1 bash tools/profile_bench.sh cfg2 gpurun_out/prof_cfg2_f32 > gpurun_out/prof_cfg2_f32.log 2>&1
bash tools/profile_bench.sh cfg3 gpurun_out/prof_cfg3 > gpurun_out/prof_cfg3.log 2>&1
bash tools/profile_bench.sh cfg5 gpurun_out/prof_cfg5 > gpurun_out/prof_cfg5.log 2>&1
HALF=1 bash tools/profile_bench.sh cfg2 gpurun_out/prof_half > gpurun_out/prof_half.log 2>&1
for s in 0 -1 0 -1; do python tools/upload_prof.py --schedule $s; done 2>&1 | grep frames
