# GPU-box verification: the whole -m gpu suite, the default bench line, rocprofv3 summaries + PMC traffic of the same command (both maths)
python -m pytest tests -m gpu -x -q > gpurun_out/verify_tests.log 2>&1; grep -E "passed|failed" gpurun_out/verify_tests.log | tail -2
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.json
PMC=1 bash tools/profile_bench.sh cfg2 gpurun_out/prof_cfg2 > gpurun_out/prof_cfg2.log 2>&1
MATH=f32 PMC=1 bash tools/profile_bench.sh cfg2 gpurun_out/prof_cfg2_f32 > gpurun_out/prof_cfg2_f32.log 2>&1
ls gpurun_out/prof_cfg2 gpurun_out/prof_cfg2_f32 | head -40
