# GPU-box verification of the serialized schedule: parity tests, default bench line, rocprofv3 summaries + PMC traffic (both maths)
python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_pipeline.py tests/test_gpu_video_detect.py tests/test_gpu_runtime.py -x -q > gpurun_out/verify_tests.log 2>&1; tail -3 gpurun_out/verify_tests.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
PMC=1 bash tools/profile_bench.sh cfg2 gpurun_out/prof_cfg2 > gpurun_out/prof_cfg2.log 2>&1
MATH=f32 PMC=1 bash tools/profile_bench.sh cfg2 gpurun_out/prof_cfg2_f32 > gpurun_out/prof_cfg2_f32.log 2>&1
ls gpurun_out/prof_cfg2 gpurun_out/prof_cfg2_f32
