python -m pytest tests/test_gpu_assoc.py tests/test_gpu_pipeline.py tests/test_gpu_dropin.py tests/test_gpu_bench_shape.py tests/test_gpu_video_detect.py -x -q 2>&1 | tail -15
for c in cfg2 cfg5; do python tools/upload_prof.py --hbm --config $c; done
python tools/lsap_bench.py 2>&1 | tail -8
