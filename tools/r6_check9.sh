#!/bin/bash
mkdir -p gpurun_out/r6h
export DET_LOOP_B=68 PASSES=20
for rep in 1 2; do
for tag in "" vd1; do
  echo "== tag [$tag]"; YDS_BUILD_TAG=$tag tools/det_kstats.sh conv_stem2 conv_block1; grep "detector pass" /tmp/det_out.txt
done; done 2>&1 | tee gpurun_out/r6h/valu_diet_ab.txt
YDS_BUILD_TAG=vd1 timeout 1800 python -m pytest tests/test_gpu_detector.py tests/test_gpu_conv_variants.py tests/test_gpu_wide_range.py -x -q 2>&1 | tail -5 | tee gpurun_out/r6h/tests_vd1.txt
