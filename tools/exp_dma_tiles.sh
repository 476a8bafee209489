#!/bin/bash
# experiment (run on the GPU box): alternative LDS-DMA tile shapes (libydsort_exp1.so remaps variant ids) against the product's on the
# layers of the LDS-DMA class, 68 frames:  id 19 = <64,128,2x2,NS>, 14 = <128,128,..,3>, 18 = fv 11, 12 = fv 5, 11 = <128,128,2x2,2>, 13 = <128,256,2x4,3>
mkdir -p gpurun_out/r6k
run() {  # tag id shape
  echo -n "[$1] id $2 $3: "; YDS_BUILD_TAG=$1 YDS_CONV_FORCE=$2 python tools/conv_bench.py --only $3 --batch 68 --iters 30 | tail -2 | head -1 | awk '{print $10, $11, $12}'
}
for shape in "19,19,1024,512,1,1,1,0" "38,38,512,256,1,1,1,0" "76,76,256,128,1,1,1,0" "152,152,64,128,3,1,1,1" "152,152,128,256,3,2,1,0" "76,76,256,512,3,2,1,0" "38,38,512,1024,3,2,1,0" "19,19,1024,255,1,1,0,0"; do
  echo "== $shape"
  run "" "" $shape                 # the tuner's pick
  for id in 19 14 18 12 11 13; do run "" $id $shape; done
  for id in 19 14 18 12; do run exp1 $id $shape; done
done 2>&1 | tee gpurun_out/r6k/dma_tiles.txt
