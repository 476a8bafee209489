#!/bin/bash
# experiment (GPU box): order of the tiles inside an XCD's rectangle - YDS_TILE_GN filter tiles walked together (1 = rounds 1-5) - on the
# layers with more than one filter tile, 68 frames, the tuner's pick per setting; needs the tag build `gn` (conv_common.h experiment switch)
mkdir -p gpurun_out/r6l
for shape in "76,76,128,256,3,1,1,1" "38,38,256,512,3,1,1,1" "19,19,512,1024,3,1,1,1" "19,19,512,1024,3,1,1,0" "38,38,512,256,1,1,1,0" "19,19,1024,512,1,1,1,0" "76,76,256,512,3,2,1,0" "38,38,512,1024,3,2,1,0" "152,152,128,256,3,2,1,0"; do
  for rep in 1 2; do
  for gn in 1 2 4 8; do
    echo -n "$shape gn=$gn: "; YDS_BUILD_TAG=gn YDS_TILE_GN=$gn python tools/conv_bench.py --only $shape --batch 68 --iters 40 | tail -2 | head -1 | awk '{print $10, $11, $12}'
  done; done
done 2>&1 | tee gpurun_out/r6l/tile_order.txt
