#!/bin/bash
# Kill-gate probe for Winograd F(2x2,3x3) on the three 109-GFLOP 3x3 shapes of yolov3-608 (VERDICT r4 'next' #3), run on the GPU box.
# The 16 position GEMMs of a Winograd layer, looped position-outermost over one output tile, ARE a 1x1 convolution with K = 16 * Cin over
# M = N * ceil(H/2) * ceil(W/2) tiles - operand delivery of the existing LDS-DMA kernel, the longest K loops it has ever seen, no epilogue
# fold.  This times that proxy (an upper bound of what the GEMM stage could reach with this kernel family) next to the direct 3x3 layer
# it would replace; the input transform (reads the tensor once, writes 4x its bytes) and the output fold come on top.
B=${1:-32}
for s in "19,19,512,1024,3,1,1,0 10,10,8192,1024,1,1,1,0" "38,38,256,512,3,1,1,0 19,19,4096,512,1,1,1,0" "76,76,128,256,3,1,1,0 38,38,2048,256,1,1,1,0"; do
  set -- $s
  echo "== direct 3x3: $1   |   winograd GEMM proxy (1x1, K = 16 Cin, M = tiles): $2"
  python tools/conv_bench.py --only $1 --batch $B --iters 30 | tail -2
  python tools/conv_bench.py --only $2 --batch $B --iters 30 | tail -2
done
