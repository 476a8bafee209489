# window kernel (id 15) vs LDS-DMA 128x128 (11) vs two-workgroup window kernel (20) on the 3x3 stride-1 shapes (run on the GPU box)
B=${1:-16}
for shape in 76,76,128,256,3,1,1,1 76,76,128,256,3,1,1,0 38,38,256,512,3,1,1,1 19,19,512,1024,3,1,1,1 19,19,512,1024,3,1,1,0 76,76,128,128,3,1,2,1 38,38,256,256,3,1,2,1 19,19,512,512,3,1,2,1; do
  echo "== $shape  batch $B"
  for v in ${VARS:-15 11 20}; do
    for tag in "" $TAGS; do
      printf "%-6s" "[$tag]"
      YDS_BUILD_TAG=$tag YDS_CONV_FORCE=$v python tools/conv_bench.py --only $shape --batch $B --iters 20 | tail -2 | head -1
    done
  done
done
