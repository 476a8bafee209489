# A/B of an experiment build (TAGS) of the two-workgroup window kernel at several batch sizes (run on the GPU box)
for B in ${BATCHES:-16 2}; do
for shape in 76,76,128,256,3,1,1,1 38,38,256,512,3,1,1,1 19,19,512,1024,3,1,1,0 76,76,128,128,3,1,2,1 19,19,512,512,3,1,2,1; do
  echo "== $shape  batch $B"
  for tag in "" $TAGS; do
    printf "%-6s" "[$tag]"
    YDS_BUILD_TAG=$tag YDS_CONV_FORCE=${V:-20} python tools/conv_bench.py --only $shape --batch $B --iters 20 | tail -2 | head -1
  done
done
done
