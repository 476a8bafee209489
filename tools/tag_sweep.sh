# compare experiment builds (YDS_BUILD_TAG) on the dominant layer shapes: TAGS="a b" VARS="7 11" tools/tag_sweep.sh
for shape in 76,76,128,256,3,1,1,0 38,38,256,512,3,1,1,0 19,19,512,1024,3,1,1,0; do
  for v in ${VARS:-7 11}; do
    for tag in "" $TAGS; do
      printf "%-8s" "[$tag]"
      YDS_BUILD_TAG=$tag YDS_CONV_FORCE=$v python tools/conv_bench.py --only $shape --batch 16 --iters 20 | tail -2 | head -1
    done
  done
done
