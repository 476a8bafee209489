# time every f16x3 variant (ids 7-17, see yds_conv_variant_name; inapplicable ones fall back to the autotuned pick) on the
# dominant layer shapes (run on the GPU box): tools/force_sweep.sh [batch]
B=${1:-16}
for shape in 76,76,128,256,3,1,1,0 38,38,256,512,3,1,1,0 19,19,512,1024,3,1,1,0 76,76,256,128,1,1,1,0 152,152,64,128,3,1,1,1 64,32,64,64,3,1,3,0; do
  echo "== $shape"
  for v in ${VARS:-7 8 9 10 11 12 13 14 15 16 17}; do
    YDS_CONV_FORCE=$v python tools/conv_bench.py --only $shape --batch $B --iters 20 | tail -2 | head -1
  done
done
