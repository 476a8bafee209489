for shape in ${SHAPES:-19,19,512,512,1,1,2,0 38,38,256,256,1,1,2,0 19,19,1024,512,1,1,1,0 38,38,512,256,1,1,1,0 76,76,256,128,1,1,1,0 76,76,128,128,1,1,2,0 152,152,128,64,1,1,1,0 38,38,768,256,1,1,1,0}; do
  echo "== $shape"
  for v in ${VARS:-7 8 9 10 11 12 13 14 18 19 21 22 23 24}; do
    echo -n "v$v: "; YDS_CONV_FORCE=$v python tools/conv_bench.py --only $shape --batch 16 --iters 30 2>&1 | tail -2 | head -1 | awk '{print $10, $11, $12}'
  done
done
