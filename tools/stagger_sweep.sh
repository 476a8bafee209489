# phase offset between the two workgroups of a CU in the two-workgroup window kernel (run on the GPU box)
B=${1:-16}
for shape in 76,76,128,256,3,1,1,1 76,76,128,256,3,1,1,0 38,38,256,512,3,1,1,1 19,19,512,1024,3,1,1,1 76,76,128,128,3,1,2,1; do
  echo "== $shape  batch $B"
  for st in ${STS:-0 2 4 6 8 10}; do
    printf "stagger %-3s" $st
    YDS_WIN2_STAGGER=$st YDS_CONV_FORCE=20 python tools/conv_bench.py --only $shape --batch $B --iters 20 | tail -2 | head -1
  done
done
