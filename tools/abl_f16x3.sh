# ablation builds of the f16x3 conv kernel (YDS_F16_ABL, see conv_f16x3.hip); run on the GPU box
for shape in 76,76,128,256,3,1,1,0 19,19,512,1024,3,1,1,0 76,76,256,128,1,1,1,0; do
  for tag in "" ${ABLS:-abl1 abl2 abl3 abl4 abl5}; do
    echo "== $shape tag=$tag"
    YDS_BUILD_TAG=$tag YDS_CONV_FORCE=7 python tools/conv_bench.py --only $shape --batch 16 --iters 20 | tail -2 | head -1
  done
done
