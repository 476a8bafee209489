# exact-fp32 kernel A/B (run on the GPU box): TAGS="old" tools/ab_f32.sh
export YDS_CONV_MATH=f32
for spec in "76,76,128,256,3,1,1,1 32" "38,38,256,512,3,1,1,0 32" "19,19,512,1024,3,1,1,1 32" "76,76,256,128,1,1,1,0 32" "152,152,64,128,3,1,1,1 32" "304,304,64,128,3,2,1,0 32" "32,16,128,128,3,1,3,0 960"; do
  set -- $spec
  for tag in $TAGS ""; do
    echo -n "$1 b$2 [$tag]: "
    YDS_BUILD_TAG=$tag python tools/conv_bench.py --only $1 --batch $2 --iters 10 | tail -2 | head -1 | awk '{print $10, $11, $12}'
  done
done
for tag in $TAGS ""; do echo -n "yolov3 b32 [$tag]: "; YDS_BUILD_TAG=$tag python tools/conv_bench.py --net yolov3 --batch 32 --iters 5 | tail -1; done
