# A/B of tagged builds (tools/tagbuild.sh, or a copy of an older libydsort.so as libydsort_<tag>.so) on window-kernel layer shapes
# (run on the GPU box): TAGS="old" [BATCH=32] tools/ab_tags.sh      ("" = the main build)
B=${BATCH:-32}
for spec in "76,76,128,256,3,1,1,1 $B 15" "38,38,256,512,3,1,1,1 $B 15" "19,19,512,1024,3,1,1,1 $B 15" "76,76,128,128,3,1,2,1 $B 15"; do
  set -- $spec
  for rep in 1 2; do
  for tag in $TAGS ""; do
    echo -n "$1 b$2 [$tag]: "
    YDS_BUILD_TAG=$tag YDS_CONV_FORCE=$3 python tools/conv_bench.py --only $1 --batch $2 --iters 30 | tail -2 | head -1 | awk '{print $10, $11}'
  done
  done
done
