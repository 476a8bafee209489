# A/B of tagged builds (tools/tagbuild.sh, or a copy of an older libydsort.so as libydsort_<tag>.so) on the dominant window-kernel
# layer shapes (run on the GPU box): TAGS="old" tools/ab_tags.sh      ("" = the main build)
for spec in "76,76,128,256,3,1,1,1 16 15" "38,38,256,512,3,1,1,1 16 15" "19,19,512,1024,3,1,1,0 16 15" "76,76,128,128,3,1,2,1 16 15" "64,32,64,64,3,1,3,0 480 17" "32,16,128,128,3,1,3,0 480 15" "8,4,512,512,3,1,3,2 480 15"; do
  set -- $spec
  for rep in 1 2; do
  for tag in $TAGS ""; do
    echo -n "$1 b$2 [$tag]: "
    YDS_BUILD_TAG=$tag YDS_CONV_FORCE=$3 python tools/conv_bench.py --only $1 --batch $2 --iters 30 | tail -2 | head -1 | awk '{print $10, $11}'
  done
  done
done
