# the 16x16x32 window kernel's scheduling experiments (tagged builds, tools/tagbuild.sh) against the 32x32x16 form:
#   TAGS="p1 p2" tools/ab_win16_tags.sh      ("" = the main build, WIN32 = conv_win.hip's 32x32x16 kernel through YDS_WIN32=1)
for spec in "76,76,128,256,3,1,1,1 16 15" "38,38,256,512,3,1,1,1 16 15" "19,19,512,1024,3,1,1,0 16 15" "76,76,128,128,3,1,2,1 16 15" "64,32,64,64,3,1,3,0 480 16" "32,16,128,128,3,1,3,0 480 15"; do
  set -- $spec
  for rep in 1 2; do
  for tag in WIN32 "" $TAGS; do
    if [ "$tag" = WIN32 ]; then export YDS_WIN32=1; t=""; else unset YDS_WIN32; t=$tag; fi
    echo -n "$1 b$2 [$tag]: "
    YDS_BUILD_TAG=$t YDS_CONV_FORCE=$3 python tools/conv_bench.py --only $1 --batch $2 --iters 30 | tail -2 | head -1 | awk '{print $10, $11}'
  done
  done
done
