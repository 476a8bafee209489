# A/B of the window kernel's two MFMA forms on the dominant layer shapes (run on the GPU box): tools/ab_win16.sh
# YDS_WIN32=1 = v_mfma_f32_32x32x16_f16 (conv_win.hip), default = v_mfma_f32_16x16x32_f16 (conv_win16.hip)
for spec in "76,76,128,256,3,1,1,1 16 15" "76,76,128,256,3,1,1,0 16 15" "38,38,256,512,3,1,1,1 16 15" "19,19,512,1024,3,1,1,0 16 15" "76,76,128,128,3,1,2,1 16 15" "64,32,64,64,3,1,3,0 480 16" "64,32,64,64,3,1,3,0 480 17" "32,16,128,128,3,1,3,0 480 15" "8,4,512,512,3,1,3,0 480 15"; do
  set -- $spec
  for rep in 1 2; do
    for mode in 1 0; do
      if [ $mode = 1 ]; then export YDS_WIN32=1; else unset YDS_WIN32; fi
      echo -n "$1 b$2 v$3 win32=$mode: "
      YDS_CONV_FORCE=$3 python tools/conv_bench.py --only $1 --batch $2 --iters 30 | tail -2 | head -1
    done
  done
done
