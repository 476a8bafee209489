# Socket power and clocks while the dominant window layer runs back to back (run on the GPU box): tools/power_sample.sh
R=$PWD
run() {   # $1 label, env assignments follow
  label=$1; shift
  env "$@" YDS_CONV_FORCE=15 python $R/tools/conv_bench.py --only 76,76,128,256,3,1,1,1 --batch 16 --iters ${ITERS:-40000} > /tmp/ps_$label.txt 2>&1 &
  pid=$!
  sleep 4
  for i in 1 2 3 4 5; do
    p=$(rocm-smi --showpower 2>/dev/null | grep -i -E "power" | grep -o -E "[0-9]+\.[0-9]+" | head -1)
    c=$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | grep -o -E "\([0-9]+Mhz\)" | head -1)
    echo "$label: power ${p} W  sclk ${c}"
    sleep 1
  done
  wait $pid
  tail -2 /tmp/ps_$label.txt | head -1
}
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|power" | head -3
run idle_then_rand YDS_BENCH_DATA=rand
run zero YDS_BENCH_DATA=zero
