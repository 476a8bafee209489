# whole-net conv time of experiment builds side by side: TAGS="a b" tools/cmp_tags.sh
for rep in 1 2; do for tag in "" $TAGS; do printf "%-8s" "[$tag]"; YDS_BUILD_TAG=$tag python tools/conv_bench.py --batch 16 --iters 10 2>&1 | tail -1; done; done
for tag in "" $TAGS; do printf "%-8s" "[$tag]"; YDS_BUILD_TAG=$tag python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
