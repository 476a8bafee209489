# LDS / barrier counters of the window kernel, default arithmetic vs cross8 (run on the GPU box): what the K step is made of once the
# matrix pipe no longer fills it.  tools/pmc_lds.sh [iters]
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/pmc_lds; mkdir -p $out; cd /tmp
IT=${1:-20}
for x in 0 1; do
  for pass in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -c1-14 | tr ' ' '_')
    YDS_CONV_CROSS8=$x YDS_BENCH_DATA=zero YDS_CONV_FORCE=15 rocprofv3 --pmc $pass --output-format csv -d $out -o x${x}_$tag -- python $R/tools/conv_bench.py --only 76,76,128,256,3,1,1,0 --batch 16 --iters $IT > $out/log_${x}_$tag.txt 2>&1
  done
done
cd $R
python - <<PY
import csv, collections, glob
for x in (0, 1):
    agg = collections.defaultdict(list)
    for f in glob.glob("$out/x%d_*counter_collection.csv" % x):
        for r in csv.DictReader(open(f)):
            if "conv3x3_f16x3_win" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("cross8 = %d" % x)
    for k, v in sorted(agg.items()):
        print("   %-28s mean %.5g  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
rm -f $out/*agent_info.csv
