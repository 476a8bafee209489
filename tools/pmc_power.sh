# PMC evidence for the power experiment (run on the GPU box): the window kernel on random and on zero operands - cycle and
# instruction counters must agree, only the wall time (i.e. the clock) differs.  tools/pmc_power.sh [iters]
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/pmc_power; mkdir -p $out; cd /tmp
IT=${1:-30}
for d in rand zero; do
  YDS_BENCH_DATA=$d YDS_CONV_FORCE=15 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o $d -- python $R/tools/conv_bench.py --only 76,76,128,256,3,1,1,0 --batch 16 --iters $IT > /dev/null 2>&1
done
cd $R
python - <<PY
import csv, collections
for d in ("rand", "zero"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$out/%s_counter_collection.csv" % d)):
        if "conv3x3_f16x3_win" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open("$out/%s_kernel_trace.csv" % d)) if "conv3x3_f16x3_win" in r["Kernel_Name"]]
    dur = dur[3:]
    print("%s operands: %d launches, mean duration %.1f us (profiled launches are serialised)" % (d, len(dur), sum(dur) / len(dur)))
    for k, v in sorted(agg.items()):
        print("   %-28s mean %.5g" % (k, sum(v) / len(v)))
    g = sum(agg["GRBM_GUI_ACTIVE"]) / len(agg["GRBM_GUI_ACTIVE"])
    print("   -> shader clock = GRBM_GUI_ACTIVE / 8 XCDs / duration = %.2f GHz; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.2f" % (g / 8 / (sum(dur) / len(dur)) / 1e3, sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(agg["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * g / 8)))
PY
rm -f $out/*_kernel_trace.csv $out/*agent_info.csv
