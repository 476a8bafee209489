# kernel statistics of the detector-only loop (run on the GPU box): tools/prof_det.sh <outdir>
export TMPDIR=/tmp
R=$PWD; out=$R/${1:-gpurun_out/prof_det}; mkdir -p $out; cd /tmp
DET_LOOP_N=20 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python $R/tools/det_loop.py > $out/log.txt 2>&1
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/s_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:14]: print("%-100s %5s %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
