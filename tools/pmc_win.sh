export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/pmc_win; mkdir -p $out; cd /tmp
for v in 15 11; do
run() { YDS_CONV_FORCE=$v rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o v${v}_$tag -- python $R/tools/conv_bench.py --only 76,76,128,256,3,1,1,0 --batch 16 --iters 5 > /dev/null 2>&1; }
tag=lds; run SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES
tag=sq; run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$out/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1])
    for k,v in agg.items(): print('   %-28s n=%d mean=%.5g' % (k, len(v), sum(v)/len(v)))
PY
