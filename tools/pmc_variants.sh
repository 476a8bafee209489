export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/pmc2; mkdir -p $out; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TA_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*" | sort -u | tr '\n' ' ' > $out/counters.txt
for v in 7 11 12; do
run() { YDS_CONV_FORCE=$v rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o v${v}_$tag -- python $R/tools/conv_bench.py --only 76,76,128,256,3,1,1,0 --batch 16 --iters 5 > /dev/null 2>&1; }
tag=sq; run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
tag=sq2; run SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_LDS
tag=tcc; run TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
tag=ta; run TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
tag=tcp; run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$out/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-1])
    for k,v in agg.items(): print('   %-32s n=%d mean=%.5g' % (k, len(v), sum(v)/len(v)))
PY
