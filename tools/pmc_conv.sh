#!/bin/bash
# usage: tools/pmc_conv.sh <shape> <batch> <outdir>   (run on the GPU box)
export TMPDIR=/tmp
R=$PWD; shape=$1; batch=$2; out=$R/$3; mkdir -p $out; cd /tmp
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o p_$tag -- python $R/tools/conv_bench.py --only $shape --batch $batch --iters 5 > /dev/null 2>&1; }
tag=sq; run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
tag=tcc; run TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
tag=fetch; run FETCH_SIZE
tag=lds; run SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$out/p_*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print('%-28s n=%d mean=%.5g' % (k, len(v), sum(v)/len(v)))
for f in sorted(glob.glob('$out/p_sq_kernel_trace.csv')):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'conv_igemm' in r['Kernel_Name']]
    print('durations us', [round(x,1) for x in d])
PY
