# rocprofv3 kernel statistics of a detector-only loop (run on the GPU box): tools/det_kstats.sh <kernel name substring>...
export TMPDIR=/tmp
R=$PWD; cd /tmp; rm -rf /tmp/rp_det
DET_NET=${DET_NET:-yolov3} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_det -o r -- python $R/tools/det_loop.py ${PASSES:-30} > /tmp/det_out.txt 2>&1
cd $R; python tools/kstats.py /tmp/rp_det "$@"
