# sample shader clock / socket power while the detector loop runs (run on the GPU box): DET_LOOP_N=1500 tools/clk_probe.sh
(python tools/det_loop.py > /tmp/det.log 2>&1 &)
for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | tr -s '\t ' ' ' | tr '\n' ' '; echo; sleep 0.5; done
wait
cat /tmp/det.log
