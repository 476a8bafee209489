#!/bin/bash
# build the micro-probes for gfx950 (binaries travel to the GPU box with gpurun, they are not tracked)
cd "$(dirname "$0")"
for p in bw_probe mfma_probe placement_probe fp8_cross_probe; do hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result -o $p $p.hip; done
