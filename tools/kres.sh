#!/bin/bash
# kernel resource usage (VGPRs, spills, scratch, occupancy) of one csrc file: tools/kres.sh conv_win16.hip [grep pattern]
cd "$(dirname "$0")/../yolo_deepsort_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -ffp-contract=off $YDS_EXTRA_FLAGS -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
  python3 -c "
import sys,re,subprocess
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l) or re.search(r' Name: (\S+)',l)
    if m:
        cur=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip(); d={}
        continue
    m=re.search(r'(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill): (\d+)',l)
    if m and cur:
        d[m.group(1)]=int(m.group(2))
        if m.group(1)=='VGPRs Spill':
            name=re.sub(r'yds::\(anonymous namespace\)::|yds::|\(yds::ConvKernelArgs.*','',cur)
            print(f\"{name[:70]:70s} vgpr {d.get('VGPRs')} agpr {d.get('AGPRs')} spill {d.get('VGPRs Spill')} scratch {d.get('ScratchSize [bytes/lane]')} occ {d.get('Occupancy [waves/SIMD]')}\")
" | grep "${2:-.}"
