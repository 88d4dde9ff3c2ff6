#!/bin/bash
# register / spill / occupancy table of the conv kernels (CPU-side cross compile)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Iinclude -Iamt-saga_amd/csrc -c amt-saga_amd/csrc/amt_rdcnn.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
out=sys.stdin.read()
pat=sys.argv[1] if len(sys.argv)>1 else 'conv'
for b in out.split('Function Name: ')[1:]:
    name=b.split()[0]
    if pat not in name: continue
    g=lambda k: re.search(k+r': (\d+)', b).group(1)
    print(name[:72], 'VGPR',g(r'\bVGPRs'),'AGPR',g('AGPRs'),'spill',g('VGPRs Spill'),'scratch',g(r'ScratchSize \[bytes/lane\]'),'occ',g(r'Occupancy \[waves/SIMD\]'))
" "$1"
