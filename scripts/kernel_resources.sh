#!/bin/bash
# per-kernel register / scratch / occupancy report of pais_kernels.hip (compiler view); extra flags: $@
cd "$(dirname "$0")/../pais_mvs_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c pais_kernels.hip -o /tmp/k_res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for key in ('VGPRs','SGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]','VGPR Spill'):
        m=re.search(key+r': (\d+)',l)
        if m and cur: rows[cur][key.split(' ')[0]]=int(m.group(1))
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.split('(')[0].replace('void ','')
    print('%-28s vgpr %4d sgpr %4d scratch %4d occ %d'%(name,v.get('VGPRs',-1),v.get('SGPRs',-1),v.get('ScratchSize',-1),v.get('Occupancy',-1)))
"
