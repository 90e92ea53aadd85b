#!/bin/bash
# prints VGPR / scratch / occupancy per kernel of one padded size (hipcc -Rpass-analysis=kernel-resource-usage); writes nothing into the tree
# usage: lcqpow_amd/csrc/resusage.sh [NCH=2] [-DFLAGS...]
cd "$(dirname "$0")"
NCH=${1:-2}; shift
for f in "lcqp_nch.hip -DLCQP_TU_NCH=$NCH" "lcqp_sparse.hip"; do
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -mllvm -disable-machine-licm -c -o /tmp/resusage_$$.o $f -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); rows[cur]={}
    for k,s in (('VGPRs:','v'),('AGPRs:','a'),('ScratchSize','scr'),('Occupancy','occ'),('VGPRs Spill','spill'),('LDS Size','lds')):
        m=re.search(re.escape(k)+r'[^:]*:? (\d+)',line)
        if m and cur: rows[cur][s]=m.group(1)
    if 'error' in line: print(line.rstrip())
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().replace('(anonymous namespace)::','').split('(')[0]
    print(f'{name:40s}', ' '.join(f'{a}={b}' for a,b in v.items()))
"
done
rm -f /tmp/resusage_$$.o
