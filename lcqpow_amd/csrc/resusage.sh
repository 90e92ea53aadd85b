#!/bin/bash
# prints VGPR / scratch / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -o ../liblcqpow_hip.so lcqp_hip.hip -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); rows[cur]={}
    for k,s in (('VGPRs:','v'),('ScratchSize','scr'),('Occupancy','occ'),('VGPRs Spill','spill'),('LDS Size','lds')):
        m=re.search(re.escape(k)+r'[^:]*:? (\d+)',line)
        if m and cur: rows[cur][s]=m.group(1)
    if 'error' in line or 'warning' in line: print(line.rstrip())
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().split('(')[0]
    print(f'{name:28s}', ' '.join(f'{a}={b}' for a,b in v.items()))
"
