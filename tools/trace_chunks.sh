#!/bin/bash
# kernel timeline of one chunked step (start / end of every kernel relative to the first of the step)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-trace_chunks}
K=${2:-2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LCQP_RUN_CHUNKS=$K
rocprofv3 --kernel-trace -d $O/tr$K --output-format csv -- python3 $R/bench.py --no-sparse --no-pipelined --no-resident --no-backsolve --cpu-sample 0 --steps 2 --warmup 1 > $O/bench_$K.json 2> $O/err_$K.txt
python3 - $O/tr$K <<'PY' | tee $O/timeline_$K.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if 'synth' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find last k_prepare groups
idx=[i for i,r in enumerate(rows) if 'k_prepare' in r['Kernel_Name']]
import re
K=int(__import__('os').environ.get('LCQP_RUN_CHUNKS','1'))
start=idx[-K]
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    nm=re.sub(r'<.*','',r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::',''))
    print(f"{nm:16s} q={r.get('Queue_Id','?'):>3s} grid={r.get('Grid_Size_X', r.get('Grid_Size','?')):>8s} start {(int(r['Start_Timestamp'])-t0)/1e6:8.3f} ms  end {(int(r['End_Timestamp'])-t0)/1e6:8.3f} ms  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:7.3f}")
PY
