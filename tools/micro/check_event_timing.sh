R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/forkchk
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-pipelined --no-resident --no-sparse --no-backsolve > $O/bench.json 2>> $O/err
python3 $R/tools/prof_summary.py $O | head -12
python3 -c "
import json;b=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('events: setup',b['config']['setup_ms_per_step'],'kernel',b['config']['homotopy_kernel_ms_per_step'],'wall',b['ms_per_step'])"
