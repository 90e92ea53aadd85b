#!/bin/bash
# Dense-workload profiles (run on the GPU box from the repo root): kernel trace + stats, FETCH_SIZE / WRITE_SIZE / SQ in their own passes.
# usage: tools/run_profiles_dense.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D="--cpu-sample 0 --no-pipelined --no-resident --no-backsolve --no-sparse"
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $D > $O/bench_under_rocprof.json 2>> $O/rocprof.err
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $O/sq --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D > /dev/null 2>> $O/rocprof.err
python3 $R/tools/prof_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
