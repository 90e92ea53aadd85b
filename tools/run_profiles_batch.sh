#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_lcqp_run at several batch sizes: usage tools/run_profiles_batch.sh <tag> B1 B2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$1; shift
cd /tmp && export TMPDIR=/tmp
D="--cpu-sample 0 --no-pipelined --no-resident --no-backsolve --no-sparse"
for B in "$@"; do
  O=$R/gpurun_out/$T/b$B
  mkdir -p $O
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --batch $B $D > $O/bench_under_rocprof.json 2>> $O/rocprof.err
  rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --batch $B $D > /dev/null 2>> $O/rocprof.err
  echo "== batch $B"; python3 $R/tools/prof_summary.py $O | grep -E "k_lcqp_run|bench under"
done
