#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_lcqp_run for a library variant: usage tools/micro/traffic_of_variant.sh path/to/lib.so tag
# (the variant is selected through LCQPOW_HIP_LIBRARY, read by lcqpow_amd/capi.py: the product library is never overwritten)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/var_$2
mkdir -p $O
export LCQPOW_HIP_LIBRARY=$(readlink -f $1)
cd /tmp && export TMPDIR=/tmp
D="--cpu-sample 0 --no-pipelined --no-resident --no-sparse --no-backsolve"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D > $O/bench_under_rocprof.json 2>> $O/err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D > /dev/null 2>> $O/err
python3 $R/tools/prof_summary.py $O | grep -E "k_lcqp_run|bench under"
