#!/bin/bash
# Fuzz seeds beyond the five sets of tools/run_fuzz_sets.sh (run on the GPU box from the repo root): batched device loop on seeds 2, 4, 6, 7, 8, 9 (400 each),
# host loop over SubsolverHIP on seeds 5 and 11 (300 each).  usage: tools/run_fuzz_more.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-fuzz_more}
mkdir -p $O
cd $R
for s in 2 4 6 7 8 9; do python3 tools/gpu_fuzz.py 400 $s > $O/fuzz_batched_seed${s}_400.log 2>&1; done
for s in 5 11; do python3 tools/gpu_fuzz.py 300 $s host > $O/fuzz_host_seed${s}_300.log 2>&1; done
tail -n 4 $O/fuzz_*.log
