#!/bin/bash
# The fuzz sets the parity claim is held to (run on the GPU box from the repo root): batched device loop on seeds 1 (600), 11 (400), 5 (150),
# host loop over SubsolverHIP on seeds 1 and 3 (300 each).  usage: tools/run_fuzz_sets.sh <tag>   (writes gpurun_out/<tag>/fuzz_*.log)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-fuzz}
mkdir -p $O
cd $R
python3 tools/gpu_fuzz.py 600 1 > $O/fuzz_batched_seed1_600.log 2>&1
python3 tools/gpu_fuzz.py 400 11 > $O/fuzz_batched_seed11_400.log 2>&1
python3 tools/gpu_fuzz.py 150 5 > $O/fuzz_batched_seed5_150.log 2>&1
python3 tools/gpu_fuzz.py 300 1 host > $O/fuzz_host_seed1_300.log 2>&1
python3 tools/gpu_fuzz.py 300 3 host > $O/fuzz_host_seed3_300.log 2>&1
tail -n 3 $O/fuzz_*.log
