#!/bin/bash
# Kernel times of the dense workload at several batch sizes (rocprofv3 --kernel-trace --stats): is a setup kernel bound by the chain of one
# instance (time independent of the batch) or by the machine?   usage (GPU box, repo root): bash tools/micro/setup_kernel_times.sh 256 512 1024
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/setup_times
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  rocprofv3 --kernel-trace --stats -d $O/b$b --output-format csv -- python3 $R/bench.py --batch $b --steps 3 --warmup 1 --cpu-sample 0 --no-pipelined --no-resident --no-sparse > $O/b$b.json 2> $O/b$b.err
  echo "B = $b"
  python3 $R/tools/micro/setup_kernel_times.py $O/b$b
  rm -rf $O/b$b
done
