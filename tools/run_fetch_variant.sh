#!/bin/bash
# FETCH_SIZE of the dense workload's kernels with another build of the library (run on the GPU box from the repo root): tools/run_fetch_variant.sh <tag> <lib.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-fetchvar}
mkdir -p $O
export LCQPOW_HIP_LIBRARY=$R/$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-pipelined --no-resident --no-sparse > $O/bench.json 2> $O/err.txt
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE": acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "k_lcqp_run" in k: print(k, "FETCH_SIZE KiB mean", sum(v) / len(v), "launches", len(v))
PY
