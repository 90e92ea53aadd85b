#!/bin/bash
# Instruction-cache counters of the dense kernels (k_lcqp_run is 290 KB of code, the instruction cache 64 KB): tools/run_icache.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-icache}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/avail.txt 2>&1
grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*" $O/avail.txt | sort -u > $O/names.txt
cat $O/names.txt
D="--steps 1 --warmup 0 --cpu-sample 0 --no-pipelined --no-resident --no-sparse --no-backsolve"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/a --output-format csv -- python3 $R/bench.py $D > /dev/null 2>> $O/err.txt
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_IFETCH_LEVEL -d $O/b --output-format csv -- python3 $R/bench.py $D > /dev/null 2>> $O/err.txt
python3 - $O <<'PY' | tee $O/icache.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()): print(f"{k:28s} {c:34s} {sum(v)/len(v):.5e}")
PY
tail -5 $O/err.txt
