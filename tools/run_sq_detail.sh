#!/bin/bash
# What do the waves of k_lcqp_run wait for?  Issue / LDS / vector-memory counters of the dense workload, one pass per group
# (run on the GPU box from the repo root): tools/run_sq_detail.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-sqdetail}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D="--steps 1 --warmup 0 --cpu-sample 0 --no-pipelined --no-resident --no-sparse --no-backsolve"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d $O/a --output-format csv -- python3 $R/bench.py $D > /dev/null 2>> $O/err.txt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS -d $O/b --output-format csv -- python3 $R/bench.py $D > /dev/null 2>> $O/err.txt
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU -d $O/c --output-format csv -- python3 $R/bench.py $D > /dev/null 2>> $O/err.txt
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_TAG_STALL_sum -d $O/d --output-format csv -- python3 $R/bench.py $D > /dev/null 2>> $O/err.txt
python3 - $O <<'PY' | tee $O/sq_detail.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "k_lcqp_run" in k or "k_factor" in k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()): print(f"{k:28s} {c:34s} {sum(v)/len(v):.5e}")
PY
tail -3 $O/err.txt
