#!/bin/bash
# NOT USABLE AS IS on this pool: the texture-addresser / L1 counter groups slowed the 13 s sparse run beyond a 25-minute limit (nothing came
# back).  Kept as a record of the counter names; run single groups under `timeout` on a small batch if at all.
# Which unit bounds k_sparse_sched?  Texture-addresser / L1 / L2 counters of the sparse workload, one rocprofv3 --pmc pass per group.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sp_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S="--workload sparse --steps 1 --warmup 0 --cpu-sample 0"
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/g$i --output-format csv -- python3 $R/bench.py $S > /dev/null 2>> $O/rocprof.err
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/g*/*/*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "k_sparse_sched" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, v in acc.items():
        print(f"{k:40s} {v:.6g}")
PY
grep -i "error\|invalid\|not found" $O/rocprof.err | head
