#!/bin/bash
# sparse arm against the pool size of the phase machine (LCQP_SPARSE_POOL) at moderate batch sizes (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-sppools}
mkdir -p $O
cd $R
for cfg in "1024 0" "1024 64" "1024 128" "1024 256" "4096 0" "4096 128" "4096 256" "4096 512" "4096 1024" "16384 0" "16384 512" "16384 1024" "65536 0" "65536 1024"; do
  set -- $cfg; B=$1; P=$2
  if [ $P -eq 0 ]; then unset LCQP_SPARSE_POOL; else export LCQP_SPARSE_POOL=$P; fi
  timeout 300 python3 bench.py --workload sparse --batch $B --steps 2 --warmup 1 --cpu-sample 0 > $O/sp_${B}_$P.json 2>> $O/err.txt
  python3 - $B $P $O/sp_${B}_$P.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[3]))
    print(f"B {sys.argv[1]:>6s} pool {sys.argv[2]:>5s}: {d['value']:8.0f} LCQPs/s  ms/step {d['ms_per_step']:9.1f} frac {d['roofline']['frac']:.3f} solved {d['config']['solved']}")
except Exception as e:
    print(f"B {sys.argv[1]} pool {sys.argv[2]}: failed ({e})")
PY
done 2>&1 | tee $O/sparse_pools.log
