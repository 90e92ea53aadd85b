#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-spwaves2}
mkdir -p $O
cd $R
for cfg in "1024 0" "1024 512" "4096 0" "4096 2048" "16384 0" "65536 0"; do
  set -- $cfg; B=$1; W=$2
  if [ $W -eq 0 ]; then unset LCQP_SPARSE_WAVES; else export LCQP_SPARSE_WAVES=$W; fi
  python3 bench.py --workload sparse --batch $B --steps 2 --warmup 1 --cpu-sample 0 > $O/sp_${B}_$W.json 2>> $O/err.txt
  python3 - $B $W $O/sp_${B}_$W.json <<'PY'
import json,sys
d=json.load(open(sys.argv[3]))
print(f"B {sys.argv[1]:>6s} waves {sys.argv[2]:>5s}: {d['value']:8.0f} LCQPs/s  ms/step {d['ms_per_step']:9.1f} frac {d['roofline']['frac']:.3f} solved {d['config']['solved']}")
PY
done 2>&1 | tee $O/sparse_waves2.log
