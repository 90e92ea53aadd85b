#!/bin/bash
# scheduler wavefronts of the sparse engine at moderate batches, with the step-size rule in place (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-spwaves4}
mkdir -p $O
cd $R
for cfg in "1024 0" "1024 512" "1024 1024" "2048 0" "2048 1024" "2048 2048" "4096 0" "4096 1536" "4096 2048" "8192 0" "8192 1024" "256 0" "256 512"; do
  set -- $cfg; B=$1; W=$2
  if [ $W -eq 0 ]; then unset LCQP_SPARSE_WAVES; else export LCQP_SPARSE_WAVES=$W; fi
  for rep in 1 2; do
  timeout 300 python3 bench.py --workload sparse --batch $B --steps 1 --warmup 1 --cpu-sample 0 > $O/sp_${B}_$W.json 2>> $O/err.txt
  python3 - $B $W $O/sp_${B}_$W.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[3]))
    print(f"B {sys.argv[1]:>6s} waves {sys.argv[2]:>5s} (0 = the library's rule): {d['value']:8.0f} LCQPs/s  frac {d['roofline']['frac']:.3f} solved {d['config']['solved']}")
except Exception as e:
    print(f"B {sys.argv[1]} waves {sys.argv[2]}: failed ({e})")
PY
  done
done 2>&1 | tee $O/sparse_waves_with_step_rule.log
