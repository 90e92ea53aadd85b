#!/bin/bash
# small sparse batches against the number of scheduler wavefronts (LCQP_SPARSE_WAVES); run on the GPU box from the repo root
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-spwaves3}
mkdir -p $O
cd $R
for cfg in "64 0" "64 16" "64 32" "64 64" "256 0" "256 64" "256 128" "256 256" "512 0" "512 128" "512 256" "512 512" "2048 0" "2048 512" "2048 1024" "8192 0" "8192 2048"; do
  set -- $cfg; B=$1; W=$2
  if [ $W -eq 0 ]; then unset LCQP_SPARSE_WAVES; else export LCQP_SPARSE_WAVES=$W; fi
  python3 bench.py --workload sparse --batch $B --steps 2 --warmup 1 --cpu-sample 0 > $O/sp_${B}_$W.json 2>> $O/err.txt
  python3 - $B $W $O/sp_${B}_$W.json <<'PY'
import json,sys
d=json.load(open(sys.argv[3]))
print(f"B {sys.argv[1]:>6s} waves {sys.argv[2]:>5s}: {d['value']:8.0f} LCQPs/s  ms/step {d['ms_per_step']:9.1f} frac {d['roofline']['frac']:.3f} solved {d['config']['solved']}")
PY
done 2>&1 | tee $O/sparse_waves3.log
