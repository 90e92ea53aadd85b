#!/bin/bash
# Sparse arm against the number of scheduler wavefronts (LCQP_SPARSE_WAVES), one parametrised script (run on the GPU box from the repo root).
# usage: tools/run_sparse_ab.sh <tag> [reps] "B W" ["B W" ...]     W = 0: the library's rule
#   the round-5 sweeps:  small batches  "64 0" "64 16" "64 32" "64 64" "256 0" "256 64" "256 128" "256 256" "512 0" "512 128" "512 256" "512 512"
#                        moderate       "1024 0" "1024 512" "1024 1024" "2048 0" "2048 1024" "4096 0" "4096 1536" "4096 2048" "8192 0" "8192 1024"
#                        large          "16384 0" "65536 0"
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-spab}; shift
REPS=1
if [[ "$1" =~ ^[0-9]+$ ]]; then REPS=$1; shift; fi
mkdir -p $O
cd $R
for cfg in "$@"; do
  set -- $cfg; B=$1; W=$2
  if [ "$W" -eq 0 ]; then unset LCQP_SPARSE_WAVES; else export LCQP_SPARSE_WAVES=$W; fi
  for rep in $(seq $REPS); do
  timeout 600 python3 bench.py --workload sparse --batch $B --steps 2 --warmup 1 --cpu-sample 0 > $O/sp_${B}_$W.json 2>> $O/err.txt
  python3 - $B $W $O/sp_${B}_$W.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print(f"B {sys.argv[1]:>6s} waves {sys.argv[2]:>5s} (0 = the library's rule): {d['value']:8.0f} LCQPs/s  ms/step {d['ms_per_step']:9.1f} frac {d['roofline']['frac']:.3f} solved {d['config']['solved']}")
except Exception as e:
    print(f"B {sys.argv[1]} waves {sys.argv[2]}: failed ({e})")
PY
  done
done 2>&1 | tee $O/sparse_waves.log
