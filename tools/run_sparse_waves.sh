#!/bin/bash
# sparse arm at moderate batch sizes with different numbers of scheduler wavefronts (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-spwaves}
mkdir -p $O
cd $R
for B in 1024 4096 16384; do
for W in 0 256 512 1024 2048; do
  if [ $W -eq 0 ]; then unset LCQP_SPARSE_WAVES; else export LCQP_SPARSE_WAVES=$W; fi
  python3 bench.py --workload sparse --batch $B --steps 2 --warmup 1 --cpu-sample 0 > $O/sp_${B}_$W.json 2>> $O/err.txt
  python3 - $B $W $O/sp_${B}_$W.json <<'PY'
import json,sys
d=json.load(open(sys.argv[3]))
print(f"B {sys.argv[1]:>6s} waves {sys.argv[2]:>5s}: {d['value']:8.0f} LCQPs/s  ms/step {d['ms_per_step']:9.1f} frac {d['roofline']['frac']:.3f} solved {d['config']['solved']}")
PY
done
done 2>&1 | tee $O/sparse_waves.log
