#!/bin/bash
# lcqp_hip_batch_run in 1 ... 6 chunks on the BASELINE batch (run on the GPU box from the repo root): LCQPs/s, setup ms not hidden, homotopy span
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-chunks}
mkdir -p $O
cd $R
for rep in 1 2; do
for k in 1 2 3 4 6 8; do
  LCQP_RUN_CHUNKS=$k python3 bench.py --no-sparse --no-pipelined --no-resident --no-backsolve --cpu-sample 0 --steps 10 --warmup 3 > $O/chunks_$k.json 2>> $O/chunks.err
  python3 - $k $O/chunks_$k.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
c=d["config"]
print(f"chunks {sys.argv[1]}: {d['value']:.0f} LCQPs/s  ms/step {d['ms_per_step']:.2f}  setup(not hidden) {c['setup_ms_per_step']:.2f}  homotopy span {c['homotopy_kernel_ms_per_step']:.2f}  solved {c['solved']}")
PY
done
done 2>&1 | tee $O/chunks_ab.log
