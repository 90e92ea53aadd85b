#!/bin/bash
# A larger fuzz campaign for statistics (run on the GPU box from the repo root): batched device loop, seeds 14 .. 33, 400 problems each.
# usage: tools/run_fuzz_campaign.sh <tag>   (writes gpurun_out/<tag>/fuzz_campaign.log: one summary line per seed and every divergence)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-fuzzc}
mkdir -p $O
cd $R
for s in $(seq 14 33); do python3 tools/gpu_fuzz.py 400 $s 2>&1 | grep -v "branch QP\|^$"; done > $O/fuzz_campaign.log
python3 - $O/fuzz_campaign.log <<'PY' | tee -a $O/fuzz_campaign.log
import re, sys
tot = {}
n = 0
for l in open(sys.argv[1]):
    m = re.search(r"fuzz\[.*?\]: (\d+) problems \(seed (\d+)\): (\{.*?\});", l)
    if m:
        n += int(m.group(1))
        for k, v in eval(m.group(3)).items():
            tot[k] = tot.get(k, 0) + v
print(f"TOTAL over {n} problems: {tot}")
PY
