#!/bin/bash
# instances per streaming step of the sparse phase machine: the fixed 16 of round 4 (a library built with it: build/ab/fuse2.so) against the
# rule "unfinished instances per SIMD" (build/ab/wbdyn.so), and that rule with other divisors (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-spwide}
mkdir -p $O
cd $R
(
for B in 256 1024 2048 4096 8192 16384 65536; do
  python3 tools/micro/sparse_variant_bench.py $B fixed16=build/ab/fuse2.so rule=build/ab/wbdyn.so
done
for cfg in "2048 512" "2048 2048" "4096 512" "4096 2048" "8192 256" "8192 1024"; do
  set -- $cfg
  echo "-- B = $1, LCQP_SPARSE_WIDE_DIV = $2"
  LCQP_SPARSE_WIDE_DIV=$2 python3 tools/micro/sparse_variant_bench.py $1 rule_div$2=build/ab/wbdyn.so
done
) 2>&1 | tee $O/sparse_wide_batch_rule.log
