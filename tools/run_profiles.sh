#!/bin/bash
# Profiles of a round (run on the GPU box from the repo root): kernel trace + stats, then FETCH_SIZE / WRITE_SIZE / SQ / MFMA counters each in
# its own pass (MI355X_MICROARCH.md, HBM section), for the dense workload and for the sparse one.  usage: tools/run_profiles.sh <tag>
# (writes gpurun_out/<tag>; tools/collect_profiles.py <tag> <round> copies the summaries to profiles/<round>/final)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
D="--cpu-sample 0 --no-pipelined --no-resident --no-sparse"
S="--workload sparse --steps 1 --warmup 0 --cpu-sample 0"
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $D > $O/bench_under_rocprof.json 2>> $O/rocprof.err
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $O/sq --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D --no-backsolve > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU -d $O/mfma --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $D --no-backsolve > /dev/null 2>> $O/rocprof.err
python3 $R/bench.py --workload sparse --steps 2 --warmup 1 > $O/bench_sparse.json 2> $O/bench_sparse.err
rocprofv3 --kernel-trace --stats -d $O/trace_sparse --output-format csv -- python3 $R/bench.py $S > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc FETCH_SIZE -d $O/fetch_sparse --output-format csv -- python3 $R/bench.py $S > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc WRITE_SIZE -d $O/write_sparse --output-format csv -- python3 $R/bench.py $S > /dev/null 2>> $O/rocprof.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/sq_sparse --output-format csv -- python3 $R/bench.py $S > /dev/null 2>> $O/rocprof.err
$R/tools/run_profiles_sizes.sh ${1:-r3} > /dev/null 2>&1
python3 $R/tools/prof_trim.py $O
find $O -name "*.csv" | head -80 > $O/files.txt
tail -5 $O/rocprof.err
python3 $R/tools/prof_summary.py $O 2>&1 | head -60
