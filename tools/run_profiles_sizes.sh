#!/bin/bash
# One kernel-trace line (registers, scratch, LDS, duration) for the instantiations the default workloads never launch (run on the GPU box from the
# repo root): the dense kernels at np = 1024 / 2048 / 4096 and the sparse scheduler with 16 / 32 / 64 lanes per instance.
# usage: tools/run_profiles_sizes.sh <tag>   (writes gpurun_out/<tag>/other_instantiations.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r5}
mkdir -p $O/sizes
cd /tmp && export TMPDIR=/tmp
if [ "$2" != "--summary-only" ]; then
X="--steps 1 --warmup 1 --cpu-sample 0 --no-backsolve --no-pipelined --no-resident --no-sparse"
rocprofv3 --kernel-trace -d $O/sizes/np1024 --output-format csv -- python3 $R/bench.py --batch 64 --n 1000 --nC 600 --nComp 200 $X > $O/sizes/np1024.json 2>> $O/sizes/err.txt
rocprofv3 --kernel-trace -d $O/sizes/np2048 --output-format csv -- python3 $R/bench.py --batch 16 --n 2000 --nC 800 --nComp 300 $X > $O/sizes/np2048.json 2>> $O/sizes/err.txt
rocprofv3 --kernel-trace -d $O/sizes/np4096 --output-format csv -- python3 $R/bench.py --batch 4 --n 4000 --nC 1000 --nComp 400 $X > $O/sizes/np4096.json 2>> $O/sizes/err.txt
for G in 16 32 64; do
  export LCQP_SPARSE_LANES=$G
  rocprofv3 --kernel-trace -d $O/sizes/G$G --output-format csv -- python3 $R/bench.py --workload sparse --batch 2048 --steps 1 --warmup 1 --cpu-sample 0 > $O/sizes/G$G.json 2>> $O/sizes/err.txt
done
unset LCQP_SPARSE_LANES
fi
python3 - $O <<'PY' | tee $O/other_instantiations.txt
import csv, glob, json, os, sys
O = sys.argv[1]
print("Instantiations the default workloads do not launch: one run each under rocprofv3 --kernel-trace (tools/run_profiles_sizes.sh); last launch of each kernel")
for tag, what in (("np1024", "dense n = 1000, nC = 600, nComp = 200, B = 64"), ("np2048", "dense n = 2000, nC = 800, nComp = 300, B = 16"), ("np4096", "dense n = 4000, nC = 1000, nComp = 400, B = 4"),
                  ("G16", "sparse n = 4096, B = 2048, LCQP_SPARSE_LANES=16"), ("G32", "sparse, 32 lanes per instance"), ("G64", "sparse, 64 lanes per instance")):
    try:
        b = json.load(open(os.path.join(O, "sizes", tag + ".json")))
        head = f"{tag}: {what}: {b['value']:.1f} LCQPs/s, {b['ms_per_step']:.1f} ms per step, solved {b['config']['solved']}/{b['config']['global_batch']}"
    except Exception as e:
        head = f"{tag}: {what}: no JSON line ({e})"
    print(head)
    f = glob.glob(os.path.join(O, "sizes", tag, "**", "*kernel_trace.csv"), recursive=True)
    if not f:
        print("   no kernel trace"); continue
    last = {}
    for r in csv.DictReader(open(f[0])):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if any(k in nm for k in ("k_lcqp_run", "k_sparse_sched<", "k_sparse_setup", "k_factor", "k_trsm", "k_build_M")):
            last[nm] = r
    for nm, r in sorted(last.items()):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        print(f"   {nm:28s} VGPR {r.get('VGPR_Count', r.get('Arch_VGPR_Count', '?')):>4s} AGPR {r.get('Accum_VGPR_Count', '?'):>4s} scratch {r.get('Scratch_Size', r.get('Private_Segment_Size', '?')):>5s} B  LDS {r.get('LDS_Block_Size', r.get('Group_Segment_Size', '?')):>6s} B  {dur:10.3f} ms")
PY
