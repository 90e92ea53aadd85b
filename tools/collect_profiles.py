"""Copy the summaries of tools/run_profiles.sh (gpurun_out/<tag>) into profiles/<round>/final and write profiles/latest_traffic.json
(HBM bytes per k_lcqp_run launch by the guide's recipe, tagged with the hash of the kernel sources it was measured on).
usage: python tools/collect_profiles.py [tag=r3] [round=round3]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
TAG = sys.argv[1] if len(sys.argv) > 1 else "r3"
ROUND = sys.argv[2] if len(sys.argv) > 2 else "round3"
src = os.path.join(ROOT, "gpurun_out", TAG)
dst = os.path.join(ROOT, "profiles", ROUND, "final")
os.makedirs(dst, exist_ok=True)
for name, pat in (("kernel_stats.csv", "trace/*/*kernel_stats.csv"), ("kernel_trace.csv", "trace/*/*kernel_trace.csv"),
                  ("kernel_stats_sparse.csv", "trace_sparse/*/*kernel_stats.csv"), ("pmc_fetch_size.csv", "fetch/*/*counter_collection.csv"),
                  ("pmc_write_size.csv", "write/*/*counter_collection.csv"), ("pmc_sq.csv", "sq/*/*counter_collection.csv"),
                  ("pmc_fetch_size_sparse.csv", "fetch_sparse/*/*counter_collection.csv"), ("pmc_write_size_sparse.csv", "write_sparse/*/*counter_collection.csv"),
                  ("pmc_sq_sparse.csv", "sq_sparse/*/*counter_collection.csv"), ("pmc_mfma.csv", "mfma/*/*counter_collection.csv")):
    f = glob.glob(os.path.join(src, pat))
    if not f:
        print("missing:", pat); continue
    f = max(f, key=os.path.getmtime)      # (several runs of the profile script may have been merged into one directory: the newest)
    rows = list(csv.DictReader(open(f)))
    if "counter_collection" in f or "kernel_trace" in f:      # keep the product kernels only (the copies of the generator run are noise)
        rows = [r for r in rows if any(k in r.get("Kernel_Name", "") for k in ("k_lcqp_run", "k_backsolve", "k_build", "k_factor", "k_trsm", "k_prepare", "k_sparse"))]
    with open(os.path.join(dst, name), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for name in ("bench_default.json", "bench_under_rocprof.json", "bench_sparse.json", "other_instantiations.txt"):
    if os.path.exists(os.path.join(src, name)): shutil.copy(os.path.join(src, name), os.path.join(dst, name))
def mean_counter(fname, kernel, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(dst, fname))) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v)
fetch, write = mean_counter("pmc_fetch_size.csv", "k_lcqp_run", "FETCH_SIZE"), mean_counter("pmc_write_size.csv", "k_lcqp_run", "WRITE_SIZE")
out = {"kernel": "k_lcqp_run<2>", "workload": ["dense", 1024, 256, 512, 64], "source_hash": bench.kernel_source_hash("dense"),
       "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
       "traffic_bytes_guide_recipe": (2 * fetch + write) * 1024, "traffic_bytes_uncorrected": (fetch + write) * 1024,
       "backsolve_FETCH_SIZE_KiB": mean_counter("pmc_fetch_size.csv", "k_backsolve", "FETCH_SIZE"),
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-pipelined "
               "--no-resident` (tools/run_profiles.sh); traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (HBM); source_hash = "
               "bench.kernel_source_hash(workload) of the profiled sources (dense: the translation units of the dense kernels; sparse: all): bench.py reports the figure only while it matches"}
bs = json.load(open(os.path.join(dst, "bench_sparse.json")))
Bs = bs["config"]["global_batch"]
def sum_counter(fname, counter):      # k_sparse_setup + k_sparse_sched of the one profiled step
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(dst, fname))) if "k_sparse" in r["Kernel_Name"] and r["Counter_Name"] == counter)
fs, ws = sum_counter("pmc_fetch_size_sparse.csv", "FETCH_SIZE"), sum_counter("pmc_write_size_sparse.csv", "WRITE_SIZE")
sparse = {"kernel": "k_sparse_setup + k_sparse_sched", "workload": ["sparse", Bs, 4096, 2048, 512], "source_hash": bench.kernel_source_hash(),
          "FETCH_SIZE_KiB": fs, "WRITE_SIZE_KiB": ws, "traffic_bytes_guide_recipe": (2 * fs + ws) * 1024, "traffic_bytes_uncorrected": (fs + ws) * 1024,
          "note": "same recipe, `python3 bench.py --workload sparse --steps 1 --warmup 0 --cpu-sample 0`"}
out["entries"] = [dict(out), sparse]
json.dump(out, open(os.path.join(ROOT, "profiles", "latest_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# summary.txt: written from the files beside it (nothing copied from a log), one source state per file
def stats_rows(fname):
    f = os.path.join(dst, fname)
    return list(csv.DictReader(open(f))) if os.path.exists(f) else []
KEYS = ("k_lcqp_run", "k_backsolve", "k_build", "k_factor", "k_trsm", "k_prepare", "k_sparse", "k_compress")
L = [f"profiles/{ROUND}/final -- summary of the files in this directory, written by tools/collect_profiles.py",
     f"kernel sources: bench.kernel_source_hash() = {bench.kernel_source_hash()}, dense translation units alone {bench.kernel_source_hash('dense')} (every pass below was taken on these sources in ONE tools/run_profiles.sh run: {TAG})", ""]
for title, fname in (("kernel durations, dense default workload (rocprofv3 --kernel-trace --stats, `bench.py --steps 3 --warmup 1`): kernel_stats.csv", "kernel_stats.csv"),
                     ("kernel durations, sparse workload (`bench.py --workload sparse --steps 1 --warmup 0`): kernel_stats_sparse.csv", "kernel_stats_sparse.csv")):
    L.append(title)
    for r in stats_rows(fname):
        if any(k in r["Name"] for k in KEYS):
            L.append(f"  {r['Name'][:64]:64s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e6:10.4f} ms  total {float(r['TotalDurationNs']) / 1e6:10.3f} ms")
    L.append("")
tr = stats_rows("kernel_trace.csv")
if tr:
    L.append("registers / scratch / LDS of the product kernels (kernel_trace.csv; first launch of each):")
    seen = set()
    for r in tr:
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if nm in seen or not any(k in nm for k in KEYS):
            continue
        seen.add(nm)
        L.append(f"  {nm[:64]:64s} VGPR {r.get('VGPR_Count', r.get('Arch_VGPR_Count', '?')):>4s} AGPR {r.get('Accum_VGPR_Count', '?'):>4s} SGPR {r.get('SGPR_Count', '?'):>4s} scratch {r.get('Scratch_Size', r.get('Private_Segment_Size', '?')):>5s} B  LDS {r.get('LDS_Block_Size', r.get('Group_Segment_Size', '?')):>6s} B  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}")
    L.append("")
def counters(fname):
    acc = {}
    for r in stats_rows(fname):
        if any(k in r["Kernel_Name"] for k in KEYS):
            acc.setdefault((r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return acc
for title, fname in (("FETCH_SIZE (KiB per launch, dense)", "pmc_fetch_size.csv"), ("WRITE_SIZE (KiB per launch, dense)", "pmc_write_size.csv"), ("SQ counters (dense)", "pmc_sq.csv"),
                     ("matrix-core counters (dense setup kernels)", "pmc_mfma.csv"), ("FETCH_SIZE (KiB, sparse)", "pmc_fetch_size_sparse.csv"),
                     ("WRITE_SIZE (KiB, sparse)", "pmc_write_size_sparse.csv"), ("SQ counters (sparse)", "pmc_sq_sparse.csv")):
    acc = counters(fname)
    if not acc:
        continue
    L.append(title + ": " + fname)
    for (kn, cn), v in sorted(acc.items()):
        L.append(f"  {kn:48s} {cn:30s} mean {sum(v) / len(v):.6g} over {len(v)} launch(es)")
    L.append("")
L.append(f"HBM traffic by the guide's recipe, (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (traffic.json):")
L.append(f"  k_lcqp_run, B = 1024: {out['traffic_bytes_guide_recipe'] / 1e9:.2f} GB per launch")
L.append(f"  k_sparse_setup + k_sparse_sched, B = {Bs}: {sparse['traffic_bytes_guide_recipe'] / 1e12:.3f} TB per launch")
for name in ("bench_default.json", "bench_under_rocprof.json", "bench_sparse.json"):
    f = os.path.join(dst, name)
    if os.path.exists(f):
        b_ = json.load(open(f))
        L.append(f"{name}: {b_['value']:.0f} {b_['unit']}, {b_['ms_per_step']:.2f} ms per step, roofline.frac {b_['roofline']['frac']:.3f}, algorithmic bytes per launch {b_['roofline']['algorithmic_bytes_per_launch'] / 1e9:.2f} GB")
open(os.path.join(dst, "summary.txt"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
