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
for name in ("bench_default.json", "bench_under_rocprof.json", "bench_sparse.json", "summary.txt"):
    if os.path.exists(os.path.join(src, name)): shutil.copy(os.path.join(src, name), os.path.join(dst, name))
def mean_counter(fname, kernel, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(dst, fname))) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v)
fetch, write = mean_counter("pmc_fetch_size.csv", "k_lcqp_run", "FETCH_SIZE"), mean_counter("pmc_write_size.csv", "k_lcqp_run", "WRITE_SIZE")
out = {"kernel": "k_lcqp_run<2>", "workload": ["dense", 1024, 256, 512, 64], "source_hash": bench.kernel_source_hash(),
       "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
       "traffic_bytes_guide_recipe": (2 * fetch + write) * 1024, "traffic_bytes_uncorrected": (fetch + write) * 1024,
       "backsolve_FETCH_SIZE_KiB": mean_counter("pmc_fetch_size.csv", "k_backsolve", "FETCH_SIZE"),
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-pipelined "
               "--no-resident` (tools/run_profiles.sh); traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (HBM); source_hash = "
               "bench.kernel_source_hash() of the profiled sources: bench.py reports the figure only while it matches"}
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
