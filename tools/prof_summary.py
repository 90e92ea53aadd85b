"""Print what a tools/run_profiles.sh / run_profiles_dense.sh directory holds: average duration per kernel, FETCH_SIZE / WRITE_SIZE / SQ counters of the
product kernels (per launch), the traffic by the guide's recipe.  usage: python tools/prof_summary.py gpurun_out/<tag>"""
import csv, glob, json, os, sys
d = sys.argv[1]
def rows(pat):
    f = glob.glob(os.path.join(d, pat))
    return list(csv.DictReader(open(f[0]))) if f else []
KEYS = ("k_lcqp_run", "k_backsolve", "k_build", "k_factor", "k_trsm", "k_prepare", "k_sparse", "k_compress", "k_setup")
for r in rows("trace/*/*kernel_stats.csv") + rows("trace_sparse/*/*kernel_stats.csv"):
    if any(k in r["Name"] for k in KEYS):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e6:9.4f} ms  total {float(r['TotalDurationNs']) / 1e6:9.3f} ms")
out = {}
for tag in ("fetch", "write", "sq", "mfma", "fetch_sparse", "write_sparse", "sq_sparse"):
    acc = {}
    for r in rows(f"{tag}/*/*counter_collection.csv"):
        if any(k in r["Kernel_Name"] for k in KEYS):
            key = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
            acc.setdefault(key, []).append(float(r["Counter_Value"]))
    for (kn, cn), v in sorted(acc.items()):
        print(f"{kn:40s} {cn:22s} mean {sum(v) / len(v):.6g} over {len(v)} launches")
        out[(kn, cn)] = sum(v) / len(v)
for kn in sorted({k for k, _ in out}):
    if (kn, "FETCH_SIZE") in out and (kn, "WRITE_SIZE") in out:
        f, w = out[(kn, "FETCH_SIZE")], out[(kn, "WRITE_SIZE")]
        print(f"{kn}: traffic (2*FETCH + WRITE)*1024 = {(2 * f + w) * 1024 / 1e9:.2f} GB per launch (read {2 * f * 1024 / 1e9:.2f}, written {w * 1024 / 1e9:.2f})")
try:
    b = json.load(open(os.path.join(d, "bench_under_rocprof.json")))
    print("bench under rocprof:", b["value"], "LCQPs/s; algorithmic GB per launch", b["roofline"]["algorithmic_bytes_per_launch"] / 1e9, "kernel ms", b["config"]["homotopy_kernel_ms_per_step"])
except Exception as e:
    print("no bench json", e)
