"""Keep only the product kernels' rows in the rocprofv3 CSVs under a directory (the loaders of a 65 536-instance sparse batch launch tens of
thousands of copy kernels: the untrimmed output exceeds what gpurun copies back).  usage: python tools/prof_trim.py <dir>"""
import csv, glob, os, sys
KEEP = ("k_lcqp_run", "k_backsolve", "k_build", "k_factor", "k_trsm", "k_prepare", "k_compress", "k_sparse", "k_qp_solve")
for f in glob.glob(os.path.join(sys.argv[1], "**", "*.csv"), recursive=True):
    if not (f.endswith("counter_collection.csv") or f.endswith("kernel_trace.csv")):
        continue
    rows = list(csv.DictReader(open(f)))
    if not rows or "Kernel_Name" not in rows[0]:
        continue
    kept = [r for r in rows if any(k in r["Kernel_Name"] for k in KEEP)]
    with open(f, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(kept)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*"), recursive=True):
    if os.path.isfile(f) and os.path.getsize(f) > (8 << 20):
        os.remove(f)      # (databases / traces nobody reads)
