"""Keep only the product kernels' rows in the rocprofv3 CSVs under a directory (the loaders of a 65 536-instance sparse batch launch tens of
thousands of copy kernels: the untrimmed output exceeds what gpurun copies back) and drop the profiler's databases / traces nobody reads.
usage: python tools/prof_trim.py <dir>   (a directory under gpurun_out/ or profiles/; anything else is refused)"""
import csv, glob, os, sys
KEEP = ("k_lcqp_run", "k_backsolve", "k_build", "k_factor", "k_trsm", "k_prepare", "k_compress", "k_sparse", "k_qp_solve")
DROP_EXT = (".db", ".rocpd", ".pftrace", ".otf2", ".json")      # what rocprofv3 writes beside its CSVs
if len(sys.argv) != 2 or not sys.argv[1]:
    sys.exit(__doc__)
root = os.path.abspath(sys.argv[1])
if not os.path.isdir(root) or not any(part in ("gpurun_out", "profiles") for part in root.split(os.sep)):
    sys.exit(f"prof_trim: {root} is not a directory under gpurun_out/ or profiles/ -- nothing touched")
for f in glob.glob(os.path.join(root, "**", "*.csv"), recursive=True):
    if not (f.endswith("counter_collection.csv") or f.endswith("kernel_trace.csv")):
        continue
    rows = list(csv.DictReader(open(f)))
    if not rows or "Kernel_Name" not in rows[0]:
        continue
    kept = [r for r in rows if any(k in r["Kernel_Name"] for k in KEEP)]
    with open(f, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(kept)
for f in glob.glob(os.path.join(root, "**", "*"), recursive=True):
    if os.path.isfile(f) and f.endswith(DROP_EXT) and os.path.getsize(f) > (8 << 20):
        os.remove(f)
