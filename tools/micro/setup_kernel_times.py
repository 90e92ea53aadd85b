import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
if not fs: sys.exit("no kernel_stats.csv under " + sys.argv[1])
for r in csv.DictReader(open(fs[0])):
    if any(k in r["Name"] for k in ("k_factor", "k_trsm", "k_build_M", "k_build_C", "k_lcqp_run", "k_compress_C", "k_prepare")):
        print("   %-30s calls %3s  avg %8.3f ms" % (r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e6))
