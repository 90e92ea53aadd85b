"""prints the objects of a bench.py JSON line in short form: python tools/show_bench.py <file>"""
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print(f"value {d['value']:.0f} {d['unit']}  ms/step {d['ms_per_step']:.2f}  setup {c.get('setup_ms_per_step', 0):.2f}  kernel {c.get('homotopy_kernel_ms_per_step', 0):.2f}  frac {d['roofline']['frac']:.3f}  traffic {d['roofline'].get('traffic')}  solved {c['solved']}")
for k in ("pipelined", "resident_8192", "backsolve_kernel"):
    print(k, {a: b for a, b in d.get(k, {}).items() if a in ("value", "frac", "roofline_frac", "error", "ms", "bitwise_equal_to_sequential")})
s = d.get("sparse_config5", {})
print("sparse", {a: s.get(a) for a in ("value", "batch", "ms_per_step", "error", "note", "solved")}, "frac", s.get("roofline", {}).get("frac"),
      "| B=4096:", s.get("batch_4096", {}).get("value"), "frac", s.get("batch_4096", {}).get("roofline", {}).get("frac"))
print("sparse cpu", {a: b for a, b in s.get("cpu_baseline", {}).items() if a != "sample"})
print("cpu", {a: b for a, b in d.get("cpu_baseline", {}).items() if a in ("value", "cores", "gpu_over_cpu", "error", "max_abs_dx_vs_gpu", "best_placement")})
