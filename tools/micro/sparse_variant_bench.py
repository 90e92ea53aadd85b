"""Throughput of library variants on the sparse workload, interleaved: python tools/micro/sparse_variant_bench.py B name=lib.so ..."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = sys.argv[1]
for rnd in range(2):
    for a in sys.argv[2:]:
        name, so = a.split("=", 1)
        env = dict(os.environ, LCQPOW_HIP_LIBRARY=os.path.abspath(so))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "sparse", "--batch", B, "--steps", "1", "--warmup", "1", "--cpu-sample", "0"],
                           capture_output=True, text=True, env=env, timeout=900)
        try:
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
            print(f"{name:12s} B={B}: {d['value']:8.0f} LCQPs/s  solved {d['config']['solved']}  frac {d['roofline']['frac']:.3f}  iter {d['config']['mean_lcqp_iterates']:.2f}", flush=True)
        except Exception as e:
            print(name, "failed", r.returncode, r.stderr[-500:], flush=True)
