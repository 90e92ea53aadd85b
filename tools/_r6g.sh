O=gpurun_out/r6g; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
(time python3 bench.py > $O/bench.json 2> $O/bench.err); tail -3 $O/bench.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r6g/bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], "errors", d.get("errors"))
s = d["sparse_config5"]; g = s["grid_128"]
print("sparse", s["value"], s["roofline"]["frac"], "B4096", s["batch_4096"]["value"], "single ms", s.get("single_instance_ms"))
print("grid", {k: g[k] for k in ("value", "batch", "ms_per_step", "solved", "mean_lcqp_iterates", "max_lcqp_iterates")}, g["roofline"]["frac"], {k: g["cpu_baseline"][k] for k in ("value", "threads", "gpu_over_cpu", "single_core_value", "max_abs_dx_vs_gpu")})
print("in situ", d["backsolve_in_situ"]["frac"], "pipelined", d["pipelined"]["value"], "resident", d["resident_8192"]["value"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_over_cpu"])
PY
