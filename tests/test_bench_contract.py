"""bench.py prints ONE JSON line with the keys the driver's contract names (small workload, GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16",
                        "--n", "64", "--nC", "96", "--nComp", "16", "--cpu-sample", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "LCQPs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["solved"] == 16
    assert abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None          # the PMC figure belongs to the BASELINE workload only
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["max_abs_dx_vs_gpu"] < 1e-9


def test_bench_refuses_a_smaller_run_than_asked_for():
    """`python bench.py --gpus N` without a launcher drives N devices itself and must not report a run on fewer: with no (or one)
    visible device it exits non-zero and prints no JSON line (runs on CPU and on a one-GPU box alike)"""
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("TORCHELASTIC_RUN_ID", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0", "--batch", "4", "--n", "64",
                        "--nC", "96", "--nComp", "16", "--cpu-sample", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_sparse_workload_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "sparse", "--steps", "1", "--warmup", "1", "--batch", "8",
                        "--n", "512", "--nC", "256", "--nComp", "64", "--cpu-sample", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["unit"] == "LCQPs/s" and d["config"]["solved"] == 8 and "sparse" in d["config"]["workload"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0 and d["roofline"]["traffic"] is None
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["max_abs_dx_vs_gpu"] < 1e-8
