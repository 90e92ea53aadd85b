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
    # no object of the line may hide a failure (bench.py exits non-zero and lists them under "errors" when one does)
    assert "errors" not in d, d["errors"]
    for k in ("pipelined", "backsolve_kernel", "cpu_baseline"):
        assert k in d and "error" not in d[k], (k, d.get(k))
    assert d["pipelined"]["bitwise_equal_to_sequential"] and d["backsolve_kernel"]["frac"] > 0
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["solved"] == 16
    assert d["config"]["max_lcqp_iterates"] >= d["config"]["mean_lcqp_iterates"] >= 1      # (a batch is as fast as its slowest instance: both are reported)
    assert abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None          # the PMC figure belongs to the BASELINE workload only
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "threads", "kind", "sample", "runs", "value_one_worker_per_core", "single_core_value", "host_stream_read_GBps_all_cores"):
        assert k in cb, k
    assert cb["kind"] == "port" and 1 <= cb["cores"] <= cb["threads"] and cb["value"] > 0 and cb["max_abs_dx_vs_gpu"] < 1e-9
    assert cb["value"] == max(r_["value"] for r_ in cb["runs"]) and all(r_["solved"] == r_["workers"] * r_["instances_per_worker"] for r_ in cb["runs"])


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
    assert d["config"]["max_lcqp_iterates"] >= d["config"]["mean_lcqp_iterates"] >= 1
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0 and d["roofline"]["traffic"] is None
    assert "errors" not in d and "error" not in d["cpu_baseline"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["max_abs_dx_vs_gpu"] < 1e-8


@pytest.mark.gpu
def test_bench_under_the_drivers_launcher():
    """the launcher path on a GPU box: bench.py as a fresh child of `python -m torch.distributed.run` with one rank --
    gloo process group (control plane) and HIP in the same process -- prints the one JSON line with n_gpus = 1"""
    port = 29600 + (os.getpid() % 300)
    env = dict(os.environ); env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16",
                        "--cpu-sample", "0", "--no-backsolve"],      # (the default shape: torch.distributed.run's own parser would take "--n" as a prefix of its options)
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["config"]["solved"] == 16 and d["config"]["global_batch"] == 16
    assert "no collective" in d["config"]["parallelism"] and d["scaling"] == "weak" and d["roofline"]["achieved"] > 0


def _one_json_line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_shards_rehearsed_on_one_device():
    """The N = 2 path of bench.py on a one-GPU box, so that an 8-GPU lease is not spent on a plumbing bug:
    `--gpus 2 --devices 0,0` runs two shards through the same run_devices / aggregation / JSON code as `--gpus 2` on two devices -- shard r
    solves instance ids [r B, (r + 1) B), max-over-devices time, summed solved counts.  Once from one process, once under the driver's
    launcher with two ranks that both take device 0 (gloo control plane: barrier, max of the step time, sum of the solved counts).
    The shards are different instances: the mean iterate count of rank 0's shard in the two-shard run equals the one-shard run's."""
    common = ["--steps", "2", "--warmup", "1", "--batch", "16", "--cpu-sample", "0", "--no-backsolve", "--no-pipelined", "--no-resident", "--no-sparse"]
    env = dict(os.environ); env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    one = _one_json_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=900, env=env))
    two = _one_json_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0"] + common,
                                        capture_output=True, text=True, timeout=900, env=env))
    port = 29700 + (os.getpid() % 200)
    launched = _one_json_line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                              "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0"] + common,
                                             capture_output=True, text=True, timeout=900, env=env))
    for d in (two, launched):
        assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
        assert d["config"]["global_batch"] == 32 and d["config"]["solved"] == 32
        assert "no collective" in d["config"]["parallelism"] and "rehearsal" in d["config"]["parallelism"]
        assert abs(d["value"] - 32 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
        assert d["config"]["mean_lcqp_iterates"] == one["config"]["mean_lcqp_iterates"]      # (the per-instance means are those of shard 0 = the one-shard run's instances)
    assert one["n_gpus"] == 1 and one["config"]["solved"] == 16


@pytest.mark.gpu
def test_cpp_shards_on_one_device_equal_one_batch():
    """examples/multi_gpu_batch (the sharding of SURVEY.md §8e from C++: one host thread, batch object and stream per shard): two shards of
    1024 instances on one device give the checksum of a single 2048-instance batch -- shard boundaries change nothing"""
    exe = os.path.join(ROOT, "examples", "bin", "multi_gpu_batch")
    if not os.path.exists(exe):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        ge.build_examples()
    out = {}
    for shards in (1, 2):
        r = subprocess.run([exe, "2048", str(shards)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        line = [l for l in r.stdout.splitlines() if "checksum" in l][-1]
        assert "2048/2048 LCQPs solved" in line, line
        out[shards] = float(line.split("checksum")[1].split(",")[0].strip())
    assert abs(out[1] - out[2]) <= 1e-10 * abs(out[1]), out      # (the sum of 2048 x 256 solution entries, associated per shard)


@pytest.mark.gpu
def test_bench_in_situ_backsolve_object():
    """the default workload's line carries the back-solves of the PRODUCT path (wg_trsv inside k_lcqp_run) beside the stand-alone kernel: bytes
    counted by the kernel, time from the stamped build of the same sources, results of the stamped build bit-identical to the product's"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--no-pipelined",
                        "--no-resident", "--no-sparse"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "errors" not in d
    b = d["backsolve_in_situ"]
    assert "error" not in b and b["bitwise_equal_to_product"]
    assert 50 < b["triangular_solves_per_lcqp"] < 400 and 0.02 < b["share_of_instance_cycles"] < 0.6 and 0.5 < b["mean_busy_share_of_launch"] <= 1.0
    assert abs(b["algorithmic_bytes_per_launch"] - b["triangular_solves_per_lcqp"] * 1024 * 8 * 256 * 258 / 2) < 1e-6 * b["algorithmic_bytes_per_launch"]
    assert b["frac"] > 0.1
