"""The profile tooling does not touch what it was not given (ADVICE, round 4): tools/prof_trim.py refuses directories outside gpurun_out/ and profiles/."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_prof_trim_refuses_foreign_directories(tmp_path):
    big = tmp_path / "precious.db"
    big.write_bytes(b"x" * (9 << 20))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_trim.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "nothing touched" in (r.stdout + r.stderr)
    assert big.exists()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_trim.py")], capture_output=True, text=True)
    assert r.returncode != 0


def test_prof_trim_keeps_product_kernels_and_drops_databases(tmp_path):
    d = tmp_path / "gpurun_out" / "tag" / "trace"
    d.mkdir(parents=True)
    (d / "1_kernel_trace.csv").write_text("Kernel_Name,Start_Timestamp\nvoid k_lcqp_run<2, true>(x),1\nsome_copy_kernel,2\n")
    (d / "1_results.db").write_bytes(b"x" * (9 << 20))
    (d / "small.json").write_text("{}")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_trim.py"), str(tmp_path / "gpurun_out" / "tag")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = (d / "1_kernel_trace.csv").read_text().splitlines()
    assert len(rows) == 2 and "k_lcqp_run" in rows[1]
    assert not (d / "1_results.db").exists() and (d / "small.json").exists()
