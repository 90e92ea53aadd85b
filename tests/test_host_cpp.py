"""The C++ host layer (lcqpow_amd/csrc/host: LCQProblem, Subsolver, SubsolverHIP, Options, OutputStatistics,
Utilities, BatchLCQProblem) through its own C++ test driver tests/cpp/host_tests.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_tests")


def _exe():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build_hip(); g.build_host(); g.build_host_tests()
    return EXE


def test_host_layer_cpu():
    """Utilities known answers (test/RunUnitTests.cpp:33-246), Options validation, OutputStatistics"""
    r = subprocess.run([_exe(), "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_host_layer_gpu():
    """RunWarmUp, CheckQPReturnFlag, example programs, OptimizeOnCircle and BatchLCQProblem via SubsolverHIP"""
    r = subprocess.run([_exe(), "gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("circle xOpt")][0]
    assert "i = 26; k = 8; rho = 2.56" in line, line     # same iterate counts as the oracle / device loop


@pytest.mark.gpu
def test_host_file_loader_example_data(tmp_path, oracle):
    """LCQProblem::loadLCQP(file names) (src/LCQProblem.cpp:147-387, Utilities::readFromFile :341-366) on the
    reference's example_data fixture (one value per line, inf accepted), solved through SubsolverHIP and
    compared with the oracle."""
    import numpy as np
    import problems as P
    z = np.load(os.path.join(P.GOLDEN, "example_data.npz"))
    for k in z.files:
        with open(tmp_path / (k + ".txt"), "w") as f:
            for v in np.ravel(z[k]):
                f.write("Inf\n" if v == np.inf else "-Inf\n" if v == -np.inf else repr(float(v)) + "\n")
    d = P.example_data()
    r = subprocess.run([_exe(), "files", str(tmp_path), str(d["nV"]), str(d["nC"]), str(d["nComp"])], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert "ret = 0; i = 34; k = 8; rho = 2.56" in lines[0], lines[0]
    x = np.array([float(t) for t in lines[1].split()[2:]])
    ro = P.oracle_solve(oracle, d, oracle.default_options(perturbStep=0))
    assert np.abs(x - ro["x"]).max() < 1e-7


@pytest.mark.gpu
def test_example_programs():
    """examples/warm_up.cpp and examples/batch_synthetic.cpp run to completion on the GPU"""
    import __graft_entry__ as g
    g.build_examples()
    bindir = os.path.join(ROOT, "examples", "bin")
    r = subprocess.run([os.path.join(bindir, "warm_up")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "xOpt = [" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([os.path.join(bindir, "batch_synthetic"), "64", "64", "96", "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "64/64 LCQPs solved" in r.stdout, r.stdout + r.stderr
    # batch sharding from C++: 4 shards (threads, batch objects, streams) vs one, same instance ids -> same checksum;
    # on a one-GPU box the shards share the device, which exercises the thread safety of the library
    outs = []
    for shards in ("1", "4"):
        r = subprocess.run([os.path.join(bindir, "multi_gpu_batch"), "128", shards, "64", "96", "16"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "128/128 LCQPs solved" in r.stdout, r.stdout + r.stderr
        outs.append(float(r.stdout.split("checksum")[1].split(",")[0]))
    assert abs(outs[0] - outs[1]) <= 1e-10 * abs(outs[0]), outs      # the sum over shards is taken in another order


@pytest.mark.gpu
def test_reference_example_programs(tmp_path):
    """examples/optimize_on_circle.cpp and examples/solve_lcqp_from_file.cpp: the other two programs of the reference's examples/ directory
    (OptimizeOnCircle.cpp, solve_lcqp_from_file.cpp) written against this backend's LCQProblem"""
    import re
    import numpy as np
    import problems as P
    import __graft_entry__ as g
    g.build_examples()
    bindir = os.path.join(ROOT, "examples", "bin")
    for args in (["100"], ["100", "dense"], ["400"]):
        r = subprocess.run([os.path.join(bindir, "optimize_on_circle")] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"xOpt = \[ ([-0-9.e]+), ([-0-9.e]+) \]; \|xOpt\| = ([0-9.]+)", r.stdout)
        x = (float(m.group(1)), float(m.group(2)))
        # the two local minimisers the reference's program accepts (examples/OptimizeOnCircle.cpp:144-145), on the polygon around the circle
        assert min(abs(x[0] - 0.1811) + abs(x[1] + 0.9835), abs(x[0] - 0.9764) + abs(x[1] + 0.2183)) < 2e-3, r.stdout
        assert abs(float(m.group(3)) - 1.0) < 2e-3
    z = np.load(os.path.join(P.GOLDEN, "example_data.npz"))
    for k in z.files:
        with open(tmp_path / (k + ".txt"), "w") as f:
            for v in np.ravel(z[k]):
                f.write("Inf\n" if v == np.inf else "-Inf\n" if v == -np.inf else repr(float(v)) + "\n")
    r = subprocess.run([os.path.join(bindir, "solve_lcqp_from_file"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nV = 151, nC = 50, nComp = 100" in r.stdout and "xOpt =" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([os.path.join(bindir, "solve_lcqp_from_file"), str(tmp_path / "nothing")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1
