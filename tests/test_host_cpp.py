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
