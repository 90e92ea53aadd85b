"""CPU checks of the general sparse LDL' of the sparse arm: the symbolic analysis the product runs once per pattern
(lcqpow_amd/csrc/lcqp_sparse_general.hpp) with a scalar restatement of the device's numeric loops against a dense solve
(tests/cpp/general_ldl_test.cpp), and the oracle's own general LDL' (oracle/lcqp_oracle_sparse.c, w = -1) against its band LDL'."""
import os
import subprocess

import numpy as np
import pytest

import problems as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("gen") / "general_ldl_test")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-o", out, os.path.join(ROOT, "tests", "cpp", "general_ldl_test.cpp")])
    return out


@pytest.mark.parametrize("grid,leaf", [(6, 32), (12, 32), (30, 32), (44, 16), (60, 32), (60, 8)])
def test_symbolic_analysis_and_front_loops_against_a_dense_solve(exe, grid, leaf):
    """fronts in postorder, update rows behind the pivots and inside the parent's front, every entry of K assembled exactly once; K x = b
    solved through the fronts leaves a residual at rounding level after one refinement step, with every row in and with a third of them out"""
    r = subprocess.run([exe, str(grid), str(leaf)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "structure failures 0" in r.stdout


def test_symbolic_analysis_at_the_size_of_the_largest_test(exe):
    """128 x 128 grid (N = 21 845 nodes): the analysis the GPU test of that size relies on stays small -- the largest front fits a wavefront's
    panel, the factor a few MB"""
    r = subprocess.run([exe, "128", "32"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    import re
    m = re.search(r"fronts (\d+), largest front (\d+), nnz\(L\) (\d+)", r.stdout)
    assert m and int(m.group(2)) <= 576 and int(m.group(3)) < 2_000_000


@pytest.mark.parametrize("g,nK,nC", [(20, 60, 40), (44, 300, 200)])
def test_oracle_general_ldl_matches_its_band_ldl(oracle, g, nK, nC):
    """the sparse oracle on a 2-D grid problem: band LDL' in a reverse Cuthill-McKee ordering (half bandwidth ~ 2 g) against the general
    up-looking LDL' in a nested-dissection ordering -- two factorisations of the same KKT matrices, the same homotopy to 1e-12"""
    d = P.grid_lcqp(g, nK, nC)
    n = d["nV"]
    Qc, Ec = d["Q"].tocsr(), d["E"].tocsr()
    perm, w, kb = oracle.kkt_ordering(n, Qc.indptr, Qc.indices, Ec.indptr, Ec.indices, wmax=10 ** 6, kbmax=0)
    opt = oracle.default_options(perturbStep=0)
    rb = oracle.sparse_lcqp_solve(n, nC, nK, Qc, d["g"], Ec, lbA=d["lbA"], ubA=d["ubA"], perm=perm, w=w, kb=0, opt=opt)
    pg = oracle.kkt_ordering_general(n, Qc.indptr, Qc.indices, Ec.indptr, Ec.indices)
    rg = oracle.sparse_lcqp_solve(n, nC, nK, Qc, d["g"], Ec, lbA=d["lbA"], ubA=d["ubA"], perm=pg, w=-1, kb=0, opt=opt)
    assert rb["ret"] == rg["ret"] == 0 and rb["stats"]["iterTotal"] == rg["stats"]["iterTotal"]
    assert np.abs(rb["x"] - rg["x"]).max() < 1e-12 and np.abs(rb["y"] - rg["y"]).max() < 1e-10
