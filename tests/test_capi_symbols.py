"""The C-ABI library loads on a CPU-only box and exports every symbol include/lcqp_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "lcqp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lcqp_hip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_declared_symbols():
    import lcqpow_amd
    L = ctypes.CDLL(lcqpow_amd.library_path())
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n


def test_options_default_matches_reference_defaults():
    import lcqpow_amd as la
    o = la.default_options()
    assert o.complementarityTolerance == 1e3 * 2.221e-16 and o.stationarityTolerance == 1e6 * 2.221e-16
    assert (o.initialPenaltyParameter, o.penaltyUpdateFactor, o.maxPenaltyParameter) == (0.01, 2.0, 1e8)
    assert (o.solveZeroPenaltyFirst, o.perturbStep, o.maxIterations, o.nDynamicPenalty) == (1, 1, 1000, 3)


def test_option_structs_have_identical_layout():
    import lcqpow_amd as la
    import oracle_py
    assert [f[0] for f in la.Options._fields_] == [f[0] for f in oracle_py.Options._fields_]
    assert ctypes.sizeof(la.Options) == ctypes.sizeof(oracle_py.Options)
    assert ctypes.sizeof(la.Stats) == ctypes.sizeof(oracle_py.Stats)


def test_no_product_import_of_oracle():
    """the shipped path must never route through the oracle"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lcqpow_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"#\s*include[^\n]*oracle", txt), f
                assert not re.search(r"\bimport\s+oracle_py|from\s+oracle_py", txt), f
                assert "liblcqp_oracle" not in txt and not re.search(r"\borc_[a-z_]+\s*\(", txt), f
