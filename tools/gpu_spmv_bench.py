"""SpMV rate of the device CSC products (lcqp_hip_csc_apply) at the sizes of BASELINE config 5 (n = 4096) and beyond.
Bytes per product: 12 per non-zero (8 value + 4 index) + 4 per column pointer + 8 per input and output entry.
usage: python tools/gpu_spmv_bench.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lcqpow_amd as la


def csc_from_dense_pattern(m, n, per_col, rng):
    p = [0]; i = []; x = []
    for c in range(n):
        rows = np.sort(rng.choice(m, size=min(per_col, m), replace=False))
        i.extend(rows.tolist()); x.extend(rng.standard_normal(rows.size).tolist()); p.append(len(i))
    return np.array(p, dtype=np.int32), np.array(i, dtype=np.int32), np.array(x)


rng = np.random.default_rng(0)
print("shape, nnz, per product: microseconds, GB/s (algorithmic bytes), fraction of 8 TB/s")
for (m, n, per_col) in [(4096, 4096, 3), (4096, 4096, 20), (6142, 4096, 3), (65536, 65536, 20), (262144, 262144, 20), (1048576, 1048576, 16)]:
    p, i, x = csc_from_dense_pattern(m, n, per_col, rng)
    M = la.CSCMatrix(m, n, p, i, x)
    b = rng.standard_normal(n)
    d = M.apply(b, repeat=50); ms = M.last_ms
    ref = np.zeros(m); np.add.at(ref, i, x * np.repeat(b, np.diff(p)))
    assert np.abs(d - ref).max() < 1e-9 * (1 + np.abs(ref).max())
    bt = rng.standard_normal(m)
    dt = M.apply(bt, transposed=True, repeat=50); mst = M.last_ms
    nbytes = 12.0 * len(x) + 4.0 * (n + 1) + 8.0 * (m + n)
    for tag, t in (("A b ", ms), ("A'b ", mst)):
        print(f"{m}x{n} nnz {len(x):9d} {tag}: {1e3 * t:9.1f} us  {nbytes / (t * 1e-3) / 1e9:8.1f} GB/s  {nbytes / (t * 1e-3) / 8e12:6.3f}")
    M.close()
