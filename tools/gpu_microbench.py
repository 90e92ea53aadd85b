import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lcqpow_amd as la
L = la.lib()
L.lcqp_hip_bench_rows.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
for (B, m, n) in ((1024, 640, 256), (4096, 640, 256), (1024, 140, 256)):
    for mode, name in ((1, "dots  A x "), (2, "axpy  A'y "), (3, "both      ")):
        ms = C.c_float(0)
        rc = L.lcqp_hip_bench_rows(B, m, n, mode, 20, C.byref(ms))
        by = B * m * n * 8.0
        print(f"B={B} m={m} n={n} {name}: rc {rc} {ms.value:.4f} ms  {by / ms.value / 1e6:.0f} GB/s")
