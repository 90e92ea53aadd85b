"""HBM read bandwidth of the sparse engine's access pattern (8 streams per wavefront, 64 B each) vs one 512-B stream per wavefront."""
import ctypes as C, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "stream_pattern.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "stream_pattern.hip")])
L = C.CDLL(so)
L.run.argtypes = [C.c_int, C.c_long, C.c_int, C.c_int, C.POINTER(C.c_float)]
for waves in (1024, 2048, 4096, 8192):
    per = (8 << 30) // (waves * 8 * 8)          # 8 GiB in total
    per -= per % 32
    for inter in (0, 1):
        ms = C.c_float()
        rc = L.run(waves, per, inter, 3, C.byref(ms))
        gb = waves * 8 * per * 8 / 1e9
        print(f"waves {waves:5d} per-instance {per * 8 / 1e6:7.2f} MB  {'interleaved 512 B' if inter else '8 x 64 B streams '}: rc {rc} {ms.value:8.3f} ms -> {gb / ms.value:7.2f} TB/s")

L.run_multi.argtypes = [C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
print("a wavefront alternating between K far-apart arrays (total 32 GiB):")
for waves in (2048, 4096):
    per = (32 << 30) // (waves * 8 * 8)
    per -= per % (32 * 12)
    for K, blocked in ((1, 0), (4, 0), (12, 0), (12, 1)):
        ms = C.c_float()
        rc = L.run_multi(waves, per, K, blocked, 2, C.byref(ms))
        gb = waves * 8 * per * 8 / 1e9
        print(f"waves {waves:5d} K {K:2d} {'pieces of a wavefront adjacent' if blocked else 'arrays of B instances      '}: rc {rc} {ms.value:8.3f} ms -> {gb / ms.value:7.2f} TB/s")
