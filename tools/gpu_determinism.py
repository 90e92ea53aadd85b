import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lcqpow_amd as la
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bt = la.BatchLCQP(B, 256, 512, 64, opt=la.default_options(perturbStep=0))
bt.generate_synthetic(0)
res = []
for r in range(4):
    bt.run(); x, y, st = bt.solution(); res.append((x.copy(), y.copy(), st))
for r in range(1, 4):
    dx = np.abs(res[r][0] - res[0][0]); dy = np.abs(res[r][1] - res[0][1])
    bad = np.nonzero(dx.max(axis=1) > 0)[0]
    sd = [b for b in range(B) if res[r][2][b] != res[0][2][b]]
    print(f"run {r} vs 0: max dx {dx.max():.3e} max dy {dy.max():.3e} instances differing {len(bad)} (first {bad[:8]}), stats differing {sd[:8]}")
    if len(bad):
        b = bad[0]; i = np.argmax(dx[b]); print("   e.g. inst", b, "coord", i, res[0][0][b, i], res[r][0][b, i], res[0][2][b], res[r][2][b])
