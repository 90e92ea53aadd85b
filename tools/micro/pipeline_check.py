"""Do two batch objects in flight overlap?  Sequential steps, free-running alternation (no waits), and lcqpow_amd.BatchPipeline (waits for the
oldest batch before reusing its object), each for 10 steps of the BASELINE batch.  usage: python tools/micro/pipeline_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lcqpow_amd as la
B, n, nC, nK, K = 1024, 256, 512, 64, 10
opt = la.default_options(perturbStep=0, printLevel=0)
extra = []
if "--extra" in sys.argv:      # a third, idle batch object as in bench.py (more HIP streams in the process)
    e = la.BatchLCQP(B, n, nC, nK, opt=opt); e.generate_synthetic(0); e.run(); e.synchronize(); extra.append(e)
pipe = la.BatchPipeline(2, B, n, nC, nK, opt=opt)
for s in pipe.slots:
    s.generate_synthetic(0); s.run(); s.synchronize()
a, b = pipe.slots
t = time.perf_counter()
for k in range(K):
    a.run(); a.synchronize()
print(f"sequential           {B * K / (time.perf_counter() - t):8.0f} LCQPs/s")
t = time.perf_counter()
for k in range(K):
    (a, b)[k % 2].run()
a.synchronize(); b.synchronize()
print(f"free-running pair    {B * K / (time.perf_counter() - t):8.0f} LCQPs/s")
t = time.perf_counter()
for k in range(K):
    bt, done = pipe.acquire()
    pipe.launch(bt)
for bt in pipe.drain():
    pass
print(f"BatchPipeline        {B * K / (time.perf_counter() - t):8.0f} LCQPs/s")
