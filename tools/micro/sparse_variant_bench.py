import sys, os, json
sys.path.insert(0, os.getcwd())
so, B = sys.argv[1], int(sys.argv[2])
import lcqpow_amd.capi as la
la._SO = os.path.abspath(so)
sys.argv = ["bench.py", "--workload", "sparse", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--batch", str(B)]
import runpy
runpy.run_path("bench.py", run_name="__main__")
