mkdir -p gpurun_out/r6b; O=gpurun_out/r6b
LCQPOW_HIP_LIBRARY=build/ab/trace.so python3 tools/fuzz_case.py --trace 8 370 > $O/case370.log 2>&1
LCQPOW_HIP_LIBRARY=build/ab/trace.so python3 tools/fuzz_case.py --trace 8 46 > $O/case46.log 2>&1
python3 tools/fuzz_case.py 8 46 370 > $O/cases_seed8.log 2>&1
python3 tools/fuzz_case.py 2 134 254 > $O/cases_seed2.log 2>&1
python3 tools/fuzz_case.py 9 243 > $O/cases_seed9.log 2>&1
python3 tools/gpu_fuzz.py 400 8 > $O/fuzz_batched_seed8_400.log 2>&1
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -2 $O/fuzz_batched_seed8_400.log; tail -3 $O/pytest_gpu.log; grep -h "orc:\|hip:" $O/cases_seed8.log
