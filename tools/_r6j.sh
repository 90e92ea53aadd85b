O=gpurun_out/r6j; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python3 tools/micro/general_timing.py 44 64 90 128 > $O/general_ldl_timing.log 2>&1
LCQPOW_HIP_LIBRARY=build/ab/genprof.so python3 tools/micro/general_timing.py 128 > $O/general_ldl_profile.log 2>&1
cat $O/general_ldl_timing.log $O/general_ldl_profile.log | cut -c1-400
bash tools/run_profiles.sh r6r > $O/run_profiles.log 2>&1; grep -v "^void" $O/run_profiles.log | tail -6
