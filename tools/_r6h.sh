O=gpurun_out/r6h; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
bash tools/run_profiles.sh r6p > $O/run_profiles.log 2>&1; tail -30 $O/run_profiles.log
