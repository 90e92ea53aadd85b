O=gpurun_out/r6e; mkdir -p $O
(time timeout 1200 python3 -m pytest tests/test_python_api.py -x -q -k "neither_banded" --durations=5) > $O/pytest_grid.log 2>&1
tail -12 $O/pytest_grid.log
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
