O=gpurun_out/r6d; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_sparse.py -x -q -k "general or dense_rows or too_dense or grid" > $O/pytest_general.log 2>&1
tail -25 $O/pytest_general.log
