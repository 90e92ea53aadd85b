O=gpurun_out/r6d; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_sparse.py -x -q -k "general or dense_rows or too_dense or grid" > $O/pytest_general.log 2>&1
tail -15 $O/pytest_general.log
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "mixed" > $O/pytest_mixed.log 2>&1; tail -2 $O/pytest_mixed.log
timeout 600 python3 tools/gpu.py phase_profile 1024 --so=lcqpow_amd/liblcqpow_hip_prof.so > $O/phase_profile_B1024.log 2>&1
timeout 600 python3 tools/gpu.py phase_profile 1 --so=lcqpow_amd/liblcqpow_hip_prof.so > $O/phase_profile_B1.log 2>&1
cat $O/phase_profile_B1.log | head -30
