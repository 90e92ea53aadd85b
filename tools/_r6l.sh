O=gpurun_out/r6l; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
bash tools/run_profiles.sh r6s > $O/run_profiles.log 2>&1; grep -v "^void" $O/run_profiles.log | tail -4
python3 -c "
import json; d=json.load(open('gpurun_out/r6s/bench_sparse.json')); print('sparse', d['value'], d['roofline']['frac'], d['config']['solved'], d['config']['mean_lcqp_iterates'])"
