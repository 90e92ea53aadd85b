O=gpurun_out/r6c; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
bash tools/run_fuzz_sets.sh r6c > /dev/null 2>&1
bash tools/run_fuzz_more.sh r6c > /dev/null 2>&1
bash tools/run_fuzz_campaign.sh r6c > /dev/null 2>&1
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_gpu.log; grep -h "^fuzz\[" $O/fuzz_*seed*.log | cut -c1-260; tail -1 $O/fuzz_campaign.log; grep -h "differ\|other stationary point:" $O/fuzz_*.log | cut -c1-200
