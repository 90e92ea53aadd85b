O=gpurun_out/r6k; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
(time python3 bench.py > $O/bench.json 2> $O/bench.err); tail -3 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d.get('errors')); s=d['sparse_config5']; print(s['value'], s['roofline']['traffic'], s['grid_128']['value'], s['grid_128']['cpu_baseline']['gpu_over_cpu'])"
python3 -c "import __graft_entry__ as g; g.smoke()"
