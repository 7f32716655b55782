#!/bin/bash
# what the driver runs at round end, on one box: GPU tests, smoke(), the bench line
O=gpurun_out/r3final; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_all.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; python -c "
import json; d=json.load(open('$O/bench_k20.json')); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['dropin_call']['ms_per_call']); print({k:(v.get('ms_per_call') or v.get('sharded',{}).get('ms_per_step') or v.get('ms_per_step')) for k,v in d['legs'].items()}); print(d['legs']['c1'])"
