#!/bin/bash
O=gpurun_out/r3b4; mkdir -p $O
python -m pytest tests -x -q -m gpu -s > $O/pytest_all.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_all.txt | tail -5
grep -E "poses above|rounding units" $O/pytest_all.txt
python tools/readme_case.py > $O/readme_case.txt 2>&1; grep -v amdgpu.ids $O/readme_case.txt | cut -c1-300
python tools/scalar_probe.py > $O/scalar_probe.txt 2>&1; grep -v amdgpu.ids $O/scalar_probe.txt
CQ_MARGINS="0.05,-0.001,9" python tools/cq_sweep.py > $O/cq.txt 2>&1; grep -v amdgpu.ids $O/cq.txt
bash tools/valu_counts.sh r3b4 > $O/valu_counts_stdout.txt 2>&1; cat $O/valu_counts_stdout.txt | head -60
cp $O/valu_counts.json profiles/r03_valu_counts.json
PVAMD_LIB=tools/variants/libpvamd_stats.so python tools/exact_pairs.py > $O/exact_pairs.json 2> $O/exact_pairs.err; cat $O/exact_pairs.json; cp $O/exact_pairs.json profiles/r03_exact_pairs.json
python tools/rule_exposure.py > $O/rule_exposure.txt 2>&1; grep -v amdgpu.ids $O/rule_exposure.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; python -c "
import json; d=json.load(open('$O/bench_k20.json')); print({k: d[k] for k in ('value','ms_per_step')}); print(d['roofline']['frac'], d['roofline']['dropin_call']); print({k:(v.get('ms_per_call') or v.get('sharded',{}).get('ms_per_step') or v.get('ms_per_step')) for k,v in d['legs'].items()}); print(d.get('large_batch'), d.get('mid_batch')); print(d['legs']['c5'].get('roofline')); print(d['legs']['c4']['sharded'].get('roofline'))"
python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; tail -c 1500 $O/configs.json
