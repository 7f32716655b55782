#!/bin/bash
O=gpurun_out/r3final2; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_all.txt | tail -3
PVAMD_FUZZ_SCALE=40 python -m pytest tests/test_fuzz_gpu.py -x -q -m gpu -k "composed or cached" > $O/fuzz.txt 2>&1; grep -E "passed|failed|rror" $O/fuzz.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
bash tools/valu_counts.sh r3final2 > $O/valu_stdout.txt 2>&1; cp $O/valu_counts.json profiles/r03_valu_counts.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; python -c "
import json; d=json.load(open('$O/bench_k20.json')); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['dropin_call']['ms_per_call']); print({k:(v.get('ms_per_call') or v.get('sharded',{}).get('ms_per_step') or v.get('ms_per_step')) for k,v in d['legs'].items()}); print(d['legs']['c4']['sharded']['roofline']['valu_inst_per_step'], d['legs']['c4']['sharded']['roofline']['frac'])"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print({k: d[k] for k in ('value','ms_per_step','steps')}, d['roofline']['frac'], d['large_batch']['frac_of_8TBs'])"
