#!/bin/bash
O=gpurun_out/r3b6; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_all.txt | tail -5
PVAMD_FUZZ_SCALE=30 python -m pytest tests/test_fuzz_gpu.py -x -q -m gpu > $O/fuzz.txt 2>&1; grep -E "passed|failed|rror" $O/fuzz.txt | tail -3
python tools/coherent_probe.py > $O/coherent.txt 2>&1; grep -v amdgpu.ids $O/coherent.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print({k: d[k] for k in ('value','ms_per_step','steps')}); print(d['roofline']['frac'], d['roofline']['dropin_call']['ms_per_call']); print(d.get('large_batch'));print(d['cpu_baseline']['value'], d['cpu_baseline']['torch_opforop'])"
