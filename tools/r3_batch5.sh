#!/bin/bash
O=gpurun_out/r3b5; mkdir -p $O
python tools/coherent_probe.py > $O/coherent.txt 2>&1; grep -v amdgpu.ids $O/coherent.txt
bash tools/profile_bench.sh r3b5 > $O/profile_bench.txt 2>&1; tail -30 $O/profile_bench.txt
python tools/prof_sjc.py > $O/prof_sjc.txt 2>&1; head -45 $O/prof_sjc.txt
