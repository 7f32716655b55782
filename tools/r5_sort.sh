#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5sort; mkdir -p $O
timeout 600 python -m pytest tests/test_sort_gpu.py tests/test_chamfer_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python tools/sort_probe.py > $O/sort_probe.txt 2>&1; cat $O/sort_probe.txt
