#!/bin/bash
# round 5, batch 1: the changed tests + the full bench (compact line check)
export TMPDIR=/tmp
O=gpurun_out/r5b1; mkdir -p $O
timeout 900 python -m pytest tests/test_robot_gpu.py tests/test_wrench_fullsize_gpu.py tests/test_bench_launch.py -m gpu -x -q -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
( time timeout 600 python bench.py --detail $O/bench_detail.json > $O/bench.out 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; cat $O/bench.time
tail -1 $O/bench.out | wc -c
tail -1 $O/bench.out
