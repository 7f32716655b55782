#!/bin/bash
# round 5: the per-lane composed kernel with two points per lane
export TMPDIR=/tmp
O=gpurun_out/r5composed; mkdir -p $O
timeout 300 python tools/composed_ab.py c3 c4 2>&1 | grep "^C3\|^C4\|README" > $O/variants5.txt; cat $O/variants5.txt
timeout 900 python -m pytest tests/test_composed_gpu.py tests/test_composed_queue_gpu.py tests/test_robot_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed"
