#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4b14; mkdir -p $O
timeout 900 python -m pytest tests/test_composed_queue_gpu.py tests/test_composed_gpu.py tests/test_index_rules.py tests/test_robot_gpu.py -q -m gpu > $O/pytest_composed.txt 2>&1; grep -E "^FAILED|passed|failed|AssertionError: " $O/pytest_composed.txt | head -30
PVAMD_FUZZ_SCALE=4 timeout 600 python -m pytest tests/test_fuzz_gpu.py tests/test_golden_gpu.py tests/test_dist_gloo.py -q -m gpu > $O/pytest_more.txt 2>&1; grep -E "^FAILED|passed|failed|AssertionError: " $O/pytest_more.txt | head -20
timeout 600 python tools/composed_ab.py c4 c3 2>&1 | grep "^C\|^README" | tee $O/composed_ab.txt
