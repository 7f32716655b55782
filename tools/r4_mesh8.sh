#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_mesh_gpu.py tests/test_sort_gpu.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -4
python tools/ab_mesh.py 2>&1 | tail -1
tools/trace_c1.sh r4trace_c1b 2>&1 | tail -8 | head -4
