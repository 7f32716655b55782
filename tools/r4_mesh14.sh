#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_mesh_gpu.py tests/test_sort_gpu.py tests/test_sampler_gpu.py tests/test_cached_gpu.py tests/test_float64_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4
python tools/ab_mesh.py 2>&1 | tail -1
PYTHONPATH=tools python tools/tune_c1.py 2>&1 | grep auto | head -1
tools/trace_c1.sh r4trace_c1c 2>&1 | tail -8 | head -5
