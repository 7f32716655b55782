#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_mesh_gpu.py tests/test_sort_gpu.py tests/test_chamfer_gpu.py tests/test_cached_gpu.py tests/test_sampler_gpu.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -4
PVAMD_FUZZ_SCALE=10 timeout 1200 python -m pytest tests/test_fuzz_gpu.py -x -q -m gpu -k "mesh or chamfer" 2>&1 | grep "passed\|failed\|Error" | tail -3
python tools/ab_mesh.py 2>&1 | tail -1
