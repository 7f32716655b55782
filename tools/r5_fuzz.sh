#!/bin/bash
# the long differential campaign on the round's final kernels
export TMPDIR=/tmp
O=gpurun_out/r5fuzz; mkdir -p $O
{ echo "# PVAMD_FUZZ_SCALE=100 python -m pytest tests/test_fuzz_gpu.py -m gpu -q  (round 5, final kernels: $(date -u))"
  ( time PVAMD_FUZZ_SCALE=100 timeout 2400 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl\|amdgpu.ids\|^\.*$\|^\.\.\..*%\]$" ) 2>&1
  echo "# tests/test_composed_gpu.py tests/test_composed_queue_gpu.py tests/test_cached_gpu.py tests/test_mesh_gpu.py tests/test_sort_gpu.py"
  timeout 1200 python -m pytest tests/test_composed_gpu.py tests/test_composed_queue_gpu.py tests/test_cached_gpu.py tests/test_mesh_gpu.py tests/test_sort_gpu.py -m gpu -q 2>&1 | grep "passed\|failed"
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
