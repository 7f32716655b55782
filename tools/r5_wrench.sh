#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5wrench; mkdir -p $O
timeout 600 python -m pytest tests/test_wrench_fullsize_gpu.py -m gpu -q -s > $O/pytest.log 2>&1; grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl\|amdgpu.ids" $O/pytest.log | tail -12
