#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5cq; mkdir -p $O
for v in "" cq_split "" cq_split; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  echo "== ${v:-shipped}"
  PVAMD_LIB=$lib timeout 300 python tools/c2_floor_probe.py 2>&1 | grep -v amdgpu | head -4
  PVAMD_LIB=$lib CQ_LOGP=20,22,23 timeout 300 python tools/cq_sweep.py 2>&1 | grep -v amdgpu
done > $O/split.txt 2>&1
cat $O/split.txt
PVAMD_LIB=tools/variants/libpvamd_cq_split.so timeout 600 python -m pytest tests/test_cached_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed"
