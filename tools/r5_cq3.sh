#!/bin/bash
# round 5: warming the XCD's L2 with the grid behind the point loads (PVAMD_CQ_WARM = 1 KB pieces per wave)
export TMPDIR=/tmp
O=gpurun_out/r5cq; mkdir -p $O
for v in cq_warm0 cq_warm1 "" cq_warm4 cq_warm0 ""; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  echo "== ${v:-shipped (warm 2)}"
  PVAMD_LIB=$lib timeout 300 python tools/c2_floor_probe.py 2>&1 | grep -v amdgpu | head -4
  PVAMD_LIB=$lib CQ_LOGP=20,23,26 timeout 300 python tools/cq_sweep.py 2>&1 | grep -v amdgpu
done > $O/warm.txt 2>&1
cat $O/warm.txt
timeout 600 python -m pytest tests/test_cached_gpu.py tests/test_index_rules.py tests/test_float64_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed"
