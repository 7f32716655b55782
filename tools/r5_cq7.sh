#!/bin/bash
# round 5: mixed tiles compacted (shipped) against the direct path for every tile (cq_nocompact)
export TMPDIR=/tmp
O=gpurun_out/r5cq; mkdir -p $O
timeout 900 python -m pytest tests/test_cached_gpu.py tests/test_index_rules.py tests/test_float64_gpu.py tests/test_golden_gpu.py tests/test_cabi_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | head -5
for v in cq_nocompact "" cq_nocompact ""; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  echo "== ${v:-shipped (compacted mixed tiles)}"
  PVAMD_LIB=$lib timeout 300 python tools/c2_floor_probe.py 2>&1 | grep -v amdgpu | head -4
  PVAMD_LIB=$lib CQ_LOGP=20,22,23 timeout 300 python tools/cq_sweep.py 2>&1 | grep -v amdgpu
done > $O/compact.txt 2>&1
cat $O/compact.txt
