#!/bin/bash
# round 5: leaves taken in chunks (several value gathers in flight per wave) in the per-lane composed kernel
export TMPDIR=/tmp
O=gpurun_out/r5composed; mkdir -p $O
for v in chunk1 chunk2 "" chunk8; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  echo "== variant ${v:-shipped (chunk 4)}"
  PVAMD_LIB=$lib timeout 300 python tools/composed_ab.py c3 c4 big 2>&1 | grep "^C3\|^C4\|README"
done > $O/variants2.txt 2>&1
cat $O/variants2.txt
timeout 900 python -m pytest tests/test_composed_gpu.py tests/test_composed_queue_gpu.py tests/test_robot_gpu.py tests/test_golden_gpu.py tests/test_float64_gpu.py -m gpu -x -q 2>&1 | tail -4
