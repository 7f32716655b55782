#!/bin/bash
# round 5: med3 range test + one-compare estimate check (shipped) vs per-axis estimate check; tests + fuzz on the shipped build
export TMPDIR=/tmp
O=gpurun_out/r5composed; mkdir -p $O
for v in "" peraxis; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  echo "== variant ${v:-shipped (med3 range test, max3 estimate check)}"
  PVAMD_LIB=$lib timeout 300 python tools/composed_ab.py c3 c4 big 2>&1 | grep "^C3\|^C4\|README"
done > $O/variants3.txt 2>&1
cat $O/variants3.txt
timeout 1200 python -m pytest tests/test_composed_gpu.py tests/test_composed_queue_gpu.py tests/test_robot_gpu.py tests/test_golden_gpu.py tests/test_float64_gpu.py tests/test_cached_gpu.py tests/test_index_rules.py tests/test_chamfer_gpu.py -m gpu -x -q 2>&1 | tail -4
PVAMD_FUZZ_SCALE=20 timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q -s 2>&1 | tail -8 | tee $O/fuzz20.txt
