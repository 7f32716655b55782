#!/bin/bash
# round 5: cached_lookup with the composed kernels' statements (med3 range test, one exact-index branch, med3 bounding-box vector, v_sqrt + residual test)
export TMPDIR=/tmp
O=gpurun_out/r5cq; mkdir -p $O
for v in cq_oldlookup ""; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  echo "== ${v:-shipped (new look-up)}"
  PVAMD_LIB=$lib timeout 300 python tools/c2_floor_probe.py 2>&1 | grep -v amdgpu | head -4
  PVAMD_LIB=$lib CQ_LOGP=20,23,26 CQ_MARGINS="0.05,-0.001,9" timeout 300 python tools/cq_sweep.py 2>&1 | grep -v amdgpu
done > $O/lookup.txt 2>&1
cat $O/lookup.txt
timeout 900 python -m pytest tests/test_cached_gpu.py tests/test_index_rules.py tests/test_float64_gpu.py tests/test_golden_gpu.py tests/test_chamfer_gpu.py tests/test_cabi_gpu.py tests/test_voxelgrid_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed"
PVAMD_FUZZ_SCALE=30 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q -k "cached or composed" 2>&1 | grep "passed\|failed"
