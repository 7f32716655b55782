#!/bin/bash
O=gpurun_out/r4cq; mkdir -p $O
for v in "" cq_np16 "" cq_np16; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  PVAMD_LIB=$lib CQ_MARGINS="0.05,-0.001,9" CQ_LOGP="20,23,26" timeout 300 python tools/cq_sweep.py 2>&1 | grep "2^20"
done | tee $O/cq_w16.txt
PVAMD_LIB=tools/variants/libpvamd_cq_np16.so timeout 600 python -m pytest tests/test_cached_gpu.py tests/test_index_rules.py -q -m gpu 2>&1 | tail -2 | tee -a $O/cq_w16.txt
