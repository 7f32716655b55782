#!/bin/bash
# round 4, the one gfx950-specific experiment on the streaming regime of the C2 kernel: LDS-DMA point loads, 1-3 tiles in flight
O=gpurun_out/r4cq; mkdir -p $O
for v in "" cqdma1 cqdma2 cqdma3 cqdma2w4 cqdma2w16 cqdma1all cqdma2all; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  PVAMD_LIB=$lib CQ_MARGINS="0.05,-0.001,9" timeout 300 python tools/cq_sweep.py 2>&1 | grep "2^20"
done | tee $O/cq_dma.txt
# parity of the variant that would ship
PVAMD_LIB=tools/variants/libpvamd_cqdma2all.so timeout 600 python -m pytest tests/test_cached_gpu.py -q -m gpu 2>&1 | tail -2 | tee -a $O/cq_dma.txt
