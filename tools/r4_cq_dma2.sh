#!/bin/bash
O=gpurun_out/r4cq; mkdir -p $O
for v in "" cq_np16 cqdma2w16 cq_dma1w16 cq_dma3w16 cq_dma4w8 cq_dma2w12 cq_dma2w16plain "" cqdma2w16; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  PVAMD_LIB=$lib CQ_MARGINS="0.05,-0.001,9" CQ_LOGP="20,26" timeout 300 python tools/cq_sweep.py 2>&1 | grep "2^20"
done | tee $O/cq_dma2.txt
