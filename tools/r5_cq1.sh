#!/bin/bash
# round 5: the C2 kernel at 1M points -- launch geometry (waves per workgroup x block cap = tiles per wave): does overlapping the
# read and the write phase inside a wave (grid-stride + prefetch) beat one tile per wave?
export TMPDIR=/tmp
O=gpurun_out/r5cq; mkdir -p $O
for v in "" cq_w8b256 cq_w8b512 cq_w8b1024 cq_w4b256 cq_w4b512 cq_w4b1024 cq_w16b128 cq_w2b1024 cq_w2b2048 ""; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  PVAMD_LIB=$lib CQ_LOGP=20,22,23 timeout 300 python tools/cq_sweep.py 2>&1 | grep -v amdgpu
done > $O/geometry.txt 2>&1
cat $O/geometry.txt
