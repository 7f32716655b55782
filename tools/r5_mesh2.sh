#!/bin/bash
# round 5: what the seed + greedy phase of the mesh scan costs on its own (onlyseed: the scan returns after it -- wrong results, timing only)
export TMPDIR=/tmp
O=gpurun_out/r5mesh; mkdir -p $O
for v in "" onlyseed nogreedy ""; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  PVAMD_LIB=$lib timeout 300 python tools/ab_mesh.py 2>&1 | grep -v amdgpu
done > $O/seed_cost.txt 2>&1
cat $O/seed_cost.txt
