#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5cq; mkdir -p $O
{ timeout 200 python tools/launch_floor.py 2>&1 | grep -v amdgpu
  for v in "" cq_abl1; do lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so; echo "== ${v:-shipped}"; PVAMD_LIB=$lib timeout 300 python tools/c2_floor_probe.py 2>&1 | grep -v amdgpu | head -4; done; } > $O/floor2.txt 2>&1
cat $O/floor2.txt
