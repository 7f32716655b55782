#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5cq; mkdir -p $O
timeout 300 python tools/c2_floor_probe.py 2>&1 | grep -v amdgpu > $O/floor.txt; cat $O/floor.txt
