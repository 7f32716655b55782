#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5mesh; mkdir -p $O
PVAMD_LIB=tools/variants/libpvamd_stats.so timeout 600 python tools/mesh_stats.py 21 > $O/stats_c5.txt 2>&1
cat $O/stats_c5.txt
