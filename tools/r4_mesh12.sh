#!/bin/bash
cd /root/repo
python tools/ab_mesh.py 2>&1 | tail -1 | cut -c1-140
PVAMD_LIB=tools/variants/libpvamd_ob18.so python tools/ab_mesh.py 2>&1 | tail -1 | cut -c1-160
python tools/ab_mesh.py 2>&1 | tail -1 | cut -c1-140
PVAMD_LIB=tools/variants/libpvamd_ob18.so tools/trace_c5.sh r4trace_c5c 2>&1 | tail -12 | head -5
