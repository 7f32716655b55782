#!/bin/bash
cd /root/repo
for v in noex default noex default; do if [ $v = default ]; then python tools/ab_mesh.py 2>&1 | tail -1; else PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/ab_mesh.py 2>&1 | tail -1; fi; done
echo "#### stats"; PVAMD_LIB=tools/variants/libpvamd_stats.so python tools/mesh_stats.py 2>&1 | grep -v "^$" | grep -v " - " | head -60
timeout 900 python -m pytest tests/test_mesh_gpu.py tests/test_chamfer_gpu.py -x -q -m gpu 2>&1 | tail -5
