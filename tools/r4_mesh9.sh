#!/bin/bash
cd /root/repo
python tools/ab_mesh.py 2>&1 | tail -1 | cut -c60-400
for v in aw1 aw2 aw1m64 aw2m64; do PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/ab_mesh.py 2>&1 | tail -1 | cut -c60-400; done
python tools/ab_mesh.py 2>&1 | tail -1 | cut -c60-400
PVAMD_LIB=tools/variants/libpvamd_aw1m64.so timeout 600 python -m pytest tests/test_mesh_gpu.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tail -4
