#!/bin/bash
# tangent bound A/B: timings, counters, and the mesh parity tests on the variant
cd /root/repo
for v in tan0 tan tan0 tan; do PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/ab_mesh.py 2>&1 | tail -1; done
for v in stats0 stats; do echo "#### $v"; PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/mesh_stats.py 2>&1 | grep -v "^$" | head -40; done
PVAMD_LIB=tools/variants/libpvamd_tan.so timeout 900 python -m pytest tests/test_mesh_gpu.py tests/test_chamfer_gpu.py -x -q -m gpu 2>&1 | tail -5
