#!/bin/bash
cd /root/repo
echo "## round-3 state: Z-order points, Z-order triangles, (tangent on)"; PVAMD_TRI_ORDER=morton PVAMD_LIB=tools/variants/libpvamd_zorder.so python tools/ab_mesh.py 2>&1 | tail -1
echo "## Hilbert points, Z-order triangles"; PVAMD_TRI_ORDER=morton python tools/ab_mesh.py 2>&1 | tail -1
echo "## Z-order points, patch triangles"; PVAMD_LIB=tools/variants/libpvamd_zorder.so python tools/ab_mesh.py 2>&1 | tail -1
echo "## Hilbert points, patch triangles, tangent on"; python tools/ab_mesh.py 2>&1 | tail -1
echo "## Hilbert points, patch triangles, tangent off"; PVAMD_LIB=tools/variants/libpvamd_tan0.so python tools/ab_mesh.py 2>&1 | tail -1
echo "## Hilbert points, patch triangles, tangent on"; python tools/ab_mesh.py 2>&1 | tail -1
echo "## Hilbert points, patch triangles, tangent off"; PVAMD_LIB=tools/variants/libpvamd_tan0.so python tools/ab_mesh.py 2>&1 | tail -1
echo "#### stats"; PVAMD_LIB=tools/variants/libpvamd_stats.so python tools/mesh_stats.py 2>&1 | grep -v "^$" | head -60
timeout 900 python -m pytest tests/test_mesh_gpu.py tests/test_chamfer_gpu.py tests/test_sort_gpu.py -x -q -m gpu 2>&1 | tail -5
