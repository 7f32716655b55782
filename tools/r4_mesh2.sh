#!/bin/bash
cd /root/repo
for v in tan0 tanr tan0 tanr; do PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/ab_mesh.py 2>&1 | tail -1; done
for v in statsr; do echo "#### $v"; PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/mesh_stats.py 2>&1 | grep -v "^$" | head -40; done
