#!/bin/bash
cd /root/repo
for v in h6 h5 h7 h8 h10 h6; do PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/ab_mesh.py 2>&1 | tail -1 | cut -c1-120; PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/c5_handover_probe.py 2>&1 | tail -1; done
