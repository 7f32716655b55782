#!/bin/bash
cd /root/repo
python tools/ab_mesh.py 2>&1 | tail -1 | cut -c60-400
for v in mp16 mp24 mp48 mp62 pw6 pw8; do PVAMD_LIB=tools/variants/libpvamd_$v.so python tools/ab_mesh.py 2>&1 | tail -1 | cut -c60-400; done
python tools/ab_mesh.py 2>&1 | tail -1 | cut -c60-400
