#!/bin/bash
export TMPDIR=/tmp
PVAMD_LIB= timeout 300 python tools/c2_floor_probe.py 2>&1 | grep -v amdgpu | head -4
CQ_LOGP=20,22,23 timeout 300 python tools/cq_sweep.py 2>&1 | grep -v amdgpu
