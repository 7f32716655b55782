#!/bin/bash
# round 5: seed + greedy with its loads issued together (shipped) against the serial chains (seedold)
export TMPDIR=/tmp
O=gpurun_out/r5mesh; mkdir -p $O
for v in seedold "" seedold ""; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libpvamd_$v.so
  PVAMD_LIB=$lib timeout 300 python tools/ab_mesh.py 2>&1 | grep -v amdgpu
done > $O/seed_mlp.txt 2>&1
cat $O/seed_mlp.txt
timeout 900 python -m pytest tests/test_mesh_gpu.py tests/test_chamfer_gpu.py tests/test_sampler_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed"
