#!/bin/bash
# round 3, second GPU batch: the any-P cached kernel (own-4 layout), variants for the 64M-point regime, host fast paths
O=gpurun_out/r3b2; mkdir -p $O
python -m pytest tests/test_cached_gpu.py tests/test_composed_gpu.py tests/test_robot_gpu.py tests/test_golden_gpu.py tests/test_float64_gpu.py tests/test_cabi_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
CQ_MARGINS="0.05,-0.001,9" python tools/cq_sweep.py > $O/cq.txt 2>&1
for v in tools/variants/libpvamd_cq_*.so; do PVAMD_LIB=$v CQ_MARGINS="0.05,-0.001,9" python tools/cq_sweep.py >> $O/cq.txt 2>&1; done
grep -v amdgpu.ids $O/cq.txt
python tools/readme_case.py > $O/readme_case.txt 2>&1; grep -v amdgpu.ids $O/readme_case.txt
python tools/outlier_probe.py > $O/outlier.txt 2>&1; grep -v amdgpu.ids $O/outlier.txt
