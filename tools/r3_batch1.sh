#!/bin/bash
# round 3, first GPU batch: parity of the any-P wave-tile kernel, the README case, the 192 us outlier
mkdir -p gpurun_out/r3b1
python -m pytest tests/test_composed_gpu.py tests/test_robot_gpu.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r3b1/pytest_composed.txt 2>&1
tail -5 gpurun_out/r3b1/pytest_composed.txt
python tools/readme_case.py > gpurun_out/r3b1/readme_case.txt 2>&1; cat gpurun_out/r3b1/readme_case.txt
python tools/outlier_probe.py > gpurun_out/r3b1/outlier.txt 2>&1; cat gpurun_out/r3b1/outlier.txt
python tools/latency.py > gpurun_out/r3b1/latency.txt 2>&1; cat gpurun_out/r3b1/latency.txt
python tools/scalar_probe.py > gpurun_out/r3b1/scalar_probe.txt 2>&1; cat gpurun_out/r3b1/scalar_probe.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r3b1/pytest_all.txt 2>&1
tail -5 gpurun_out/r3b1/pytest_all.txt
