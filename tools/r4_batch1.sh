#!/bin/bash
# round 4, GPU batch 1: the queued composed leaf loop (parity + A/B timing), the host stall traced
export TMPDIR=/tmp
O=gpurun_out/r4b1; mkdir -p $O
timeout 900 python -m pytest tests/test_composed_queue_gpu.py tests/test_composed_gpu.py -q -m gpu -x > $O/pytest_composed.txt 2>&1; tail -15 $O/pytest_composed.txt
PVAMD_FUZZ_SCALE=3 timeout 600 python -m pytest tests/test_fuzz_gpu.py tests/test_robot_gpu.py tests/test_index_rules.py -q -m gpu -k "composed or robot or rule or fuzz" > $O/pytest_more.txt 2>&1; tail -5 $O/pytest_more.txt
timeout 600 python tools/composed_ab.py > $O/composed_ab.txt 2>&1; cat $O/composed_ab.txt
timeout 300 python tools/stall_trace.py > $O/stall_plain.txt 2>&1; tail -40 $O/stall_plain.txt
timeout 600 rocprofv3 --hip-trace --hsa-trace --output-format csv -d /tmp/st -o st -- python tools/stall_trace.py > $O/stall_traced.txt 2>&1; tail -12 $O/stall_traced.txt
python tools/hip_trace_top.py /tmp/st > $O/stall_api_top.txt 2>&1; head -60 $O/stall_api_top.txt
