#!/bin/bash
O=gpurun_out/r3b3; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_all.txt | tail -5
python tools/stall_probe.py > $O/stall.txt 2>&1; grep -v amdgpu.ids $O/stall.txt
python tools/readme_case.py > $O/readme_case.txt 2>&1; grep -v amdgpu.ids $O/readme_case.txt
python tools/scalar_probe.py > $O/scalar_probe.txt 2>&1; grep -v amdgpu.ids $O/scalar_probe.txt
for v in default tools/variants/libpvamd_cq2_*.so; do if [ $v = default ]; then CQ_LOGP=20,26 python tools/cq_sweep.py; else PVAMD_LIB=$v CQ_LOGP=20,26 python tools/cq_sweep.py; fi; done > $O/cq2.txt 2>&1; grep -v amdgpu.ids $O/cq2.txt
rocprofv3 -L > $O/counters_list.txt 2>&1
bash tools/pmc_cq64.sh r3b3 > /dev/null 2>&1; cat $O/pmc_cq64_default.md
bash tools/pmc_cq64.sh r3b3 tools/variants/libpvamd_cq2_np_w8.so > /dev/null 2>&1; cat $O/pmc_cq64_libpvamd_cq2_np_w8.md
