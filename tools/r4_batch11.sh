#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4b11; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.txt 2>&1; grep -E "^FAILED|passed|failed|Error" $O/pytest_all.txt | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
