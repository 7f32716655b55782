#!/bin/bash
# the whole -m gpu suite, as the driver runs it
export TMPDIR=/tmp
O=gpurun_out/r5suite; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
